#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X ORB front-end and local-BA back-end (BASELINE.json configs[1] + the
north_star BA target).

One "step" = `--batches-per-step` (default 8) passes of the front-end over a batch of B = 512 synthetic 640x480 frames
per GPU: ORB extraction (8-level pyramid, FAST cells, quadtree, blur, rBRIEF) + brute-force Hamming matching of the B
consecutive frame pairs (k, k+1) -- 4096 frames per GPU per step, so that the timed region of the default run is
well over half a second.  Inputs are resident in HBM before the timed region starts; outputs stay in HBM.  With N > 1
GPUs every rank processes its own frames (weak scaling) and the one exchange step of the path is an RCCL all-gather of
the first frame slab of every rank (the last local frame is matched against the first frame of the next rank).

The second half of BASELINE's metric, local-BA windows/s (100 keyframes x 5000 points, 5+10 LM iterations), is
measured after the front-end on every rank -- windows are independent units sharded by rank, no collective -- and
reported under `local_ba` with its own roofline, CPU baseline and the one-window-in-flight latency.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (driver contract) with `roofline` (dominant kernel, HIP-event durations measured in a
separate pass after the timed region) and, at N == 1, `cpu_baseline` (the CPU oracle -- a port of the reference path --
timed on a bounded sample on this box's host cores).  The oracle is also the CHECKER of the timed regions' own outputs:
`parity` (slab rows and match rows of the last timed batch, bit-exact) and `local_ba.parity` (windows of the last
streamed batch within 1e-5); a mismatch makes the exit code 3.  `single_stream` is the reference's real call pattern
(one frame at a time through the drop-in C++ classes, tests/cpp/tracking_main.cpp) run as a child process before this
one creates its HIP queues; `tracking_loop` the same chain with Python in the loop, one stream per rank.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480
NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH = 1000, 1.2, 8, 20, 7
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_GINST = 1024 * 1.2  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz
MFMA_I8_PEAK_TOPS = 5000.0  # dense i8 MFMA: twice the ~2.5 PFLOP/s bf16 peak (MI355X_MICROARCH.md, Matrix cores)


def algorithmic_bytes_per_frame(level_sizes, n_kp):
    """per-stage ALGORITHMIC bytes for one frame (aprons excluded)."""
    ws, hs = level_sizes
    P = int(sum(int(a) * int(b) for a, b in zip(ws, hs)))
    return {
        "pyramid": W * H + P,               # input read + pyramid write
        "fast": P,                          # every level read once
        "quadtree": 0,
        "blur": 2 * P,                      # level read + blurred level write
        "describe": (749 + 512 + 60) * n_kp,  # orientation gather + descriptor gather + 28 B kp + 32 B desc
        "match_knn": 32 * 2 * n_kp + 64 * n_kp,  # both descriptor sets in, eight 8-byte candidate keys per query out
        "match_resolve": 64 * n_kp + 56 * n_kp + 4 * n_kp,  # candidate keys + two keypoint sets in, vnMatches12 out
    }


SINGLE_KERNEL_STAGES = ("fast", "quadtree", "blur", "describe", "match_knn", "match_resolve")
# stage -> the kernel symbol(s) behind it (csrc/*.hip) and the source file the stage's PMC figures depend on
STAGE_KERNEL = {"pyramid": "k_copy_level0 + k_resize x7", "fast": "k_fast_cells_wave", "quadtree": "k_quadtree",
                "blur": "k_blur", "describe": "k_describe", "match_knn": "k_match_knn_mfma", "match_resolve": "k_match_init"}
STAGE_SOURCE = {"pyramid": "orb_pyramid.hip", "blur": "orb_pyramid.hip", "fast": "orb_fast.hip", "quadtree": "orb_quadtree.hip",
                "describe": "orb_describe.hip", "match_knn": "orb_match.hip", "match_resolve": "orb_match.hip",
                "ba_iteration": "ba_api.hip"}


def stage_source_hashes():
    """sha256/16 of every kernel source file (+ orb_common.h, which they all include): stamped into
    profiles/traffic.json by tools/collect_traffic.py when the PMC passes are collected"""
    import hashlib
    d = os.path.join(ROOT, "orbslam_in_practice_amd", "csrc")
    common = open(os.path.join(d, "orb_common.h"), "rb").read()
    return {f: hashlib.sha256(common + open(os.path.join(d, f), "rb").read()).hexdigest()[:16]
            for f in sorted(set(STAGE_SOURCE.values()))}


def load_traffic():
    """profiles/traffic.json = PMC figures (HBM bytes, VALU instructions per launch) from separate rocprofv3 --pmc passes:
    they are NOT measured in this run, so every stage whose kernel source has changed since the passes were collected is
    dropped (-> null in the line) instead of going stale silently.  Returns (figures still valid, source tag)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return {}, None
    stamped, now = tj.get("_src_sha16", {}), stage_source_hashes()
    tj["_stale_valu"] = {}
    for st, src in STAGE_SOURCE.items():
        if stamped.get(src) != now[src]:
            tj.pop(st, None)
            if "valu_insts_" + st in tj:   # kept aside: still good enough to NAME the dominant kernel, never reported
                tj["_stale_valu"][st] = tj.pop("valu_insts_" + st)
    return tj, "profiles/traffic.json@%s (separate rocprofv3 --pmc passes; stages whose csrc changed since are null)" % tj.get("_collected", "?")


def cpu_baseline(n_frames):
    """the CPU oracle (oracle/, a port of the reference's single-threaded path) on a bounded sample"""
    from oracle import oracle as orc
    from orbslam_in_practice_amd import synth
    frames = synth.make_stream(12345, n_frames + 1, W, H)
    o = orc.OrbOracle(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH)
    o.extract(frames[0])  # warm-up
    t0 = time.perf_counter()
    feats = [o.extract(f) for f in frames]
    bounds = (0.0, float(W), 0.0, float(H))
    for k in range(n_frames):
        orc.search_for_initialization(feats[k][0], feats[k][1], feats[k + 1][0], feats[k + 1][1], bounds, None, -1, 0.9,
                                      True, 0)
    dt = time.perf_counter() - t0
    one = {"value": round(n_frames / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames 640x480 extract + brute-force match(k,k+1), oracle/ (scalar C++ -O2 port of the "
                     "reference's single-threaded path, no SIMD), %.1f s" % (n_frames, dt)}
    # the embarrassingly parallel upper bound: one frame (and its match) per host core
    from concurrent.futures import ThreadPoolExecutor
    ncores = os.cpu_count() or 1
    nthr = min(ncores, 64)
    per = max(4, n_frames // 4)

    def worker(t):
        oo = orc.OrbOracle(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH)   # one extractor per thread (not re-entrant)
        fs = [oo.extract(frames[(t + i) % len(frames)]) for i in range(per + 1)]
        for i in range(per):
            orc.search_for_initialization(fs[i][0], fs[i][1], fs[i + 1][0], fs[i + 1][1], bounds, None, -1, 0.9, True, 0)
        return per

    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as ex:   # ctypes releases the GIL inside the oracle calls
        done = sum(ex.map(worker, range(nthr)))
    dta = time.perf_counter() - t0
    allc = {"value": round(done / dta, 2), "unit": "frames/s", "cores": nthr, "kind": "port",
            "sample": "%d threads x %d frames, one oracle instance per thread, %.1f s" % (nthr, per, dta)}
    return one, allc


def frontend_parity(frames, j, snap, chk, B, window):
    """the oracle as the CHECKER of the timed region: frames `chk` of the last timed batch (batch j of the step = the
    synthetic stream translated circularly by (7j, 13j) px) are extracted and matched by oracle/ and compared with the
    slab rows and vnMatches12 rows the device left in HBM -- bit for bit (src/ORBextractor.cpp:1001-1065,
    src/ORBmatcher.cpp:9-126 with the brute-force window)"""
    from oracle import oracle as orc
    o = orc.OrbOracle(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH)
    bounds = (0.0, float(W), 0.0, float(H))
    feats = {i: o.extract(np.ascontiguousarray(np.roll(frames[i], (7 * j, 13 * j), axis=(0, 1)))) for i in chk}
    res = {"frames": len(chk), "pairs": 0, "keypoints": 0, "keypoints_equal": True, "descriptors_equal": True,
           "matches_equal": True, "matches": 0,
           "checked": "frames %s of the LAST timed batch (slab rows + vnMatches12 rows left in HBM) vs oracle/, bit-exact" % chk}
    for i in chk:
        cnt, kps, desc, m12, nm = snap[i]
        ok_, od = feats[i]
        res["keypoints"] += cnt
        same_n = cnt == len(ok_)
        res["keypoints_equal"] &= bool(same_n and np.array_equal(kps[:cnt].reshape(-1).view(orc.KP_DTYPE), ok_))
        res["descriptors_equal"] &= bool(same_n and np.array_equal(desc[:cnt], od))
        if i + 1 in feats:     # pair (i, i+1)
            on, om12, _ = orc.search_for_initialization(ok_, od, feats[i + 1][0], feats[i + 1][1], bounds, None, window, 0.9, True)
            res["pairs"] += 1
            res["matches"] += int(on)
            res["matches_equal"] &= bool(same_n and nm == on and np.array_equal(m12[:cnt], om12))
    res["ok"] = bool(res["keypoints_equal"] and res["descriptors_equal"] and res["matches_equal"] and res["pairs"] > 0)
    return res


def tracking_loop(n_frames, device=0, rank=0, world=1, barrier=None, max_over_ranks=None):
    """BASELINE configs[4] in the shape this repo can give it (orbslam_in_practice_amd/sequence.py): ONE stream per GPU
    through extract -> SearchByProjection -> PoseOptimization per frame and a local BA per keyframe, the calls chained
    the way Tracking / LocalMapping chain them, host-pointer API, Python glue included; ATE against the synthetic
    ground truth.  At N > 1 every rank runs its own stream (seed 5 + rank) between two barriers: value = world x frames
    / the slowest rank's time.  (The same loop on the CPU oracle -- identical matches and trajectory -- is
    tests/test_gpu_sequence.py's business.)"""
    from orbslam_in_practice_amd import sequence as sq
    frames, gt, K, _ = sq.make_plane_sequence(5 + rank, n_frames)
    be = sq.GpuBackend(NFEATURES, frames.shape[2], frames.shape[1], device=device)
    sq.run_sequence(be, frames[:6], K, gt[0])  # warm-up
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    r = sq.run_sequence(be, frames, K, gt[0])
    dt = time.perf_counter() - t0
    if barrier is not None:
        barrier()
        dt = max_over_ranks(dt)
    be.close()
    # the same loop with a frame's three calls as ONE device chain and the next frame fetched ahead (Tracker, csrc/orb_track.hip)
    be2 = sq.TrackerBackend(NFEATURES, frames.shape[2], frames.shape[1], device=device)
    sq.run_sequence(be2, frames[:6], K, gt[0])
    be2.reset()
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    r2 = sq.run_sequence(be2, frames, K, gt[0])
    dt2 = time.perf_counter() - t0
    if barrier is not None:
        barrier()
        dt2 = max_over_ranks(dt2)
    be2.close()
    chain = {"value": round(world * n_frames / dt2, 1), "unit": "frames/s", "ate_rmse_m": round(sq.ate_rmse(r2["poses"], gt), 5),
             "same_matches": bool(np.array_equal(r2["n_matches"], r["n_matches"])),
             "what": "the same loop through Tracker.track(next_gray=...): one device chain per frame, the next frame extracted beside it"}
    return {"value": round(world * n_frames / dt, 1), "unit": "frames/s", "streams": world, "device_chain": chain,
            "ate_rmse_m": round(sq.ate_rmse(r["poses"], gt), 5),
            "local_ba_runs": int(r["ba_runs"]), "map_points": int(r["n_points"]), "mean_matches": round(float(r["n_matches"][1:].mean()), 1),
            "workload": "%d synthetic frames %dx%d of a textured plane per stream, %d features, keyframe every 4 frames, local BA "
                        "over 5 keyframes; one stream per GPU, one frame at a time, Python glue in the loop"
                        % (n_frames, frames.shape[2], frames.shape[1], NFEATURES)}


def single_stream(n_frames):
    """The reference's REAL call pattern, without Python in the loop: tests/cpp/tracking_main.cpp is a Tracking +
    LocalMapping style loop in C++ on the drop-in classes with the reference's class names and signatures --
    Frame(im, &extractor) (src/Frame.cpp:77), ORBmatcher(0.9, true).SearchByProjection(cur, last, th, true)
    (src/Tracking.cpp:341-352), Optimizer::PoseOptimization(&cur) (:355), Optimizer::LocalBundleAdjustment(kf, &stop, &map)
    per keyframe -- one frame at a time through host pointers.  Run as a child process on the same GPU (the bench's own
    kernels are idle by then); built by __graft_entry__.build(), or here if the binary is missing."""
    import subprocess
    import tempfile
    from orbslam_in_practice_amd import sequence as sq
    import __graft_entry__ as ge
    try:
        exe = ge.build_cpp_loop("tracking_main")
        exe2 = ge.build_cpp_loop("tracker_main")
        td = tempfile.mkdtemp()
        frames, gt, K, _ = sq.make_plane_sequence(5, n_frames)
        frames.tofile(td + "/f.raw")
        gt.astype(np.float64).tofile(td + "/gt.bin")
        # the child is not part of a profiling session of this process (rocprofv3 preloads its tool through the environment)
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
        rec_dt = np.dtype([("pose", "<f8", 7), ("nm", "<i4"), ("inl", "<i4")])

        def run(binary, out, *extra):
            best = None
            for _ in range(4):   # the first run pages the binary and the library in, and an idle GPU takes a few hundred ms to clock up
                p = subprocess.run([binary, str(frames.shape[2]), str(frames.shape[1]), str(n_frames), td + "/f.raw", td + "/gt.bin",
                                    *("%.17g" % v for v in K), out, *extra], capture_output=True, text=True, timeout=300, env=env)
                if p.returncode != 0:
                    raise RuntimeError("%s exited with %d: %s" % (os.path.basename(binary), p.returncode, p.stderr[-200:]))
                j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
                if best is None or j["secs"] < best["secs"]:
                    best = j
            return best, np.frombuffer(open(out, "rb").read(), rec_dt, n_frames, 4)

        j, rec = run(exe, td + "/out.bin")
        track_s = j["secs"] - j["t_first"] - j["t_ba"]
        res = {"value": round((n_frames - 1) / track_s, 1), "unit": "frames/s",
               "what": "tracking alone: Frame(im,&extractor) -> SearchByProjection -> PoseOptimization per frame, C++ on the drop-in classes",
               "with_local_ba_frames_per_s": round((n_frames - 1) / (j["secs"] - j["t_first"]), 1),
               "per_call_ms": {"Frame(im,&extractor)": round(j["ms_extract"], 4), "SearchByProjection": round(j["ms_sbp"], 4),
                               "PoseOptimization": round(j["ms_po"], 4),
                               "LocalBundleAdjustment": round(j["t_ba"] / max(j["ba_runs"], 1) * 1e3, 3)},
               "local_ba_runs": j["ba_runs"], "map_points": j["map_points"],
               "ate_rmse_m": round(sq.ate_rmse(rec["pose"].copy(), gt), 5), "mean_matches": round(float(rec["nm"][1:].mean()), 1),
               "workload": "%d synthetic frames %dx%d, %d features, keyframe every 4 frames, local BA over the covisibility window"
                           % (n_frames, frames.shape[2], frames.shape[1], NFEATURES)}
        # the same three steps per frame as ONE device chain (orb_tracker_track: one round trip instead of three; no local BA)
        j2, rec2 = run(exe2, td + "/out2.bin")
        res["device_chain"] = {"value": round((n_frames - 1) / j2["track_secs"], 1), "unit": "frames/s",
                               "ms_per_frame": round(j2["track_secs"] / (n_frames - 1) * 1e3, 4),
                               "what": "orb_tracker_track per frame (tests/cpp/tracker_main.cpp): extraction, SearchByProjection with the walk "
                                       "on the device and PoseOptimization of the matches chained on one stream, one synchronisation",
                               "ate_rmse_m": round(sq.ate_rmse(rec2["pose"].copy(), gt), 5),
                               "mean_matches": round(float(rec2["nm"][1:].mean()), 1),
                               "mean_inliers": round(float(rec2["inl"][1:].mean()), 1)}
        # ... and with the look-ahead of a recorded sequence: frame k + 1 is uploaded and extracted on a second stream beside
        # the search and the pose problem of frame k (orb_tracker_track_prefetch); same matches, same poses
        j3, rec3 = run(exe2, td + "/out3.bin", "prefetch")
        res["device_chain"]["with_lookahead"] = {
            "value": round((n_frames - 1) / j3["track_secs"], 1), "unit": "frames/s",
            "ms_per_frame": round(j3["track_secs"] / (n_frames - 1) * 1e3, 4),
            "what": "orb_tracker_track_prefetch: the next frame's upload + extraction overlap this frame's search and pose problem",
            "same_results": bool(np.array_equal(rec3["nm"], rec2["nm"]) and np.array_equal(rec3["inl"], rec2["inl"])
                                 and np.array_equal(rec3["pose"], rec2["pose"]))}
        return res
    except Exception as e:   # a missing host compiler must not cost the driver its line
        return {"error": "%s: %s" % (type(e).__name__, e)}


def ba_algorithmic_bytes(w, iterations, lin_solves):
    """bytes one window moves over its whole solve -- linearise/build once per LM iteration;
    Schur complement, back-substitution and update + re-evaluation once per linear solve (LM trial).  No PCG term:
    the reduced system is solved by the in-kernel Cholesky, whose matrix stays in L2.  Schur output = the non-zero
    6x6 blocks of S (co-observing free camera pairs), not the dense Nc^2 bound."""
    Nc, Np, E = len(w["poses"]), len(w["points"]), len(w["obs"])
    vis = np.zeros((Np, Nc), np.float32)
    vis[w["pt_idx"], w["cam_idx"]] = 1.0
    vis[:, w["fixed"] != 0] = 0.0
    nnzb = int(np.count_nonzero(vis.T @ vis))
    build = (E * 32 + Nc * 56 + Np * 24) + (Nc * 288 + Np * 72 + E * 144 + Nc * 48 + Np * 24)
    per_solve = (E * 144 + Np * 72) + nnzb * 288 + (E * 144 + Np * 120 + Nc * 48) + (E * 32 + 2 * (Nc * 56 + Np * 24))
    return iterations * build + lin_solves * per_solve


def ba_bench(dev_index, rank, world, n_windows, nc, npt, reps, barrier, max_over_ranks, with_cpu):
    """second half of BASELINE.json's metric: local-BA windows/s at the north_star target size (100 KFs x 5000 points,
    5+10 LM iterations).  Windows are independent units: rank r solves its own n_windows per batch (weak scaling, no
    collective); 32 distinct windows per rank.  `value` is the steady state of a stream of batches through
    ba_submit_batch / ba_wait (host staging and PCIe of batch k+1 under the kernels of batch k), host side included."""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    n_uniq = min(32, n_windows)
    uniq = [synth.make_ba_window(900 + 64 * rank + i, nc, npt) for i in range(n_uniq)]
    wins = [uniq[i % n_uniq] for i in range(n_windows)]
    E = max(len(w["obs"]) for w in uniq)
    opt = oa.Optimizer(max_cams=nc, max_points=npt, max_obs=E, max_batch=n_windows, device=dev_index)
    opt.solve_batch(wins, (5, 10))  # warm-up at full size: device / pinned buffers and staging vectors reach steady state
    DEPTH = 3                       # batches in flight (the solver has three slots)
    for _ in range(DEPTH):
        opt.submit_batch(wins, (5, 10))
    for _ in range(DEPTH):
        opt.wait()                  # ... for every job slot
    barrier()
    t0 = time.perf_counter()
    pending = 0
    res = None
    for _ in range(reps):
        if pending == DEPTH:
            res = opt.wait()
            pending -= 1
        opt.submit_batch(wins, (5, 10))
        pending += 1
    while pending:
        res = opt.wait()
        pending -= 1
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    streamed = res   # outputs of the timed region's last batch: checked against the oracle below
    # one synchronous call, HIP events around the device work of every optimize() round (team == 0: the batch ran as
    # phase kernels, k_ba_ph_*; otherwise one k_ba_iteration launch per round): the roofline of the BA kernels
    opt.set_profiling(True)
    t1 = time.perf_counter()
    res = opt.solve_batch(wins, (5, 10))
    sync_dt = time.perf_counter() - t1
    iter_ms, launches, team = opt.iteration_ms()
    opt.set_profiling(False)
    by_uniq = [ba_algorithmic_bytes(uniq[i], res[i]["iterations"], res[i]["lin_solves"]) for i in range(n_uniq)]
    total_bytes = sum(by_uniq[i % n_uniq] for i in range(n_windows))
    a = total_bytes / max(launches, 1) / (iter_ms * 1e-3) / 1e9 if iter_ms > 0 else 0.0
    tj, traffic_src = load_traffic()
    traffic = None
    if "ba_iteration" in tj:
        traffic = int(tj["ba_iteration"] * n_windows / float(tj.get("_ba_batch", n_windows)))
    roof = {"bound": "hbm", "kernel": "k_ba_ph_* (the phase kernels of one optimize() round)" if team == 0 else "k_ba_iteration", "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(a / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
            "algorithmic_bytes_per_launch": int(total_bytes / max(launches, 1)), "windows_per_launch": n_windows,
            "launches": launches, "avg_launch_ms": round(iter_ms, 4), "workgroups_per_window": team,
            "lin_solves_per_window": round(float(np.mean([r["lin_solves"] for r in res])), 2)}
    # ONE window in flight: the LocalMapping call pattern (one local BA per new keyframe)
    one = opt.solve_batch([uniq[0]], (5, 10))
    t2 = time.perf_counter()
    nrep1 = 5
    for i in range(nrep1):
        one = opt.solve_batch([uniq[i % n_uniq]], (5, 10))
    single_ms = (time.perf_counter() - t2) / nrep1 * 1e3
    opt.set_profiling(True)
    opt.solve_batch([uniq[0]], (5, 10))
    s_ms, s_launches, s_team = opt.iteration_ms()
    opt.close()
    out = {"value": round(world * n_windows * reps / dt, 2), "unit": "windows/s", "dtype": "f64", "n_gpus": world,
           "scaling": "weak", "roofline": roof,
           "workload": "local BA, %d windows per GPU per batch (%d distinct) of %d KFs x %d points (E~%d), 5+10 LM "
                       "iterations, %d batches through ba_submit_batch/ba_wait (three in flight), host staging + PCIe included"
                       % (n_windows, n_uniq, nc, npt, E, reps),
           "sync_call_windows_per_s": round(n_windows / sync_dt, 2),
           "single_window_ms": round(single_ms, 3), "single_window_windows_per_s": round(1e3 / single_ms, 2),
           "single_window": {"workgroups_per_window": s_team, "avg_iteration_launch_ms": round(s_ms, 4), "launches": s_launches},
           "chi2_ratio": round(float(np.mean([r["chi2_final"] / r["chi2_initial"] for r in res])), 5)}
    if rank == 0:
        # the CPU oracle (dense Cholesky on the Schur complement, g2o semantics) on a bounded sample of the same windows
        # (N > 1: two windows, as the checker only -- the other ranks wait at the final barrier meanwhile)
        from oracle import oracle as orc
        ncpu = min(8 if with_cpu else 2, n_uniq)
        t3 = time.perf_counter()
        refs = [orc.ba_solve(w, rounds=(5, 10)) for w in uniq[:ncpu]]
        cpu_dt = (time.perf_counter() - t3) / ncpu
        if with_cpu:
            out["cpu_baseline"] = {"value": round(1.0 / cpu_dt, 3), "unit": "windows/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the same windows through oracle/ba_oracle.cpp, %.1f s" % (ncpu, cpu_dt * ncpu)}
        # ... and the SAME solves are the checker of the timed stream: `streamed` holds the results of the last batch of
        # the timed region (ba_submit_batch / ba_wait, phase kernels), window i of it = uniq[i]
        rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
        par = {"windows": ncpu, "launch_shape": "phase kernels" if team == 0 else "team of %d" % team,
               "max_rel_points": 0.0, "max_rel_trans": 0.0, "max_abs_quat": 0.0, "max_rel_chi2": 0.0,
               "outliers_equal": True, "iterations_equal": True, "tolerance": 1e-5,
               "checked": "windows 0..%d of the LAST timed batch vs oracle.ba_solve" % (ncpu - 1)}
        for i, r in enumerate(refs):
            g = streamed[i]
            par["max_rel_points"] = max(par["max_rel_points"], rel(g["points"], r["points"]))
            par["max_rel_trans"] = max(par["max_rel_trans"], rel(g["poses"][:, 4:], r["poses"][:, 4:]))
            par["max_abs_quat"] = max(par["max_abs_quat"], float(np.abs(g["poses"][:, :4] - r["poses"][:, :4]).max()))
            par["max_rel_chi2"] = max(par["max_rel_chi2"], abs(g["chi2_final"] - r["chi2_final"]) / r["chi2_final"])
            par["outliers_equal"] &= bool(np.array_equal(g["outlier"], r["outlier"]))
            par["iterations_equal"] &= bool(g["iterations"] == r["iterations"])
        par["ok"] = bool(par["outliers_equal"] and par["iterations_equal"] and
                         max(par["max_rel_points"], par["max_rel_trans"], par["max_abs_quat"], par["max_rel_chi2"]) < 1e-5)
        out["parity"] = par
    return out


def main():
    global W, H, NFEATURES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per launch group")
    ap.add_argument("--batches-per-step", type=int, default=8, help="batches of --batch frames one step processes")
    ap.add_argument("--cpu-frames", type=int, default=200, help="CPU baseline sample size (0 = skip); 200 frames = ~9 s on one core")
    ap.add_argument("--size", default="640x480", help="frame size WxH (configs[1] = 640x480; 1280x720 = configs[3] shape)")
    ap.add_argument("--window", type=int, default=-1, help="match window (-1 = brute force; 100 = the reference's "
                    "SearchForInitialization call, src/Tracking.cpp:189)")
    ap.add_argument("--nfeatures", type=int, default=NFEATURES)
    ap.add_argument("--split", type=int, default=2, help="1..4 launch lanes: equal parts of a batch on separate stream pairs")
    ap.add_argument("--ba-windows", type=int, default=256, help="local-BA windows per GPU per batch (0 = skip)")
    ap.add_argument("--ba-size", default="100x5000", help="keyframes x points of a window (configs[2] = 50x2000)")
    ap.add_argument("--ba-reps", type=int, default=48, help="batches in the timed BA stream (the pipeline fill of three slots weighs 1/16)")
    ap.add_argument("--sequence-frames", type=int, default=40, help="frames of the one-stream-per-GPU tracking loop (0 = skip)")
    args = ap.parse_args()
    W, H = (int(v) for v in args.size.lower().split("x"))
    NFEATURES = args.nfeatures

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    # The C++ one-stream loop is a CHILD process on the same GPU: it runs first, before this process creates its own HIP
    # queues (two processes' queues on one device are time-sliced by the hardware scheduler: 3.5 ms per frame instead of
    # 1.2 when the loop ran after the bench's streams existed)
    single = None
    if args.sequence_frames > 0 and rank == 0 and world == 1:
        single = single_stream(max(args.sequence_frames, 80))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU fallback)")
    # REHEARSAL ONLY (one-GPU box): BENCH_REHEARSE=1 puts every rank on cuda:0 and swaps RCCL for gloo, to
    # exercise the multi-rank control flow; numbers from such a run are meaningless and say so.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        # several PROCESSES on one GPU: a workgroup team (cooperative launch) assumes it has the device to itself
        os.environ["ORBSLAM_AMD_BA_TEAM"] = "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import shard, synth

    B = args.batch
    S = max(1, args.batches_per_step)
    nsplit = max(1, min(args.split, 4))
    # one extractor/matcher pair per launch lane: the lanes of a batch run on their own streams, so the latency-bound
    # stages of one (quadtree, match resolution) overlap the throughput-bound stages of the other (pyramid, FAST, blur)
    exts = [oa.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, max_w=W, max_h=H, max_batch=B, device=local_rank)
            for _ in range(nsplit)]
    ext = exts[0]
    cap = ext.cap
    matchers = [oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B, device=local_rank) for _ in range(nsplit)]
    matcher = matchers[0]

    # S distinct batches per step, all resident in HBM: batch 0 is a synthetic stream of B frames, batch j the same
    # stream translated by (7j, 13j) pixels (circularly) -- different pixels under every FAST cell and pyramid tap
    frames = synth.make_stream(1000 + rank, B, W, H)
    d_base = torch.from_numpy(frames).to(dev)
    d_imgs = [d_base] + [torch.roll(d_base, shifts=(7 * j, 13 * j), dims=(1, 2)).contiguous() for j in range(1, S)]
    rows = shard.halo_rows(world, B)
    pa_h, pb_h = shard.halo_pairs(rank, world, B)  # (k, k+1) in LOCAL slab rows; the last one points into the halo
    pa = torch.from_numpy(pa_h).to(dev)
    pb = torch.from_numpy(pb_h).to(dev)
    bounds = (0.0, float(W), 0.0, float(H))
    exchange = shard.HaloExchange(world, B)

    # Two stream pairs per lane, double-buffered feature slabs: the (latency-bound) match of batch k runs on stream B
    # while stream A already extracts batch k+1.
    class Slab:
        def __init__(self):
            self.kps = torch.zeros((rows, cap, 28), dtype=torch.uint8, device=dev)
            self.desc = torch.zeros((rows, cap, 32), dtype=torch.uint8, device=dev)
            self.cnt = torch.zeros((rows,), dtype=torch.int32, device=dev)
            self.m12 = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
            self.nm = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.extracted = [torch.cuda.Event() for _ in range(nsplit)]
            self.matched = [torch.cuda.Event() for _ in range(nsplit)]
            self.gathered = torch.cuda.Event()

    slabs = [Slab(), Slab()]
    sAs = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    sBs = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    Bh = B // nsplit  # frames per launch
    assert Bh * nsplit == B
    batch_no = [0]

    def one_batch(d_in):
        sl = slabs[batch_no[0] & 1]
        batch_no[0] += 1
        for hh in range(nsplit):
            sA = sAs[hh]
            with torch.cuda.stream(sA):
                for ev in sl.matched:
                    sA.wait_event(ev)  # the matches that last read this slab have finished
                f0 = hh * Bh
                exts[hh].extract_batch_device(d_in.data_ptr() + f0 * W * H, Bh, W, H, W, W * H,
                                              sl.kps.data_ptr() + f0 * cap * 28, sl.desc.data_ptr() + f0 * cap * 32, cap,
                                              sl.cnt.data_ptr() + f0 * 4, C.c_void_p(sA.cuda_stream))
                sl.extracted[hh].record(sA)
        for hh in range(nsplit):
            sB = sBs[hh]
            with torch.cuda.stream(sB):
                for ev in sl.extracted:
                    sB.wait_event(ev)  # pair (k, k+1) may straddle the lanes
                if world > 1:
                    # the one exchange step of the path: the first frame slab of every rank over xGMI (RCCL
                    # all-gather, 3 x world x ~20 KB), on lane 0's MATCH stream, so the next batch's extraction
                    # (streams A) overlaps it; the other lanes' matches wait for its event.
                    if hh == 0:
                        exchange(sl.kps, sl.desc, sl.cnt)
                        sl.gathered.record(sB)
                    else:
                        sB.wait_event(sl.gathered)
                p0 = hh * Bh
                matchers[hh].match_batch_device(sl.kps.data_ptr(), sl.desc.data_ptr(), sl.cnt.data_ptr(), cap,
                                                pa.data_ptr() + p0 * 4, pb.data_ptr() + p0 * 4, Bh, bounds, args.window,
                                                sl.m12.data_ptr() + p0 * cap * 4, sl.nm.data_ptr() + p0 * 4,
                                                C.c_void_p(sB.cuda_stream))
                sl.matched[hh].record(sB)

    def step():
        for j in range(S):
            one_batch(d_imgs[j])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    L = oa._lib.lib()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # outputs of the timed region's LAST batch (batch S-1 of the step, in the slab it went to): a few frames at the start
    # of the batch and on both sides of the lane boundary are copied out now and checked against the oracle below
    last_slab, last_j = slabs[(batch_no[0] - 1) & 1], S - 1
    chk = sorted(set(i for i in list(range(3)) + list(range(Bh - 1, Bh + 2)) if i < B))
    snap = {i: (int(last_slab.cnt[i]), last_slab.kps[i].cpu().numpy(), last_slab.desc[i].cpu().numpy(),
                last_slab.m12[i].cpu().numpy(), int(last_slab.nm[i])) for i in chk} if rank == 0 else {}

    # per-stage durations: a SEPARATE pass after the timed region, HIP events on the launch streams between the stages
    oa._lib.check(L.orb_set_profiling(ext._h, 1))
    oa._lib.check(L.orb_matcher_set_profiling(matcher._m, 1))
    for _ in range(3):
        step()
    barrier()
    ms5 = (C.c_float * 5)()
    oa._lib.check(L.orb_get_stage_ms(ext._h, ms5))
    mm = (C.c_float * 2)()
    oa._lib.check(L.orb_matcher_get_stage_ms(matcher._m, mm))
    oa._lib.check(L.orb_set_profiling(ext._h, 0))
    oa._lib.check(L.orb_matcher_set_profiling(matcher._m, 0))
    stage_ms = dict(zip(["pyramid", "fast", "quadtree", "blur", "describe"], [float(v) for v in ms5]))
    stage_ms["match_knn"], stage_ms["match_resolve"] = float(mm[0]), float(mm[1])
    counts = slabs[0].cnt[:B].cpu().numpy()
    nmatch = slabs[0].nm.cpu().numpy()
    n_kp = float(counts.mean())

    local_ba = None
    if args.ba_windows > 0:
        nc, npt = (int(v) for v in args.ba_size.lower().split("x"))
        local_ba = ba_bench(local_rank, rank, world, args.ba_windows, nc, npt, max(2, args.ba_reps), barrier, max_over_ranks,
                            with_cpu=(rank == 0 and world == 1))

    tl = None
    if args.sequence_frames > 0:
        tl = tracking_loop(args.sequence_frames, local_rank, rank, world, barrier, max_over_ranks)
    rc = 0
    if rank == 0:
        fps = world * B * S * args.steps / elapsed
        ab = algorithmic_bytes_per_frame(ext.level_sizes(W, H), n_kp)
        tj, traffic_src = load_traffic()
        # The dominant KERNEL.  The two launch lanes keep the GPU's vector ALUs ~95 % busy (profiles/r02_pmc_sq_*.json), so
        # the kernel that bounds the pipeline is the one that issues the most VALU instructions per batch (PMC
        # SQ_INSTS_VALU, profiles/traffic.json), not the one whose in-flight HIP-event interval happens to be longest
        # -- the latency-bound single-wave stages (quadtree, match resolution) stretch to ~0.8 ms beside the others
        # while occupying a few percent of the machine.  Without the PMC file: the longest single-launch stage.
        # ("pyramid" is several launches -- level-0 copy + resizes -- and is listed under roofline_stages only)
        timed = {k: v for k, v in stage_ms.items() if ab.get(k, 0) > 0 and k in SINGLE_KERNEL_STAGES}
        by_valu = {k: tj.get("valu_insts_" + k, tj.get("_stale_valu", {}).get(k)) for k in timed}
        by_valu = {k: v for k, v in by_valu.items() if v is not None}
        dom = max(by_valu, key=by_valu.get) if by_valu else max(timed, key=timed.get)

        def stage_roofline(k):
            # one launch covers Bh frames; PMC traffic was collected per launch at tj['_batch'] frames and the
            # per-frame traffic is batch independent
            a = ab[k] * Bh / (stage_ms[k] * 1e-3) / 1e9
            t = int(tj[k] * Bh / float(tj.get("_batch", Bh))) if k in tj else None
            r = {"bound": "hbm", "kernel": STAGE_KERNEL[k], "stage": k, "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(a / HBM_PEAK_GBS, 5), "traffic": t, "traffic_source": traffic_src if t is not None else None,
                 "algorithmic_bytes_per_launch": int(ab[k] * Bh),
                 "frames_per_launch": Bh, "avg_launch_ms": round(stage_ms[k], 4)}
            vk = "valu_insts_" + k  # SQ_INSTS_VALU per launch (PMC pass, profiles/traffic.json): issue-rate view of a VALU-bound kernel
            if vk in tj:
                ginst = tj[vk] * Bh / float(tj.get("_batch", Bh)) / 1e9
                r["valu"] = {"source": traffic_src, "insts_per_launch": int(ginst * 1e9), "achieved_ginst_s": round(ginst / (stage_ms[k] * 1e-3), 1),
                             "peak_ginst_s": VALU_PEAK_GINST, "frac": round(ginst / (stage_ms[k] * 1e-3) / VALU_PEAK_GINST, 4)}
            if k == "match_knn":
                # the all-pairs Hamming product on the matrix cores (v_mfma_i32_32x32x32_i8): 2 x 256 integer ops per
                # (query, F2 keypoint) pair of 32-row tiles; priced against the dense i8 peak (2 x the bf16 peak)
                tops = 2.0 * 256.0 * (32.0 * np.ceil(n_kp / 32.0)) ** 2 * Bh / (stage_ms[k] * 1e-3) / 1e12
                r["mfma"] = {"dtype": "i8", "achieved": round(tops, 2), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                             "frac": round(tops / MFMA_I8_PEAK_TOPS, 5),
                             "note": "in-flight interval beside the other lane's kernels; the kernel is bound by looking at the "
                                     "results (3 VALU instructions per 64 distances), not by the products"}
            return r
        pipe_bytes = sum(ab.values()) * world * B * S * args.steps
        pipe = pipe_bytes / elapsed / 1e9
        out = {
            "metric": "frames/s ORB extract+match @%dx%d (local-BA windows/s under local_ba)" % (W, H),
            "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "rehearsal": rehearse, "timed_region_s": round(elapsed, 4),
            "config": {"workload": "%s: ORB extract (8 levels, %d features, FAST 20/7) + %s match of consecutive "
                                   "frames, %dx%d stream" % ("configs[1]" if (W, H, args.window) == (640, 480, -1) else
                                                             "variant", NFEATURES,
                                                             "brute-force Hamming" if args.window < 0 else
                                                             "SearchForInitialization(window=%d)" % args.window, W, H),
                       "frames_per_gpu_per_step": B * S, "pairs_per_gpu_per_step": B * S, "frames_per_batch": B,
                       "batches_per_step": S, "launch_lanes": nsplit,
                       "keypoints_per_frame": round(n_kp, 1), "matches_per_pair": round(float(nmatch.mean()), 1),
                       "exchange": ("rccl all_gather of the first frame slab of every rank (%d B per rank per batch)"
                                    % exchange.bytes_per_rank(cap)) if world > 1 else "none"},
            "roofline": stage_roofline(dom),
            "roofline_stages": [stage_roofline(k) for k in stage_ms if ab.get(k, 0) > 0],
            "roofline_pipeline": {"bound": "hbm", "achieved": round(pipe, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                  "frac": round(pipe / (HBM_PEAK_GBS * world), 5),
                                  "algorithmic_bytes_per_frame": int(sum(ab.values()))},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(args.cpu_frames)
        out["parity"] = frontend_parity(frames, last_j, snap, chk, B, args.window)
        if local_ba is not None:
            out["local_ba"] = local_ba
        if tl is not None:
            out["tracking_loop"] = tl
        if single is not None:
            out["single_stream"] = single
        print(json.dumps(out), flush=True)
        bad = [k for k, blk in (("parity", out["parity"]), ("local_ba.parity", (local_ba or {}).get("parity"))) if blk and not blk["ok"]]
        if bad:
            sys.stderr.write("bench.py: outputs of the timed region differ from the oracle: %s\n" % ", ".join(bad))
            rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
