#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X ORB front-end (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch: ORB extraction (8-level pyramid, FAST cells,
quadtree, blur, rBRIEF) of B synthetic 640x480 frames per GPU + brute-force Hamming matching of the
B consecutive frame pairs (k, k+1).  Inputs are resident in HBM before the timed region starts;
outputs stay in HBM.  With N > 1 GPUs every rank processes its own B frames (weak scaling) and the
per-frame feature slabs are exchanged with ONE round of RCCL all-gathers so that the last local frame
is matched against the first frame of the next rank.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see the driver contract) with `roofline` (dominant stage, live HIP-event
timing) and, at N == 1, `cpu_baseline` (the CPU oracle -- a port of the reference path -- timed on a
bounded sample on this box's host cores).
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


def ba_throughput(device, n_windows):
    """second half of BASELINE.json's metric: local-BA windows/s at the north_star target size
    (100 KFs x 5000 points, 5+10 LM iterations), a batch of independent windows on this GPU"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    uniq = [synth.make_ba_window(900 + i, 100, 5000) for i in range(4)]
    wins = [uniq[i % 4] for i in range(n_windows)]
    E = max(len(w["obs"]) for w in uniq)
    opt = oa.Optimizer(max_cams=100, max_points=5000, max_obs=E, max_batch=n_windows, device=device)
    opt.solve_batch(wins, (5, 10))  # warm-up at full size: device / pinned buffers and staging vectors reach steady state
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        res = opt.solve_batch(wins, (5, 10))
    dt = (time.perf_counter() - t0) / reps
    opt.close()
    # the CPU oracle (dense Cholesky on the Schur complement, g2o semantics) on a bounded sample of the same windows
    from oracle import oracle as orc
    t1 = time.perf_counter()
    for w in uniq:
        orc.ba_solve(w, rounds=(5, 10))
    cpu_dt = (time.perf_counter() - t1) / len(uniq)
    return {"value": round(n_windows / dt, 2), "unit": "windows/s", "dtype": "f64",
            "cpu_baseline": {"value": round(1.0 / cpu_dt, 3), "unit": "windows/s", "cores": 1, "kind": "port",
                             "sample": "%d of the same windows through oracle/ba_oracle.cpp, %.1f s" % (len(uniq), cpu_dt * len(uniq))},
            "workload": "local BA, %d windows of 100 KFs x 5000 points (E~%d), 5+10 LM iterations, host staging "
                        "included, mean of %d calls" % (n_windows, E, reps),
            "chi2_ratio": round(float(np.mean([r["chi2_final"] / r["chi2_initial"] for r in res])), 5)}


def main():
    global W, H, NFEATURES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=200, help="CPU baseline sample size (0 = skip); 200 frames = ~9 s on one core")
    ap.add_argument("--size", default="640x480", help="frame size WxH (configs[1] = 640x480; 1280x720 = configs[3] shape)")
    ap.add_argument("--window", type=int, default=-1, help="match window (-1 = brute force; 100 = the reference's "
                    "SearchForInitialization call, src/Tracking.cpp:189)")
    ap.add_argument("--nfeatures", type=int, default=NFEATURES)
    ap.add_argument("--split", type=int, default=2, help="1..4 launch lanes: equal parts of a batch on separate stream pairs")
    ap.add_argument("--ba-windows", type=int, default=256, help="windows in the local-BA side measurement (0 = skip)")
    args = ap.parse_args()
    W, H = (int(v) for v in args.size.lower().split("x"))
    NFEATURES = args.nfeatures

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU fallback)")
    # REHEARSAL ONLY (one-GPU box): BENCH_REHEARSE=1 puts every rank on cuda:0 and swaps RCCL for gloo, to
    # exercise the multi-rank control flow; numbers from such a run are meaningless and say so.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def all_gather_flat(out, inp):
        if rehearse:  # gloo: list form
            parts = list(out.view(world, -1).unbind(0))
            dist.all_gather(parts, inp.view(-1))
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.view(-1))

    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth

    B = args.batch
    ext = oa.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, max_w=W, max_h=H, max_batch=B,
                          device=local_rank)
    cap = ext.cap
    matcher = oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B, device=local_rank)
    # a second extractor/matcher pair lets the two HALVES of a step's batch run on their own streams: the
    # latency-bound stages of one half (quadtree, match resolution) then overlap the throughput-bound stages of
    # the other (pyramid, FAST, blur)
    nsplit = max(1, min(args.split, 4))
    more_ext = [oa.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, max_w=W, max_h=H, max_batch=B, device=local_rank)
                for _ in range(nsplit - 1)]
    more_match = [oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B, device=local_rank) for _ in range(nsplit - 1)]

    frames = synth.make_stream(1000 + rank, B, W, H)
    d_imgs = torch.from_numpy(frames).to(dev)
    nG = world * B  # frames visible to the matcher after the exchange
    lo = rank * B
    from orbslam_in_practice_amd import shard
    pa_h, pb_h = shard.stream_pairs(rank, world, B)  # (k, k+1) in the global stream; the last one wraps
    pa = torch.from_numpy(pa_h).to(dev)
    pb = torch.from_numpy(pb_h).to(dev)
    bounds = (0.0, float(W), 0.0, float(H))

    # Two HIP streams, double-buffered feature slabs: the (latency-bound) match of batch k runs on stream B
    # while stream A already extracts batch k+1.
    class Slab:
        def __init__(self):
            self.kps = torch.zeros((nG, cap, 28), dtype=torch.uint8, device=dev)
            self.desc = torch.zeros((nG, cap, 32), dtype=torch.uint8, device=dev)
            self.cnt = torch.zeros((nG,), dtype=torch.int32, device=dev)
            if world > 1:  # separate send buffers: the all-gather never aliases its own output
                self.k_loc = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
                self.d_loc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
                self.c_loc = torch.zeros((B,), dtype=torch.int32, device=dev)
            else:
                self.k_loc, self.d_loc, self.c_loc = self.kps, self.desc, self.cnt
            self.m12 = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
            self.nm = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.extracted = torch.cuda.Event()
            self.matched = torch.cuda.Event()
            self.gathered = torch.cuda.Event()

    slabs = [Slab(), Slab()]
    exts = [ext] + more_ext
    matchers = [matcher] + more_match
    sAs = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    sBs = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    for sl in slabs:
        sl.extracted = [torch.cuda.Event() for _ in range(nsplit)]
        sl.matched = [torch.cuda.Event() for _ in range(nsplit)]
    Bh = B // nsplit  # frames per launch
    assert Bh * nsplit == B
    step_no = [0]

    def step():
        sl = slabs[step_no[0] & 1]
        step_no[0] += 1
        for hh in range(nsplit):
            sA = sAs[hh]
            with torch.cuda.stream(sA):
                for ev in sl.matched:
                    sA.wait_event(ev)  # the matches that last read this slab have finished
                f0 = hh * Bh
                exts[hh].extract_batch_device(d_imgs.data_ptr() + f0 * W * H, Bh, W, H, W, W * H,
                                              sl.k_loc.data_ptr() + f0 * cap * 28, sl.d_loc.data_ptr() + f0 * cap * 32, cap,
                                              sl.c_loc.data_ptr() + f0 * 4, C.c_void_p(sA.cuda_stream))
                sl.extracted[hh].record(sA)
        for hh in range(nsplit):
            sB = sBs[hh]
            with torch.cuda.stream(sB):
                for ev in sl.extracted:
                    sB.wait_event(ev)  # pair (k, k+1) may straddle the halves
                if world > 1:
                    # the one exchange step of the path: fixed-capacity feature slabs over xGMI (RCCL all-gather),
                    # once per step for the rank's whole block.  It lives on lane 0's MATCH stream, so the next
                    # batch's extraction (streams A) overlaps it; the other lane's match waits for its event.
                    if hh == 0:
                        all_gather_flat(sl.kps, sl.k_loc)
                        all_gather_flat(sl.desc, sl.d_loc)
                        all_gather_flat(sl.cnt, sl.c_loc)
                        sl.gathered.record(sB)
                    else:
                        sB.wait_event(sl.gathered)
                p0 = hh * Bh
                matchers[hh].match_batch_device(sl.kps.data_ptr(), sl.desc.data_ptr(), sl.cnt.data_ptr(), cap,
                                                pa.data_ptr() + p0 * 4, pb.data_ptr() + p0 * 4, Bh, bounds, args.window,
                                                sl.m12.data_ptr() + p0 * cap * 4, sl.nm.data_ptr() + p0 * 4,
                                                C.c_void_p(sB.cuda_stream))
                sl.matched[hh].record(sB)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    L = oa._lib.lib()
    if True:
        for _ in range(args.warmup):
            step()
        barrier()
        oa._lib.check(L.orb_set_profiling(ext._h, 1))
        oa._lib.check(L.orb_matcher_set_profiling(matcher._m, 1))
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # live per-stage durations (HIP events on the launch stream, averaged over the timed steps)
    ms5 = (C.c_float * 5)()
    oa._lib.check(L.orb_get_stage_ms(ext._h, ms5))
    mm = (C.c_float * 2)()
    oa._lib.check(L.orb_matcher_get_stage_ms(matcher._m, mm))
    stage_ms = dict(zip(["pyramid", "fast", "quadtree", "blur", "describe"], [float(v) for v in ms5]))
    stage_ms["match_knn"], stage_ms["match_resolve"] = float(mm[0]), float(mm[1])
    counts = slabs[0].c_loc.cpu().numpy()
    nmatch = slabs[0].nm.cpu().numpy()
    n_kp = float(counts.mean())

    if rank == 0:
        fps = world * B * args.steps / elapsed
        ab = algorithmic_bytes_per_frame(ext.level_sizes(W, H), n_kp)
        # the dominant KERNEL: the single-launch stage with the longest average duration in the timed region
        # ("pyramid" is eight launches -- level-0 copy + seven resizes -- and is listed under roofline_stages only)
        timed = {k: v for k, v in stage_ms.items() if ab.get(k, 0) > 0 and k in SINGLE_KERNEL_STAGES}
        dom = max(timed, key=timed.get)
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
            except Exception:
                tj = {}

        def stage_roofline(k):
            # one launch covers Bh frames; PMC traffic was collected per launch at tj['_batch'] frames and the
            # per-frame traffic is batch independent
            a = ab[k] * Bh / (stage_ms[k] * 1e-3) / 1e9
            t = int(tj[k] * Bh / float(tj.get("_batch", Bh))) if k in tj else None
            return {"bound": "hbm", "kernel": k, "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBS, 5), "traffic": t, "algorithmic_bytes_per_launch": int(ab[k] * Bh),
                    "frames_per_launch": Bh, "avg_launch_ms": round(stage_ms[k], 4)}
        pipe_bytes = sum(ab.values()) * world * B * args.steps
        pipe = pipe_bytes / elapsed / 1e9
        out = {
            "metric": "frames/s ORB extract+match @%dx%d" % (W, H), "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "rehearsal": rehearse,
            "config": {"workload": "%s: ORB extract (8 levels, %d features, FAST 20/7) + %s match of consecutive "
                                   "frames, %dx%d stream" % ("configs[1]" if (W, H, args.window) == (640, 480, -1) else
                                                             "variant", NFEATURES,
                                                             "brute-force Hamming" if args.window < 0 else
                                                             "SearchForInitialization(window=%d)" % args.window, W, H),
                       "frames_per_gpu_per_step": B, "pairs_per_gpu_per_step": B, "launch_lanes": nsplit,
                       "keypoints_per_frame": round(n_kp, 1), "matches_per_pair": round(float(nmatch.mean()), 1),
                       "exchange": "rccl all_gather of kp/desc/count slabs" if world > 1 else "none"},
            "roofline": stage_roofline(dom),
            "roofline_stages": [stage_roofline(k) for k in stage_ms if ab.get(k, 0) > 0],
            "roofline_pipeline": {"bound": "hbm", "achieved": round(pipe, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                  "frac": round(pipe / (HBM_PEAK_GBS * world), 5),
                                  "algorithmic_bytes_per_frame": int(sum(ab.values()))},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(args.cpu_frames)
        if world == 1 and args.ba_windows > 0:
            out["local_ba"] = ba_throughput(local_rank, args.ba_windows)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
