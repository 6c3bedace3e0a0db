"""GPU: one frame of Tracking::TrackWithMotionModel (reference src/Tracking.cpp:330-372) as ONE device chain -- the
OrbTracker entry points (csrc/orb_track.hip): image up, extraction (features stay in HBM as the next frame's "last
frame"), SearchByProjection with the walk and the rotation histogram on the device, the pose problem built from the matches
on the device, everything back in one copy, one synchronisation per frame.  Same kernels and arithmetic as the three
separate calls, so: identical keypoints, descriptors and match indices, and the same trajectory as the CPU oracle's run of
the loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tracker_chain_equals_the_three_calls_frame_by_frame(oracle):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import sequence as sq
    frames, gt, K, _ = sq.make_plane_sequence(5, 6)
    be = sq.TrackerBackend(1000, 640, 480)
    ref = sq.GpuBackend(1000, 640, 480)
    r0 = be.track(frames[0], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[0], K, 15.0)
    k0, d0 = ref.extract(frames[0])
    assert np.array_equal(r0["kps"], k0) and np.array_equal(r0["desc"], d0) and r0["n_matches"] == 0
    X0 = sq.backproject_to_plane(gt[0], K, np.stack([k0["x"], k0["y"]], -1).astype(np.float64))
    rng = np.random.default_rng(1)
    flags = np.where(rng.random(len(k0)) < 0.85, 3, 0).astype(np.uint8)
    flags[rng.random(len(k0)) < 0.1] = 1                       # some points without observations: they do not take the feature
    pose_pred = gt[1].copy()                                   # (a good prediction: hundreds of matches)
    r1 = be.track(frames[1], flags, X0, pose_pred, K, 15.0)
    k1, d1 = ref.extract(frames[1])
    assert np.array_equal(r1["kps"], k1) and np.array_equal(r1["desc"], d1)
    F = oa.matcher.FrameFeatures
    R, t = sq.quat_to_R(pose_pred[:4]), pose_pred[4:]
    nm, m, _ = ref.matcher.SearchByProjection(F(k1, d1, ref.bounds), F(k0, d0, ref.bounds), flags, X0, R, t, K[0], K[1], K[2], K[3],
                                              ref.scale_factors, np.zeros(len(k1), np.uint8), 15.0)
    assert r1["n_matches"] == nm and nm > 100 and np.array_equal(r1["match"], m)
    idx = np.where(m >= 0)[0]
    obs = np.stack([k1["x"][idx], k1["y"][idx]], -1).astype(np.float64)
    pose, outl, ninl = ref.pose_optimization(pose_pred, X0[m[idx]], obs, ref.inv_sigma2[k1["octave"][idx]], K)
    assert r1["n_inliers"] == ninl and np.array_equal(r1["outlier"][idx], outl) and not r1["outlier"][m < 0].any()
    assert np.abs(r1["pose"] - pose).max() < 1e-12             # same kernel, same correspondences in the same order
    # ... and against the oracle
    on, om, _ = oracle.search_by_projection(k1, d1, k0, d0, ref.bounds, flags, X0, R, t, K[0], K[1], K[2], K[3], ref.scale_factors,
                                            np.zeros(len(k1), np.uint8), 15.0, True)
    assert on == nm and np.array_equal(om, m)
    op, oo, oi = oracle.pose_optimization(pose_pred, X0[m[idx]], obs, ref.inv_sigma2[k1["octave"][idx]], K[0], K[1], K[2], K[3])
    assert oi == ninl and np.array_equal(oo, outl) and np.abs(op - r1["pose"]).max() < 1e-5 * max(np.abs(op).max(), 1.0)
    be.close()
    ref.close()


def test_tracking_loop_on_the_device_chain_equals_oracle(oracle):
    from orbslam_in_practice_amd import sequence as sq
    from tests.sequence_oracle import OracleBackend
    frames, gt, K, _ = sq.make_plane_sequence(5, 17)
    be = sq.TrackerBackend(1000, 640, 480)
    got = sq.run_sequence(be, frames, K, gt[0])
    be.close()
    ref = sq.run_sequence(OracleBackend(1000, 640, 480), frames, K, gt[0])
    assert got["ba_runs"] == ref["ba_runs"] >= 2 and got["n_points"] == ref["n_points"]
    assert np.array_equal(got["n_matches"], ref["n_matches"]) and got["n_matches"][1:].min() >= 50
    assert np.array_equal(got["n_inliers"], ref["n_inliers"])
    for k in range(len(frames)):
        assert np.array_equal(got["matches"][k], ref["matches"][k]), "frame %d" % k
    assert np.abs(got["poses"] - ref["poses"]).max() < 2e-5
    assert sq.ate_rmse(got["poses"], gt) < 0.03


def test_tracker_loop_in_cpp_through_the_c_abi(tmp_path):
    """tests/cpp/tracker_main.cpp: the per-frame chain through orb_tracker_track in plain C++ on the C-ABI (no cv:: types, no
    HIP in the caller) -- tracks the synthetic sequence to centimetres, with the match counts of the three-call loop"""
    import os
    import subprocess
    from orbslam_in_practice_amd import sequence as sq
    from tests.test_gpu_dropin import build_dropin
    exe = build_dropin(str(tmp_path), extra=["-O2", "-Wall", "-Wextra", "-Werror"], program="tracker_main")
    n = 33
    frames, gt, K, _ = sq.make_plane_sequence(5, n)
    frames.tofile(str(tmp_path / "frames.raw"))
    gt.astype(np.float64).tofile(str(tmp_path / "gt.bin"))
    out = subprocess.run([exe, "640", "480", str(n), str(tmp_path / "frames.raw"), str(tmp_path / "gt.bin"),
                          *("%.17g" % v for v in K), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(str(tmp_path / "out.bin"), "rb").read()
    assert np.frombuffer(raw, np.int32, 1)[0] == n
    rec = np.frombuffer(raw, np.dtype([("pose", "<f8", 7), ("nm", "<i4"), ("inl", "<i4")]), n, 4)
    assert rec["nm"][1:].min() >= 50 and (rec["inl"][1:] >= 0.8 * rec["nm"][1:]).all()
    assert sq.ate_rmse(rec["pose"].copy(), gt) < 0.03
    # frames 1..3 (before the first keyframe adds points) see exactly what the Python loop on the three separate calls sees
    gpu = sq.GpuBackend(1000, 640, 480)
    ref = sq.run_sequence(gpu, frames[:4], K, gt[0])
    gpu.close()
    assert np.array_equal(rec["nm"][:4], ref["n_matches"]) and np.array_equal(rec["inl"][:4], ref["n_inliers"])
    assert np.abs(rec["pose"][:4] - ref["poses"]).max() < 1e-7
    import json
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert (n - 1) / j["track_secs"] > 150, j      # a floor for one cold run (1300-1370 frames/s warm: bench.py single_stream.device_chain)
    # the look-ahead variant (frame k + 1 extracted beside the chain of frame k) returns the same record, byte for byte
    out2 = subprocess.run([exe, "640", "480", str(n), str(tmp_path / "frames.raw"), str(tmp_path / "gt.bin"),
                           *("%.17g" % v for v in K), str(tmp_path / "out_pf.bin"), "prefetch"], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    assert open(str(tmp_path / "out_pf.bin"), "rb").read()[:-8] == raw[:-8]      # (the file ends with the run's wall time)


def test_tracker_lookahead_python_and_dropped_prefetch():
    """Tracker.track(next_gray=...): identical results to the plain calls; a look-ahead that is not followed by its frame
    (another buffer comes next, or a reset) is dropped without a trace"""
    from orbslam_in_practice_amd import sequence as sq
    frames, gt, K, _ = sq.make_plane_sequence(5, 5)
    frames = [np.ascontiguousarray(f) for f in frames]
    be = sq.TrackerBackend(1000, 640, 480)
    r0 = be.track(frames[0], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[0], K, 15.0)
    n0 = len(r0["kps"])
    fl = np.ones(n0, np.uint8) * 3
    xw = np.random.default_rng(5).normal(size=(n0, 3)) + [0, 0, 4.0]
    plain = be.track(frames[1], fl, xw, gt[1], K, 15.0)
    be.reset()
    be.track(frames[0], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[0], K, 15.0, next_gray=frames[1])
    ahead = be.track(frames[1], fl, xw, gt[1], K, 15.0, next_gray=frames[3])          # uses the look-ahead; fetches frame 3 ahead
    for k in ("kps", "desc", "match", "outlier", "pose"):
        assert np.array_equal(plain[k], ahead[k]), k
    assert plain["n_matches"] == ahead["n_matches"] and plain["n_inliers"] == ahead["n_inliers"]
    n1 = len(ahead["kps"])
    r2a = be.track(frames[2], np.zeros(n1, np.uint8), np.zeros((n1, 3)), gt[2], K, 15.0)   # not frame 3: the look-ahead is dropped
    be.reset()
    be.track(frames[1], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[1], K, 15.0, next_gray=frames[4])
    be.reset()                                                                            # a reset drops it too
    be.track(frames[1], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[1], K, 15.0)
    r2b = be.track(frames[2], np.zeros(n1, np.uint8), np.zeros((n1, 3)), gt[2], K, 15.0)
    assert np.array_equal(r2a["kps"], r2b["kps"]) and np.array_equal(r2a["desc"], r2b["desc"])
    be.close()


def test_tracker_edge_cases():
    """no map points on the previous frame (nothing to match: the pose comes back untouched, as PoseOptimization does with
    fewer than three correspondences), a wrong n_prev (refused), reset (the next frame is a first frame again), and a
    second tracker on the same device beside the first"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import sequence as sq
    frames, gt, K, _ = sq.make_plane_sequence(5, 4)
    be = sq.TrackerBackend(1000, 640, 480)
    r0 = be.track(frames[0], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[0], K, 15.0)
    n0 = len(r0["kps"])
    assert n0 > 900 and (r0["match"] == -1).all() and not r0["outlier"].any() and np.array_equal(r0["pose"], gt[0])
    r1 = be.track(frames[1], np.zeros(n0, np.uint8), np.zeros((n0, 3)), gt[1], K, 15.0)          # no feature holds a map point
    assert r1["n_matches"] == 0 and r1["n_inliers"] == 0 and (r1["match"] == -1).all() and not r1["outlier"].any()
    q = gt[1] * (1 if gt[1][0] >= 0 else -1)
    assert np.abs(r1["pose"] - q).max() < 1e-15                                                   # (normalised, otherwise untouched)
    with pytest.raises(oa.OrbError) as ei:
        be.track(frames[2], np.zeros(n0 + 5, np.uint8), np.zeros((n0 + 5, 3)), gt[2], K, 15.0)    # not the previous frame's count
    assert ei.value.code == oa._lib.ORB_E_INVALID
    be.tracker.reset()
    r2 = be.track(frames[2], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[2], K, 15.0)
    assert r2["n_matches"] == 0 and len(r2["kps"]) > 900
    # the chain continues from the reset frame: same result as a fresh tracker fed the same two frames
    X2 = sq.backproject_to_plane(gt[2], K, np.stack([r2["kps"]["x"], r2["kps"]["y"]], -1).astype(np.float64))
    fl = np.full(len(r2["kps"]), 3, np.uint8)
    r3 = be.track(frames[3], fl, X2, gt[3], K, 15.0)
    other = sq.TrackerBackend(1000, 640, 480)
    other.track(frames[2], np.zeros(0, np.uint8), np.zeros((0, 3)), gt[2], K, 15.0)
    s3 = other.track(frames[3], fl, X2, gt[3], K, 15.0)
    assert r3["n_matches"] == s3["n_matches"] > 100 and np.array_equal(r3["match"], s3["match"]) and np.array_equal(r3["pose"], s3["pose"])
    assert np.array_equal(r3["outlier"], s3["outlier"]) and r3["n_inliers"] == s3["n_inliers"]
    other.close()
    be.close()
