#!/usr/bin/env python3
"""End-to-end rate of ONE stream through the hot path's calls chained the way Tracking / LocalMapping chain them
(orbslam_in_practice_amd/sequence.py): frames/s and absolute trajectory error of the HIP path (the same loop on the CPU
oracle -- 30 frames/s, identical matches and trajectory -- is tests/test_gpu_sequence.py's business).
    python tools/tracking_loop.py [n_frames]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam_in_practice_amd import sequence as sq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
frames, gt, K, _ = sq.make_plane_sequence(5, n)
out = {"workload": "%d frames 640x480, 1000 features, plane scene, keyframe every 4 frames, local BA over 5 keyframes" % n}
for name, mk in (("gpu", sq.GpuBackend),):
    be = mk(1000, 640, 480)
    sq.run_sequence(be, frames[:6], K, gt[0])  # warm-up
    t0 = time.perf_counter()
    r = sq.run_sequence(be, frames, K, gt[0])
    dt = time.perf_counter() - t0
    be.close()
    out[name] = {"frames_per_s": round(n / dt, 1), "ate_rmse_m": round(sq.ate_rmse(r["poses"], gt), 5), "local_ba_runs": r["ba_runs"],
                 "map_points": r["n_points"], "mean_matches": round(float(r["n_matches"][1:].mean()), 1)}
print(json.dumps(out))
