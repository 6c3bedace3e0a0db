#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer API (what the reference's Frame constructor would call):
single-frame latency of orb_extract and throughput of orb_extract_batch, 640x480."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

frames = synth.make_stream(4, 64, 640, 480)
e1 = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480, max_batch=1)
for f in frames[:3]:
    e1(f)
t0 = time.perf_counter()
for f in frames[:32]:
    e1(f)
lat = (time.perf_counter() - t0) / 32
eb = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480, max_batch=64)
eb.extract_batch(frames)
t0 = time.perf_counter()
for _ in range(3):
    eb.extract_batch(frames)
thr = 3 * 64 / (time.perf_counter() - t0)
m = oa.ORBmatcher(0.9, True, max_keypoints=2048)
k1, d1 = e1(frames[0])
k2, d2 = e1(frames[1])
F1, F2 = oa.FrameFeatures(k1, d1, (0, 640, 0, 480)), oa.FrameFeatures(k2, d2, (0, 640, 0, 480))
prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
m.SearchForInitialization(F1, F2, prev.copy(), 100)
t0 = time.perf_counter()
for _ in range(20):
    m.SearchForInitialization(F1, F2, prev.copy(), 100)
mlat = (time.perf_counter() - t0) / 20
print(json.dumps({"orb_extract_single_frame_ms": round(lat * 1e3, 3), "orb_extract_single_fps": round(1 / lat, 1),
                  "orb_extract_batch64_host_fps": round(thr, 1), "orb_match_init_window100_ms": round(mlat * 1e3, 3)}))

# pose-only optimisation the way Tracking calls it (one frame, a few hundred map-point correspondences)
w = synth.make_ba_window(11, 6, 1000, outlier_frac=0.1)
sel = w["cam_idx"] == 3
xw, obs, info, pose0 = w["gt_points"][w["pt_idx"][sel]], w["obs"][sel], w["inv_sigma2"][sel], w["poses"][3]
opt = oa.Optimizer(max_cams=8, max_points=8, max_obs=8, max_batch=1)
opt.PoseOptimization(pose0, xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
t0 = time.perf_counter()
for _ in range(20):
    opt.PoseOptimization(pose0, xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
plat = (time.perf_counter() - t0) / 20
print(json.dumps({"pose_optimization_ms": round(plat * 1e3, 3), "correspondences": int(sel.sum())}))
