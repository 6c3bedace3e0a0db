#!/usr/bin/env python3
"""PoseOptimization latency anatomy: one pose problem of N correspondences, ms per call and (ORBSLAM_AMD_PO_TIMING=1) the
sections of an LM iteration.   python tools/po_probe.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

opt = oa.Optimizer(max_cams=8, max_points=64, max_obs=64, max_batch=1)
for n in [int(a) for a in sys.argv[1:]] or [380, 850]:
    w = synth.make_ba_window(77, 6, max(4 * n, 60), outlier_frac=0.1)
    sel = np.nonzero(w["cam_idx"] == 2)[0][:n]
    xw, obs, info = w["gt_points"][w["pt_idx"][sel]], w["obs"][sel], w["inv_sigma2"][sel]
    args = (w["poses"][2], xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
    opt.PoseOptimization(*args)
    t0 = time.perf_counter()
    for _ in range(50):
        p, o, ninl = opt.PoseOptimization(*args)
    print("n %d: %.3f ms per call, %d inliers" % (len(sel), (time.perf_counter() - t0) / 50 * 1e3, ninl))
opt.close()
