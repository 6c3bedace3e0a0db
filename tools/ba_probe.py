#!/usr/bin/env python3
"""BA batch probe: device time of a 256-window batch for different numbers of distinct windows and team sizes.
Usage: python tools/ba_probe.py [n_unique ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

batch = int(os.environ.get("BATCH", "256"))
NC, NP = int(os.environ.get("NC", "100")), int(os.environ.get("NP", "5000"))
for nu in [int(a) for a in sys.argv[1:]] or [4, 32]:
    uniq = [synth.make_ba_window(900 + i, NC, NP) for i in range(nu)]
    wins = [uniq[i % nu] for i in range(batch)]
    opt = oa.Optimizer(max_cams=NC, max_points=NP, max_obs=10 * NP, max_batch=batch)
    opt.solve_batch(wins, (5, 10))
    opt.set_profiling(True)
    t0 = time.perf_counter()
    res = opt.solve_batch(wins, (5, 10))
    dt = time.perf_counter() - t0
    ms, nl, team = opt.iteration_ms()
    ls = np.array([r["lin_solves"] for r in res[:nu]])
    E = np.array([len(w["obs"]) for w in uniq])
    print("unique %d batch %d: call %.1f ms, kernel launches %d x %.2f ms = %.1f ms, team %d, lin_solves min/mean/max %d/%.2f/%d, E min/max %d/%d"
          % (nu, batch, dt * 1e3, nl, ms, nl * ms, team, ls.min(), ls.mean(), ls.max(), E.min(), E.max()))
    reps = int(os.environ.get("STREAM", "0"))
    if reps:  # a stream of batches through submit / wait (DEPTH in flight): windows/s with staging and copies overlapped
        depth = int(os.environ.get("DEPTH", "3"))
        opt.set_profiling(False)
        pending = 0
        for _ in range(depth - 1):
            opt.submit_batch(wins, (5, 10))
            pending += 1
        t0 = time.perf_counter()
        for _ in range(reps):
            opt.submit_batch(wins, (5, 10))
            opt.wait()
        dt = time.perf_counter() - t0
        while pending:
            opt.wait()
            pending -= 1
        print("  streamed (%d in flight): %d batches of %d in %.1f ms = %.0f windows/s" % (depth, reps, batch, dt * 1e3, reps * batch / dt))
    opt.close()
