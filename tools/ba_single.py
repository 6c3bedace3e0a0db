#!/usr/bin/env python3
"""One window in flight (the LocalMapping call pattern): ms per ba_solve of a 100 x 5000 (or NC x NP) window, kernel time of
its optimize() rounds, team size.   python tools/ba_single.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

NC, NP = int(os.environ.get("NC", "100")), int(os.environ.get("NP", "5000"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
uniq = [synth.make_ba_window(900 + i, NC, NP) for i in range(4)]
opt = oa.Optimizer(max_cams=NC, max_points=NP, max_obs=10 * NP, max_batch=1)
opt.solve_batch([uniq[0]], (5, 10))
t0 = time.perf_counter()
for i in range(reps):
    r = opt.solve_batch([uniq[i % 4]], (5, 10))
ms = (time.perf_counter() - t0) / reps * 1e3
opt.set_profiling(True)
opt.solve_batch([uniq[0]], (5, 10))
kms, nl, team = opt.iteration_ms()
print("one window %dx%d: %.2f ms per call; kernel %d launches x %.2f ms = %.2f ms, team %d, lin_solves %d"
      % (NC, NP, ms, nl, kms, nl * kms, team, r[0]["lin_solves"]))
opt.close()
