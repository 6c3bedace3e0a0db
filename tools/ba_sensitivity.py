#!/usr/bin/env python3
"""How well-posed is a global-BA test window at the 1e-5 tolerance?  Runs the CPU oracle twice on the same window, the
second time with the input points scaled by (1 + 1e-13), and prints how far the two runs end up apart.  A window whose
self-agreement is worse than the tolerance cannot pin ANY implementation to it (tests/test_gpu_ba.py picks its a13
cases from this table).  CPU only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

CASES = [(2, 300, 0.05, 20, 8), (2, 300, 0.05, 20, 101), (2, 300, 0.0, 20, 101), (3, 500, 0.05, 10, 102),
         (50, 2000, 0.0, 20, 149), (50, 2000, 0.05, 12, 149), (50, 2000, 0.05, 20, 2350), (100, 5000, 0.0, 20, 199)]
print("cams points outl its seed | chi2_final iterations | self-agreement(points) max|X| lin_solves(a,b)")
for nc, npt, of, k, seed in CASES:
    w = synth.make_ba_window(seed, nc, npt, outlier_frac=of)
    w2 = dict(w)
    w2["points"] = w["points"] * (1 + 1e-13)
    a, b = orc.ba_solve(w, rounds=(k,)), orc.ba_solve(w2, rounds=(k,))
    print("%4d %6d %.2f %3d %4d | %12.4f %2d | %.2e %6.1f (%d, %d)" % (
        nc, npt, of, k, seed, a["chi2_final"], a["iterations"],
        np.abs(a["points"] - b["points"]).max() / np.abs(a["points"]).max(), np.abs(a["points"]).max(),
        a["lin_solves"], b["lin_solves"]))
