#!/usr/bin/env python3
"""Local-BA throughput (BASELINE config 3 / target): windows/s for a batch of independent windows through
ba_solve_batch (5+10 LM iterations, chi2 culling in between), with the CPU oracle (dense Cholesky, 1 core)
timed beside it.  Usage: python tools/bench_ba.py [n_cams n_points batch reps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    npt = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    n_unique = min(batch, 8)
    uniq = [synth.make_ba_window(900 + i, nc, npt) for i in range(n_unique)]
    wins = [uniq[i % n_unique] for i in range(batch)]
    E = int(np.mean([len(w["obs"]) for w in uniq]))
    opt = oa.Optimizer(max_cams=max(nc, 8), max_points=npt, max_obs=E * 2, max_batch=batch)
    opt.solve_batch(wins[:2], (5, 10))  # warm-up
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        res = opt.solve_batch(wins, (5, 10))
        times.append(time.perf_counter() - t0)
    dt = min(times)
    out = {"metric": "local-BA windows/s", "config": "%d KFs x %d points, E~%d, 5+10 LM its" % (nc, npt, E),
           "batch": batch, "windows_per_s": round(batch / dt, 2), "ms_per_batch": round(dt * 1e3, 2),
           "lin_solves_per_window": float(np.mean([r["lin_solves"] for r in res])),
           "chi2_drop": float(np.mean([r["chi2_final"] / r["chi2_initial"] for r in res]))}
    # single-window latency
    t0 = time.perf_counter()
    opt.solve_batch(wins[:1], (5, 10))
    out["single_window_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    if os.environ.get("BA_CPU", "1") == "1":
        from oracle import oracle as orc
        t0 = time.perf_counter()
        n_cpu = 2
        for w in uniq[:n_cpu]:
            orc.ba_solve(w, rounds=(5, 10))
        out["cpu_oracle_windows_per_s_1core"] = round(n_cpu / (time.perf_counter() - t0), 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
