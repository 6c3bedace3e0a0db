#!/usr/bin/env python3
"""Local-BA throughput with several solver handles driven from host threads (ctypes releases the GIL inside
ba_solve_batch), so one call's host staging overlaps another call's device time.
Usage: python tools/bench_ba_threads.py [n_cams n_points batch threads reps]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    npt = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    nthr = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    uniq = [synth.make_ba_window(900 + i, nc, npt) for i in range(4)]
    wins = [uniq[i % 4] for i in range(batch)]
    E = max(len(w["obs"]) for w in uniq)
    opts = [oa.Optimizer(max_cams=nc, max_points=npt, max_obs=E, max_batch=batch) for _ in range(nthr)]
    for o in opts:
        o.solve_batch(wins[:2], (5, 10))

    def job(o):
        for _ in range(reps):
            o.solve_batch(wins, (5, 10))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(job, opts))
    dt = time.perf_counter() - t0
    print(json.dumps({"threads": nthr, "batch": batch, "windows_per_s": round(nthr * reps * batch / dt, 1),
                      "ms_per_batch": round(dt / reps * 1e3, 1)}))


if __name__ == "__main__":
    main()
