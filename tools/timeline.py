#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace run: how much of the wall time has 0, 1, 2, ... kernels in flight,
which kernel sets cover the time, and the share of the time in which only latency-bound kernels run.
Usage: python tools/timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    name = name.replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
t0 = t_lo + (t_hi - t_lo) * skip  # skip warm-up
pts = []
for s, e, n in ev:
    if e <= t0:
        continue
    pts.append((max(s, t0), 1, n))
    pts.append((e, -1, n))
pts.sort(key=lambda p: (p[0], p[1]))
active = Counter()
last = pts[0][0]
by_n = Counter()
by_set = Counter()
LIGHT = ("k_quadtree", "k_match_init", "k_copy_level0")
for t, d, n in pts:
    dt = t - last
    if dt > 0:
        k = sum(active.values())
        by_n[min(k, 6)] += dt
        names = tuple(sorted(x for x in active if active[x] > 0))
        by_set[names] += dt
    active[n] += d
    last = t
tot = sum(by_n.values())
print("window %.1f ms" % (tot / 1e6))
for k in sorted(by_n):
    print("  %d%s kernels in flight: %5.1f %%" % (k, "+" if k == 6 else "", 100.0 * by_n[k] / tot))
light = sum(v for s, v in by_set.items() if s and all(any(x.startswith(l) for l in LIGHT) for x in s))
print("  only latency-bound kernels (%s) in flight: %.1f %%" % (", ".join(LIGHT), 100.0 * light / tot))
print("top kernel sets:")
for s, v in by_set.most_common(14):
    print("  %5.1f %%  %s" % (100.0 * v / tot, " + ".join(x[:22] for x in s) if s else "(idle)"))
