#!/usr/bin/env python3
"""average per-dispatch PMC counters of kernels matching a substring: tools/pmc_kernel.py DIR/x_counter_collection.csv substr"""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float)
disp = set()
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
        disp.add(r['Dispatch_Id'])
print(sys.argv[2], "dispatches", len(disp), " ".join("%s=%.2fM" % (c, v / max(len(disp), 1) / 1e6) for c, v in sorted(agg.items())))
