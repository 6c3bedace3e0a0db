#!/usr/bin/env python3
"""rocprofv3 kernel trace -> per kernel AND launch shape: calls, average / min / max duration.
The bench process launches the same kernel in several shapes (256-frame lanes of the timed pipeline, single frames of
the tracking loop, one window / 256 windows of BA); `--stats` averages over all of them, this keeps them apart so that
the bench line's per-launch figures can be held against the trace.
    python tools/trace_by_shape.py <kernel_trace.csv> <out.json>"""
import collections
import csv
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        wg = "x".join(r[k] for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        g[(name, grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = collections.defaultdict(list)
    for (name, grid, wg), v in g.items():
        if not name.startswith("k_"):
            continue
        out[name].append(dict(grid_threads=grid, workgroup=wg, calls=len(v), avg_us=round(sum(v) / len(v) / 1e3, 2),
                              min_us=round(min(v) / 1e3, 2), max_us=round(max(v) / 1e3, 2)))
    for name in out:
        out[name].sort(key=lambda e: -e["calls"] * e["avg_us"])
    doc = {"_note": "kernel durations of one `bench.py --steps 5 --warmup 2 --cpu-frames 0 --ba-reps 2` under rocprofv3 "
                    "--kernel-trace, split by launch shape (grid in threads); the two launch lanes of the pipeline overlap, "
                    "so a kernel's duration here is its duration beside the other lane's kernels",
           "kernels": dict(sorted(out.items()))}
    json.dump(doc, open(dst, "w"), indent=1)
    print("wrote", dst, len(out), "kernels")


if __name__ == "__main__":
    main()
