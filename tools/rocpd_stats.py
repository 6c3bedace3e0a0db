#!/usr/bin/env python3
"""Per-kernel (and per-grid) duration summary from a rocprofv3 rocpd database (the default output of
`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- ...` on ROCm 7): python tools/rocpd_stats.py DIR/NAME_results.db [--grid]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    grid = "--grid" in sys.argv
    key = "s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z" if grid else "s.kernel_name"
    q = (f"select {key}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by {key} order by sum(d.end-d.start) desc")
    print("kernel,grid,calls,avg_us,min_us,max_us,total_ms" if grid else "kernel,calls,avg_us,min_us,max_us,total_ms")
    for r in c.execute(q):
        name = r[0].split("(")[0][:60]
        if grid:
            print("%s,%dx%dx%d,%d,%.1f,%.1f,%.1f,%.3f" % (name, r[1], r[2], r[3], r[4], r[5] / 1e3, r[6] / 1e3, r[7] / 1e3, r[8] / 1e6))
        else:
            print("%s,%d,%.1f,%.1f,%.1f,%.3f" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6))


if __name__ == "__main__":
    main()
