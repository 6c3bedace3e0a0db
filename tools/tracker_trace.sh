#!/bin/bash
# kernel trace of the per-frame device chain (tests/cpp/tracker_main.cpp, look-ahead on): tools/tracker_trace.sh [n_frames]
# prints the per-kernel statistics of the run (rocprofv3 --kernel-trace --stats)
n=${1:-80}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np
import __graft_entry__ as ge
from orbslam_in_practice_amd import sequence as sq
exe = ge.build_cpp_loop("tracker_main")
frames, gt, K, _ = sq.make_plane_sequence(5, $n)
os.makedirs("/tmp/tk", exist_ok=True)
frames.tofile("/tmp/tk/f.raw"); gt.astype(np.float64).tofile("/tmp/tk/gt.bin")
open("/tmp/tk/cmd", "w").write(" ".join([exe, "640", "480", "$n", "/tmp/tk/f.raw", "/tmp/tk/gt.bin", *("%.17g" % v for v in K), "/tmp/tk/out.bin", "prefetch"]))
PY
rm -rf /tmp/tk/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk/prof -- $(cat /tmp/tk/cmd) > /tmp/tk/run.log 2>&1
tail -2 /tmp/tk/run.log
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tk/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(10), r["Percentage"].rjust(6))
PY
