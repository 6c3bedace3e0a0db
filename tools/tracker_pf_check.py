#!/usr/bin/env python3
"""plain vs look-ahead runs of tests/cpp/tracker_main.cpp: first frame whose record differs"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from orbslam_in_practice_amd import sequence as sq
exe = ge.build_cpp_loop("tracker_main")
td = tempfile.mkdtemp()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
frames, gt, K, _ = sq.make_plane_sequence(5, n)
frames.tofile(td + "/f.raw"); gt.astype(np.float64).tofile(td + "/gt.bin")
dt = np.dtype([("pose", "<f8", 7), ("nm", "<i4"), ("inl", "<i4")])
recs = []
for tag, extra in (("plain", []), ("plain", []), ("prefetch", ["prefetch"]), ("prefetch", ["prefetch"])):
    out = td + "/o.bin"
    subprocess.run([exe, "640", "480", str(n), td + "/f.raw", td + "/gt.bin", *("%.17g" % v for v in K), out, *extra], capture_output=True, text=True)
    recs.append((tag, np.frombuffer(open(out, "rb").read(), dt, n, 4).copy()))
base = recs[0][1]
for tag, r in recs[1:]:
    bad = [k for k in range(n) if r[k] != base[k]]
    print(tag, "differs at frames", bad[:10], "of", n)
    for k in bad[:3]:
        print("   frame", k, "nm", base["nm"][k], r["nm"][k], "inl", base["inl"][k], r["inl"][k], "dpose", np.abs(base["pose"][k] - r["pose"][k]).max())
