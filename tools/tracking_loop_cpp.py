#!/usr/bin/env python3
"""Build and run tests/cpp/tracking_main.cpp -- the Tracking + LocalMapping style loop on the drop-in classes with the
reference's own class names and signatures -- on a synthetic plane sequence: frames/s of one stream without Python in
the loop.      python tools/tracking_loop_cpp.py [n_frames]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam_in_practice_amd import sequence as sq  # noqa: E402

lib = os.path.join(ROOT, "orbslam_in_practice_amd", "csrc")
td = tempfile.mkdtemp()
exe = os.path.join(td, "tracking_main")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "orbslam_amd"),
                       "-I", os.path.join(ROOT, "tests", "cpp", "refmodel"), os.path.join(ROOT, "tests", "cpp", "tracking_main.cpp"),
                       "-o", exe, "-L", lib, "-lorbslam_amd", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
frames, gt, K, _ = sq.make_plane_sequence(5, n)
frames.tofile(td + "/f.raw")
gt.astype(np.float64).tofile(td + "/gt.bin")
for rep in range(2):  # (the second run has warm file caches and device clocks)
    out = subprocess.run([exe, "640", "480", str(n), td + "/f.raw", td + "/gt.bin", *("%.17g" % v for v in K), td + "/out.bin"],
                         capture_output=True, text=True)
    print(out.stdout.strip())
    if out.stderr.strip():
        print(out.stderr.strip()[-int(os.environ.get("STDERR_TAIL", "300")):])
raw = open(td + "/out.bin", "rb").read()
rec = np.frombuffer(raw, np.dtype([("pose", "<f8", 7), ("nm", "<i4"), ("inl", "<i4")]), n, 4)
print("ATE %.4f m, mean matches %.0f" % (sq.ate_rmse(rec["pose"].copy(), gt), rec["nm"][1:].mean()))
