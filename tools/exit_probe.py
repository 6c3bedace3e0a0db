#!/usr/bin/env python3
"""What tools/ba_probe.py 32 does (one 256-window batch through the phase kernels, NO cooperative launch, no torch),
with markers on stderr at every step of the teardown: used by tools/exit_hang_probe.sh to find where a profiled run
stops at process exit.  MODE=hipfirst initialises the HIP runtime BEFORE liborbslam_amd.so is loaded."""
import atexit
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def say(m):
    os.write(2, ("[exit_probe.py] %s\n" % m).encode())


probe = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "exitprobe.so"))
probe.exitprobe_install()
atexit.register(lambda: say("python atexit hook (interpreter still alive)"))
if os.environ.get("MODE") == "hipfirst":
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    n = ctypes.c_int(0)
    say("hipGetDeviceCount before the library is loaded: rc %d n %d" % (hip.hipGetDeviceCount(ctypes.byref(n)), n.value))
if os.environ.get("MODE") == "torchfirst":
    import torch
    torch.cuda.init()
    say("torch %s initialised the device first" % torch.__version__)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

say("library loaded")
if os.environ.get("MODE") == "frontend":   # no BA at all: the extractor on a few frames
    ext = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480, max_batch=8)
    fr = synth.make_stream(3, 8, 640, 480, n_shapes=1500)
    r = ext.extract_batch(fr)
    say("front-end only: %d keypoints in frame 0" % len(r[0][0]))
    ext.close()
    say("end of script")
    sys.exit(0)
batch = int(os.environ.get("BATCH", "256"))
uniq = [synth.make_ba_window(900 + i, 100, 5000) for i in range(8)]
wins = [uniq[i % 8] for i in range(batch)]
opt = oa.Optimizer(max_cams=100, max_points=5000, max_obs=50000, max_batch=batch)
say("ba_create done")
res = opt.solve_batch(wins, (5, 10))
if os.environ.get("PROFILED_SOLVE", "1") == "1":   # what tools/ba_probe.py does: a second solve with HIP events around the rounds
    opt.set_profiling(True)
    res = opt.solve_batch(wins, (5, 10))
    say("profiled solve: %s" % (opt.iteration_ms(),))
say("solve_batch done: chi2 %.3f -> %.3f" % (res[0]["chi2_initial"], res[0]["chi2_final"]))
opt.close()
say("opt.close() (ba_destroy) returned")
if os.environ.get("MODE") == "reset":   # tear the HIP runtime down while the profiler tool is still alive
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    say("hipDeviceSynchronize rc %d" % hip.hipDeviceSynchronize())
    say("hipDeviceReset rc %d" % hip.hipDeviceReset())
if os.environ.get("MODE") == "drain":   # let the profiler's counter-collection thread finish its per-dispatch work first
    import time
    time.sleep(3.0)
    say("slept 3 s before leaving")
say("end of script")
