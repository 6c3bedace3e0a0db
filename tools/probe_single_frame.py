#!/usr/bin/env python3
"""Single-frame latency anatomy: per-stage HIP-event times of one 640x480 frame through orb_extract_batch_device
(eager launches, profiling on) and the end-to-end host call with and without HIP-graph replay."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

L = oa._lib.lib()
frames = synth.make_stream(4, 8, 640, 480)
e = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480, max_batch=1)
dev = torch.device("cuda", 0)
d_img = torch.from_numpy(frames[0]).to(dev)
cap = e.cap
kps = torch.zeros((1, cap, 28), dtype=torch.uint8, device=dev)
desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
oa._lib.check(L.orb_set_profiling(e._h, 1))
for _ in range(20):
    e.extract_batch_device(d_img.data_ptr(), 1, 640, 480, 640, 640 * 480, kps.data_ptr(), desc.data_ptr(), cap, cnt.data_ptr(),
                           C.c_void_p(st.cuda_stream))
torch.cuda.synchronize()
ms5 = (C.c_float * 5)()
oa._lib.check(L.orb_get_stage_ms(e._h, ms5))
print("stage ms (1 frame, eager):", dict(zip(["pyramid", "fast", "quadtree", "blur", "describe"], [round(float(v), 4) for v in ms5])))
oa._lib.check(L.orb_set_profiling(e._h, 0))
t0 = time.perf_counter()
for _ in range(50):
    e.extract_batch_device(d_img.data_ptr(), 1, 640, 480, 640, 640 * 480, kps.data_ptr(), desc.data_ptr(), cap, cnt.data_ptr(),
                           C.c_void_p(st.cuda_stream))
    st.synchronize()
print("device-resident call + sync: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
for f in frames[:3]:
    e(f)
t0 = time.perf_counter()
for f in list(frames) * 4:
    e(f)
print("host call (graph replay): %.3f ms" % ((time.perf_counter() - t0) / 32 * 1e3))
