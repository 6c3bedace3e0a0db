#!/usr/bin/env python3
"""Matcher alone: B frames extracted once, then match_batch_device(k, k+1) timed by itself (nothing else on the device).
Usage: python tools/match_probe.py [B] [window]     ORBSLAM_AMD_MATCH_NO_MFMA=1 for the xor/popcount scan"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
window = int(sys.argv[2]) if len(sys.argv) > 2 else -1
W, H = 640, 480
dev = torch.device("cuda:0")
ext = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B, device=0)
cap = ext.cap
m = oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B, device=0)
d = torch.from_numpy(synth.make_stream(1000, B, W, H)).to(dev)
kps = torch.zeros((B + 1, cap, 28), dtype=torch.uint8, device=dev)
desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros((B + 1,), dtype=torch.int32, device=dev)
m12 = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
nm = torch.zeros((B,), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
ext.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, cnt.data_ptr(), C.c_void_p(st.cuda_stream))
torch.cuda.synchronize()
pa = torch.arange(0, B, dtype=torch.int32, device=dev)
pb = torch.clamp(pa + 1, max=B - 1).to(torch.int32)
L = oa._lib.lib()
oa._lib.check(L.orb_matcher_set_profiling(m._m, 1))
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        m.match_batch_device(kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, pa.data_ptr(), pb.data_ptr(), B,
                             (0.0, float(W), 0.0, float(H)), window, m12.data_ptr(), nm.data_ptr(), C.c_void_p(st.cuda_stream))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    mm = (C.c_float * 2)()
    oa._lib.check(L.orb_matcher_get_stage_ms(m._m, mm))
    print("B %d window %d: %.3f ms per call (phase 1 %.3f, phase 2 %.3f), matches %d, mean keypoints %.0f"
          % (B, window, dt * 1e3, mm[0], mm[1], int(nm.sum()), float(cnt[:B].float().mean())))
