#!/usr/bin/env python3
"""Device time of the per-frame chain on device-resident data: extraction of one frame (orb_extract_batch_device) followed by
orb_search_by_projection_device against the previous frame -- HIP events around the enqueued chain, no copy and no wait
in between (what a device-resident tracker would run before PoseOptimization)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

W, H = 640, 480
frames = synth.make_stream(4, 8, W, H)
dev = torch.device("cuda", 0)
e = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=1)
cap = e.cap
m = oa.ORBmatcher(0.9, True, max_keypoints=cap)
d_img = [torch.from_numpy(f).to(dev) for f in frames]
slab = [dict(k=torch.zeros((cap, 28), dtype=torch.uint8, device=dev), d=torch.zeros((cap, 32), dtype=torch.uint8, device=dev),
             c=torch.zeros((1,), dtype=torch.int32, device=dev)) for _ in range(2)]
st = torch.cuda.Stream(device=dev)
sp = C.c_void_p(st.cuda_stream)
SF = (1.2 ** np.arange(8)).astype(np.float32)
bounds = (0.0, float(W), 0.0, float(H))


def extract(i, s):
    e.extract_batch_device(d_img[i].data_ptr(), 1, W, H, W, W * H, s["k"].data_ptr(), s["d"].data_ptr(), cap, s["c"].data_ptr(), sp)


extract(0, slab[0])
st.synchronize()
n_last = int(slab[0]["c"].item())
kp = slab[0]["k"].cpu().numpy().reshape(-1).view(oa._lib.KP_DTYPE)[:n_last]
xw = np.stack([(kp["x"] - 319.5) / 525.0 * 4.0, (kp["y"] - 239.5) / 525.0 * 4.0, np.full(n_last, 4.0)], axis=1)
t_fl = torch.full((cap,), 3, dtype=torch.uint8, device=dev)
t_xw = torch.zeros((cap, 3), dtype=torch.float64, device=dev)
t_xw[:n_last] = torch.from_numpy(xw).to(dev)
t_tk = torch.zeros((cap,), dtype=torch.uint8, device=dev)
t_m = torch.zeros((cap,), dtype=torch.int32, device=dev)
t_n = torch.zeros((1,), dtype=torch.int32, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tot_e = tot_s = 0.0
reps = 30
for r in range(reps + 3):
    t_tk.zero_()
    st.synchronize()
    torch.cuda.synchronize()
    ev[0].record(st)
    extract(1, slab[1])
    ev[1].record(st)
    n_cur = 1004  # (the count stays on the device in a real chain; here the known size of this frame)
    m.search_by_projection_device(slab[1]["k"].data_ptr(), slab[1]["d"].data_ptr(), n_cur, slab[0]["k"].data_ptr(), slab[0]["d"].data_ptr(),
                                  n_last, t_fl.data_ptr(), t_xw.data_ptr(), np.eye(3), np.zeros(3), 525.0, 525.0, 319.5, 239.5, SF, bounds,
                                  t_tk.data_ptr(), 15.0, t_m.data_ptr(), t_n.data_ptr(), sp)
    ev[2].record(st)
    st.synchronize()
    if r >= 3:
        tot_e += ev[0].elapsed_time(ev[1])
        tot_s += ev[1].elapsed_time(ev[2])
print("device chain per frame: extraction %.3f ms, SearchByProjection (project + top-k + walk) %.3f ms, matches %d of %d"
      % (tot_e / reps, tot_s / reps, int(t_n.item()), n_last))
