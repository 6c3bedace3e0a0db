import torch, time
x = torch.empty(512*1024*1024, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device='cuda')
for _ in range(2): d.copy_(x, non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter(); d.copy_(x, non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("H2D pinned 512MB: %.1f GB/s"%(0.512*1.048576/dt))
t0=time.perf_counter(); x.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("D2H pinned 512MB: %.1f GB/s"%(0.512*1.048576/dt))
