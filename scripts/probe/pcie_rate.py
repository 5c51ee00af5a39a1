import torch, time
n = 78*1024*1024//8
h = torch.empty(n, dtype=torch.float64).pin_memory(); h.fill_(1.0)
d = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(3):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(10): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print(f"H2D hipMemcpyAsync pinned: {n*8/dt/1e9:.1f} GB/s ({dt*1e3:.2f} ms per 78 MB)")
t=time.perf_counter()
for _ in range(10): h.copy_(d, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print(f"D2H hipMemcpyAsync pinned: {n*8/dt/1e9:.1f} GB/s")
# two streams both directions
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.float64).pin_memory(); d2=torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print(f"bidirectional: {2*n*8/dt/1e9:.1f} GB/s total")
