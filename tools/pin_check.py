import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from genefuserust_amd.stream import pinned_empty
from genefuserust_amd import _lib
n = 256 << 20
a = pinned_empty(n, np.uint8); a[:] = 1
t = torch.from_numpy(a)
print("gf_host_alloc memory: torch.is_pinned =", t.is_pinned())
tp = torch.empty(n, dtype=torch.uint8, pin_memory=True); tp[:] = 1
print("torch pinned: is_pinned =", tp.is_pinned(), " numpy view ->", torch.from_numpy(tp.numpy()[1000:]).is_pinned())
d = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
for name, src in (("gf_host_alloc", t), ("torch pinned", tp), ("pageable", torch.ones(n, dtype=torch.uint8))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        d.copy_(src, non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-14s call %.2f ms, until done %.2f ms  (%.1f GB/s)" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3, n / (t2 - t0) / 1e9))
# does a kernel on the null stream overlap with the copy? and a null-stream synchronize: does it wait for the copy?
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    d.copy_(tp, non_blocking=True)
x.add_(1)
torch.cuda.current_stream().synchronize()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("null-stream kernel + null-stream sync while a copy runs on a side stream: %.2f ms (copy done after %.2f ms)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    d.copy_(tp, non_blocking=True)
v = x[:1].cpu()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("8-byte .cpu() read-back while a copy runs on a side stream: %.2f ms" % ((t1 - t0) * 1e3))
