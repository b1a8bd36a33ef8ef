#!/usr/bin/env python3
"""dev tool: what a plain streaming kernel reaches on this box (copy, read+read+write, read-only sum), at the sizes of the step's
memory-bound kernels -- the practical ceiling their GB/s are to be read against."""
import torch, time
dev = torch.device("cuda:0")
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (56, 112, 224, 448, 1792):
    n = mb * 2**20 // 4
    a = torch.rand(n, device=dev); b = torch.rand(n, device=dev); c = torch.empty(n, device=dev)
    t = bench(lambda: c.copy_(a)); print(f"{mb:5d} MB  copy       {2*n*4/t/1e12:.2f} TB/s ({t*1e6:.1f} us)")
    t = bench(lambda: torch.add(a, b, out=c)); print(f"{mb:5d} MB  a+b->c     {3*n*4/t/1e12:.2f} TB/s ({t*1e6:.1f} us)")
    t = bench(lambda: a.sum()); print(f"{mb:5d} MB  sum        {n*4/t/1e12:.2f} TB/s ({t*1e6:.1f} us)")
    t = bench(lambda: a.mul_(1.0001)); print(f"{mb:5d} MB  a*=s       {2*n*4/t/1e12:.2f} TB/s ({t*1e6:.1f} us)")
    del a, b, c
