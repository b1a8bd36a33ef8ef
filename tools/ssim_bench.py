#!/usr/bin/env python3
"""Times gsx_l1_ssim_fwd/bwd alone at a few image sizes (dev tool)."""
import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
for H, W in ((400, 400), (800, 800), (1600, 1600), (1024, 1024)):
    x = torch.rand(3, H, W, device=dev); y = torch.rand(3, H, W, device=dev); d = torch.empty_like(x)
    loss = torch.empty(3, device=dev); g = torch.ones(1, device=dev)
    wsb = lib.gsx_l1_ssim_workspace_bytes(3, H, W); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = L.stream_ptr()
    for name, fn in (("fwd", lambda: lib.gsx_l1_ssim_fwd(p(x), p(y), 3, H, W, 0.2, p(loss), p(ws), wsb, st)),
                     ("bwd", lambda: lib.gsx_l1_ssim_bwd(p(x), p(y), 3, H, W, 0.2, p(g), p(ws), wsb, p(d), st))):
        for _ in range(5): fn()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        b.record(); torch.cuda.synchronize()
        print(f"{H}x{W} {name}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call")
