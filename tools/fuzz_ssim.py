#!/usr/bin/env python3
"""Random image shapes for the fused L1/SSIM loss against conv2d + autograd in float64 (dev tool, MI355X)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_optim import _torch_ssim_f64
from gaussian_splatting_amd import loss as gl
dev = torch.device("cuda:0")
bad = 0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for seed in range(cases):
    r = np.random.default_rng(seed)
    B = int(r.choice([1, 1, 2])); C = int(r.choice([1, 3])); H = int(r.choice([1, 2, 5, 10, 11, 12, 31, 32, 33, 42, 43, 64, 97, 200]))
    W = int(r.choice([1, 3, 9, 11, 16, 31, 32, 33, 41, 65, 128, 131, 257])); lam = float(r.choice([0.0, 0.2, 1.0]))
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(B, C, H, W, generator=g); t = (x + 0.25 * torch.randn(B, C, H, W, generator=g)).clamp(0, 1)
    xb = x.to(dev).requires_grad_(True); xd = x.double().to(dev).requires_grad_(True); td = t.to(dev)
    ld = (1 - lam) * (xd - td.double()).abs().mean() + lam * (1 - _torch_ssim_f64(xd, td.double()))
    lb = gl.l1_ssim(xb, td, lam)
    (lb * 2.0).backward(); (ld * 2.0).backward()
    ref = xd.grad.float(); scale = max(ref.abs().max().item(), 1e-12)
    e_loss = abs(float(lb.detach()) - float(ld.detach())); e_grad = (xb.grad - ref).abs().max().item() / scale
    ok = e_loss <= 2e-6 and e_grad <= 3e-5
    if not ok:
        bad += 1
        print(f"FAIL seed {seed} shape {(B, C, H, W)} lam {lam}: loss err {e_loss:.3e} grad err {e_grad:.3e}", flush=True)
print(f"done: {cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
