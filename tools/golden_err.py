#!/usr/bin/env python3
"""dev tool (GPU box): gradient errors of one golden fixture against the reference's run and against the oracle, per tensor,
with the worst element.  usage: python tools/golden_err.py <fixture> [<fixture> ...]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, rel_err, l2_rel_err, PARAMS
from test_gpu_parity import gpu_pipeline, gpu_backward
from oracle import gsx_oracle as O
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    fx = load_golden(name)
    H, W = fx["H"], fx["W"]
    order = fx["order"].astype(np.uint32) if fx["has_ties"] else None
    gp = gpu_pipeline(fx["params"], fx["poses"][0], H, W, dev, order=order)
    fw = O.forward(fx["params"], fx["poses"][0], H, W, "tiled", order=order)
    g = gpu_backward(gp, H, W, fx["d_rgb"][0], fx["d_pos2d"][0], fx["d_depths"][0], fx["d_cov2d"][0])
    go = O.backward(fx["params"], fx["poses"][0], H, W, fw, fx["d_rgb"][0], fx["d_pos2d"][0], fx["d_depths"][0], fx["d_cov2d"][0], "tiled")
    for k in PARAMS:
        a = np.asarray(g[k], np.float64); r = np.asarray(fx["g_" + k], np.float64); o = np.asarray(go[k], np.float64)
        i = np.unravel_index(np.abs(a - o).argmax(), a.shape)
        print(f"{name} {k:10s} vs ref {rel_err(a, r):.3e} (l2 {l2_rel_err(a, r):.2e})  vs oracle {rel_err(a, o):.3e} (l2 {l2_rel_err(a, o):.2e})  "
              f"oracle vs ref {rel_err(o, r):.3e}  noise {float(fx.get('ref_noise_' + k, 0)):.2e}  worst {i}: gpu {a[i]:.6e} oracle {o[i]:.6e} ref {r[i]:.6e} max {np.abs(r).max():.3e}")
