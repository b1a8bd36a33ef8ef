#!/usr/bin/env python3
"""Per-stage HIP-event timing of render forward+backward on the bench scene (dev tool)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
views = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(0, 60, 4))
dev = torch.device("cuda:0")
R._scratch.force_chunk = int(os.environ.get("GSX_FORCE_CHUNK", "0"))       # 64 / 128 / 256 pins the chunk size (0: the device picks)
params, H, W = config_scene(cfg)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
d_rgb = torch.randn(3, H, W, device=dev)
grads = [torch.empty_like(p) for p in P]
for rep in range(2):
    timer = R.StageTimer(); R.set_stage_timer(timer if rep else None)
    for v in views:
        outs, vs, _ = R.forward_view(P, poses[v], H, W)
        R.backward_view(P, vs, d_rgb, None, None, None, grads, False)
    torch.cuda.synchronize()
R.set_stage_timer(None)
tot = timer.totals_ms()
print(json.dumps({k: round(t / c, 4) for k, (t, c) in tot.items()}), "sum", round(sum(t / c for t, c in tot.values()), 4))
