#!/usr/bin/env python3
"""dev tool: trains the bench scene N steps (tools/soak.py's loop), then for a few views prints the exit search's words -- exit rank,
flags, running maximum, barrier arrivals (= rounds x 512 on a non-monotone view) -- and times the forward with CUDA events."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss
dev = torch.device("cuda:0")
params, H, W = config_scene("c3")
n = params["positions"].shape[0]
m = GaussianSplat(n); m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); m = m.to(dev)
opt = make_optimizer(m, fused_clip_norm=0.1)
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
target = torch.rand(1, 3, H, W, device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
for s in range(steps):
    train_step(m, opt, poses[s % 60:s % 60 + 1], target, l1_loss, 0.1)
torch.cuda.synchronize()
P = tuple(p.detach().contiguous() for p in (m.positions, m.scales, m.rotations, m.opacity, m.features))
for v in (10,):
    for rep in range(1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs, vs, extra = R.forward_view(P, poses[v], H, W)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = vs.state.cpu().numpy().astype(np.uint32).ravel()
    rg = vs.ranges.cpu().numpy().astype(np.int64).reshape(-1, 2)
    ln = np.sort(rg[:, 1] - rg[:, 0])[::-1]
    print("         tile list lengths: max", ln[:5].tolist(), "median", int(np.median(ln)), "tiles > 8k:", int((ln > 8192).sum()), "> 2k:", int((ln > 2048).sum()))
    print(f"view {v:2d}: forward {dt*1e3:7.3f} ms  exit rank {int(st[0]):>10d}  flags {int(st[1])}  max {int(st[2]):>10d}  arrivals {int(st[3])} (rounds {int(st[3]) / 512:.1f})  D {vs.D}")
