#!/usr/bin/env python3
"""Soak: many trainer steps over the orbit; reports time per step in windows and peak device memory (dev tool)."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
losses = []
for w in range(K // 300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(300):
        v = (w * 300 + s) % 60
        loss = train_step(m, opt, poses[v:v + 1], target, l1_loss, 0.1)
    torch.cuda.synchronize()
    losses.append(float(loss)); dsum = 0
    with torch.no_grad():
        for v in range(0, 60, 6):
            m(poses[v:v + 1], (H, W)); dsum += R.last_instance_count()
    print(f"steps {w*300}-{w*300+299}: {(time.perf_counter()-t0)/300*1e3:.4f} ms/step, loss {losses[-1]:.6f}, "
          f"mean D {dsum // 10}, scale median {float(m.scales.data.median()):.5f}, peak mem {torch.cuda.max_memory_allocated()/2**20:.0f} MiB, reserved {torch.cuda.memory_reserved()/2**20:.0f} MiB", flush=True)
assert all(np.isfinite(losses))
# per-stage times on the evolved scene
timer = R.StageTimer(); R.set_stage_timer(timer)
for v in range(60):
    opt.zero_grad()
    out = m(poses[v:v + 1], (H, W))
    l1_loss(out, target).backward()
torch.cuda.synchronize(); R.set_stage_timer(None)
print({k: round(t / c, 4) for k, (t, c) in timer.totals_ms().items()})
print("opacity <0: %.3f, >1: %.3f, >2: %.3f" % (float((m.opacity < 0).float().mean()), float((m.opacity > 1).float().mean()), float((m.opacity > 2).float().mean())))
