#!/usr/bin/env python3
"""Host-side cost of one trainer step: the same step on a tiny scene, where the kernels take no time (dev tool)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.scenes import synth_scene, create_orbit_poses
from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss

dev = torch.device("cuda:0")
n, H, W = 2000, 64, 64
params = synth_scene(n, 5, 0.01, 0.05)
model = GaussianSplat(n); model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); model = model.to(dev)
opt = make_optimizer(model, fused_clip_norm=0.1)
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
target = torch.rand(1, 3, H, W, device=dev)
for s in range(20):
    train_step(model, opt, poses[s:s + 1], target, l1_loss, 0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 300
for s in range(K):
    train_step(model, opt, poses[s % 60:s % 60 + 1], target, l1_loss, 0.1)
torch.cuda.synchronize()
print("ms/step on a tiny scene (host-bound):", round((time.perf_counter() - t0) / K * 1e3, 4))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for s in range(100):
    train_step(model, opt, poses[s % 60:s % 60 + 1], target, l1_loss, 0.1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
