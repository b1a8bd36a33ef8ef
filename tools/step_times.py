#!/usr/bin/env python3
"""Per-step wall time, instance count and allocator state over an orbit (dev tool): python tools/step_times.py c5 24"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
n = params["positions"].shape[0]
model = GaussianSplat(n); model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); model = model.to(dev)
opt = make_optimizer(model, fused_clip_norm=0.1)
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
target = torch.rand(1, 3, H, W, device=dev)
for s in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_step(model, opt, poses[s % 60:s % 60 + 1], target, l1_loss, 0.1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {s:3d} {dt*1e3:9.2f} ms  D={R.last_instance_count():>12d} cap={R._scratch.cap(dev, n):>12d} "
          f"alloc={torch.cuda.memory_allocated()/2**30:6.1f} GiB reserved={torch.cuda.memory_reserved()/2**30:6.1f} GiB "
          f"mallocs={st.get('num_device_alloc', 0)} frees={st.get('num_device_free', 0)}", flush=True)
