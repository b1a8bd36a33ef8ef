import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
dev = torch.device("cuda:0")
params, H, W = config_scene("c3")
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
Ds = []
for v in range(60):
    outs, vs, _ = R.forward_view(P, poses[v], H, W); Ds.append(vs.D)
print(sorted(Ds))
