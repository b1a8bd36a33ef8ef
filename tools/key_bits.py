import sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
dev = torch.device("cuda:0")
for cfg in ("c3", "c2"):
    params, H, W = config_scene(cfg)
    P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
    poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
    for v in (0, 7, 15, 22, 30, 45):
        mean2d, depth, cov2d, keys, pre = R.project(*P, poses[v])
        k = keys.cpu().numpy().view(np.uint32)
        a = np.bitwise_and.reduce(k); o = np.bitwise_or.reduce(k)
        print(cfg, v, hex(a), hex(o), "varying:", hex(a ^ o), float(depth.min()), float(depth.max()))
