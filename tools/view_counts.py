"""Instance count D of every orbit view of a config's scene (projection, depth sort, records, scan: no binning), to pick
views for tests:  python tools/view_counts.py c5"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
out = {}
for v in range(60):
    mean2d, depth, cov2d, keys, pre = R.project(*P, poses[v])
    order = R.depth_order(keys)
    rec, tt, rank_of = R.build_records(pre, order, H, W)
    offsets, total, (pinned, ev) = R.bin_count(tt)
    ev.synchronize()
    out[v] = int(pinned[0])
    print(v, out[v], flush=True)
json.dump(out, open(os.path.join("gpurun_out", f"view_counts_{cfg}.json"), "w"))
