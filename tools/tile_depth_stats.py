"""dev tool (GPU box): chunks per tile of the bench scene's views (what bounds the combine pass: its longest tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
dev = torch.device("cuda:0")
params, H, W = config_scene("c3")
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
for v in (0, 7, 15, 30, 45):
    outs, vs, _ = R.forward_view(P, poses[v], H, W)
    cb = vs.chunk_base.cpu().numpy().view(np.uint32).astype(np.int64)
    n_tiles = len(cb) - 2
    per = np.diff(cb[:n_tiles + 1])
    q = np.quantile(per, [0.5, 0.9, 0.99])
    print(f"view {v}: D {vs.D} chunk {cb[n_tiles + 1]} chunks {cb[n_tiles]} per tile: median {q[0]:.0f} 90% {q[1]:.0f} 99% {q[2]:.0f} max {per.max()}; tiles with > 8: {(per > 8).sum()}, > 16: {(per > 16).sum()}, > 32: {(per > 32).sum()}, empty: {(per == 0).sum()}", flush=True)
