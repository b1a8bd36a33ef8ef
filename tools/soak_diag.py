import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
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
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 900):
    train_step(m, opt, poses[s % 60:s % 60 + 1], target, l1_loss, 0.1)
torch.cuda.synchronize()
with torch.no_grad():
    for v in (0, 10, 15, 30):
        out = m(poses[v:v + 1], (H, W))
        torch.cuda.synchronize()
        st = m._last_state.cpu().numpy().astype(np.uint32)
        a = out["alpha"][0] if "alpha" in out else None
        rng = None
        print("view", v, "state", [hex(int(x)) for x in st.ravel()[:4]], "n_comp", m.last_n_composited(), "alpha min/max", float(a.min()), float(a.max()), "frac alpha>0.99", float((a > 0.99).float().mean()))
