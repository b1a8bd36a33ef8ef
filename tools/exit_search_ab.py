#!/usr/bin/env python3
"""dev tool: A/B of the exit search on ONE trained scene.  `train <steps>` trains the bench scene (tools/soak.py's loop) and saves the
parameters to /tmp/gsx_trained.pt; `time` loads them and times the forward of twelve orbit views (CUDA events, best of 3) -- run it under
tools/with_variant.sh for each library to compare."""
import sys, os
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
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
if sys.argv[1] == "train":
    opt = make_optimizer(m, fused_clip_norm=0.1)
    target = torch.rand(1, 3, H, W, device=dev)
    for s in range(int(sys.argv[2])):
        train_step(m, opt, poses[s % 60:s % 60 + 1], target, l1_loss, 0.1)
    torch.cuda.synchronize()
    torch.save(m.state_dict(), "/tmp/gsx_trained.pt")
    sys.exit(0)
m.load_state_dict(torch.load("/tmp/gsx_trained.pt", weights_only=True))
P = tuple(p.detach().contiguous() for p in (m.positions, m.scales, m.rotations, m.opacity, m.features))
tot = 0.0; line = []
for v in ([int(a) for a in sys.argv[2:]] or range(0, 60, 5)):
    best = 1e9
    for rep in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); outs, vs, extra = R.forward_view(P, poses[v], H, W); e1.record(); torch.cuda.synchronize()
        if rep: best = min(best, e0.elapsed_time(e1))
    st = vs.state.cpu().numpy().astype(np.uint32).ravel()
    line.append(f"v{v}: {best:.3f} ms (exit {int(st[0]) if st[0] != 0xFFFFFFFF else -1}, rounds {int(st[3]) / 512:.0f})")
    tot += best
print("; ".join(line)); print(f"sum over the views {tot:.3f} ms")
