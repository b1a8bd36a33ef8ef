import sys, time, torch
sys.path.insert(0, "/root/repo")
from gaussian_splatting_amd import GaussianSplat, render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
dev = torch.device("cuda:0")
params, H, W = config_scene("c3")
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
out = torch.empty(60, 3, H, W, device=dev)
for S in (1, 2, 3, 4, 6):
    R.render_orbit(*P, poses, (H, W), out=out, streams=S)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): R.render_orbit(*P, poses, (H, W), out=out, streams=S)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 180
    print(S, "streams:", round(dt * 1e3, 4), "ms/view", round(H * W / dt / 1e6, 1), "Mpix/s")
