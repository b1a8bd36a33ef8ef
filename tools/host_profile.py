import sys, os, time
import torch
sys.path.insert(0, "/root/repo")
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_ssim_loss
dev = torch.device("cuda:0")
params, H, W = config_scene("c2")
n = params["positions"].shape[0]
model = GaussianSplat(n); model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); model = model.to(dev)
opt = make_optimizer(model, fused_clip_norm=0.1)
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
target = torch.rand(1, 3, H, W, device=dev)
for s in range(120): train_step(model, opt, poses[s % 60:s % 60 + 1], target, l1_ssim_loss, 0.1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(300): train_step(model, opt, poses[s % 60:s % 60 + 1], target, l1_ssim_loss, 0.1)
torch.cuda.synchronize(); print("c2 ms/step:", round((time.perf_counter() - t0) / 300 * 1e3, 4))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for s in range(200): train_step(model, opt, poses[s % 60:s % 60 + 1], target, l1_ssim_loss, 0.1)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
