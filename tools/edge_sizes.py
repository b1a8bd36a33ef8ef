import sys; sys.path.insert(0, "/root/repo")
import torch
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.scenes import pose_tz
dev = torch.device("cuda:0")
pose = torch.from_numpy(pose_tz(3.0))[None].to(dev)
for n in (0, 1, 2):
    m = GaussianSplat(n).to(dev)
    out = m(pose, (40, 56))
    img = out["rendered_images"]
    print(n, tuple(img.shape), float(img.abs().sum()), out["positions"].shape, out["depths"].shape, out["scales"].shape)
    if n > 0:
        img.sum().backward(); print("  grads", [tuple(getattr(m, k).grad.shape) for k in ("positions", "scales", "rotations", "opacity", "features")])
    fr = m.render_frames(pose.repeat(3, 1, 1), (40, 56)); print("  frames", tuple(fr.shape))
torch.cuda.synchronize(); print("edge ok")
