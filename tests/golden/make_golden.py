#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference renderer.

Runs only in the build container (needs /root/reference); the GPU box never runs
this script -- it only reads the .npz files it wrote.  Nothing from the reference
is copied: the reference module is imported from where it lies, executed on seeded
inputs, and its inputs/outputs/gradients are stored as plain arrays.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

Every fixture holds
    params   positions scales rotations opacity features      (the five nn.Parameters)
    poses (B,4,4), H, W
    outputs  rendered_images positions2d depths cov2d           (forward dict)
    order    argsort(depths[0], descending)                     (gaussian_splatting.py:198,206)
    n_composited   number of Gaussians composited before the global early exit
                   (= number of torch.all calls in _render_chunk, gaussian_splatting.py:147)
    upstream d_rgb d_pos2d d_depths d_cov2d and the five parameter gradients of
             L = sum(out*upstream) over the four differentiable outputs.
"""
import os
import sys
import numpy as np
import torch

REF = os.environ.get("GSX_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from model.gaussian_splatting import GaussianSplat  # noqa: E402  (the reference itself)

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def scene_default(n, seed):
    """The reference's own init (gaussian_splatting.py:37-63) under manual_seed."""
    torch.manual_seed(seed)
    return GaussianSplat(n, chunk_size=10)


def scene_synth(n, seed, s_lo, s_hi, o_lo=0.0, o_hi=1.0, zq=None):
    """SURVEY 8d synthetic distribution (c2..c5 style), written into a reference module."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = GaussianSplat(n, chunk_size=10)
    with torch.no_grad():
        p = torch.empty(n, 3)
        p[:, :2] = torch.rand(n, 2, generator=g) * 2 - 1
        p[:, 2] = torch.rand(n, generator=g) + 0.5
        if zq is not None:
            p[:, 2] = torch.round(p[:, 2] * zq) / zq
        m.positions.copy_(p)
        m.scales.copy_(torch.exp(torch.rand(n, 3, generator=g) * (np.log(s_hi) - np.log(s_lo)) + np.log(s_lo)))
        q = torch.randn(n, 4, generator=g)
        m.rotations.copy_(q / q.norm(dim=1, keepdim=True))
        m.opacity.copy_(torch.rand(n, 1, generator=g) * (o_hi - o_lo) + o_lo)
        m.features.copy_(torch.rand(n, 3, generator=g))
    return m


def pose_tz(tz):
    p = torch.eye(4)
    p[2, 3] = tz
    return p


def orbit_pose(i, n_views=60, radius=4.0, elevation=45.0):
    """Arithmetic of utils/visualization.py:156-178 evaluated with plain torch."""
    import torch.nn.functional as F
    angle = 2 * np.pi * i / n_views
    x = radius * np.cos(angle) * np.cos(np.radians(elevation))
    y = radius * np.sin(angle) * np.cos(np.radians(elevation))
    z = radius * np.sin(np.radians(elevation))
    pos = torch.tensor([x, y, z], dtype=torch.float32)
    forward = F.normalize(torch.zeros(3) - pos, dim=0)
    up = torch.tensor([0., 0., 1.])
    right = F.normalize(torch.linalg.cross(forward, up), dim=0)
    up = F.normalize(torch.linalg.cross(right, forward), dim=0)
    pose = torch.eye(4)
    pose[:3, :3] = torch.stack([right, up, -forward], dim=1)
    pose[:3, 3] = pos
    return pose


def analytic_upstream(shape):
    """Closed-form upstream gradient (no RNG) used where storing it would be too big."""
    B, C, H, W = shape
    b, c, y, x = np.meshgrid(np.arange(B), np.arange(C), np.arange(H), np.arange(W), indexing="ij")
    return np.sin(0.37 * c + 0.11 * y + 0.07 * x + 0.5 * b).astype(np.float32)


def fp64_yardstick(model, pose, H, W, n_composited, d_rgb):
    """The reference's formulas (gaussian_splatting.py:241-257 projection, :284-292 covariances, :332-342 Jacobian,
    :124-144 compositing, stopped after the reference's own exit index) evaluated in float64 with autograd -- the reference
    module itself cannot run in double (its Jacobian is allocated as float32, :338).  NOT a second reference: a yardstick for
    the reference's own float32 rounding noise.  With the default init some Gaussians have world z within a few
    hundredths of 0; their 1/z Jacobian makes the 2x2 covariance nearly singular and the reference's float32 gradient of
    such a Gaussian is off by up to ~1e-3 of the tensor's maximum.  A parity tolerance cannot be tighter than that."""
    import torch.nn.functional as F
    f64 = torch.float64
    P = {k: getattr(model, k).detach().double().requires_grad_(True) for k in ("positions", "scales", "rotations", "opacity", "features")}
    M = pose.double()
    p = P["positions"]
    cam = torch.cat([p, torch.ones(p.shape[0], 1, dtype=f64)], 1) @ M.T
    depth, mu = cam[:, 2], cam[:, :2] / cam[:, 2:3]
    w, x, y, z = F.normalize(P["rotations"], dim=1).unbind(1)
    R = torch.stack([1 - 2*y*y - 2*z*z, 2*x*y - 2*w*z, 2*x*z + 2*w*y, 2*x*y + 2*w*z, 1 - 2*x*x - 2*z*z, 2*y*z - 2*w*x,
                     2*x*z - 2*w*y, 2*y*z + 2*w*x, 1 - 2*x*x - 2*y*y], 1).reshape(-1, 3, 3)
    RS = R @ torch.diag_embed(P["scales"])
    J = torch.zeros(p.shape[0], 2, 3, dtype=f64)
    J[:, 0, 0] = M[0, 0] / p[:, 2]; J[:, 1, 1] = M[1, 1] / p[:, 2]
    J[:, 0, 2] = -M[0, 0] * p[:, 0] / p[:, 2] ** 2; J[:, 1, 2] = -M[1, 1] * p[:, 1] / p[:, 2] ** 2
    cov2 = J @ (RS @ RS.transpose(1, 2)) @ J.transpose(1, 2)
    order = torch.argsort(depth, descending=True)
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, H, dtype=f64), torch.linspace(-1, 1, W, dtype=f64), indexing="ij")
    grid = torch.stack([gx, gy], -1).reshape(-1, 2)
    C = torch.zeros(3, H * W, dtype=f64); A = torch.zeros(H * W, dtype=f64)
    for k in range(n_composited):
        i = order[k]
        d = grid - mu[i]
        inv = torch.inverse(cov2[i] + 1e-6 * torch.eye(2, dtype=f64))
        al = torch.exp(-0.5 * ((d @ inv) * d).sum(-1)) * P["opacity"][i, 0]
        C = C + (1 - A) * P["features"][i][:, None] * al
        A = A + (1 - A) * al
    (C.reshape(3, H, W) * d_rgb.double()).sum().backward()
    return {k: v.grad.numpy() for k, v in P.items()}, order.numpy()


def run_case(name, model, poses, H, W, seed_up, rgb_only=False, subsample=None, analytic=False, yardstick=False):
    n = model.num_gaussians
    B = poses.shape[0]
    # count the early-exit tests the reference performs
    calls = {"n": 0}
    real_all = torch.all

    def counting_all(*a, **k):
        calls["n"] += 1
        return real_all(*a, **k)

    torch.all = counting_all
    try:
        out = model(poses, (H, W))
    finally:
        torch.all = real_all
    g = torch.Generator(device="cpu").manual_seed(seed_up)
    if analytic:
        d_rgb = torch.from_numpy(analytic_upstream((B, 3, H, W)))
    else:
        d_rgb = torch.randn(B, 3, H, W, generator=g)
    if rgb_only:
        d_pos = torch.zeros(B, n, 2)
        d_dep = torch.zeros(B, n)
        d_cov = torch.zeros(B, n, 2, 2)
    else:
        d_pos = torch.randn(B, n, 2, generator=g)
        d_dep = torch.randn(B, n, generator=g)
        d_cov = torch.randn(B, n, 2, 2, generator=g)
    L = (out["rendered_images"] * d_rgb).sum() + (out["positions"] * d_pos).sum() \
        + (out["depths"] * d_dep).sum() + (out["scales"] * d_cov).sum()
    L.backward()
    order = torch.argsort(out["depths"], dim=1, descending=True)[0]
    ys = torch.linspace(-1, 1, H)
    xs = torch.linspace(-1, 1, W)
    img = out["rendered_images"].detach().numpy()
    fx = dict(
        torch_version=np.array(torch.__version__),
        H=np.int32(H), W=np.int32(W),
        poses=poses.numpy().astype(np.float32),
        positions=model.positions.detach().numpy(), scales=model.scales.detach().numpy(),
        rotations=model.rotations.detach().numpy(), opacity=model.opacity.detach().numpy(),
        features=model.features.detach().numpy(),
        out_pos2d=out["positions"].detach().numpy(), out_depths=out["depths"].detach().numpy(),
        out_cov2d=out["scales"].detach().numpy(),
        order=order.numpy().astype(np.int64), n_composited=np.int64(calls["n"]),
        grid_x=xs.numpy(), grid_y=ys.numpy(),
        d_pos2d=d_pos.numpy(), d_depths=d_dep.numpy(), d_cov2d=d_cov.numpy(),
        g_positions=model.positions.grad.numpy(), g_scales=model.scales.grad.numpy(),
        g_rotations=model.rotations.grad.numpy(), g_opacity=model.opacity.grad.numpy(),
        g_features=model.features.grad.numpy(),
        analytic_upstream=np.int32(1 if analytic else 0),
    )
    if yardstick:
        assert rgb_only and B == 1
        g64, order64 = fp64_yardstick(model, poses[0], H, W, calls["n"], d_rgb[0])
        assert np.array_equal(order64, order.numpy()), "float64 depth order differs: the yardstick would composite another scene"
        for k in ("positions", "scales", "rotations", "opacity", "features"):
            fx["ref_noise_" + k] = np.float64(np.abs(fx["g_" + k].astype(np.float64) - g64[k]).max() / max(np.abs(g64[k]).max(), 1e-30))
    if subsample:
        fx["out_rgb_sub"] = img[:, :, ::subsample, ::subsample].copy()
        fx["subsample"] = np.int32(subsample)
        fx["out_rgb_sum"] = img.astype(np.float64).sum(axis=(2, 3))
    else:
        fx["out_rgb"] = img
    if not analytic:
        fx["d_rgb"] = d_rgb.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    if yardstick:
        print("   reference float32 noise vs float64 yardstick:", {k[10:]: f"{float(v):.1e}" for k, v in fx.items() if k.startswith("ref_noise_")})
    print(f"{name}: N={n} B={B} {W}x{H} composited={calls['n']} "
          f"max|rgb|={np.abs(img).max():.4f} -> {os.path.getsize(path)/1024:.0f} KiB", flush=True)


def run_train_case(name, model, poses, H, W, seed_target, steps=3):
    """the trainer step of /root/reference/training/trainer.py:133-178 around the
    reference render -- zero_grad, forward, loss, backward, clip_grad_norm_(0.1), Adam with the five per-tensor
    learning rates of configs/default_config.yaml:12-16 in the order of trainer.py:49-61 -- with the importable-math
    subset of model/loss.py:89-109 as the loss (clamped MSE + scale and opacity regularisers; the VGG/LPIPS terms
    need network weights).  Stores parameters before and after every step, the losses and the first step's
    clipped gradients."""
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(seed_target)
    target = torch.rand(1, 3, H, W, generator=g)
    poses_t = poses if torch.is_tensor(poses) else torch.from_numpy(poses)
    fx = {k: getattr(model, k).detach().numpy().copy() for k in ("positions", "scales", "rotations", "opacity", "features")}
    opt = torch.optim.Adam([
        {'params': model.positions, 'lr': 1e-4}, {'params': model.features, 'lr': 1e-3},
        {'params': model.scales, 'lr': 1e-4}, {'params': model.rotations, 'lr': 1e-4},
        {'params': model.opacity, 'lr': 1e-3}])
    losses, norms = [], []
    for s in range(steps):
        opt.zero_grad()
        out = model(poses_t, (H, W))
        c = torch.clamp(out['rendered_images'], 1e-8, 1.0 - 1e-8)
        o = out['opacity']
        loss = F.mse_loss(c, target) + 5e-4 * torch.mean(torch.norm(out['scales'], dim=-1)) + 5e-3 * torch.mean(o * (1 - o))
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)))
        if s == 0:
            for k in ("positions", "scales", "rotations", "opacity", "features"):
                fx["clipped_g_" + k] = getattr(model, k).grad.detach().numpy().copy()
        opt.step()
        losses.append(float(loss))
        for k in ("positions", "scales", "rotations", "opacity", "features"):
            fx[f"after{s + 1}_{k}"] = getattr(model, k).detach().numpy().copy()
    fx.update(poses=poses_t.numpy(), H=np.int64(H), W=np.int64(W), target=target.numpy(), losses=np.array(losses, np.float64),
              grad_norms=np.array(norms, np.float64), steps=np.int64(steps), torch_version=np.array(torch.__version__))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name}: losses {losses} grad norms {norms}")


def tilted_pose(tz, yaw_deg, tx=0.0, ty=0.0):
    """world->camera matrix as the reference applies it (gaussian_splatting.py:250-253): a yaw about the y axis and a shift"""
    a = np.radians(yaw_deg)
    p = torch.eye(4)
    p[0, 0], p[0, 2], p[2, 0], p[2, 2] = float(np.cos(a)), float(np.sin(a)), float(-np.sin(a)), float(np.cos(a))
    p[0, 3], p[1, 3], p[2, 3] = tx, ty, tz
    return p


# Scenes in which the reference's GLOBAL early exit (gaussian_splatting.py:147-148) fires: its default init puts world z
# near 0 for many Gaussians, the 1/z Jacobian makes those splats screen-sized, and every pixel saturates after a few
# hundred of them.  (name, N, seed, pose, H, W): different sizes, seeds, image shapes and cameras, so that the exit index
# is pinned by the reference itself and not only by the CPU restatement.
EXIT_CASES = [
    ("exit0_n200_32x24", 200, 40, pose_tz(3.0), 24, 32),
    ("exit1_n400_48x32", 400, 41, pose_tz(2.5), 32, 48),
    ("exit2_n600_64x48", 600, 42, pose_tz(3.0), 48, 64),
    ("exit3_n800_40x40", 800, 43, pose_tz(4.0), 40, 40),
    ("exit4_n1000_33x17", 1000, 44, pose_tz(2.0), 17, 33),
    ("exit5_n1200_64x36", 1200, 45, tilted_pose(3.0, 20.0), 36, 64),
    ("exit6_n1500_48x48", 1500, 46, tilted_pose(3.5, -35.0, 0.2, -0.1), 48, 48),
    ("exit7_n2000_64x48", 2000, 47, pose_tz(3.0), 48, 64),
    ("exit8_n500_16x16", 500, 48, tilted_pose(2.2, 10.0, -0.3, 0.2), 16, 16),
    ("exit9_n1000_56x40", 1000, 49, tilted_pose(5.0, 60.0), 40, 56),
    ("exit10_n300_24x32", 300, 50, pose_tz(1.5), 32, 24),
    ("exit11_n1600_64x48", 1600, 51, tilted_pose(2.8, -15.0, 0.0, 0.3), 48, 64),
]


# The same exit with opacities in (1, 3]: alpha exceeds 1, a pixel's accumulated alpha overshoots and comes back UNDER 0.99,
# so "every pixel > 0.99 at once" (the reference's test) is no longer "the last pixel's first crossing" -- on these two
# scenes the first-crossing rule gives 200 and 181 where the reference composites 207 and 198.
EXIT_OPACITY_CASES = [
    ("exitop0_n500_32x32", 500, 60, pose_tz(3.0), 32, 32),
    ("exitop1_n450_24x40", 450, 68, pose_tz(3.0), 40, 24),
]


def scene_default_high_opacity(n, seed):
    m = scene_default(n, seed)
    g = torch.Generator(device="cpu").manual_seed(seed + 1000)
    with torch.no_grad():
        m.opacity.copy_(1.0 + 2.0 * torch.rand(n, 1, generator=g))
    return m


def run_merge_case(name, n, seed, threshold):
    """merge_nearby_gaussians (:477-528) on a default-init scene.  Besides inputs and outputs the fixture keeps the
    reference's own pair list (the expressions of :491-497): above 25 rows torch.cdist takes its matmul form, whose
    self-distances round above zero for about a tenth of the points -- the reference then "merges" such a point with
    itself and removes it.  `self_pairs` counts them; `margin` is how far (float64) the nearest off-diagonal distance
    stays from the threshold, so a test can tell a real disagreement from that rounding."""
    torch.manual_seed(seed)
    model = GaussianSplat(n)
    before = {k: getattr(model, k).data.clone().numpy() for k in ("positions", "scales", "rotations", "opacity", "features")}
    d = torch.cdist(model.positions.data, model.positions.data)
    ii, jj = torch.where(torch.triu(d > 0) & torch.triu(d < threshold))
    d64 = torch.cdist(model.positions.data.double(), model.positions.data.double())
    off = ~torch.eye(n, dtype=torch.bool)
    margin = float((d64[off] - threshold).abs().min())
    model.merge_nearby_gaussians(threshold)
    out = {"threshold": np.float32(threshold), "pairs_i": ii.numpy().astype(np.int32), "pairs_j": jj.numpy().astype(np.int32),
           "self_pairs": np.int32(int((ii == jj).sum())), "margin": np.float64(margin), "n_after": np.int32(model.num_gaussians)}
    for k, v in before.items():
        out["in_" + k] = v
        out["out_" + k] = getattr(model, k).data.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {n} -> {model.num_gaussians}, {len(ii)} pairs ({int((ii == jj).sum())} self), margin {margin:.2e}")


MERGE_CASES = [("merge_n24", 24, 1, 0.3), ("merge_n300", 300, 2, 0.1), ("merge_n1500", 1500, 4, 0.03)]


def main():
    if "--merge-only" in sys.argv:
        for c in MERGE_CASES:
            run_merge_case(*c)
        return
    if "--exit-only" in sys.argv:
        for i, (name, n, seed, pose, H, W) in enumerate(EXIT_CASES):
            run_case(name, scene_default(n, seed), pose[None], H, W, 200 + i, rgb_only=True, yardstick=True)
        return
    if "--exit-opacity-only" in sys.argv:
        for i, (name, n, seed, pose, H, W) in enumerate(EXIT_OPACITY_CASES):
            run_case(name, scene_default_high_opacity(n, seed), pose[None], H, W, 300 + i, rgb_only=True, yardstick=True)
        return
    if "--train-only" in sys.argv:
        run_train_case("train3_400_32x24", scene_synth(400, 31, 0.005, 0.03), pose_tz(2.0)[None], 24, 32, 32)
        return
    big = "--big" in sys.argv
    I3 = pose_tz(3.0)[None]
    run_case("n1_8x8", scene_default(1, 11), I3, 8, 8, 101)
    run_case("n2_16x12", scene_default(2, 12), I3, 12, 16, 102)
    run_case("n64_32x24", scene_default(64, 13), I3, 24, 32, 103)
    run_case("n300_32x24", scene_default(300, 0), I3, 24, 32, 104)            # early exit fires
    run_case("n300_rgbonly", scene_default(300, 0), I3, 24, 32, 105, rgb_only=True)
    run_case("c2like_2000_64x48", scene_synth(2000, 2, 0.002, 0.01), pose_tz(2.0)[None], 48, 64, 106)
    run_case("c3like_1500_48x48_orbit", scene_synth(1500, 5, 0.001, 0.005), orbit_pose(15)[None], 48, 48, 107)
    run_case("c5like_800_64x36", scene_synth(800, 8, 0.01, 0.1, zq=64.0), pose_tz(2.0)[None], 36, 64, 108)
    # NOTE: no B>1 fixture exists because the reference cannot run B>1: `diff @ cov_inv.unsqueeze(1)`
    # (gaussian_splatting.py:132) broadcasts (B,P,2)@(B,1,2,2) to (B,B,P,2) and the following
    # .view(B,H,W) raises RuntimeError for every B>1 (observed here with B=2).
    run_case("opac_out_of_range_64", scene_synth(64, 22, 0.01, 0.05, o_lo=-0.3, o_hi=1.4), pose_tz(2.0)[None], 24, 32, 110)
    run_case("h1_w5_n8", scene_synth(8, 23, 0.05, 0.2), pose_tz(2.0)[None], 1, 5, 111)
    run_case("odd_33x17_n120", scene_synth(120, 24, 0.01, 0.08), pose_tz(1.5)[None], 17, 33, 112)
    run_train_case("train3_400_32x24", scene_synth(400, 31, 0.005, 0.03), pose_tz(2.0)[None], 24, 32, 32)
    for i, (name, n, seed, pose, H, W) in enumerate(EXIT_CASES):
        run_case(name, scene_default(n, seed), pose[None], H, W, 200 + i, rgb_only=True, yardstick=True)
    for i, (name, n, seed, pose, H, W) in enumerate(EXIT_OPACITY_CASES):
        run_case(name, scene_default_high_opacity(n, seed), pose[None], H, W, 300 + i, rgb_only=True, yardstick=True)
    for c in MERGE_CASES:
        run_merge_case(*c)
    if big:
        # BASELINE config 1: 5000 default-init Gaussians, 400x300 (README.md:85-94); ~16 GB RSS, ~10 s
        run_case("config1_5000_400x300", scene_default(5000, 0), I3, 300, 400, 0,
                 rgb_only=True, subsample=4, analytic=True)


if __name__ == "__main__":
    main()
