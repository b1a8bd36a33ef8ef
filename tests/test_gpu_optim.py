"""GPU tier: fused clip_grad_norm_ + Adam (gsx_clip_adam_step) against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam -- the reference trainer's step
(/root/reference/training/trainer.py:49-61,169-175) -- on the same parameters and gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SHAPES = [(1000, 3), (1000, 3), (1000, 3), (1000, 4), (1000, 1)]
LRS = [1e-4, 1e-3, 1e-4, 1e-4, 1e-3]


def _make(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in SHAPES]


@pytest.mark.parametrize("max_norm", [0.1, 0.0, 1e9])
def test_fused_clip_adam_matches_torch(max_norm):
    from gaussian_splatting_amd.optim import FusedClipAdam
    dev = torch.device("cuda:0")
    pa, pb = _make(dev, 1), _make(dev, 1)
    oa = torch.optim.Adam([{'params': p, 'lr': lr} for p, lr in zip(pa, LRS)])
    ob = FusedClipAdam([{'params': p, 'lr': lr} for p, lr in zip(pb, LRS)], max_grad_norm=max_norm)
    g = torch.Generator(device="cpu").manual_seed(2)
    for it in range(6):
        grads = [(torch.randn(*s, generator=g) * (10.0 if it % 2 else 0.01)).to(dev) for s in SHAPES]
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone(); q.grad = gr.clone()
        if max_norm > 0:
            tn = torch.nn.utils.clip_grad_norm_(pa, max_norm)
        oa.step(); ob.step()
        if max_norm > 0:
            norm_coef = ob.last_grad_norm().cpu().numpy()
            assert abs(norm_coef[0] - float(tn)) <= 1e-5 * float(tn)
            for p, q in zip(pa, pb):      # gradients are clipped in place, like clip_grad_norm_
                assert (p.grad - q.grad).abs().max().item() <= 1e-5 * p.grad.abs().max().item()
        for p, q in zip(pa, pb):
            err = (p.data - q.data).abs().max().item()
            assert err <= 2e-7 * max(1.0, p.data.abs().max().item()), (it, err)
    # optimizer state interchanges with torch.optim.Adam (checkpoints, trainer.py:242-271)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa['state'].keys() == sb['state'].keys()
    for k in sa['state']:
        assert float(sa['state'][k]['step']) == float(sb['state'][k]['step']) == 6
        for name in ('exp_avg', 'exp_avg_sq'):
            a, b = sa['state'][k][name], sb['state'][k][name]
            # moments scale with the clip coefficient (its fp32 norm differs from torch's by ~1e-6 rel; squared in exp_avg_sq)
            assert (a - b).abs().max().item() <= 5e-5 * a.abs().max().item(), name
    oc = torch.optim.Adam([{'params': p, 'lr': lr} for p, lr in zip(_make(dev, 1), LRS)])
    oc.load_state_dict(sb)


def test_fused_optimizer_rejects_cpu():
    from gaussian_splatting_amd.optim import FusedClipAdam
    p = torch.nn.Parameter(torch.zeros(4)); p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        FusedClipAdam([p]).step()


@pytest.mark.parametrize("shape,offset", [((1, 3, 37, 53), 0), ((1, 3, 64, 64), 0), ((3, 31, 17), 1)])
@pytest.mark.parametrize("kind", ["l1", "clamped_mse"])
def test_fused_image_loss_matches_torch(kind, shape, offset):
    """gsx_image_loss_fwd/bwd against the torch ops they replace ((x - t).abs().mean(); mse_loss(clamp(x, 1e-8, 1 - 1e-8), t),
    /root/reference/model/loss.py:135-137).  Tolerance: loss 1e-6 relative (summation order), gradient 1e-6 relative;
    the L1 gradient (sign/n) is exact."""
    from gaussian_splatting_amd import loss as gl
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    n = int(np.prod(shape))
    base = torch.randn(n + offset, generator=g) * 0.7 + 0.4        # values on both sides of the clamp range
    tbase = torch.rand(n + offset, generator=g)
    base[5:9] = tbase[5:9]                                         # exact ties: sign(0) = 0
    x = base.to(dev)[offset:].view(shape)                          # offset 1 -> pointer not 16-byte aligned
    t = tbase.to(dev)[offset:].view(shape)
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    if kind == "l1":
        la = (xa - t).abs().mean(); lb = gl.l1(xb, t)
    else:
        la = torch.nn.functional.mse_loss(xa.clamp(1e-8, 1 - 1e-8), t); lb = gl.clamped_mse(xb, t)
    (la * 3.0).backward(); (lb * 3.0).backward()
    assert abs(float(la.detach()) - float(lb.detach())) <= 1e-6 * abs(float(la.detach()))
    if kind == "l1":
        assert torch.equal(xa.grad, xb.grad)
    else:
        assert (xa.grad - xb.grad).abs().max().item() <= 1e-6 * xa.grad.abs().max().item()


def _torch_ssim(img1, img2, window_size=11, sigma=1.5):
    """The usual splatting SSIM in plain torch fp32 (per-channel 11x11 Gaussian window, zero padding): the restatement the
    fused kernels are checked against (the reference has no SSIM loss; parity unpinned by it)."""
    import math
    C = img1.shape[-3]
    gauss = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    gauss = (gauss / gauss.sum()).unsqueeze(1)
    w2 = (gauss @ gauss.t()).unsqueeze(0).unsqueeze(0)
    window = w2.expand(C, 1, window_size, window_size).contiguous().to(img1.device)
    conv = lambda z: torch.nn.functional.conv2d(z, window, padding=window_size // 2, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq; s2 = conv(img2 * img2) - mu2_sq; s12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


@pytest.mark.parametrize("shape", [(1, 3, 37, 53), (2, 3, 64, 48), (1, 3, 16, 16), (1, 1, 5, 70)])
@pytest.mark.parametrize("lam", [0.2, 1.0])
def test_fused_l1_ssim_matches_torch(shape, lam):
    """gsx_l1_ssim_fwd/bwd (BASELINE config 2's "L1/SSIM") against conv2d + autograd in fp32, with the backward run in
    float64 as the yardstick.  Tolerances: loss 2e-6 absolute; gradient 2e-5 of its max (fp32 cancellation in
    sigma = E[x^2] - mu^2 is the same on both sides; the fp32 torch backward is itself ~1e-5 from the fp64 one)."""
    from gaussian_splatting_amd import loss as gl
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(21)
    x = torch.rand(shape, generator=g); t = (x + 0.25 * torch.randn(shape, generator=g)).clamp(0, 1)
    t[..., :3, :3] = x[..., :3, :3]                               # exact ties in the L1 term
    xa = x.to(dev).requires_grad_(True); xb = x.to(dev).requires_grad_(True); xd = x.double().to(dev).requires_grad_(True)
    td = t.to(dev)
    la = (1 - lam) * (xa - td).abs().mean() + lam * (1 - _torch_ssim(xa, td))
    ld = (1 - lam) * (xd - td.double()).abs().mean() + lam * (1 - _torch_ssim_f64(xd, td.double()))
    lb, terms = gl.l1_ssim(xb, td, lam, return_terms=True)
    (la * 2.0).backward(); (lb * 2.0).backward(); (ld * 2.0).backward()
    assert abs(float(lb.detach()) - float(ld.detach())) <= 2e-6, (float(lb.detach()), float(ld.detach()), float(la.detach()))
    assert abs(float(terms[1]) - float((xd - td.double()).abs().mean())) <= 1e-6
    assert abs(float(terms[2]) - (1 - (float(ld.detach()) - (1 - lam) * float(terms[1])) / lam)) <= 1e-5
    ref = xd.grad.float()
    scale = ref.abs().max().item()
    err = (xb.grad - ref).abs().max().item(); err_torch = (xa.grad - ref).abs().max().item()
    assert err <= 2e-5 * scale, (err, err_torch, scale)


def _torch_ssim_f64(img1, img2):
    import math
    C = img1.shape[-3]
    gauss = torch.tensor([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    gauss = (gauss / gauss.sum()).double().unsqueeze(1)           # the fp32 window, applied in double
    window = (gauss @ gauss.t()).expand(C, 1, 11, 11).contiguous().to(img1.device)
    conv = lambda z: torch.nn.functional.conv2d(z, window, padding=5, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    s1 = conv(img1 * img1) - mu1 * mu1; s2 = conv(img2 * img2) - mu2 * mu2; s12 = conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


def test_config2_step_with_l1_ssim():
    """BASELINE config 2 in miniature (its distribution at 3000 Gaussians, 96x80): one trainer step with the fused L1/SSIM
    criterion against the same step with the torch restatement of the criterion on the same render path."""
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.scenes import synth_scene, CONFIGS
    from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_ssim_loss
    dev = torch.device("cuda:0")
    _, _, _, seed, s_lo, s_hi, zq = CONFIGS["c2"]
    params = synth_scene(3000, seed, s_lo * 4, s_hi * 4, zq)          # scales x4: the image is 1/8 of 800 px wide
    H, W = 80, 96
    pose = torch.eye(4); pose[2, 3] = 2.0
    target = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    torch_loss = lambda out, tgt: 0.8 * (out['rendered_images'] - tgt).abs().mean() + 0.2 * (1 - _torch_ssim(out['rendered_images'], tgt))
    res = []
    for fn in (l1_ssim_loss, torch_loss):
        m = GaussianSplat(3000)
        m.load_state_dict({k: torch.from_numpy(params[k]) for k in ("positions", "scales", "rotations", "opacity", "features")})
        m = m.to(dev)
        opt = make_optimizer(m, fused_clip_norm=0.1, write_clipped_grad=True)
        loss = train_step(m, opt, pose[None].to(dev), target, fn, max_grad_norm=0.1)
        res.append((float(loss), {k: v.grad.clone() for k, v in m.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) <= 2e-6
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item(), k


def test_fused_losses_move_a_host_target():
    from gaussian_splatting_amd import loss as gl
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.rand(1, 3, 20, 28, generator=g).to(dev); t = torch.rand(1, 3, 20, 28, generator=g)
    assert float(gl.l1(x, t)) == float(gl.l1(x, t.to(dev)))
    assert float(gl.l1_ssim(x, t)) == float(gl.l1_ssim(x, t.to(dev)))


def test_fused_image_loss_rejects_cpu():
    from gaussian_splatting_amd import loss as gl
    with pytest.raises(RuntimeError):
        gl.l1(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4))


@pytest.mark.parametrize("fused", [True, False])
def test_trainer_three_steps_match_reference(fused):
    """three trainer steps (zero_grad, forward, loss, backward, clip_grad_norm_(0.1), Adam with the
    reference's five learning rates) on the reference itself (tests/golden/train3_400_32x24.npz, made by
    make_golden.py::run_train_case) against the same steps through this path (fused HIP clip+Adam and loss, or torch's).
    Bars: losses 1e-5 relative; first step's clipped gradients 1e-3 of the tensor's max; parameters after step s:
    EVERY element whose reference gradient is well above Adam's eps within 2 % of lr per step, and 99 % of ALL elements
    within 2 % of lr PER STEP (s x 2 %: the differences add up from step to step).  Adam turns a gradient g into a step
    lr g / (|g| + 1e-8): an element whose reference gradient is of the order of eps (contributions of pairs beyond the
    footprint cut, where exp(-q/2) < 2^-18: ~1e-9 after the 1/(3HW) of the loss) moves by a fraction of lr in the reference
    where this path, with an exactly zero gradient, leaves it still -- ill-conditioned in the reference too, bounded by lr
    per step.  Measured on this tree (tools/trainer_fixture_q99.py -> profiles/r3_trainer_fixture_q99.json, cut 2^-20),
    q99 over all elements in units of lr after steps 1 / 2 / 3: scales (the worst tensor) 1.1 / 3.2 / 4.6 %, positions
    0.1 / 0.7 / 1.0 %, rotations 0.06 / 0.3 / 0.5 %, opacity and colour < 0.02 %; the well-conditioned elements (84-94 % of a
    tensor; 63 % of the rotations) stay within 0.06 % of lr.  (Round 1's 2^-24 cut: scales 0.2 % after step 2; a 2^-18 cut:
    6.6 % after step 1 -- which is why the cut stopped at 2^-20.)"""
    import os
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.trainer import make_optimizer, train_step, reference_regularised_mse, DEFAULT_LRS
    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "train3_400_32x24.npz")))
    dev = torch.device("cuda:0")
    names = ["positions", "scales", "rotations", "opacity", "features"]
    m = GaussianSplat(400)
    m.load_state_dict({k: torch.from_numpy(fx[k]) for k in names})
    m = m.to(dev)
    opt = make_optimizer(m, fused_clip_norm=0.1 if fused else None, write_clipped_grad=True)
    poses = torch.from_numpy(fx["poses"]).to(dev); target = torch.from_numpy(fx["target"]).to(dev)
    report = {}
    for s in range(int(fx["steps"])):
        loss = train_step(m, opt, poses, target, reference_regularised_mse, max_grad_norm=0.1)
        assert abs(float(loss) - fx["losses"][s]) <= 1e-5 * abs(fx["losses"][s]), (s, float(loss), fx["losses"][s])
        if s == 0:
            for k in names:
                g, gr = getattr(m, k).grad.cpu().numpy(), fx["clipped_g_" + k]
                assert np.abs(g - gr).max() <= 1e-3 * np.abs(gr).max(), k
        for k in names:
            d = np.abs(getattr(m, k).detach().cpu().numpy() - fx[f"after{s + 1}_{k}"]).ravel()
            lr = DEFAULT_LRS[k]
            assert np.quantile(d, 0.99) <= 0.02 * lr * (s + 1), (s, k, float(np.quantile(d, 0.99)), lr)
            assert d.max() <= (s + 1) * lr * 1.01, (s, k, float(d.max()))
            # the bound above is vacuous for Adam (nothing moves further than lr per step), so: EVERY element whose
            # reference gradient is well above Adam's eps (1e-8) -- where the update -lr g / (|g| + eps) is a well-conditioned
            # function of the gradient -- must land within 2 % of lr per step of the reference's parameter
            wc = np.abs(fx["clipped_g_" + k]).ravel() > 1e-6
            report[(s, k)] = (float(wc.mean()), float(d[wc].max() / lr) if wc.any() else 0.0)
            assert d[wc].max() <= 0.02 * lr * (s + 1), (s, k, report[(s, k)])
    assert np.mean([v[0] for v in report.values()]) > 0.5, report          # ... and that is most of the elements


def test_maintenance_kernels_match_cpu():
    """SURVEY 8f N3 on the GPU (gsx_keep_flags + gsx_scan_u32 + gsx_compact_rows, gsx_clamp_inplace) against the same
    module methods on CPU tensors (torch ops restating gaussian_splatting.py:346-369, :404-418): bit-equal, order kept,
    NaN handled like torch (NaN opacity is pruned, NaN scale stays NaN)."""
    from gaussian_splatting_amd import GaussianSplat
    dev = torch.device("cuda:0")
    for n in (1, 257, 100_003):
        torch.manual_seed(n)
        a = GaussianSplat(n)
        with torch.no_grad():
            a.scales.data[::5] = 3.0; a.scales.data[1::7] = 1e-6
            if n > 10:
                a.scales.data[3, 1] = float("nan"); a.opacity.data[5] = float("nan")
        b = GaussianSplat(n)
        b.load_state_dict(a.state_dict())
        b = b.to(dev)
        a.optimize_scales(); b.optimize_scales()
        assert torch.equal(a.scales.data, b.scales.data.cpu(), ) or (torch.isnan(a.scales.data) == torch.isnan(b.scales.data.cpu())).all()
        sa, sb = a.scales.data, b.scales.data.cpu()
        ok = ~torch.isnan(sa)
        assert torch.equal(sa[ok], sb[ok]) and torch.equal(torch.isnan(sa), torch.isnan(sb))
        for thr in (0.5, -1.0, 2.0):
            a2 = GaussianSplat(n); a2.load_state_dict(a.state_dict())
            b2 = GaussianSplat(n); b2.load_state_dict(a.state_dict()); b2 = b2.to(dev)
            a2.prune_gaussians(thr); b2.prune_gaussians(thr)
            assert a2.num_gaussians == b2.num_gaussians
            for k in ("positions", "rotations", "opacity", "features"):
                x, y = getattr(a2, k).data, getattr(b2, k).data.cpu()
                assert x.shape == y.shape and torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), (n, thr, k)


def test_device_prefetcher_feeds_the_trainer_step(tmp_path):
    """items staged through pinned memory on a side stream arrive intact and drive a trainer step."""
    from test_host_cpu import _write_scene
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.dataset import SceneDataset, DevicePrefetcher
    from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss
    root = str(tmp_path / "scene")
    _write_scene(root, n=4, H=32, W=48)
    ds = SceneDataset(root, "train")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = GaussianSplat(500).to(dev)
    opt = make_optimizer(m, fused_clip_norm=0.1)
    seen = 0
    for img, pose, path in DevicePrefetcher(ds, dev):
        ref = ds[seen]
        assert img.is_cuda and torch.equal(img[0].cpu(), ref["image"]) and torch.equal(pose[0].cpu(), ref["camera_pose"])
        loss = train_step(m, opt, pose, img, l1_loss, 0.1)
        assert torch.isfinite(loss)
        seen += 1
    assert seen == 4


@pytest.mark.parametrize("loss_name", ["l1", "l1_ssim", "clamped_mse"])
def test_fused_train_step_equals_modular_step(loss_name):
    """gsx_train_step (one library call per step: trainer.FusedTrainStep, which train_step routes to by itself) against the
    modular step (module forward, fused loss, autograd backward, FusedClipAdam.step): same kernels in the same order, so
    losses, gradients, parameters and optimizer state must be EQUAL after every step -- across a capacity growth (first
    step of a scene), the 32 -> 24-bit switch of the depth sort and a view that breaks the 24-bit assumption."""
    from gaussian_splatting_amd import GaussianSplat, trainer as T, render as R
    from gaussian_splatting_amd.scenes import create_orbit_poses
    dev = torch.device("cuda:0")
    loss_fn = {"l1": T.l1_loss, "l1_ssim": T.l1_ssim_loss, "clamped_mse": T.clamped_mse_loss}[loss_name]
    H, W, n = 96, 128, 20000
    poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
    near = poses[7:8].clone(); near[0, :3, 3] *= 0.55                   # camera at radius 2.2: depths straddle 2.0, the keys' top byte varies
    views = [poses[0:1], poses[13:14], poses[14:15], near.contiguous(), poses[29:30], poses[30:31]]
    target = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    runs = {}
    for fused in (False, True):
        torch.manual_seed(11)
        m = GaussianSplat(n)
        with torch.no_grad():
            m.scales.mul_(0.3)
            m.positions[:, 2] = m.positions[:, 2].abs() * 0.8 + 0.2     # the Jacobian divides by WORLD z (reference quirk): keep it off 0
        m = m.to(dev)
        opt = T.make_optimizer(m, fused_clip_norm=0.1)
        R._scratch.capacity[dev] = 4096; R._scratch.depth_bits.pop(dev, None)          # both start cold, with a block the first view outgrows
        T.FUSED_STEP = fused
        try:
            log = []
            for v in views:
                loss = T.train_step(m, opt, v, target, loss_fn, 0.1)
                log.append((loss.clone(), [p.grad.clone() for p in m.parameters()], [p.detach().clone() for p in m.parameters()],
                            R.last_instance_count(), m.last_n_composited(), R._scratch.depth_bits.get(dev)))
            assert (getattr(opt, "_fused_step", None) is not None) == fused
            runs[fused] = (log, opt.state_dict())
        finally:
            T.FUSED_STEP = True
    for s, (a, b) in enumerate(zip(runs[False][0], runs[True][0])):
        assert torch.equal(a[0], b[0]), (s, float(a[0]), float(b[0]))
        for x, y in zip(a[1], b[1]):
            assert torch.equal(x, y), s
        for x, y in zip(a[2], b[2]):
            assert torch.equal(x, y), s
        assert a[3:] == b[3:]
        assert torch.isfinite(a[0]) and all(torch.isfinite(x).all() for x in a[2]), s
    bits = [e[5] for e in runs[True][0]]
    assert bits[1] == 24 and bits[3] == 32 and bits[-1] == 24, bits     # narrowed after a quiet view, widened by the near one, narrowed again
    assert R._scratch.cap(dev, n) > 4096 and max(e[3] for e in runs[True][0]) > 4096      # the first views outgrew the initial block
    R._scratch.capacity.pop(dev, None)
    sa, sb = runs[False][1], runs[True][1]
    assert sa["param_groups"] == sb["param_groups"]
    for k in sa["state"]:
        for f in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(sa["state"][k][f], sb["state"][k][f]), (k, f)
    assert float(runs[True][0][-1][0]) != float(runs[True][0][0][0])     # the losses are per step, not one buffer seen five times


@pytest.mark.parametrize("fixture", ["exit3_n800_40x40", "exitop0_n500_32x32", "exit11_n1600_64x48"])
def test_fused_step_global_exit_equals_modular_step(fixture):
    """The one-call step enqueues the global-exit path only when the tile sort reports no empty tile (gsx.h, gsx_train_step): on the
    reference-run exit fixtures (every tile covered, the exit fires) it must run it and composite the reference's count; with the
    splats pushed off one corner (an empty tile: the exit cannot fire) it leaves the three launches out -- either way loss,
    gradients and exit count equal the modular step's, which always enqueues the path."""
    from helpers import load_golden
    from gaussian_splatting_amd import GaussianSplat, trainer as T, render as R
    dev = torch.device("cuda:0")
    fx = load_golden(fixture)
    H, W = fx["H"], fx["W"]
    n = fx["params"]["positions"].shape[0]
    pose = torch.from_numpy(fx["poses"][0])[None].to(dev)
    target = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    for shove in (False, True):
        out = {}
        for fused in (False, True):
            m = GaussianSplat(n)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in fx["params"].items()})
            m = m.to(dev)
            opt = T.make_optimizer(m, fused_clip_norm=0.1)
            T.FUSED_STEP = fused
            try:
                first = None
                if shove:      # run once as is to learn where the splats land, then move every mean that lands in tile (0, 0) out of the image
                    with torch.no_grad():
                        o = m(pose, (H, W))
                        p2 = o["positions"][0]
                        px = (p2[:, 0] + 1) * (W - 1) / 2; py = (p2[:, 1] + 1) * (H - 1) / 2
                        m.scales.mul_(1.0e-3)                                # pin-point footprints: tile (0, 0) stays empty for sure
                        near = (px < 40) & (py < 40)
                        m.positions[near, 0] += 1.0e3
                loss = T.train_step(m, opt, pose, target, T.l1_loss, 0.1)
                out[fused] = (loss.clone(), [p.grad.clone() for p in m.parameters()], m.last_n_composited(), [p.detach().clone() for p in m.parameters()])
                assert (getattr(opt, "_fused_step", None) is not None) == fused
            finally:
                T.FUSED_STEP = True
        a, b = out[False], out[True]
        assert a[2] == b[2], (fixture, shove, a[2], b[2])
        if not shove:
            assert a[2][0] == int(fx["n_composited"]) < n, "the exit fires on the fixture as the reference ran it"
        else:
            assert a[2][0] == n, "an empty tile: no exit"
        assert torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and all(torch.equal(x, y) for x, y in zip(a[3], b[3]))


def test_fused_train_step_declines_what_it_cannot_do():
    """Anything but the plain case stays on the modular path: a wrapped model, a torch optimizer, a custom loss, B > 1,
    another clip norm, a depth map."""
    from gaussian_splatting_amd import GaussianSplat, trainer as T
    from gaussian_splatting_amd.distributed import DataParallelSplat
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = GaussianSplat(300).to(dev)
    opt = T.make_optimizer(m, fused_clip_norm=0.1)
    pose = torch.eye(4, device=dev)[None].contiguous(); pose[0, 2, 3] = 3.0
    tgt = torch.rand(1, 3, 32, 48, device=dev)
    ok = lambda **kw: T._fused_step_for(kw.get("model", m), kw.get("opt", opt), kw.get("pose", pose), kw.get("tgt", tgt),
                                        kw.get("loss", T.l1_loss), kw.get("clip", 0.1))
    assert ok() is not None
    assert ok(model=DataParallelSplat(GaussianSplat(300).to(dev))) is None
    assert ok(opt=T.make_optimizer(m)) is None
    assert ok(loss=T.reference_regularised_mse) is None and ok(loss=lambda o, t: T.l1_loss(o, t)) is None
    assert ok(pose=pose.repeat(2, 1, 1)) is None and ok(clip=0.5) is None and ok(tgt=tgt.cpu()) is None
    m.render_depth_map = True
    assert ok() is None
    m.render_depth_map = False
    m.scales.requires_grad_(False)
    opt._fused_step = None
    assert ok() is None
    m.scales.requires_grad_(True)
    with torch.no_grad():
        assert ok() is None
    loss = T.train_step(m, opt, pose, tgt, T.l1_loss, 0.1)            # and the plain case runs
    assert torch.isfinite(loss) and m.positions.grad is not None and opt._fused_step is not None


def test_fused_step_writes_the_reference_forward_outputs():
    """The one-call step must produce what GaussianSplat.forward returns (gaussian_splatting.py:212-219), the per-Gaussian
    'positions' / 'depths' / 'scales' included (the timed step used to leave them out), and it follows
    the optimizer's hyper-parameters from step to step (betas / eps / clip norm were frozen at construction)."""
    from gaussian_splatting_amd import GaussianSplat, trainer as T
    from gaussian_splatting_amd.scenes import create_orbit_poses
    dev = torch.device("cuda:0")
    H, W, n = 64, 96, 4000
    poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
    target = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
    torch.manual_seed(6)
    m = GaussianSplat(n)
    with torch.no_grad():
        m.positions[:, 2] = m.positions[:, 2].abs() * 0.8 + 0.2
    m = m.to(dev)
    opt = T.make_optimizer(m, fused_clip_norm=0.1)
    with torch.no_grad():
        want = {k: v.clone() for k, v in m(poses[4:5], (H, W)).items()}
    T.train_step(m, opt, poses[4:5], target, T.l1_loss, 0.1)
    got = opt._fused_step.outputs()
    for k in ("rendered_images", "positions", "depths", "scales", "alpha"):
        assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k
    # edited hyper-parameters reach the next step: the twin runs the modular path with the same edits
    torch.manual_seed(6)
    m2 = GaussianSplat(n)
    with torch.no_grad():
        m2.positions[:, 2] = m2.positions[:, 2].abs() * 0.8 + 0.2
    m2 = m2.to(dev)
    opt2 = T.make_optimizer(m2, fused_clip_norm=0.1)
    T.FUSED_STEP = False
    try:
        T.train_step(m2, opt2, poses[4:5], target, T.l1_loss, 0.1)
        for o in (opt, opt2):
            o.max_grad_norm = 0.05
            for g in o.param_groups:
                g['betas'] = (0.8, 0.99); g['eps'] = 1e-6
        T.train_step(m2, opt2, poses[9:10], target, T.l1_loss, 0.05)
    finally:
        T.FUSED_STEP = True
    T.train_step(m, opt, poses[9:10], target, T.l1_loss, 0.05)
    assert opt._fused_step is not None and getattr(opt2, "_fused_step", None) is None
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_flat_parameters_under_the_fused_step():
    """Parameters re-seated into one buffer (GaussianSplat.flatten_parameters) in the MIDDLE of training: the
    one-call step picks the new addresses up, and the run stays bit-equal to one whose parameters never moved."""
    from gaussian_splatting_amd import GaussianSplat, trainer as T
    from gaussian_splatting_amd.scenes import create_orbit_poses
    dev = torch.device("cuda:0")
    H, W, n = 64, 96, 5000
    poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
    target = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(9)).to(dev)
    out = []
    for flatten_at in (None, 1):
        torch.manual_seed(4)
        m = GaussianSplat(n)
        with torch.no_grad():
            m.positions[:, 2] = m.positions[:, 2].abs() * 0.8 + 0.2
        m = m.to(dev)
        opt = T.make_optimizer(m, fused_clip_norm=0.1)
        for s_ in range(4):
            if s_ == flatten_at:
                flat = m.flatten_parameters()
                assert m.flat_parameters() is flat
            T.train_step(m, opt, poses[3 * s_:3 * s_ + 1], target, T.l1_loss, 0.1)
        assert opt._fused_step is not None
        out.append([p.detach().clone() for p in m.parameters()])
        if flatten_at is not None:
            assert m.flat_parameters() is not None and torch.equal(m.flat_parameters()[:4 * n].view(n, 4), m.rotations.data)
    for a, b in zip(*out):
        assert torch.equal(a, b) and torch.isfinite(a).all()
