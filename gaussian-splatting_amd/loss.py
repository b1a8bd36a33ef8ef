"""Image losses between the render forward and backward of a trainer step
(/root/reference/training/trainer.py:144-149; RGB term /root/reference/model/loss.py:126-137), as ONE
HIP pass each way (csrc/gsx_loss.hip) instead of torch's 6-7 launches.  The reference's full criterion
also needs VGG16/LPIPS weights from the network (loss.py:34,39) and is out of scope."""
import torch
from . import _lib as L

L1, CLAMPED_MSE = 0, 1


def _ptr(t):
    return t.data_ptr() if t is not None else None


_stream = L.stream_ptr


_UNIT = {}


def unit_gradient(loss):
    """A cached scalar 1.0 on the loss's device: `loss.backward(gradient=unit_gradient(loss))` spares autograd's fill
    launch, and the fused losses recognise it (by address) as "upstream = 1" and hand back the gradient they already
    wrote in their forward pass instead of launching a backward kernel."""
    key = (loss.device, loss.dtype)
    one = _UNIT.get(key)
    if one is None:
        one = _UNIT[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    return one


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind):
        if not pred.is_cuda:
            raise RuntimeError("gaussian-splatting_amd.loss runs on an MI355X only (no CPU fallback); use torch ops on CPU tensors")
        lib = L.lib()
        p = pred.detach().contiguous().float()
        t = target.detach().to(pred.device).expand_as(pred).contiguous().float()      # a host pointer must never reach a kernel
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        wsb = lib.gsx_image_loss_workspace_bytes()
        ws = torch.empty(wsb, dtype=torch.uint8, device=p.device)
        ctx.kind = kind
        st = _stream(p.device)                      # the stream of the prediction's device, not of the current one
        if ctx.needs_input_grad[0]:
            # the loss is almost always the root of the backward: write dLoss/dpred for upstream 1 in the same pass
            d = torch.empty_like(p)
            with L.on_device(p.device):
                L.check(lib.gsx_image_loss_fwd_grad(kind, _ptr(p), _ptr(t), p.numel(), _ptr(loss), _ptr(d), _ptr(ws), wsb, st),
                        "gsx_image_loss_fwd_grad")
            ctx.save_for_backward(p, t, d)
        else:
            with L.on_device(p.device):
                L.check(lib.gsx_image_loss_fwd(kind, _ptr(p), _ptr(t), p.numel(), _ptr(loss), _ptr(ws), wsb, st), "gsx_image_loss_fwd")
            ctx.save_for_backward(p, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        p, t = saved[0], saved[1]
        if len(saved) == 3:
            unit = _UNIT.get((g.device, g.dtype))
            if unit is not None and g.data_ptr() == unit.data_ptr():
                return saved[2], None, None                  # upstream is the cached 1.0: the forward already wrote this
        lib = L.lib()
        d = torch.empty_like(p)
        g = g.detach().contiguous().float()
        with L.on_device(p.device):
            L.check(lib.gsx_image_loss_bwd(ctx.kind, _ptr(p), _ptr(t), p.numel(), _ptr(g), _ptr(d), _stream(p.device)), "gsx_image_loss_bwd")
        return d, None, None


def l1(pred, target):
    """mean |pred - target| (== (pred - target).abs().mean())"""
    return _ImageLoss.apply(pred, target, L1)


def clamped_mse(pred, target):
    """F.mse_loss(clamp(pred, 1e-8, 1 - 1e-8), target) -- loss.py:135-137"""
    return _ImageLoss.apply(pred, target, CLAMPED_MSE)


class _L1Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lambda_dssim):
        if not pred.is_cuda:
            raise RuntimeError("gaussian-splatting_amd.loss runs on an MI355X only (no CPU fallback); use torch ops on CPU tensors")
        if pred.dim() < 2:
            raise ValueError("l1_ssim needs (..., H, W) images")
        lib = L.lib()
        p = pred.detach().contiguous().float()
        t = target.detach().to(pred.device).expand_as(pred).contiguous().float()      # a host pointer must never reach a kernel
        H, W = int(p.shape[-2]), int(p.shape[-1])
        planes = p.numel() // (H * W)
        loss3 = torch.empty(3, dtype=torch.float32, device=p.device)
        wsb = lib.gsx_l1_ssim_workspace_bytes(planes, H, W)
        ws = torch.empty(wsb, dtype=torch.uint8, device=p.device)
        with L.on_device(p.device):
            L.check(lib.gsx_l1_ssim_fwd(_ptr(p), _ptr(t), planes, H, W, float(lambda_dssim), _ptr(loss3), _ptr(ws), wsb, _stream(p.device)), "gsx_l1_ssim_fwd")
        ctx.lam = float(lambda_dssim)
        ctx.save_for_backward(p, t, ws)
        ctx.mark_non_differentiable(loss3)
        return loss3[0], loss3

    @staticmethod
    def backward(ctx, g, _g3):
        p, t, ws = ctx.saved_tensors
        lib = L.lib()
        H, W = int(p.shape[-2]), int(p.shape[-1])
        planes = p.numel() // (H * W)
        d = torch.empty_like(p)
        g = g.detach().contiguous().float()
        with L.on_device(p.device):
            L.check(lib.gsx_l1_ssim_bwd(_ptr(p), _ptr(t), planes, H, W, ctx.lam, _ptr(g), _ptr(ws), ws.numel(), _ptr(d), _stream(p.device)), "gsx_l1_ssim_bwd")
        return d, None, None


def l1_ssim(pred, target, lambda_dssim=0.2, return_terms=False):
    """(1 - lambda) * mean|pred - target| + lambda * (1 - SSIM(pred, target)); SSIM per plane with an 11x11 Gaussian window
    (sigma 1.5, zero padding), i.e. BASELINE config 2's "L1/SSIM".  pred: (..., H, W).  With return_terms also the
    detached tensor (loss, l1, ssim)."""
    loss, terms = _L1Ssim.apply(pred, target, lambda_dssim)
    return (loss, terms) if return_terms else loss
