"""Image losses between the render forward and backward of a trainer step
(/root/reference/training/trainer.py:144-149; RGB term /root/reference/model/loss.py:126-137), as ONE
HIP pass each way (csrc/gsx_loss.hip) instead of torch's 6-7 launches.  The reference's full criterion
also needs VGG16/LPIPS weights from the network (loss.py:34,39) and is out of scope."""
import ctypes as C
import torch
from . import _lib as L

L1, CLAMPED_MSE = 0, 1


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind):
        if not pred.is_cuda:
            raise RuntimeError("gaussian-splatting_amd.loss runs on an MI355X only (no CPU fallback); use torch ops on CPU tensors")
        lib = L.lib()
        p = pred.detach().contiguous().float()
        t = target.detach().expand_as(pred).contiguous().float()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        wsb = lib.gsx_image_loss_workspace_bytes()
        ws = torch.empty(wsb, dtype=torch.uint8, device=p.device)
        L.check(lib.gsx_image_loss_fwd(kind, _ptr(p), _ptr(t), p.numel(), _ptr(loss), _ptr(ws), wsb, _stream()), "gsx_image_loss_fwd")
        ctx.kind = kind
        ctx.save_for_backward(p, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        lib = L.lib()
        d = torch.empty_like(p)
        g = g.detach().contiguous().float()
        L.check(lib.gsx_image_loss_bwd(ctx.kind, _ptr(p), _ptr(t), p.numel(), _ptr(g), _ptr(d), _stream()), "gsx_image_loss_bwd")
        return d, None, None


def l1(pred, target):
    """mean |pred - target| (== (pred - target).abs().mean())"""
    return _ImageLoss.apply(pred, target, L1)


def clamped_mse(pred, target):
    """F.mse_loss(clamp(pred, 1e-8, 1 - 1e-8), target) -- loss.py:135-137"""
    return _ImageLoss.apply(pred, target, CLAMPED_MSE)
