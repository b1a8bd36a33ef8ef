"""The trainer step around the render path (caller a13 of SURVEY 8a): same sequence as
/root/reference/training/trainer.py:133-178 -- zero_grad, forward, loss, backward,
clip_grad_norm_, Adam step -- with the reference's five per-tensor learning rates (:49-61).
The reference's own loss needs VGG16/LPIPS weights from the network (model/loss.py:34,39) and
is out of scope; the loss is a callable here."""
import torch

# configs/default_config.yaml:12-16
DEFAULT_LRS = dict(positions=1e-4, features=1e-3, scales=1e-4, rotations=1e-4, opacity=1e-3)


def make_optimizer(model, lrs=None, fused_clip_norm=None, write_clipped_grad=False):
    """Adam with one group per tensor in the reference's order (trainer.py:49-61).
    fused_clip_norm = max_grad_norm selects the fused HIP clip+Adam step (optim.FusedClipAdam);
    train_step then skips its own clip_grad_norm_ call.  write_clipped_grad=True also scales .grad in place
    as clip_grad_norm_ does (the trainer never reads .grad after the step, so the default saves that write)."""
    lrs = dict(DEFAULT_LRS, **(lrs or {}))
    m = model.module if hasattr(model, "module") else model
    if fused_clip_norm is not None:
        from .optim import FusedClipAdam
        return FusedClipAdam([
            {'params': m.positions, 'lr': lrs['positions']},
            {'params': m.features, 'lr': lrs['features']},
            {'params': m.scales, 'lr': lrs['scales']},
            {'params': m.rotations, 'lr': lrs['rotations']},
            {'params': m.opacity, 'lr': lrs['opacity']},
        ], max_grad_norm=fused_clip_norm, write_clipped_grad=write_clipped_grad)
    return torch.optim.Adam([
        {'params': m.positions, 'lr': lrs['positions']},
        {'params': m.features, 'lr': lrs['features']},
        {'params': m.scales, 'lr': lrs['scales']},
        {'params': m.rotations, 'lr': lrs['rotations']},
        {'params': m.opacity, 'lr': lrs['opacity']},
    ])


def l1_loss(outputs, target):
    """mean |C - target|: one fused HIP pass each way on the GPU (loss.py of this package); plain torch for CPU tensors
    (host-logic tests only -- the render path itself has no CPU form)."""
    c = outputs['rendered_images']
    if c.is_cuda:
        from . import loss as _loss
        return _loss.l1(c, target)
    return (c - target).abs().mean()


def l1_ssim_loss(outputs, target, lambda_dssim=0.2):
    """BASELINE config 2's criterion: (1 - lambda) L1 + lambda (1 - SSIM), fused in HIP on the GPU (loss.l1_ssim)."""
    c = outputs['rendered_images']
    if c.is_cuda:
        from . import loss as _loss
        return _loss.l1_ssim(c, target, lambda_dssim)
    raise RuntimeError("l1_ssim_loss runs on an MI355X only (no CPU fallback)")


def reference_regularised_mse(outputs, target, w_scale=5e-4, w_opacity=5e-3):
    """The importable-math subset of model/loss.py:89-109: clamped MSE + scale and opacity
    regularisers (exercises all three gradient routes into the render backward)."""
    c = outputs['rendered_images']
    o = outputs['opacity']
    if c.is_cuda:
        from . import loss as _loss
        rgb = _loss.clamped_mse(c, target)
    else:
        rgb = ((c.clamp(1e-8, 1 - 1e-8) - target) ** 2).mean()
    return rgb + w_scale * torch.norm(outputs['scales'], dim=-1).mean() \
        + w_opacity * (o * (1 - o)).mean()


def _unit_gradient(loss):
    """The cached 1.0 of loss.py (see there): no fill launch for the backward seed, and the fused losses skip their
    backward kernel when they see it."""
    if loss.dim() != 0 or not loss.is_cuda:
        return None
    from .loss import unit_gradient
    return unit_gradient(loss)


def train_step(model, optimizer, camera_poses, target, loss_fn=l1_loss, max_grad_norm=0.1):
    """One step (trainer.py:134-175).  With a DataParallelSplat model the gradient all-reduce
    fires inside backward(); clipping then sees the averaged gradients, like DDP."""
    optimizer.zero_grad()
    outputs = model(camera_poses, target.shape[-2:])
    loss = loss_fn(outputs, target)
    loss.backward(gradient=_unit_gradient(loss))
    fused = getattr(optimizer, "max_grad_norm", None)                     # FusedClipAdam clips inside step()
    if fused is not None and fused > 0:
        if float(fused) != float(max_grad_norm):
            raise ValueError(f"train_step(max_grad_norm={max_grad_norm}) but the fused optimizer clips at {fused}: "
                             "build it with the same value (make_optimizer(fused_clip_norm=...))")
    elif max_grad_norm > 0:                                               # torch optimizer, or a fused one built without clipping
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    return loss.detach()
