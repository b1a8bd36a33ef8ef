"""The trainer step around the render path (caller a13 of SURVEY 8a): same sequence as
/root/reference/training/trainer.py:133-178 -- zero_grad, forward, loss, backward,
clip_grad_norm_, Adam step -- with the reference's five per-tensor learning rates (:49-61).
The reference's own loss needs VGG16/LPIPS weights from the network (model/loss.py:34,39) and
is out of scope; the loss is a callable here."""
import ctypes as C
import os

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


def clamped_mse_loss(outputs, target):
    """The RGB term of the reference's criterion alone (model/loss.py:135-137): mse(clamp(C, 1e-8, 1 - 1e-8), target)."""
    c = outputs['rendered_images']
    if c.is_cuda:
        from . import loss as _loss
        return _loss.clamped_mse(c, target)
    return ((c.clamp(1e-8, 1 - 1e-8) - target) ** 2).mean()


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


# GSX_FUSED_STEP=0 (or trainer.FUSED_STEP = False) keeps train_step on the modular path (module forward, autograd, optimizer)
FUSED_STEP = os.environ.get("GSX_FUSED_STEP", "1") != "0"
_FUSED_LOSSES = {}            # loss callable -> (GSX_LOSS_* kind, lambda_dssim); filled below


class FusedTrainStep:
    """train_step's plain case -- one view, a bare GaussianSplat on the GPU, one of this package's fused image losses,
    FusedClipAdam -- as ONE library call (gsx_train_step): the same kernels in the same order as the module forward,
    the loss, autograd's backward and optimizer.step(), hence the same bits, without the ~0.3 ms of interpreter,
    autograd-engine and allocator time per step that bound a small scene (BASELINE config 2).  Everything a step needs
    lives in buffers this object keeps; `.grad` of the five parameters are views of one flat buffer, as after the
    modular backward; the optimizer's state (exp_avg, exp_avg_sq, step counts) is the FusedClipAdam's own, so
    checkpoints and a later optimizer.step() see nothing unusual.  train_step() routes here by itself."""

    def __init__(self, model, optimizer, kind, lambda_dssim, dp=None):
        from . import _lib as L, render as R
        self.L, self.R = L, R
        self.model, self.optimizer, self.kind = model, optimizer, kind
        # a data-parallel wrapper around `model` (distributed.DataParallelSplat), or None: with one, the step runs as its two
        # halves (gsx_train_step_grads, gsx_train_step_apply) with the wrapper's one collective over `flat` between them
        self.dp = dp
        self.params = [model.positions, model.scales, model.rotations, model.opacity, model.features]
        n = self.n = self.params[0].shape[0]
        dev = self.dev = self.params[0].device
        f32 = torch.float32
        flat = self.flat = torch.empty(14 * n + R.GRAD_TAIL, dtype=f32, device=dev)           # RenderFunction.backward's layout
        self.grads = [flat[4 * n:7 * n].view(n, 3), flat[7 * n:10 * n].view(n, 3), flat[0:4 * n].view(n, 4),
                      flat[10 * n:11 * n].view(n, 1), flat[11 * n:14 * n].view(n, 3)]
        S = self.S = L.GsxStep()
        group_of = {}
        for k, group in enumerate(optimizer.param_groups):
            for p in group['params']:
                group_of[id(p)] = k
        order = sorted(range(5), key=lambda t: group_of[id(self.params[t])])                # the order optimizer.step() walks them in
        for k in range(5):
            S.param[k] = self.params[k].data_ptr(); S.grad[k] = self.grads[k].data_ptr(); S.adam_order[k] = order[k]
        S.N = n
        S.loss_kind, S.lambda_dssim = kind, lambda_dssim
        self.group_of = group_of
        self.shape = None
        self.cap = -1

    def _bind_image(self, H, W):
        L, S, dev, n = self.L, self.S, self.dev, self.n
        f32 = torch.float32
        self.shape = (H, W)
        self.rgb = torch.empty(3, H, W, dtype=f32, device=dev); self.alpha = torch.empty(1, H, W, dtype=f32, device=dev)
        self.d_rgb = torch.empty(3, H, W, dtype=f32, device=dev)
        self.state = torch.empty(4, dtype=torch.int32, device=dev)
        lib = L.lib()
        wsb = lib.gsx_l1_ssim_workspace_bytes(3, H, W) if self.kind == 2 else lib.gsx_image_loss_workspace_bytes()
        self.loss_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        S.H, S.W = H, W
        # the reference's forward returns these per-Gaussian tensors ('positions', 'depths', 'scales', gaussian_splatting.py:212-219)
        # whatever the criterion looks at: the step writes them too (28 B per Gaussian; `outputs` hands them out)
        self.mean2d = torch.empty(1, n, 2, dtype=f32, device=dev); self.depth = torch.empty(1, n, dtype=f32, device=dev)
        self.cov2d = torch.empty(1, n, 2, 2, dtype=f32, device=dev)
        S.mean2d, S.depth, S.cov2d = self.mean2d.data_ptr(), self.depth.data_ptr(), self.cov2d.data_ptr()
        S.rgb, S.alpha, S.d_rgb, S.state = self.rgb.data_ptr(), self.alpha.data_ptr(), self.d_rgb.data_ptr(), self.state.data_ptr()
        S.loss_ws, S.loss_ws_bytes = self.loss_ws.data_ptr(), wsb
        self.cap = -1

    def _bind_capacity(self, cap):
        R, S = self.R, self.S
        H, W = self.shape
        self.arena = None                                   # the old block goes back to the allocator before the new one is asked for
        nbytes, _, _ = R._view_layout(self.n, H, W, cap, R._scratch.force_chunk, True)
        self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        S.arena, S.arena_bytes, S.capacity, S.chunk_force = self.arena.data_ptr(), nbytes, cap, R._scratch.force_chunk
        self.cap = cap

    @torch.no_grad()
    def __call__(self, pose, target):
        L, R, S, opt, dev = self.L, self.R, self.S, self.optimizer, self.dev
        H, W = int(target.shape[-2]), int(target.shape[-1])
        if self.shape != (H, W):
            self._bind_image(H, W)
        # what optimizer.step() would read this step (a caller may edit param_groups or max_grad_norm between steps)
        g0 = opt.param_groups[0]
        S.beta1, S.beta2 = g0['betas']
        S.eps = g0['eps']
        S.max_grad_norm = opt.max_grad_norm
        S.write_clipped_grad = 1 if opt.write_clipped_grad else 0
        # the optimizer's own bookkeeping (FusedClipAdam.step): state on first use, one step count for all five
        step = None
        for k, p in enumerate(self.params):
            S.param[k] = p.data_ptr()                       # a parameter's storage may have been re-seated since the last step
            st = opt.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            s_ = opt._steps.get(p)
            if s_ is None:
                s_ = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
            if step is None:
                step = s_
            elif step != s_:
                raise RuntimeError("FusedTrainStep: parameters are at different step counts")
            S.exp_avg[k] = st['exp_avg'].data_ptr(); S.exp_avg_sq[k] = st['exp_avg_sq'].data_ptr()
            S.lr[k] = float(opt.param_groups[self.group_of[id(p)]]['lr'])
        S.step = step + 1
        # room behind the optimizer's own workspace for one double per 256 Gaussians: gsx_train_step then takes the clip norm's sums
        # of squares from the last backward kernel instead of reading the gradients back (gsx.h)
        ws_need = L.lib().gsx_clip_adam_workspace_bytes() + 8 * ((self.n + 255) // 256)
        if opt._ws is None or opt._ws.device != dev or opt._ws.numel() < ws_need:
            opt._ws = torch.empty(ws_need, dtype=torch.uint8, device=dev)
        S.adam_ws, S.adam_ws_bytes = opt._ws.data_ptr(), opt._ws.numel()
        loss = torch.empty(3, dtype=torch.float32, device=dev)           # fresh per step: the caller may keep what is returned
        S.loss, S.pose, S.target = loss.data_ptr(), pose.data_ptr(), target.data_ptr()
        lib = L.lib()
        with L.on_device(dev):
            pinned, _ev, pinned_ptr, ev_handle = R._scratch.count_slot_raw(dev)
            S.counts_pinned_host, S.count_event = pinned_ptr, ev_handle
            stream = L.stream_ptr(dev)
            while True:
                cap = R._scratch.cap(dev, self.n)
                if cap != self.cap or S.chunk_force != R._scratch.force_chunk:
                    self._bind_capacity(cap)
                S.depth_bits = R._scratch.depth_bits.get(dev, 32)
                status = (lib.gsx_train_step if self.dp is None else lib.gsx_train_step_grads)(C.byref(S), stream)
                if status == L.GSX_E_CAPACITY and int(pinned[0]) > cap:   # nothing was applied: grow the block, same step again
                    R._scratch.grow(dev, int(pinned[0]))
                    continue
                if status == L.GSX_E_AGAIN:                              # the depth keys' top byte is no longer constant
                    R._scratch.depth_bits[dev] = 32
                    continue
                L.check(status, "gsx_train_step")
                break
            if self.dp is not None:
                # the exchange (DDP's gradient average, scripts/train.py:134-139) between backward and clip + Adam, on the
                # stream both halves are enqueued on: the render buffer as it is, presence flags behind the gradients
                self.dp.last_sync_was_flat, self.dp.last_sync_copied_bytes = True, 0
                self.dp.exchange_flat(self.flat[:14 * self.n + _FLAG_WORDS], 14 * self.n, _ALL_FIVE)
                L.check(lib.gsx_train_step_apply(C.byref(S), L.stream_ptr(dev)), "gsx_train_step_apply")
        R._scratch.depth_bits[dev] = R.DEPTH_CHECK_BIT if int(pinned[1]) == 0 else 32
        R._last_D = int(pinned[0])
        for p, g in zip(self.params, self.grads):
            if p.grad is not g:
                p.grad = g
            opt._steps[p] = step + 1
        torch.autograd.graph.increment_version(self.params)              # written through raw pointers: autograd must see it
        self.model._last_state = self.state[0:1]
        return loss[0]

    def outputs(self):
        """The last step's forward outputs under the reference's keys (gaussian_splatting.py:212-219), as views of this
        object's buffers (overwritten by the next step)."""
        return {'rendered_images': self.rgb.unsqueeze(0), 'positions': self.mean2d, 'depths': self.depth, 'scales': self.cov2d,
                'features': self.model.features, 'opacity': self.model.opacity, 'alpha': self.alpha.unsqueeze(0)}


def _fused_step_for(model, optimizer, camera_poses, target, loss_fn, max_grad_norm):
    """The FusedTrainStep that can stand in for this call of train_step, or None.  Cheap: runs every step."""
    spec = _FUSED_LOSSES.get(loss_fn)
    if spec is None or not FUSED_STEP or not torch.is_grad_enabled():
        return None
    from .model import GaussianSplat
    from .optim import FusedClipAdam
    from .distributed import DataParallelSplat
    from . import render as R
    dp = None
    if isinstance(model, DataParallelSplat):          # the wrapper (either face): the step runs as its two halves around the exchange
        dp, model = model, model.module
        if not dp.active:
            dp = None                                 # world 1 without forced collectives: no hooks, nothing to exchange
        elif not dp._sync_enabled:
            return None                               # inside no_sync(): gradients accumulate, which the one-call step does not do
    if type(optimizer) is not FusedClipAdam or not isinstance(model, GaussianSplat) or R._timer is not None or R.STAGED:
        return None
    if float(optimizer.max_grad_norm) != float(max_grad_norm) or getattr(model, "render_depth_map", False):
        return None
    pos = model.positions
    dev = pos.device
    if (not pos.is_cuda or camera_poses.shape != (1, 4, 4) or camera_poses.device != dev or camera_poses.dtype != torch.float32
            or not camera_poses.is_contiguous() or target.device != dev or target.dtype != torch.float32 or not target.is_contiguous()
            or tuple(target.shape[:-2]) not in ((1, 3), (3,)) or pos.shape[0] == 0):
        return None
    params = [model.positions, model.scales, model.rotations, model.opacity, model.features]
    for p in params:                  # a user's hooks only fire on the autograd path; the wrapper's own are what `dp` replaces
        if (p.dtype != torch.float32 or not p.is_contiguous() or not p.requires_grad or p.device != dev
                or getattr(p, "_backward_hooks", None)):
            return None
        hooks = getattr(p, "_post_accumulate_grad_hooks", None)
        if hooks and not (dp is not None and all(dp.owns_hook(fn) for fn in hooks.values())):
            return None
    if dp is not None and len(dp._dp_params()) != 5:
        return None
    hit = getattr(optimizer, "_fused_step", None)
    if (hit is not None and hit.model is model and hit.dp is dp and hit.kind == spec[0] and hit.n == pos.shape[0]
            and all(a is b for a, b in zip(hit.params, params))):
        return hit
    grouped = [p for g in optimizer.param_groups for p in g['params']]
    if len(grouped) != 5 or {id(p) for p in grouped} != {id(p) for p in params}:
        return None
    optimizer._fused_step = hit = FusedTrainStep(model, optimizer, spec[0], spec[1], dp)
    return hit


def train_step(model, optimizer, camera_poses, target, loss_fn=l1_loss, max_grad_norm=0.1):
    """One step (trainer.py:134-175).  With a DataParallelSplat model the gradient all-reduce
    fires inside backward(); clipping then sees the averaged gradients, like DDP.  The plain single-view case (a
    GaussianSplat, bare or inside a DataParallelSplat, FusedClipAdam, l1_loss / l1_ssim_loss / clamped_mse_loss) runs
    through the library's one-call step (FusedTrainStep; in two halves around the exchange when data-parallel): same bits."""
    fused_step = _fused_step_for(model, optimizer, camera_poses, target, loss_fn, max_grad_norm)
    if fused_step is not None:
        return fused_step(camera_poses, target)
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


def train_epoch(model, optimizer, views, loss_fn=l1_loss, max_grad_norm=0.1):
    """One pass over `views` -- an iterable of (image (1,3,H,W), camera_pose (1,4,4), ...) already on the model's device,
    i.e. what dataset.DevicePrefetcher yields -- with train_step per view; the loop of trainer.py:128-207 without its
    logging.  Returns the mean loss like the reference, but sums on the device and reads it back once at the end: the
    reference's per-step loss.item() (:178) would stall the stream every step."""
    model.train()
    total, n = None, 0
    for view in views:
        loss = train_step(model, optimizer, view[1], view[0], loss_fn, max_grad_norm)
        total = loss.clone() if total is None else total.add_(loss)        # the fused step returns a view of its own buffer
        n += 1
    return float(total) / n if n else 0.0


@torch.no_grad()
def evaluate(model, views):
    """Mean L1, MSE and PSNR (of images in [0,1]: -10 log10 MSE, per view, then averaged) over `views` (as for
    train_epoch), under no_grad -- the image metrics of the reference's validation pass (trainer.py:209-240) that need
    no network weights.  One read-back at the end."""
    was_training = model.training
    model.eval()
    acc, n = None, 0
    for view in views:
        image, pose = view[0], view[1]
        c = model(pose, image.shape[-2:])['rendered_images']
        d = c - image
        mse = (d * d).mean()
        row = torch.stack([d.abs().mean(), mse, -10.0 * torch.log10(mse.clamp_min(1e-12))])
        acc = row if acc is None else acc.add_(row)
        n += 1
    model.train(was_training)
    if not n:
        return {'l1': 0.0, 'mse': 0.0, 'psnr': 0.0, 'views': 0}
    l1, mse, psnr = (acc / n).tolist()
    return {'l1': l1, 'mse': mse, 'psnr': psnr, 'views': n}


def save_checkpoint(path, model, optimizer, epoch, loss, scheduler=None, **extra):
    """The reference's checkpoint file (trainer.py:242-271): a dict with 'epoch', 'model_state_dict' (prefix-free, also
    for a data-parallel wrapper), 'optimizer_state_dict', 'loss' and, when given, 'scheduler_state_dict'; tensors and
    plain containers only, so torch.load(weights_only=True) reads it.  (The reference also pickles its config object into
    the file; pass what you need of it as plain values through **extra.)  FusedClipAdam's state has torch.optim.Adam's
    layout, so either optimizer resumes from the other's file."""
    inner = model.module if hasattr(model, "module") else model
    ckpt = {'epoch': int(epoch), 'model_state_dict': inner.state_dict(), 'optimizer_state_dict': optimizer.state_dict(),
            'loss': float(loss)}
    if scheduler is not None:
        ckpt['scheduler_state_dict'] = scheduler.state_dict()
    ckpt.update(extra)
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location=None):
    """Restores what save_checkpoint wrote and returns the epoch to resume from, like trainer.py:273-295.  Keys saved
    from inside a DistributedDataParallel ('module.' prefix) are accepted; a Gaussian count other than the model's raises,
    as load_state_dict does (GaussianSplat.load_state is the call that resizes).
    A file written by the REFERENCE's trainer cannot be read as it is: it pickles its OmegaConf config object into the
    dict (trainer.py:257), and this loader executes nothing from the file (weights_only=True), so the whole file is
    refused -- with an error that says so.  Re-save it on the reference side without the 'config' key."""
    import pickle
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        raise RuntimeError(
            f"{path}: not loadable with weights_only=True ({str(e).splitlines()[0][:160]}).  Checkpoints of the reference's "
            "trainer hold a pickled config object under 'config' (training/trainer.py:257); this loader executes nothing "
            "from the file.  Re-save the checkpoint without that key (tensors and plain values only) and load it again.") from e
    inner = model.module if hasattr(model, "module") else model
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ckpt['model_state_dict'].items()}
    inner.load_state_dict(sd)
    if optimizer is not None and 'optimizer_state_dict' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    if scheduler is not None and 'scheduler_state_dict' in ckpt:
        scheduler.load_state_dict(ckpt['scheduler_state_dict'])
    return int(ckpt['epoch']) + 1


_FLAG_WORDS = 8                                   # distributed.FLAG_WORDS == render.GRAD_TAIL
_ALL_FIVE = (1.0,) * 5 + (0.0,) * 3               # presence flags of a backward that reached all five tensors
_FUSED_LOSSES.update({l1_loss: (0, 0.0), clamped_mse_loss: (1, 0.0), l1_ssim_loss: (2, 0.2)})
