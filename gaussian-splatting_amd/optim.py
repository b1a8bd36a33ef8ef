"""Fused clip_grad_norm_ + Adam for the splat parameters: one C-ABI call
(`gsx_clip_adam_step`, three kernel launches) instead of torch's ~25 foreach kernels.  A
`torch.optim.Optimizer`, so `param_groups[i]['lr']` (ReduceLROnPlateau, trainer.py:64-70) and
`state_dict()` (checkpoints, trainer.py:242-271) behave and interchange with `torch.optim.Adam`:
state per parameter = {step, exp_avg, exp_avg_sq}."""
import ctypes as C
import torch
from . import _lib as L


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0, write_clipped_grad=True):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self.max_grad_norm = float(max_grad_norm)
        self.write_clipped_grad = bool(write_clipped_grad)
        self._ws = None
        self._steps = {}
        betas_set = {tuple(g['betas']) for g in self.param_groups}; eps_set = {g['eps'] for g in self.param_groups}
        if len(betas_set) != 1 or len(eps_set) != 1:
            raise ValueError("FusedClipAdam: betas and eps must be the same for every group (only lr is per group)")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # The step count lives in a Python int per parameter; the `step` tensors torch.optim.Adam keeps in its state are
        # refreshed only when the state is read (state_dict): five CPU tensor increments + .item() calls per step were a
        # third of this method's host time, which is what bounds a small scene's trainer step.
        ps, gs, ms, vs, ns, lrs, touched = [], [], [], [], [], [], []
        step = None
        for group in self.param_groups:
            lr = float(group['lr'])
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous():
                    raise RuntimeError("FusedClipAdam needs contiguous fp32 parameters and gradients on the GPU (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                s = self._steps.get(p)
                if s is None:                      # first step, or state loaded from a checkpoint
                    s = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
                s += 1
                self._steps[p] = s
                if step is None:
                    step = s
                elif step != s:
                    raise RuntimeError("FusedClipAdam: parameters are at different step counts")
                touched.append(p)
                ps.append(p.data_ptr()); gs.append(g.data_ptr()); ms.append(st['exp_avg'].data_ptr())
                vs.append(st['exp_avg_sq'].data_ptr()); ns.append(p.numel()); lrs.append(lr)
        if not ps:
            return loss
        k = len(ps)
        if k > 8:
            raise RuntimeError("FusedClipAdam handles at most 8 tensors per step")
        lib = L.lib()
        dev = touched[0].device
        if any(p.device != dev for p in touched):
            raise RuntimeError("FusedClipAdam: all parameters must live on one device")
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(lib.gsx_clip_adam_workspace_bytes(), dtype=torch.uint8, device=dev)
        arr = lambda v: (C.c_void_p * k)(*v)
        b1, b2 = self.param_groups[0]['betas']
        with L.on_device(dev):
            L.check(lib.gsx_clip_adam_step(k, arr(ps), arr(gs), arr(ms), arr(vs), (C.c_int64 * k)(*ns), (C.c_float * k)(*lrs),
                                           b1, b2, self.param_groups[0]['eps'], step, self.max_grad_norm,
                                           1 if self.write_clipped_grad else 0, self._ws.data_ptr(), self._ws.numel(),
                                           L.stream_ptr(dev)), "gsx_clip_adam_step")
        # the kernel wrote the parameters through raw pointers: tell autograd (a backward that saved them must refuse to run)
        torch.autograd.graph.increment_version(touched)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """set_to_none=True (torch's default, and what the trainer uses): drop the five gradients without the generic
        machinery (foreach grouping, hooks, profiler ranges: ~20 us per call, a twentieth of a small scene's step)."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group['params']:
                p.grad = None

    def _flush_steps(self):
        for p, s in self._steps.items():
            st = self.state.get(p)
            if st:
                st['step'] = torch.tensor(float(s))

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps = {}                           # re-read from the loaded `step` tensors at the next step

    def last_grad_norm(self):
        """(total_norm, clip_coefficient) of the last step, as a device tensor view (no sync)."""
        return None if self._ws is None else self._ws[8192:8200].view(torch.float32)
