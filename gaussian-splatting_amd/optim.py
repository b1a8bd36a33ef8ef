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
        betas_set = {tuple(g['betas']) for g in self.param_groups}; eps_set = {g['eps'] for g in self.param_groups}
        if len(betas_set) != 1 or len(eps_set) != 1:
            raise ValueError("FusedClipAdam: betas and eps must be the same for every group (only lr is per group)")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps, gs, ms, vs, ns, lrs = [], [], [], [], [], []
        step = None
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedClipAdam needs contiguous fp32 parameters and gradients on the GPU (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                s = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
                if step is None:
                    step = s
                elif step != s:
                    raise RuntimeError("FusedClipAdam: parameters are at different step counts")
                ps.append(p.data_ptr()); gs.append(p.grad.data_ptr()); ms.append(st['exp_avg'].data_ptr())
                vs.append(st['exp_avg_sq'].data_ptr()); ns.append(p.numel()); lrs.append(float(group['lr']))
        if not ps:
            return loss
        k = len(ps)
        if k > 8:
            raise RuntimeError("FusedClipAdam handles at most 8 tensors per step")
        lib = L.lib()
        dev = self.param_groups[0]['params'][0].device
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(lib.gsx_clip_adam_workspace_bytes(), dtype=torch.uint8, device=dev)
        arr = lambda v: (C.c_void_p * k)(*v)
        b1, b2 = self.param_groups[0]['betas']
        L.check(lib.gsx_clip_adam_step(k, arr(ps), arr(gs), arr(ms), arr(vs), (C.c_int64 * k)(*ns), (C.c_float * k)(*lrs),
                                       b1, b2, self.param_groups[0]['eps'], step, self.max_grad_norm,
                                       1 if self.write_clipped_grad else 0, C.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                       L.stream_ptr()), "gsx_clip_adam_step")
        return loss

    def last_grad_norm(self):
        """(total_norm, clip_coefficient) of the last step, as a device tensor view (no sync)."""
        return None if self._ws is None else self._ws[8192:8200].view(torch.float32)
