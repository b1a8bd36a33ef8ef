"""Data-parallel gradient path: replaces DistributedDataParallel + NCCL of the reference
(/root/reference/scripts/train.py:65-73,134-139; training/trainer.py:83-97) with ONE RCCL
all-reduce over the flat 14*N gradient buffer the render backward already produces.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm; "gloo" on CPU for the
tests).  The path shards over camera views; parameters are replicated; the only exchange step is
the gradient average after the local backward.  There is nothing to overlap it with: every
Gaussian's gradient is final only when the last tile of the last local view has finished.
"""
import os
import torch
import torch.distributed as dist

PARAM_ORDER = ("rotations", "positions", "scales", "opacity", "features")   # layout of the flat buffers


def shard_views(n_views, rank, world):
    """view v -> rank v mod world."""
    return list(range(rank, n_views, world))


def _flat_view_of(tensors):
    """If the tensors are consecutive views into one contiguous storage (as RenderFunction.backward
    lays the gradients out), return that flat tensor without copying; else None."""
    t0 = tensors[0]
    base = t0.untyped_storage().data_ptr()
    expect = t0.data_ptr()
    total = 0
    for t in tensors:
        if not t.is_contiguous() or t.untyped_storage().data_ptr() != base or t.data_ptr() != expect:
            return None
        expect += t.numel() * t.element_size(); total += t.numel()
    off = (t0.data_ptr() - base) // t0.element_size()
    flat_all = torch.empty(0, dtype=t0.dtype, device=t0.device).set_(t0.untyped_storage())
    return flat_all[off:off + total]


class DataParallelSplat(torch.nn.Module):
    """Wraps a GaussianSplat the way the reference wraps it in DDP (`.module` holds the model so
    trainer code written against DDP keeps working), broadcasts rank 0's parameters once
    (trainer.py:83-87) and averages gradients with one all-reduce (DDP semantics: SUM / world).

    auto_sync=True fires the all-reduce from autograd hooks as soon as all five parameter
    gradients of a backward are final -- like DDP, the caller only calls loss.backward().
    """

    def __init__(self, module, process_group=None, auto_sync=True, broadcast=True, force_collectives=False, schedule=None):
        super().__init__()
        self.module = module
        # "allreduce": one RCCL all-reduce (default).  "mesh": the direct schedule of SURVEY 8e -- every rank sends slice j of
        # its gradients straight to rank j (all-to-all: all seven xGMI links of a GPU carry S/8 at once), sums the eight
        # slices it received in rank order, and the averaged slices are all-gathered.  2*(S/8) per link instead of a
        # ring's 2*(7/8)*S through every link; whether that beats RCCL's own choice needs the 8-GPU node to tell
        # (GSX_DP_SCHEDULE=mesh or bench.py --dp-schedule mesh), so it is opt-in.  "auto": the first exchange times both
        # on a scratch buffer of the gradients' size and every rank adopts the same winner (tune_schedule below).
        self.schedule = schedule or os.environ.get("GSX_DP_SCHEDULE", "allreduce")
        if self.schedule not in ("allreduce", "mesh", "auto"):
            raise ValueError(f"unknown data-parallel schedule {self.schedule!r}")
        self.tuned_ms = None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())   # world 1: rehearsal only
        self._params = [getattr(module, k) for k in PARAM_ORDER]
        self._pending = 0
        self.last_allreduce_bytes = 0
        self.last_sync_was_flat = None
        if broadcast and self.active:
            self.broadcast_parameters()
        if auto_sync and self.active:
            for p in self._params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, camera_poses, image_size):
        return self.module(camera_poses, image_size)

    def broadcast_parameters(self, src=0):
        flat = torch.cat([p.data.reshape(-1) for p in self._params])
        dist.broadcast(flat, src, group=self.group)
        off = 0
        for p in self._params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p)); off += p.numel()

    def _on_grad(self, _p):
        self._pending += 1
        if self._pending == len(self._params):
            self._pending = 0
            self.sync_gradients()

    def sync_gradients(self):
        """Average .grad of the five parameters over all ranks in one collective."""
        if not self.active:
            return
        grads = [p.grad for p in self._params]
        if any(g is None for g in grads):
            raise RuntimeError("sync_gradients: a parameter has no gradient")
        flat = _flat_view_of(grads)
        packed = flat is None
        self.last_sync_was_flat = not packed
        if packed:
            flat = torch.cat([g.reshape(-1) for g in grads])
        if self.schedule == "auto":
            self.tune_schedule(flat.numel(), flat.device)
        self._average(flat, self.schedule)
        self.last_allreduce_bytes = flat.numel() * flat.element_size()
        if packed:
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()


def _average_impl(self, flat, schedule):
    if schedule == "mesh":
        _mesh_average_impl(flat, self.world, self.group)
    elif flat.is_cuda and dist.get_backend(self.group) == "nccl":
        # RCCL averages inside the collective: no separate 1/world pass over the 14*N buffer
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
    else:                                   # gloo (CPU tests) has no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)


def _tune_schedule_impl(self, numel, device, reps=3):
    """Times both exchange schedules on a scratch buffer of `numel` floats (one untimed call each, then `reps` timed ones
    between barriers) and adopts the faster.  The two timings are MAX-reduced over the ranks before they are compared, so
    every rank takes the same decision -- ranks on different schedules would deadlock in the next exchange."""
    import time
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    sync = torch.cuda.synchronize if buf.is_cuda else (lambda: None)
    times = []
    for sched in ("allreduce", "mesh"):
        try:
            self._average(buf, sched)
            sync(); dist.barrier(group=self.group); sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                self._average(buf, sched)
            sync()
            times.append((time.perf_counter() - t0) / reps)
        except RuntimeError as e:          # a backend without all_to_all_single refuses on every rank alike: keep the all-reduce
            if sched == "allreduce":
                raise
            self.tune_error = str(e).splitlines()[0][:200]
            times.append(float("inf"))
    t = torch.tensor(times, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
    t = [float(x) for x in t.cpu()]
    self.tuned_ms = {"allreduce": round(t[0] * 1e3, 4), "mesh": round(t[1] * 1e3, 4) if t[1] != float("inf") else None}
    self.schedule = "allreduce" if t[0] <= t[1] else "mesh"
    return self.schedule


def _mesh_average_impl(flat, world, group):
    """flat <- mean over ranks, by all-to-all + local sum + all-gather.  Every rank reduces its own slice in rank order, so
    all ranks end up with the same bits (the replicas must not drift apart)."""
    n = flat.numel()
    chunk = (n + world - 1) // world
    padded = chunk * world != n
    send = flat
    if padded:
        send = flat.new_zeros(chunk * world); send[:n] = flat
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)            # recv[j*chunk:(j+1)*chunk] = rank j's copy of MY slice
    mine = recv.view(world, chunk).sum(0)
    mine.mul_(1.0 / world)
    dist.all_gather_into_tensor(send, mine, group=group)
    if padded:
        flat.copy_(send[:n])


DataParallelSplat._mesh_average = lambda self, flat: _mesh_average_impl(flat, self.world, self.group)
DataParallelSplat._average = _average_impl
DataParallelSplat.tune_schedule = _tune_schedule_impl


def reduce_metric(value, device=None, group=None):
    """Scalar metric averaged over ranks (trainer.py:89-97)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)
