"""Data-parallel gradient path: replaces DistributedDataParallel + NCCL of the reference
(/root/reference/scripts/train.py:65-73,134-139; training/trainer.py:83-97) with ONE RCCL
all-reduce over the flat 14*N gradient buffer the render backward already produces.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm; "gloo" on CPU for the
tests).  The path shards over camera views; parameters are replicated; the only exchange step is
the gradient average after the local backward.  There is nothing to overlap it with: every
Gaussian's gradient is final only when the last tile of the last local view has finished.

Two wrappers, same behaviour, for the two sides of the reference trainer's `isinstance(model,
DistributedDataParallel)` switch (training/trainer.py:49-61, 80-81, 91, 228-235, 251-253, 286-289):
  DataParallelSplat        isinstance is False: the trainer's plain-model branches run, so the wrapper forwards the
                           model's attributes (`model.positions` ...) and speaks the model's own prefix-free
                           state_dict; metrics are NOT reduced by the trainer on that branch (`reduce_metrics` is there).
  ddp_compatible(...)      the same wrapper as a DistributedDataParallel subclass (DDP's constructor never runs: no
                           reducer, no buckets): isinstance is True, every DDP branch of the trainer -- `model.module.*`,
                           `_sync_model_params`, `_reduce_tensor`, checkpoint unwrap -- runs unedited.
"""
import contextlib
import os
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

PARAM_ORDER = ("rotations", "positions", "scales", "opacity", "features")   # layout of the flat buffers
FLAG_WORDS = 8             # per-tensor presence flags ride behind the gradients (render.GRAD_TAIL spare floats)


def shard_views(n_views, rank, world):
    """view v -> rank v mod world."""
    return list(range(rank, n_views, world))


def _adopt_strays(params, grads, tail):
    """The keep-flat path: the render backward lays all five gradients out in ONE buffer, but a parameter with a second
    autograd route (the opacity regulariser of the reference's criterion, model/loss.py:100-108) ends up with a fresh
    tensor -- autograd summed the two routes somewhere else -- while the other four are still views of that buffer.
    Rather than gathering all 56 N bytes into a new buffer, copy the strays (4 N bytes for the opacity) into their own
    slots of the buffer the others live in.  Returns (flat view incl. `tail`, bytes copied) or (None, 0)."""
    offs, off = [], 0
    for p in params:
        offs.append(off); off += p.numel()
    need = off + tail
    home = None
    for g, o in zip(grads, offs):
        if g is None or not g.is_contiguous():
            continue
        base = g.storage_offset() - o
        st = g.untyped_storage()
        # only the render path's own buffer qualifies -- a storage of exactly 14 N + GRAD_TAIL floats with the gradients laid
        # out from its start (render.RenderFunction.backward, trainer.FusedTrainStep): its tail is spare by construction.
        # Gradients that merely sit somewhere inside a larger caller-owned storage keep what lies behind them (packed path).
        if base == 0 and need * g.element_size() == st.nbytes():
            home = (st.data_ptr(), base, g)
            break
    if home is None:
        return None, 0
    ptr, base, like = home
    flat_all = torch.empty(0, dtype=like.dtype, device=like.device).set_(like.untyped_storage())
    flat = flat_all[base:base + need]
    copied = 0
    for p, g, o in zip(params, grads, offs):
        slot = flat[o:o + p.numel()]
        if g is None:
            slot.zero_()
        elif not (g.is_contiguous() and g.untyped_storage().data_ptr() == ptr and g.storage_offset() == base + o):
            if g.dtype != flat.dtype or g.device != flat.device:
                return None, 0
            slot.copy_(g.reshape(-1)); copied += p.numel() * g.element_size()
    return flat, copied


class BalancedViewSampler(torch.utils.data.Sampler):
    """Stands where the reference puts `DistributedSampler(train_dataset)` (scripts/train.py:106-110): every rank draws
    its share of the views of an epoch, all views are visited once (the list is padded by repetition to a multiple of
    the world size, as DistributedSampler does), `set_epoch` reshuffles, `len()` is the per-rank count.

    What it adds is the reason it exists: a data-parallel step ends when its SLOWEST rank ends (the gradient exchange
    waits for all), and a view's cost is proportional to its instance count, which differs 2x between the views of
    one orbit.  With `costs` (one number per view -- the instance counts D the render path reports,
    `render.last_instance_count()`; any monotone proxy will do) the views are ranked by cost and each step's `world`
    views are NEIGHBOURS in that ranking, so the ranks of a step finish together; which group comes when is shuffled
    per epoch (or kept cost-descending with shuffle=False), and the members of a group rotate over the ranks from
    epoch to epoch so that no rank keeps the costliest view of its groups.  Without costs the order is the plain strided
    one (view v -> rank v mod world).  Every rank computes the same table from the same costs: no communication here
    (`gather_costs` is the one collective, for ranks that each measured only their own views)."""

    def __init__(self, data_source, num_replicas=None, rank=None, shuffle=True, seed=0, costs=None):
        self.n = data_source if isinstance(data_source, int) else len(data_source)
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} outside a world of {num_replicas}")
        self.world, self.rank, self.shuffle, self.seed, self.epoch = int(num_replicas), int(rank), bool(shuffle), int(seed), 0
        self.costs = None
        if costs is not None:
            self.set_costs(costs)

    def set_costs(self, costs):
        costs = [float(c) for c in costs]
        if len(costs) != self.n:
            raise ValueError(f"{len(costs)} costs for {self.n} views")
        self.costs = costs

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return (self.n + self.world - 1) // self.world

    def steps(self):
        """The epoch as a list of steps, each a list of `world` view indices (entry r is rank r's view)."""
        if self.n == 0:
            return []
        if self.costs is None:
            order = list(range(self.n))
        else:
            order = sorted(range(self.n), key=lambda v: (-self.costs[v], v))
        order += order[:len(self) * self.world - self.n]                 # pad by repetition
        groups = [order[i:i + self.world] for i in range(0, len(order), self.world)]
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        if self.shuffle:
            groups = [groups[i] for i in torch.randperm(len(groups), generator=g).tolist()]
        if self.costs is not None:
            groups = [grp[(k + self.epoch) % self.world:] + grp[:(k + self.epoch) % self.world] for k, grp in enumerate(groups)]
        return groups

    def __iter__(self):
        return iter([grp[self.rank] for grp in self.steps()])

    @staticmethod
    def gather_costs(local, n_views, group=None, device=None):
        """local: {view: cost} this rank measured.  Returns the full list (MAX over ranks per view; views nobody
        measured get the mean of the measured ones) -- one small all-reduce."""
        t = torch.full((n_views,), -1.0, dtype=torch.float64)
        for v, c in local.items():
            t[v] = float(c)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            on_dev = device is not None and dist.get_backend(group) == "nccl"
            if on_dev:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            t = t.cpu()
        known = t[t >= 0]
        fill = float(known.mean()) if known.numel() else 0.0
        return [float(c) if c >= 0 else fill for c in t.tolist()]

    def imbalance(self):
        """mean over the steps of (max cost in the step / mean cost in the step): 1.0 = the ranks finish together."""
        if self.costs is None:
            return None
        r = [max(self.costs[v] for v in grp) / max(sum(self.costs[v] for v in grp) / len(grp), 1e-30) for grp in self.steps()]
        return sum(r) / len(r) if r else 1.0


def _flat_view_of(tensors, tail=0):
    """If the tensors are consecutive views into one contiguous storage (as RenderFunction.backward
    lays the gradients out) with at least `tail` more elements behind the last, return that flat tensor
    (tail included) without copying; else None."""
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
    if off + total + tail > flat_all.numel():
        return None
    # a tail is only spare in the render path's own buffer: exactly the gradients + GRAD_TAIL floats, from the storage's start
    if tail and (off != 0 or flat_all.numel() != total + tail):
        return None
    return flat_all[off:off + total + tail]


class DataParallelSplat(torch.nn.Module):
    """Wraps a GaussianSplat the way the reference wraps it in DDP (`.module` holds the model), broadcasts rank 0's
    parameters once (trainer.py:83-87) and averages gradients with one collective (DDP semantics: SUM / world).

    auto_sync=True fires the exchange from autograd, ONCE per backward, after the last gradient of that backward has
    been accumulated -- like DDP, the caller only calls loss.backward().  What is exchanged is fixed by the model, not by
    the backward: the gradients of every parameter with requires_grad, in PARAM_ORDER, plus one presence flag per
    tensor.  A parameter the local backward did not reach contributes zeros; one that NO rank reached keeps grad None
    (what DDP(find_unused_parameters=True) does, scripts/train.py:135-139), so a frozen tensor or a regulariser-only loss
    neither skips the exchange nor leaves the ranks waiting for each other.
    """

    def __init__(self, module, process_group=None, auto_sync=True, broadcast=True, force_collectives=False, schedule=None):
        super().__init__()
        self._setup(module, process_group, auto_sync, broadcast, force_collectives, schedule)

    def _setup(self, module, process_group, auto_sync, broadcast, force_collectives, schedule):
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
        self.tune_error = None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())   # world 1: rehearsal only
        self._names = [k for k in PARAM_ORDER if hasattr(module, k)]
        self._sync_enabled = True
        self._callback_queued = False
        self._mesh_scratch = {}
        self._flag_cache = {}
        self._host_stage = None
        self.last_allreduce_bytes = 0
        self.last_sync_was_flat = None
        self.last_sync_copied_bytes = 0      # bytes moved to make the gradients one buffer (0: the render's own buffer as it was)
        self.time_exchange = False              # True: HIP events around every exchange, collected in exchange_events
        self.exchange_events = []
        if broadcast and self.active:
            self.broadcast_parameters()
        if auto_sync and self.active:
            for k in self._names:               # a tensor frozen at wrap time has no hook (and is not exchanged)
                if getattr(module, k).requires_grad:
                    getattr(module, k).register_post_accumulate_grad_hook(self._on_grad)

    # ---- the plain-model face of the wrapper (the reference trainer's non-DDP branches) ---------------------------
    def __getattr__(self, name):
        """`model.positions`, `.num_gaussians`, `.chunk_size`, maintenance methods ...: whatever the wrapper does not
        have itself is the wrapped model's (trainer.py:56-60 builds the optimizer from `model.<parameter>`)."""
        try:
            return super().__getattr__(name)
        except AttributeError:
            modules = self.__dict__.get("_modules") or {}
            if name != "module" and "module" in modules:
                return getattr(modules["module"], name)
            raise

    def state_dict(self, *args, **kwargs):
        """The wrapped model's own keys (`positions`, ... -- no `module.` prefix): trainer.py:253 saves
        `self.model.state_dict()` on this branch, and the checkpoint must load into a plain GaussianSplat."""
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        """Accepts the model's own keys and -- for checkpoints a DDP-wrapped run wrote with `model.state_dict()` --
        `module.`-prefixed ones."""
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        return self.module.load_state_dict(state_dict, *args, **kwargs)

    def forward(self, camera_poses, image_size):
        self._callback_queued = False           # a backward that raised never ran its callback
        return self.module(camera_poses, image_size)

    def exchange_ms(self, clear=True):
        """mean duration of the timed exchanges (time_exchange = True) in ms, from their HIP events; synchronises"""
        if not self.exchange_events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.exchange_events) / len(self.exchange_events)
        if clear:
            self.exchange_events = []
        return ms

    def _dp_params(self):
        """the parameters in the exchange: requires_grad, in PARAM_ORDER -- the same set on every rank by construction
        (replicated model), whatever a particular backward touched"""
        return [getattr(self.module, k) for k in self._names if getattr(self.module, k).requires_grad]

    def broadcast_parameters(self, src=0):
        params = [getattr(self.module, k) for k in self._names]
        own = None
        if hasattr(self.module, "flatten_parameters") and len(params) == len(PARAM_ORDER):
            try:
                own = self.module.flatten_parameters()
            except RuntimeError:                              # parameters on several devices / dtypes: pack them below
                own = None
        if own is not None:                                   # the parameters ARE one buffer: one collective, nothing to pack or unpack
            if own.is_cuda and dist.get_backend(self.group) != "nccl":
                host = own.cpu(); dist.broadcast(host, src, group=self.group); own.copy_(host)
            else:
                dist.broadcast(own, src, group=self.group)
            return
        flat = torch.cat([p.data.reshape(-1) for p in params])
        if flat.is_cuda and dist.get_backend(self.group) != "nccl":       # gloo: through the host (see _average)
            host = flat.cpu()
            dist.broadcast(host, src, group=self.group)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src, group=self.group)
        off = 0
        for p in params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p)); off += p.numel()

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation over several backwards without an exchange (DistributedDataParallel.no_sync)."""
        old = self._sync_enabled
        self._sync_enabled = False
        try:
            yield
        finally:
            self._sync_enabled = old

    def _on_grad(self, _p):
        # Counting hook firings (five = done) breaks as soon as a backward reaches fewer tensors, and the stale count then
        # fires the collective in the middle of the next backward.  The first gradient of a backward queues ONE callback
        # that the autograd engine runs when that backward has finished.
        if self._callback_queued or not self._sync_enabled:
            return
        self._callback_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(self._after_backward)

    def _after_backward(self):
        self._callback_queued = False
        self.sync_gradients()

    def sync_gradients(self):
        """Average .grad of the exchanged parameters over all ranks in one collective."""
        if not self.active:
            return
        params = self._dp_params()
        if not params:
            return
        grads = [p.grad for p in params]
        k = len(params)
        dev = params[0].device
        flat = None
        if all(g is not None for g in grads):
            flat = _flat_view_of(grads, FLAG_WORDS)          # the render backward's own buffer: no copy
        adopted = False
        self.last_sync_copied_bytes = 0
        if flat is None and k == len(PARAM_ORDER) and sum(g is not None for g in grads) >= 2:
            flat, self.last_sync_copied_bytes = _adopt_strays(params, grads, FLAG_WORDS)
            adopted = flat is not None
        packed = flat is None
        self.last_sync_was_flat = not packed
        total = sum(p.numel() for p in params)
        if packed:
            # a second autograd route into a parameter (the opacity regulariser, trainer.py:56-67 of this package) makes
            # AccumulateGrad own a fresh tensor; a frozen or unreached tensor leaves a hole: gather into one buffer
            flat = torch.empty(total + FLAG_WORDS, dtype=params[0].dtype, device=dev)
            off = 0
            for p, g in zip(params, grads):
                dst = flat[off:off + p.numel()]
                if g is None:
                    dst.zero_()
                else:
                    dst.copy_(g.reshape(-1))
                off += p.numel()
            self.last_sync_copied_bytes = total * flat.element_size()
        pattern = tuple(1.0 if (i < k and grads[i] is not None) else 0.0 for i in range(FLAG_WORDS))
        flags = self.exchange_flat(flat, total, pattern)
        if packed or adopted:
            # the averaged gradients stay where they are: .grad becomes a view of the exchange buffer (no copy back), and
            # the five are consecutive again for whoever walks them next (clip_grad_norm_, the fused optimizer)
            seen = flags.tolist() if any(g is None for g in grads) else None
            off = 0
            for i, p in enumerate(params):
                if seen is None or seen[i] != 0.0:           # reached on some rank
                    p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()                             # reached on no rank: grad stays None, like DDP

    def exchange_flat(self, flat, total, pattern):
        """The one collective of a data-parallel step on a buffer that is already flat: `total` gradient floats in
        PARAM_ORDER followed by FLAG_WORDS presence flags (`pattern`, written here).  sync_gradients ends here, and the
        one-call trainer step (trainer.FusedTrainStep) calls it between gsx_train_step_grads and gsx_train_step_apply on
        the stream both are enqueued on.  Returns the flag words (a view of `flat`)."""
        dev = flat.device
        flags = flat[total:total + FLAG_WORDS]
        src = self._flag_cache.get((pattern, dev))
        if src is None:                                      # one host-to-device copy per pattern, not per step
            src = self._flag_cache[(pattern, dev)] = torch.tensor(pattern, dtype=flat.dtype).to(dev)
        flags.copy_(src)
        if self.schedule == "auto":
            self.tune_schedule(flat.numel(), dev)
        ev = None
        if self.time_exchange and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._average(flat, self.schedule)
        if ev is not None:
            ev[1].record()
            self.exchange_events.append(ev)
        self.last_allreduce_bytes = flat.numel() * flat.element_size()
        return flags

    def owns_hook(self, fn):
        """is `fn` (an entry of a parameter's post-accumulate-grad hooks) this wrapper's own exchange trigger"""
        return getattr(fn, "__self__", None) is self and getattr(fn, "__func__", None) is DataParallelSplat._on_grad

    def reduce_metrics(self, metrics, device=None):
        """trainer.py:228-233 for the plain-model branch: scalar entries averaged over the ranks, the rest passed through."""
        return {k: reduce_metric(v, device, self.group) if isinstance(v, (float, int)) else v for k, v in metrics.items()}


class _DDPFace(DataParallelSplat, DistributedDataParallel):
    """DataParallelSplat that also answers `isinstance(model, DistributedDataParallel)`.  DistributedDataParallel's
    constructor is never run -- there is no reducer and there are no buckets; every method that matters is
    DataParallelSplat's (MRO) -- so the only thing inherited is the type, which is what the reference trainer tests.
    state_dict()/load_state_dict() keep DDP's `module.`-prefixed form here, as a DDP-wrapped model's would."""

    def __init__(self, module, **kwargs):
        torch.nn.Module.__init__(self)
        self._setup(module, kwargs.pop("process_group", None), kwargs.pop("auto_sync", True), kwargs.pop("broadcast", True),
                    kwargs.pop("force_collectives", False), kwargs.pop("schedule", None))
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        # attributes DistributedDataParallel's inherited helpers look at
        self.require_backward_grad_sync = True
        self.device_ids = None
        self.process_group = self.group

    state_dict = torch.nn.Module.state_dict
    load_state_dict = torch.nn.Module.load_state_dict

    def __getstate__(self):
        raise TypeError("a data-parallel wrapper is per process: pickle .module instead")


def ddp_compatible(module, **kwargs):
    """DataParallelSplat for callers that switch on isinstance(model, DistributedDataParallel) -- the reference's trainer
    (training/trainer.py:49-61,80-81,91,228,251-253,286-289) then takes its DDP branches unedited."""
    return _DDPFace(module, **kwargs)


def _average_impl(self, flat, schedule):
    if schedule == "mesh":
        _mesh_average_impl(flat, self.world, self.group, self._mesh_scratch)
    elif flat.is_cuda and dist.get_backend(self.group) == "nccl":
        # RCCL averages inside the collective: no separate 1/world pass over the 14*N buffer
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
    elif flat.is_cuda:
        # gloo with device tensors (rehearsals of the N>1 path on one GPU, tests): staged through pinned host memory HERE.
        # gloo's own device-tensor path hung intermittently on this image with both ranks inside the all-reduce of a
        # 56 MB buffer (r2, two ranks sharing one MI355X); host tensors are the path gloo is built for.
        host = self._host_stage
        if host is None or host.numel() < flat.numel() or host.dtype != flat.dtype:
            host = self._host_stage = torch.empty(flat.numel(), dtype=flat.dtype).pin_memory()
        host = host[:flat.numel()]
        host.copy_(flat)                    # synchronises with the producing stream
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        host.mul_(1.0 / self.world)
        flat.copy_(host, non_blocking=True)
    else:                                   # gloo on CPU tensors (the CPU tests) has no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)


def _mesh_supported(group, device):
    """LOCAL answer to "can this rank run the mesh schedule" (all_to_all_single + all_gather_into_tensor on `device`),
    given without entering a collective: a rank that found out by trying would leave its peers inside the all-to-all."""
    try:
        backend = dist.get_backend(group)
    except Exception:
        return False
    if device is not None and torch.device(device).type == "cuda":
        return backend == "nccl" or backend.endswith("nccl")
    return backend in ("gloo", "mpi") or "gloo" in backend


def _tune_schedule_impl(self, numel, device, reps=3):
    """Times both exchange schedules on a scratch buffer of `numel` floats (one untimed call each, then `reps` timed ones
    between barriers) and adopts the faster.  Every decision is collective.  Whether the mesh schedule may be tried at all
    is each rank's LOCAL capability answer MIN-reduced over the ranks, before anybody enters it (a rank-local failure
    inside a trial -- out of memory, a communicator error -- is not caught: its peers are inside the collective by then,
    so the only honest outcome is the job's failure); the timings are MAX-reduced before they are compared.  Ranks on
    different schedules, or in different collectives, would deadlock."""
    import time
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    sync = torch.cuda.synchronize if buf.is_cuda else (lambda: None)
    ok = 1.0 if _mesh_supported(self.group, device) else 0.0
    ctrl = device if dist.get_backend(self.group) == "nccl" else "cpu"     # control words: device tensors only for RCCL
    flag = torch.tensor([ok], dtype=torch.float32, device=ctrl)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)          # every rank reaches this, whatever it answered
    if ok == 0.0:
        self.tune_error = "mesh schedule not supported by this rank's backend"
    candidates = ("allreduce", "mesh") if float(flag.item()) > 0.0 else ("allreduce",)
    times = []
    for sched in candidates:
        self._average(buf, sched)
        sync(); dist.barrier(group=self.group); sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            self._average(buf, sched)
        sync()
        times.append((time.perf_counter() - t0) / reps)
    times += [float("inf")] * (2 - len(times))
    t = torch.tensor(times, dtype=torch.float64, device=ctrl)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
    t = [float(x) for x in t.cpu()]
    self.tuned_ms = {"allreduce": round(t[0] * 1e3, 4), "mesh": round(t[1] * 1e3, 4) if t[1] != float("inf") else None}
    self.schedule = "allreduce" if t[0] <= t[1] else "mesh"
    self._mesh_scratch.clear()
    return self.schedule


def _mesh_average_impl(flat, world, group, scratch=None):
    """flat <- mean over ranks, by all-to-all + local sum + all-gather.  Every rank reduces its own slice in rank order, so
    all ranks end up with the same bits (the replicas must not drift apart).  `scratch` (a dict the caller keeps) holds
    the receive / padding buffers from one exchange to the next: the sizes repeat every step."""
    n = flat.numel()
    chunk = (n + world - 1) // world
    padded = chunk * world != n
    key = (n, world, flat.dtype, flat.device)
    bufs = scratch.get(key) if scratch is not None else None
    if bufs is None:
        bufs = (torch.empty(chunk * world, dtype=flat.dtype, device=flat.device),
                torch.zeros(chunk * world, dtype=flat.dtype, device=flat.device) if padded else None,
                torch.empty(chunk, dtype=flat.dtype, device=flat.device))
        if scratch is not None:
            scratch.clear(); scratch[key] = bufs
    recv, pad, mine = bufs
    send = flat
    if padded:
        send = pad; send[:n] = flat
    dist.all_to_all_single(recv, send, group=group)            # recv[j*chunk:(j+1)*chunk] = rank j's copy of MY slice
    torch.sum(recv.view(world, chunk), 0, out=mine)
    mine.mul_(1.0 / world)
    dist.all_gather_into_tensor(send, mine, group=group)
    if padded:
        flat.copy_(send[:n])


DataParallelSplat._mesh_average = lambda self, flat: _mesh_average_impl(flat, self.world, self.group, self._mesh_scratch)
DataParallelSplat._average = _average_impl
DataParallelSplat.tune_schedule = _tune_schedule_impl


def reduce_metric(value, device=None, group=None):
    """Scalar metric averaged over ranks (trainer.py:89-97)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)
