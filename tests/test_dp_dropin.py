"""Seam S2: the data-parallel wrapper under a REFERENCE-SHAPED trainer, gloo world 2, on CPU.

`RefShapedTrainer` below restates, in behaviour, the code paths of /root/reference/training/trainer.py that touch the
model object: optimizer groups built from `model.<param>` or `model.module.<param>` depending on
`isinstance(model, DistributedDataParallel)` (:49-61), `_sync_model_params` (:80-87), `_reduce_tensor` (:89-97) and its use
on validation metrics (:228-235), the checkpoint dict (:249-258) and its loading (:283-295).  The reference file itself
cannot be imported (wandb, skimage, lpips are absent).  The render path has no CPU form, so the model under the wrapper
is a GaussianSplat whose forward is a small differentiable torch expression of all five parameters -- what is tested
here is the plumbing around the gradients, and tests/test_gpu_dp.py runs the real kernels under the same wrapper.
"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, io, torch, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel
sys.path.insert(0, sys.argv[1])
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.distributed import DataParallelSplat, ddp_compatible, PARAM_ORDER
import gaussian_splatting_amd.distributed as D

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
N = 40
LRS = dict(positions=1e-4, features=1e-3, scales=1e-4, rotations=1e-4, opacity=1e-3)
NAMES = ("positions", "scales", "rotations", "opacity", "features")


class CpuSplat(GaussianSplat):
    """forward = a differentiable stand-in that reaches all five parameters (the HIP path needs a GPU)"""
    def forward(self, camera_poses, image_size):
        H, W = int(image_size[0]), int(image_size[1])
        k = camera_poses[:, :3, 3].sum()
        tone = self.features.mean(0) * self.opacity.mean() + 1e-2 * (self.positions * k).sum() \
            + 1e-2 * (self.scales * self.rotations[:, :3]).sum()
        img = tone.view(1, 3, 1, 1).expand(camera_poses.shape[0], 3, H, W) * (1.0 + k)
        return {"rendered_images": img, "opacity": self.opacity, "features": self.features}


class RefShapedTrainer:
    def __init__(self, model, device="cpu"):
        self.model, self.device = model, device
        wrapped = isinstance(model, DistributedDataParallel)
        groups = []
        for name in ("positions", "features", "scales", "rotations", "opacity"):         # trainer.py:49-61
            tensor = getattr(model.module, name) if wrapped else getattr(model, name)
            groups.append({"params": tensor, "lr": LRS[name]})
        self.optimizer = torch.optim.Adam(groups)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=0.5, patience=5)
        if wrapped:                                                                        # :80-87
            for p in self.model.parameters():
                if p.requires_grad:
                    dist.broadcast(p.data, 0)

    def reduce_tensor(self, t):                                                            # :89-97
        if not isinstance(self.model, DistributedDataParallel):
            return t
        rt = t.clone(); dist.all_reduce(rt, op=dist.ReduceOp.SUM); rt /= dist.get_world_size()
        return rt

    def reduced_metrics(self, metrics):                                                    # :228-235
        if isinstance(self.model, DistributedDataParallel):
            return {k: self.reduce_tensor(torch.tensor(v, device=self.device)).item() if isinstance(v, (float, int)) else v
                    for k, v in metrics.items()}
        return metrics

    def step(self, pose, target):                                                          # :134-175 with an importable loss
        self.optimizer.zero_grad()
        out = self.model(pose, target.shape[-2:])
        o = out["opacity"]
        loss = ((out["rendered_images"] - target) ** 2).mean() + 5e-3 * (o * (1 - o)).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), 0.1)
        self.optimizer.step()
        return float(loss)

    def checkpoint(self, epoch=3, loss=0.5):                                               # :249-258
        return {"epoch": epoch,
                "model_state_dict": self.model.module.state_dict() if isinstance(self.model, DistributedDataParallel)
                else self.model.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(), "scheduler_state_dict": self.scheduler.state_dict(),
                "loss": loss}

    def load(self, ck):                                                                    # :283-295
        if isinstance(self.model, DistributedDataParallel):
            self.model.module.load_state_dict(ck["model_state_dict"])
        else:
            self.model.load_state_dict(ck["model_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"]); self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        return ck["epoch"] + 1


def make(seed):
    torch.manual_seed(seed)
    return CpuSplat(N)


def pose_of(r):
    p = torch.eye(4)[None].clone(); p[0, :3, 3] = torch.tensor([0.1 * (r + 1), -0.2, 0.3 + r]); return p


target = torch.full((1, 3, 6, 8), 0.25)
def roundtrip(ck):          # through torch.save / torch.load, like a checkpoint file
    buf = io.BytesIO(); torch.save(ck, buf); buf.seek(0); return torch.load(buf, weights_only=True)


# what one data-parallel step must produce: both ranks' gradients, averaged, then clip + Adam on identical replicas
want = make(100)
ref = RefShapedTrainer(want)
gsum = None
for r in range(world):
    ref.optimizer.zero_grad()
    out = want(pose_of(r), target.shape[-2:]); o = out["opacity"]
    (((out["rendered_images"] - target) ** 2).mean() + 5e-3 * (o * (1 - o)).mean()).backward()
    g = [getattr(want, k).grad.clone() for k in NAMES]
    gsum = g if gsum is None else [a + b for a, b in zip(gsum, g)]
for k, g in zip(NAMES, gsum):
    getattr(want, k).grad = g / world
torch.nn.utils.clip_grad_norm_(want.parameters(), 0.1); ref.optimizer.step()

checkpoints = {}
for kind in ("dp", "ddp"):
    m = make(100 + rank)                                   # ranks start with DIFFERENT parameters
    model = DataParallelSplat(m) if kind == "dp" else ddp_compatible(m)
    assert isinstance(model, DistributedDataParallel) == (kind == "ddp")
    # the attributes the reference's plain-model branches (and its other callers) read off the model object
    assert model.positions is m.positions and model.opacity is m.opacity and model.num_gaussians == N and model.chunk_size == 500
    assert [k for k, _ in model.named_parameters()] == ["module." + k for k in NAMES]
    t = RefShapedTrainer(model)                            # dp: `model.positions` (:56); ddp: `model.module.positions` (:50)
    assert t.optimizer.param_groups[0]["params"][0] is m.positions and t.optimizer.param_groups[4]["params"][0] is m.opacity
    t.step(pose_of(rank), target)
    for k in NAMES:                                        # identical on every rank and equal to the averaged-gradient step
        got = getattr(m, k).data
        assert torch.allclose(got, getattr(want, k).data, rtol=0, atol=1e-9), (kind, k, float((got - getattr(want, k).data).abs().max()))
        both = [torch.empty_like(got) for _ in range(world)]; dist.all_gather(both, got)
        assert torch.equal(both[0], both[1]), (kind, k)
    # metrics: the DDP branch reduces inside the trainer (:91-97); the plain branch returns them unreduced (:234-235) and
    # the wrapper's reduce_metrics is the one line a caller adds
    mets = t.reduced_metrics({"psnr": float(rank), "name": "x"})
    if kind == "ddp":
        assert abs(mets["psnr"] - (world - 1) / 2) < 1e-6 and mets["name"] == "x"
    else:
        assert mets["psnr"] == float(rank)
        red = model.reduce_metrics({"psnr": float(rank), "name": "x"})
        assert abs(red["psnr"] - (world - 1) / 2) < 1e-12 and red["name"] == "x"
    ck = roundtrip(t.checkpoint())
    assert sorted(ck["model_state_dict"]) == sorted(NAMES), sorted(ck["model_state_dict"])      # prefix-free either way
    checkpoints[kind] = ck

# checkpoints interchange: wrapper -> plain model, plain -> wrappers, wrapper -> wrapper
plain = make(7); pt = RefShapedTrainer(plain)
for kind in ("dp", "ddp"):
    assert pt.load(checkpoints[kind]) == 4
    for k in NAMES:
        assert torch.equal(getattr(plain, k).data, getattr(want, k).data if False else checkpoints[kind]["model_state_dict"][k])
plain_ck = roundtrip(pt.checkpoint())
for kind, other in (("dp", "ddp"), ("ddp", "dp")):
    m = make(9); model = DataParallelSplat(m) if kind == "dp" else ddp_compatible(m)
    t = RefShapedTrainer(model)
    t.load(plain_ck)
    for k in NAMES:
        assert torch.equal(getattr(m, k).data, plain_ck["model_state_dict"][k]), (kind, k)
    t.load(checkpoints[other])
    assert t.optimizer.state_dict()["state"][0]["step"] == 1
# a DDP-style `module.`-prefixed state dict loads into the plain-face wrapper too
m = make(11); model = DataParallelSplat(m)
model.load_state_dict({"module." + k: v for k, v in plain_ck["model_state_dict"].items()})
assert torch.equal(m.features.data, plain_ck["model_state_dict"]["features"])

# ---- one exchange per backward, whatever the backward reaches (ADVICE r1: the hook counter) --------------------------
m = make(100); dp = DataParallelSplat(m)
reg = lambda: (m.opacity * (1 - m.opacity)).mean() * float(rank + 1)
reg().backward()                                           # a regulariser-only loss: ONE gradient
want_o = torch.autograd.grad((m.opacity * (1 - m.opacity)).mean(), m.opacity)[0] * (sum(r + 1 for r in range(world)) / world)
assert torch.allclose(m.opacity.grad, want_o, atol=1e-9)
assert all(getattr(m, k).grad is None for k in NAMES if k != "opacity")      # reached on no rank: stays None (DDP find_unused)
for k in NAMES: getattr(m, k).grad = None
out = dp(pose_of(rank), (6, 8)); (out["rendered_images"].sum() * float(rank + 1)).backward()      # the next, full backward
full = make(100); (full(pose_of(rank), (6, 8))["rendered_images"].sum() * float(rank + 1)).backward()
for k in NAMES:
    both = [torch.empty_like(getattr(full, k).grad) for _ in range(world)]; dist.all_gather(both, getattr(full, k).grad)
    assert torch.allclose(getattr(m, k).grad, sum(both) / world, atol=1e-9), k
# ranks that reach DIFFERENT sets: rank 0 only the regulariser, rank 1 everything -- same collective on both, no hang
for k in NAMES: getattr(m, k).grad = None
if rank == 0:
    reg().backward()
else:
    out = dp(pose_of(1), (6, 8)); (out["rendered_images"].sum() + reg()).backward()
assert all(getattr(m, k).grad is not None for k in NAMES)
total = torch.stack([getattr(m, k).grad.abs().sum() for k in NAMES]); both = [torch.empty_like(total) for _ in range(world)]
dist.all_gather(both, total); assert torch.equal(both[0], both[1])
# a frozen tensor is simply not part of the exchange
m = make(100); m.rotations.requires_grad_(False); dp = DataParallelSplat(m)
out = dp(pose_of(rank), (6, 8)); (out["rendered_images"].sum() * float(rank + 1)).backward()
assert m.rotations.grad is None and dp.last_allreduce_bytes == (10 * N + 8) * 4
for k in ("positions", "scales", "opacity", "features"):
    both = [torch.empty_like(getattr(m, k).grad) for _ in range(world)]; dist.all_gather(both, getattr(m, k).grad)
    assert torch.equal(both[0], both[1]) and float(both[0].abs().sum()) > 0, k
# no_sync: accumulate locally, exchange on the last backward
m = make(100); dp = DataParallelSplat(m)
with dp.no_sync():
    (dp(pose_of(rank), (6, 8))["rendered_images"].sum() * float(rank + 1)).backward()
    local = m.features.grad.clone()
both = [torch.empty_like(local) for _ in range(world)]; dist.all_gather(both, local)
assert not torch.equal(both[0], both[1])                   # nothing was exchanged yet

# ---- tune_schedule decides collectively (ADVICE r1) ------------------------------------------------------------------
m = make(100); dp = DataParallelSplat(m, schedule="auto")
if rank == 1:                                              # ONE rank cannot run the mesh schedule: nobody may enter it
    D._mesh_supported = lambda group, device: False
dp.tune_schedule(14 * N + 8, torch.device("cpu"))
picks = [None] * world; dist.all_gather_object(picks, (dp.schedule, dp.tuned_ms["mesh"]))
assert picks == [("allreduce", None)] * world, picks
assert (dp.tune_error is not None) == (rank == 1)
# ---- keep-flat: four gradients still in the render backward's buffer, the fifth (opacity: second autograd route) astray --
m = make(100); dp = DataParallelSplat(m)
flat = torch.zeros(14 * N + 8)
g_ = torch.Generator().manual_seed(40 + rank)
slots = dict(rotations=(0, 4), positions=(4, 3), scales=(7, 3), opacity=(10, 1), features=(11, 3))
for k_, (o_, w_) in slots.items():
    v_ = flat[o_ * N:(o_ + w_) * N].view(N, w_); v_.copy_(torch.randn(N, w_, generator=g_)); getattr(m, k_).grad = v_
stray = m.opacity.grad + 0.25                               # what autograd's sum of two routes leaves: a tensor of its own
m.opacity.grad = stray
mine = torch.cat([getattr(m, k_).grad.reshape(-1) for k_ in PARAM_ORDER]).clone()
both = [torch.empty_like(mine) for _ in range(world)]; dist.all_gather(both, mine)
dp.sync_gradients()
assert dp.last_sync_was_flat and dp.last_sync_copied_bytes == 4 * N and dp.last_allreduce_bytes == (14 * N + 8) * 4
got = torch.cat([getattr(m, k_).grad.reshape(-1) for k_ in PARAM_ORDER])
assert torch.equal(got, (both[0] + both[1]) / world)
assert m.opacity.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()     # back in its slot
assert m.opacity.grad.data_ptr() == flat.data_ptr() + 10 * N * 4

# ---- the view sampler: each rank measured only its own views, one collective completes the table ---------------------
from gaussian_splatting_amd.distributed import BalancedViewSampler
true_cost = [100.0 + 50.0 * ((v * 7) % 12) for v in range(12)]
costs = BalancedViewSampler.gather_costs({v: true_cost[v] for v in range(rank, 12, world)}, 12)
assert costs == true_cost
mine = list(BalancedViewSampler(12, costs=costs, seed=3))          # world and rank from the process group
both = [None] * world; dist.all_gather_object(both, mine)
assert sorted(both[0] + both[1]) == list(range(12))                # disjoint, covering
for a_, b_ in zip(both[0], both[1]):                               # the two views of a step are neighbours in cost
    assert abs(true_cost[a_] - true_cost[b_]) <= 50.0
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_wrapper_under_reference_shaped_trainer_gloo(tmp_path):
    script = tmp_path / "dropin_worker.py"
    script.write_text(_WORKER)
    port = "29641"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert r.stdout.count("ok") == 2


def test_wrapper_without_process_group_is_transparent():
    """world 1, no process group: attribute forwarding and the prefix-free state dict still hold (single-GPU runs of a
    trainer written for the wrapper)"""
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.distributed import DataParallelSplat
    m = GaussianSplat(6)
    w = DataParallelSplat(m)
    assert not w.active and w.scales is m.scales and w.num_gaussians == 6
    assert sorted(w.state_dict()) == sorted(m.state_dict())
    with pytest.raises(AttributeError):
        w.no_such_attribute
    m2 = GaussianSplat(6)
    DataParallelSplat(m2).load_state_dict(w.state_dict())
    assert torch.equal(m2.positions.data, m.positions.data)


def test_launches_follow_the_tensors_device(monkeypatch):
    """the reference's launcher never calls torch.cuda.set_device (scripts/train.py:76,132), so the
    stream handed to the library must be the current stream OF THE TENSORS' DEVICE, and that device must be current while
    the library launches.  No GPU here: the three torch hooks of _lib are replaced by a fake 8-device runtime."""
    from gaussian_splatting_amd import _lib as L
    state = {"cur": 0, "log": []}
    monkeypatch.setattr(L, "_cur_device", lambda: state["cur"])

    def exchange(i):
        prev, state["cur"] = state["cur"], i
        state["log"].append(i)
        return prev
    monkeypatch.setattr(L, "_exchange_device", exchange)
    monkeypatch.setattr(L, "_raw_stream", lambda i: 1000 + i)
    assert L.stream_ptr() == 1000                                   # current device
    assert L.stream_ptr(torch.device("cuda", 5)) == 1005 and L.stream_ptr(3) == 1003
    with L.on_device(torch.device("cuda", 3)):
        assert state["cur"] == 3 and L.stream_ptr() == 1003         # what every `_stream()` call inside the guard sees
        with L.on_device(torch.device("cuda", 3)):                  # nested, same device: nothing to do
            assert state["cur"] == 3
    assert state["cur"] == 0 and state["log"] == [3, 0]
    with L.on_device(torch.device("cuda", 0)):                      # already current: no exchange at all
        pass
    assert state["log"] == [3, 0]
    with pytest.raises(ValueError):
        with L.on_device(torch.device("cuda", 2)):
            raise ValueError("restored on the way out too")
    assert state["cur"] == 0


def test_balanced_view_sampler():
    """The stand-in for the reference's DistributedSampler (scripts/train.py:106-110): per-rank shares that cover the
    views once per epoch, identical tables on every rank, views of similar cost in the same step."""
    import math
    from gaussian_splatting_amd.distributed import BalancedViewSampler as S
    from torch.utils.data.distributed import DistributedSampler
    n = 60
    costs = [2.0 + math.cos(2 * math.pi * v / 30) + 0.3 * math.sin(v) for v in range(n)]      # an orbit's 2x swing
    for world in (1, 2, 4, 8):
        shares = [list(S(n, world, r, costs=costs, seed=1)) for r in range(world)]
        per_rank = (n + world - 1) // world
        assert all(len(sh) == per_rank == len(S(n, world, 0)) for sh in shares)
        seen = sorted(v for sh in shares for v in sh)
        assert set(seen) == set(range(n)) and len(seen) == per_rank * world             # padded by repetition only
        tables = [S(n, world, r, costs=costs, seed=1).steps() for r in range(world)]
        assert all(t == tables[0] for t in tables)                                      # no communication needed
        assert [[t[r] for t in tables[0]] for r in range(world)] == shares
        bal = S(n, world, 0, costs=costs).imbalance()
        strided = [[((s * world + r) * 7) % n for r in range(world)] for s in range(n)]
        plain = sum(max(costs[v] for v in g) / (sum(costs[v] for v in g) / len(g)) for g in strided) / len(strided)
        assert bal <= plain + 1e-12 and (world == 1 or bal < 1.0 + 0.4 * (plain - 1.0)), (world, bal, plain)
    a = S(n, 4, 1, costs=costs, seed=1); e0 = list(a); a.set_epoch(1); e1 = list(a)
    assert e0 != e1 and sorted(e0) != sorted(e1)                        # groups reshuffled, members rotate over the ranks
    b = S(n, 4, 1, costs=costs, shuffle=False)
    assert [costs[v] for v in b] == sorted((costs[v] for v in b), reverse=True)         # cost-descending when not shuffled
    # without costs: DistributedSampler's unshuffled assignment (view v -> rank v mod world)
    for r in range(4):
        assert list(S(n, 4, r, shuffle=False)) == list(DistributedSampler(range(n), num_replicas=4, rank=r, shuffle=False))
    with pytest.raises(ValueError):
        S(n, 4, 4)
    with pytest.raises(ValueError):
        S(n, 4, 0, costs=[1.0] * 3)
    assert list(S(0, 2, 0)) == []
