"""GPU tier: the data-parallel gradient path with the REAL render kernels under it.

One MI355X cannot host two RCCL ranks, so the exchange is covered from both sides:
  * two processes on cuda:0 over gloo (device tensors staged through the host by gloo): each rank renders ITS view
    forward + backward, DataParallelSplat averages inside backward(); the result must equal, bit for bit, the mean of the
    two views' gradients computed in one process without any collective (the kernels are deterministic, x0.5 is exact);
  * one process over RCCL with world 1 and the collectives forced (all-reduce, mesh and auto schedules): averaging over
    one rank is the identity, so the gradients must come back unchanged and through the zero-copy flat buffer.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_COMMON = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.distributed import DataParallelSplat, PARAM_ORDER
from gaussian_splatting_amd.scenes import synth_scene, create_orbit_poses
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
N, H, W = 3000, 96, 128
params = synth_scene(N, 11, 0.004, 0.03)
poses = torch.from_numpy(create_orbit_poses(4.0, 8, 45.0)).to(dev)
gen = torch.Generator(device="cpu").manual_seed(5)
upstream = [torch.randn(1, 3, H, W, generator=gen).to(dev) for _ in range(2)]

def fresh():
    m = GaussianSplat(N)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return m.to(dev)

def local_grads(view):
    m = fresh()
    (m(poses[view:view + 1], (H, W))["rendered_images"] * upstream[view]).sum().backward()
    return {k: getattr(m, k).grad.clone() for k in PARAM_ORDER}
'''

_GLOO_WORKER = _COMMON + r'''
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
m = fresh()
dp = DataParallelSplat(m)
(dp(poses[rank:rank + 1], (H, W))["rendered_images"] * upstream[rank]).sum().backward()     # exchange fires from the hooks
torch.cuda.synchronize()
assert dp.last_sync_was_flat and dp.last_allreduce_bytes == (14 * N + 8) * 4
g0, g1 = local_grads(0), local_grads(1)
for k in PARAM_ORDER:
    want = (g0[k] + g1[k]) * 0.5
    got = getattr(m, k).grad
    assert torch.equal(got, want), (k, float((got - want).abs().max()))
    assert float(want.abs().max()) > 0
dist.destroy_process_group()
print("rank", rank, "ok")
'''

_STEP_WORKER = _COMMON + r'''
# the one-call trainer step in its two halves around the exchange (gsx_train_step_grads -> collective -> gsx_train_step_apply)
# against the modular data-parallel step (module forward, autograd, hooks, optimizer.step()): same kernels, same collective
import gaussian_splatting_amd.trainer as T
from gaussian_splatting_amd.distributed import ddp_compatible
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
target = torch.rand(1, 3, H, W, generator=gen).to(dev)
def run(fused, wrap):
    T.FUSED_STEP = fused
    m = fresh(); dp = wrap(m)
    opt = T.make_optimizer(dp, fused_clip_norm=0.1)
    losses = []
    for s in range(3):
        v = (2 * s + rank) % 8
        losses.append(float(T.train_step(dp, opt, poses[v:v + 1], target, T.l1_loss, max_grad_norm=0.1)))
    torch.cuda.synchronize()
    drv = getattr(opt, "_fused_step", None)
    assert (drv is not None and drv.dp is dp) == fused, (fused, drv)
    assert dp.last_sync_was_flat and dp.last_allreduce_bytes == (14 * N + 8) * 4
    return m, opt, losses
ma, oa, la = run(False, DataParallelSplat)
for wrap in (DataParallelSplat, ddp_compatible):
    mb, ob, lb = run(True, wrap)
    assert la == lb, (la, lb)
    for k in PARAM_ORDER:
        assert torch.equal(getattr(ma, k), getattr(mb, k)), (k, float((getattr(ma, k) - getattr(mb, k)).abs().max()))
        assert torch.equal(getattr(ma, k).grad, getattr(mb, k).grad), k
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(oa.state[getattr(ma, k)][key], ob.state[getattr(mb, k)][key]), (k, key)
T.FUSED_STEP = True
# both ranks must hold the same parameters (replicas must not drift)
flat = torch.cat([getattr(mb, k).detach().reshape(-1) for k in PARAM_ORDER]).cpu()
both = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1])
dist.destroy_process_group()
print("rank", rank, "ok")
'''

_RCCL_WORKER = _COMMON + r'''
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
want = local_grads(1)
for schedule in ("allreduce", "mesh", "auto"):
    m = fresh()
    dp = DataParallelSplat(m, force_collectives=True, schedule=schedule)
    (dp(poses[1:2], (H, W))["rendered_images"] * upstream[1]).sum().backward()
    torch.cuda.synchronize()
    assert dp.last_sync_was_flat and dp.last_allreduce_bytes == (14 * N + 8) * 4, schedule
    if schedule == "auto":
        assert dp.schedule in ("allreduce", "mesh") and dp.tuned_ms is not None
    for k in PARAM_ORDER:
        assert torch.equal(getattr(m, k).grad, want[k]), (schedule, k)
dist.destroy_process_group()
print("rccl ok")
'''


def test_two_ranks_average_real_render_gradients(tmp_path):
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(_GLOO_WORKER)
    port = "29671"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2


def test_one_call_step_between_exchange_equals_modular_dp_step(tmp_path):
    """data-parallel ranks take the one-call step too (reference: training/trainer.py:133-175 under the DDP
    wrap of scripts/train.py:134-139).  Two gloo ranks on the one GPU, three trainer steps each way: losses, parameters,
    gradients and Adam moments bit-equal, replicas equal across the ranks."""
    script = tmp_path / "dp_step_worker.py"
    script.write_text(_STEP_WORKER)
    port = "29675"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2


def test_rccl_world1_exchange_is_identity(tmp_path):
    script = tmp_path / "dp_rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT, "29673"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl ok" in r.stdout


_COMM_WORKER = r'''
import ctypes as C, sys, torch
sys.path.insert(0, sys.argv[1])
from gaussian_splatting_amd import _lib as L
lib = L.lib()
assert lib.gsx_comm_available() == 1
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
ident = C.create_string_buffer(128)
L.check(lib.gsx_comm_unique_id(ident), "gsx_comm_unique_id")
assert any(ident.raw)
comm = C.c_void_p()
L.check(lib.gsx_comm_init(0, 1, ident, C.byref(comm)), "gsx_comm_init")
n = 14 * 1000 + 8
g = torch.randn(n, device=dev); want = g.clone()
st = L.stream_ptr(dev)
L.check(lib.gsx_allreduce_avg(comm, g.data_ptr(), n, st), "gsx_allreduce_avg")       # mean over one rank: unchanged
L.check(lib.gsx_bcast(comm, g.data_ptr(), n, 0, st), "gsx_bcast")
torch.cuda.synchronize()
assert torch.equal(g, want)
assert lib.gsx_allreduce_avg(comm, None, 5, st) == L.GSX_E_INVALID and lib.gsx_bcast(comm, g.data_ptr(), n, 3, st) == L.GSX_E_INVALID
assert lib.gsx_comm_init(2, 2, ident, C.byref(C.c_void_p())) == L.GSX_E_INVALID
L.check(lib.gsx_comm_destroy(comm), "gsx_comm_destroy")
print("comm ok")
'''


def test_comm_entry_points_world1(tmp_path):
    """gsx_comm_* (the exchange for hosts without torch.distributed): RCCL opened on first use, a communicator of
    one rank on the one GPU there is, average and broadcast in place.  Two ranks need two GPUs (RCCL refuses two ranks on one
    device): that leg is the driver's 8-GPU node's."""
    script = tmp_path / "comm_worker.py"
    script.write_text(_COMM_WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "comm ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
