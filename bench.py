#!/usr/bin/env python3
"""bench.py -- fwd+bwd splat-render throughput of the gfx950 path on BASELINE.json's metric
configuration: 1M Gaussians, 1024x1024, the 60-view orbit of config 3 (SURVEY 8d seed 5).

    python bench.py --gpus 1 --steps 60 --warmup 5
    python bench.py --gpus N ...          # no torchrun environment: spawns its N ranks itself (as child processes,
                                          # before anything touches a GPU) and exits with their status
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --backend gloo     # rehearsal of the N>1 path on ONE GPU (ranks share the card, gloo exchange)

A "step" is one trainer step on one view per GPU (trainer.py:134-175 sequence): zero_grad ->
render forward -> L1 loss -> render backward -> [N>1: one RCCL all-reduce of the flat 14*N
gradient buffer, or the all-to-all + all-gather mesh schedule if that timed faster at start-up] -> clip_grad_norm_(0.1) -> Adam step.
N = 1: the view of step s is (s * 7) mod 60: the stride (7 is coprime with 60) makes ANY window of steps a sample of the
whole orbit -- the orbit's views differ 2x in instance count, so walking them in order made the value depend on which
stretch `--steps` happened to time.
N > 1: views are sharded over GPUs, per-GPU work is fixed (weak scaling): one view per GPU per step, every view of the
orbit once per epoch of 60/N steps.  A step ends when its slowest rank ends (the gradient exchange waits for all), so WHICH
views share a step matters: `distributed.BalancedViewSampler` (the package's stand-in for the reference's
DistributedSampler) groups views of similar instance count into the same step; the counts come from one untimed forward
pass over the orbit at start-up.  `--view-order strided` gives the plain ((s*N + r) * 7) mod 60 assignment instead; the
line reports the cost imbalance (max / mean over the ranks of a step) of both.
value = N * K * H * W / max-over-ranks wall time / 1e6 (Mpix/s), inputs resident in HBM.

Rank 0 also reports (same JSON line):
  roofline         dominant kernel's algorithmic HBM bytes per launch / its HIP-event duration vs 8 TB/s
  cpu_baseline     the CPU oracle (tiled restatement, "port") timed on the host on a few views of the same workload,
                   the same step (render fwd + L1 + bwd + clip + Adam)
  cpu_baseline_c1  BASELINE.md row CPU-A: the reference ALGORITHM (dense restatement: every Gaussian x every pixel, global
                   early exit) at config 1 (5000 Gaussians, 400x300) on the same host
  rccl             N>1 (or --force-dp): backend, world size seen by torch.distributed, bytes per exchange, and the
                   exchange's duration from HIP events around it inside the timed loop
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
PMC_FILE = os.path.join("profiles", "r3_pmc_hbm_traffic.json")
SQ_FILE = os.path.join("profiles", "r3_pmc_sq_counters.json")
# stage -> EXACT name (as tools/pmc_traffic.py writes it: arguments stripped) of the kernel whose PMC traffic (separate
# rocprofv3 --pmc passes on the final tree) belongs to it
STAGE_KERNEL = {"composite_bwd": "gsx::chunk_bwd_kernel", "composite_fwd": "gsx::chunk_fwd_wg_kernel<false>",
                "preprocess_bwd": "gsx::preprocess_bwd_kernel", "build_records": "gsx::build_records_kernel",
                "project_fwd": "gsx::project_fwd_kernel"}
N_VIEWS = 60
VIEW_STRIDE = 7
T_START = time.perf_counter()


def pmc_traffic_bytes(stage):
    """HBM bytes per launch of the stage's kernel from the committed PMC summary (None if absent)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            kernels = json.load(f)["kernels"]
        want = STAGE_KERNEL[stage]
        if want in kernels:
            return int(kernels[want]["hbm_total_MB"] * 1024 * 1024)
    except (OSError, KeyError, ValueError):
        pass
    return None


def algorithmic_bytes(stage, n, d, p):
    """HBM bytes one launch must move assuming perfect on-chip reuse inside the kernel."""
    return {
        # rank 4 + record 48 per instance; rgb 12 + alpha 4 + depth 4 per pixel; ranges 8 per tile
        "composite_fwd": 52 * d + 20 * p,
        # rank 4 + record 48 + offset 4 + partial row 36 per instance; rgb 12 + d_rgb 12 per pixel
        "composite_bwd": 92 * d + 24 * p,
        "project_fwd": (40 + 32) * n,
        "depth_sort": 4 * (8 + 8 + 8) * n,                # 4 passes: hist read 4, scatter read 8 + write 8 (+ index)
        "build_records": (4 + 28 + 16 + 48 + 4) * n,
        "bin_instances": 8 * n + 8 * d + 2 * (4 + 8 + 8) * d + 8 * d,
        "preprocess_bwd": (4 + 40 + 16 + 28 + 56) * n + 36 * d,
    }[stage]


def view_of(step, rank, world):
    return ((step * world + rank) * VIEW_STRIDE) % N_VIEWS


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torch.distributed.run and hand its exit
    status on.  Nothing in this process has touched a GPU (a process that has must not be replaced or forked into ranks)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run(cmd, env=env)
    if r.returncode != 0:
        print(f"[bench] a rank failed: torch.distributed.run exited with {r.returncode}", file=sys.stderr)
    return r.returncode


def cpu_baselines(args, params, H, W, target_np):
    """The two CPU rows of BASELINE.md section 3, on this host's cores (rank 0, N = 1 only)."""
    import torch
    from oracle import gsx_oracle as O          # checker / reported baseline only
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.scenes import create_orbit_poses, pose_tz
    from gaussian_splatting_amd.trainer import DEFAULT_LRS
    O.lib()
    names = ("positions", "scales", "rotations", "opacity", "features")

    def host_step(p, pose, h, w, target, mode, adam):
        """one trainer step on the host: render fwd, L1, render bwd, clip_grad_norm_(0.1), Adam"""
        fw = O.forward(p, pose, h, w, mode)
        d_rgb = (np.sign(fw["rgb"] - target) / target.size).astype(np.float32)
        g = O.backward(p, pose, h, w, fw, d_rgb, None, None, None, mode)
        total = np.sqrt(sum(float((g[k] ** 2).sum()) for k in names))
        coef = min(1.0, 0.1 / (total + 1e-6))
        adam["t"] += 1
        t = adam["t"]
        for k in names:
            gk = (g[k] * coef).astype(np.float32).reshape(p[k].shape)
            m = adam.setdefault("m" + k, np.zeros_like(p[k])); v = adam.setdefault("v" + k, np.zeros_like(p[k]))
            m *= 0.9; m += 0.1 * gk
            v *= 0.999; v += 0.001 * gk * gk
            p[k] -= DEFAULT_LRS[k] * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
        return fw

    # CPU-B: the tiled restatement on a bounded sample of the bench workload
    all_poses = create_orbit_poses(4.0, N_VIEWS, 45.0)
    p = {k: v.copy() for k, v in params.items()}
    # SURVEY 8d protocol: 1 warm-up + median of >= 5.  Six trainer steps on views spread over the orbit (~4 s of 16-thread
    # CPU work each): the first is the warm-up (page faults, thread pool), the median of the other five is reported
    sample_views, ds, adam, times_b = (0, 10, 20, 30, 40, 50), [], {"t": 0}, []
    for i, v in enumerate(sample_views):
        t1 = time.perf_counter()
        d_v = int(host_step(p, all_poses[v], H, W, target_np, "tiled", adam)["D"])
        if i:
            times_b.append(time.perf_counter() - t1); ds.append(d_v)
    tc = sorted(times_b)[len(times_b) // 2]
    cpu = {"value": round(H * W / tc / 1e6, 4), "unit": "Mpix/s", "cores": O.num_threads(), "kind": "port",
           "includes": ["render_fwd", "l1_loss", "render_bwd", "clip_grad_norm", "adam"], "like_for_like_with_value": True,
           "sample": f"1 warm-up + median of {len(times_b)} trainer steps (orbit views {sample_views[1:]}, D={ds}) of the same workload on "
                     f"the tiled CPU restatement: {tc:.2f} s per step ({sum(times_b):.1f} s timed); the reference's own dense loop "
                     f"cannot run at this size (16 ns*N*P, SURVEY 6)"}

    # CPU-A: the reference algorithm itself (dense, global early exit) at config 1 -- same seeded init as the reference's
    # `torch.manual_seed(0); GaussianSplat(5000)`, pose = I with t_z = 3
    torch.manual_seed(0)
    m1 = GaussianSplat(5000)
    p1 = {k: getattr(m1, k).detach().numpy().copy() for k in names}
    h1, w1 = 300, 400
    tgt1 = np.random.default_rng(1).random((3, h1, w1), dtype=np.float32)
    times, k_exit = [], None
    for i in range(6):                             # 1 warm-up + 5 timed
        q = {k: v.copy() for k, v in p1.items()}
        t1 = time.perf_counter()
        fw = host_step(q, pose_tz(3.0), h1, w1, tgt1, "dense", {"t": 0})
        if i:
            times.append(time.perf_counter() - t1)
        k_exit = int(fw["n_composited"])
    med = sorted(times)[len(times) // 2]
    c1 = {"value": round(h1 * w1 / med / 1e6, 4), "unit": "Mpix/s", "cores": O.num_threads(), "kind": "port",
          "algorithm": "dense (the reference's: every Gaussian x every pixel, global early exit)",
          "sample": f"config 1: 5000 Gaussians (reference init, seed 0), 400x300, one trainer step, median of {len(times)}: {med:.3f} s; "
                    f"early exit after {k_exit} Gaussians",
          "true_reference_measured_in_survey": {"value": 0.0126, "unit": "Mpix/s", "cores": 8,
                                                "note": "BASELINE.md section 2: the reference's PyTorch loop, fwd+bwd 9.54 s, survey container"}}
    return cpu, c1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", help="scene config (c2/c3/c4/c5); the metric is quoted on c3")
    ap.add_argument("--loss", default="l1", choices=("l1", "l1_ssim"),
                    help="criterion between forward and backward: l1 (headline) or config 2's L1/SSIM (fused HIP either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time render fwd+bwd(+all-reduce) only")
    ap.add_argument("--inference", action="store_true",
                    help="informational: forward-only orbit rendering (SURVEY 8f N1, render_frames); not the headline metric")
    ap.add_argument("--force-dp", action="store_true",
                    help="rehearse the multi-GPU path on one GPU: init RCCL with world 1 and run the collectives anyway")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend for N>1: nccl (= RCCL, one rank per GPU) or gloo (rehearsal: the ranks may "
                         "share a GPU, gradients are exchanged through the host)")
    ap.add_argument("--dp-schedule", default=None, choices=("allreduce", "mesh", "auto"),
                    help="gradient exchange: one RCCL all-reduce (default), all-to-all + local sum + all-gather (SURVEY 8e's mesh "
                         "schedule), or auto = time both on a scratch buffer at the first exchange and keep the faster")
    ap.add_argument("--view-order", default="balanced", choices=("balanced", "strided"),
                    help="N>1: which views share a step -- balanced = views of similar instance count (BalancedViewSampler), "
                         "strided = ((s*N + r) * 7) mod 60")
    ap.add_argument("--verbose", action="store_true", help="phase progress lines on stderr")
    ap.add_argument("--hang-dump", type=int, default=0, metavar="SECONDS",
                    help="dump every thread's Python stack and exit if the run is still going after SECONDS")
    ap.add_argument("--no-fused-step", action="store_true",
                    help="keep the trainer step on the modular path (module forward, autograd, optimizer.step()) instead of the "
                         "one-call gsx_train_step that train_step() uses for the plain single-GPU case; same kernels, same bits")
    ap.add_argument("--no-inference-extra", action="store_true",
                    help="skip the informational forward-only orbit pass behind the timed region (profiling runs: its two-stream "
                         "overlap inflates the per-kernel averages of a kernel trace)")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="use torch clip_grad_norm_ + torch.optim.Adam instead of the fused HIP clip+Adam step")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    # stdout carries ONE line, the JSON: whatever a native library writes to file descriptor 1 on the way (RCCL prints a version banner
    # there when its first communicator comes up) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.hang_dump == 0 and args.gpus > 1:
        args.hang_dump = 900                         # multi-rank default: a hang leaves a stack trace instead of silence
    if args.hang_dump > 0:                           # a rank stuck in a collective prints where, then exits non-zero
        import faulthandler
        faulthandler.dump_traceback_later(args.hang_dump, exit=True)

    def phase(msg):
        if args.verbose:
            print(f"[bench rank {os.environ.get('RANK', '0')}] {msg} (+{time.perf_counter() - T_START:.1f} s)", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        sys.exit(f"bench.py --gpus {world} needs {world} GPUs for RCCL, this node shows {n_dev} (--backend gloo rehearses on fewer)")
    local_dev = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1 or args.force_dp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd import render as R
    from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
    from gaussian_splatting_amd.distributed import DataParallelSplat, BalancedViewSampler
    from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss, l1_ssim_loss
    loss_fn = l1_ssim_loss if args.loss == "l1_ssim" else l1_loss
    if args.no_fused_step:
        import gaussian_splatting_amd.trainer as _trainer
        _trainer.FUSED_STEP = False

    params, H, W = config_scene(args.config)
    n = params["positions"].shape[0]
    model = GaussianSplat(n)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev)
    use_dp = world > 1 or args.force_dp
    # N>1 default "auto": both exchange schedules are timed once, here, on a scratch buffer of the gradients' size
    # (outside warm-up and timed region alike) and the faster is kept; the line reports the choice and both timings
    schedule = args.dp_schedule or os.environ.get("GSX_DP_SCHEDULE", "auto")
    net = DataParallelSplat(model, force_collectives=args.force_dp, schedule=schedule) if use_dp else model
    phase("wrapper built")
    if use_dp and net.schedule == "auto":
        net.tune_schedule(14 * n + R.GRAD_TAIL, dev)
        phase(f"schedule tuned: {net.schedule} {net.tuned_ms}")
    opt = make_optimizer(net, fused_clip_norm=None if args.torch_optimizer else 0.1)
    poses = torch.from_numpy(create_orbit_poses(4.0, N_VIEWS, 45.0)).to(dev)      # README.md:129-137
    g = torch.Generator(device="cpu").manual_seed(3)
    target_cpu = torch.rand(1, 3, H, W, generator=g)
    target = target_cpu.to(dev)

    # ---- which view a rank renders at step s ---------------------------------------------------------------
    balance = None
    schedule_views = None
    if world > 1 and args.view_order == "balanced" and not args.inference:
        local = {}
        with torch.no_grad():
            for v in range(rank, N_VIEWS, world):                   # each rank counts its share of the orbit, untimed
                model(poses[v:v + 1], (H, W))
                local[v] = R.last_instance_count()
        costs = BalancedViewSampler.gather_costs(local, N_VIEWS, device=dev)
        sampler = BalancedViewSampler(N_VIEWS, world, rank, shuffle=True, seed=0, costs=costs)
        schedule_views, need, epoch = [], args.warmup + args.steps + N_VIEWS, 0
        while len(schedule_views) < need:
            sampler.set_epoch(epoch); schedule_views += list(sampler); epoch += 1
        strided = [[view_of(st, r, world) for r in range(world)] for st in range(N_VIEWS)]
        imb = lambda groups: sum(max(costs[v] for v in g_) / (sum(costs[v] for v in g_) / len(g_)) for g_ in groups) / len(groups)
        balance = {"sampler": "BalancedViewSampler(costs = instance counts of one untimed forward pass over the orbit)",
                   "cost_imbalance_max_over_mean": round(sampler.imbalance(), 4), "strided_order_would_be": round(imb(strided), 4),
                   "instances_min_max": [int(min(costs)), int(max(costs))]}
        phase(f"view costs gathered, imbalance {balance}")

    def view_for(s):
        return schedule_views[s] if schedule_views is not None else view_of(s, rank, world)

    def step(s):
        v = view_for(s)
        if args.no_optimizer:
            opt.zero_grad()
            out = net(poses[v:v + 1], (H, W))
            loss_fn(out, target).backward()
        else:
            train_step(net, opt, poses[v:v + 1], target, loss_fn, max_grad_norm=0.1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def orbit_pass(first, count):
        """forward-only: `count` views of this rank's share of the orbit in ONE render_frames call (views pipelined over streams)"""
        idx = torch.tensor([view_for(first + i) for i in range(count)], device=dev)
        model.render_frames(poses[idx], (H, W))

    def run(first, count):
        if args.inference:
            done = 0
            while done < count:
                k = min(N_VIEWS, count - done)
                orbit_pass(first + done, k); done += k
        else:
            for s in range(count):
                step(first + s)

    phase("model ready")
    run(0, args.warmup)
    fence()
    phase("warm-up done")
    if use_dp:
        net.time_exchange = True                  # two HIP events per exchange, read after the timed region
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    phase("timed region done")
    exchange_ms = None
    if use_dp:
        net.time_exchange = False
        exchange_ms = net.exchange_ms()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel durations (HIP events on the launch stream) over one more orbit pass.  Every rank runs
    # the loop (the gradient all-reduce inside backward is collective); rank 0 records the events. -----------
    roofline = None
    stages = {}
    timer = R.StageTimer() if rank == 0 else None
    R.set_stage_timer(timer)
    d_sum = 0
    k_prof = 0 if args.inference else min(args.steps, N_VIEWS)
    for s in range(k_prof):
        v = view_for(args.warmup + s)                 # the views the timed region rendered
        opt.zero_grad()
        out = net(poses[v:v + 1], (H, W))
        d_sum += R.last_instance_count()
        loss_fn(out, target).backward()
    R.set_stage_timer(None)
    fence()
    phase("per-kernel pass done")
    d_avg = d_sum / k_prof if k_prof else 0
    if rank == 0 and k_prof:
        tot = timer.totals_ms()
        stages = {k: round(t / c, 4) for k, (t, c) in tot.items()}
        dom = max(("composite_fwd", "composite_bwd", "depth_sort", "bin_instances", "project_fwd", "build_records",
                   "preprocess_bwd"), key=lambda k: stages.get(k, 0.0))
        alg = algorithmic_bytes(dom, n, d_avg, H * W)
        ach = alg / (stages[dom] * 1e-3) / 1e9
        traffic = pmc_traffic_bytes(dom) if args.config == "c3" else None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_missing": traffic is None,
                    "algorithmic_bytes_per_launch": int(alg),
                    "avg_launch_ms": stages[dom], "avg_instances_D": int(d_avg),
                    "note": f"traffic = HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                            f"({PMC_FILE}; reads x2 as the gfx950 guide prescribes; c3 only). The kernel is "
                            f"VALU-issue bound (SQ counters in {SQ_FILE}, issue rates in profiles/r3_valu_issue_rate.txt, "
                            "DESIGN.md 4): its HBM fraction is small by construction"}

    # informational, N = 1: BASELINE config 3 read literally (forward-only orbit rendering) next to the headline
    inference = None
    if world == 1 and not args.inference and not args.no_inference_extra:
        idx = torch.arange(N_VIEWS, device=dev)
        model.render_frames(poses[idx], (H, W)); torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.render_frames(poses[idx], (H, W)); torch.cuda.synchronize()
        ti = time.perf_counter() - t1
        inference = {"value": round(N_VIEWS * H * W / ti / 1e6, 1), "unit": "Mpix/s", "ms_per_view": round(ti / N_VIEWS * 1e3, 4),
                     "what": "forward only, the 60 orbit views in one render_frames call (two streams); not the headline metric"}
    cpu = cpu_c1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # N=1 only: the other ranks would sit in the teardown meanwhile
        cpu, cpu_c1 = cpu_baselines(args, params, H, W, target_cpu[0].numpy())

    rccl = None
    if use_dp:
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "device": str(dev), "gpu": torch.cuda.get_device_name(dev)})
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks": seen,
                "bytes_per_exchange": net.last_allreduce_bytes, "zero_copy_flat_buffer": net.last_sync_was_flat, "bytes_copied_to_flatten": net.last_sync_copied_bytes,
                "schedule": net.schedule, "exchange_ms_on_scratch": net.tuned_ms,
                "exchange_ms": None if exchange_ms is None else round(exchange_ms, 4),
                "exchange_ms_note": "mean over the timed steps, HIP events around the collective on rank 0's stream"}

    if rank == 0:
        value = world * args.steps * H * W / dt / 1e6
        timed_views = [view_for(args.warmup + s) for s in range(args.steps)]
        line = {
            "metric": ("forward-only (inference) " if args.inference else "fwd+bwd ") + "splat-render Mpixels/s "
                      + ("@1M Gaussians 1024^2" if args.config == "c3" else f"@{n} Gaussians {W}x{H} (not the headline configuration)"),
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ns_per_instance": round(dt / args.steps * 1e9 / d_avg, 4) if d_avg else None,
            "config": {"workload": f"{args.config}: {n} Gaussians, {W}x{H}, 60-view orbit (radius 4, elevation 45), "
                                   f"one trainer step per view per GPU"
                                   + (" (INFERENCE: forward only, not the headline metric)" if args.inference else " (render fwd+bwd only)" if args.no_optimizer else (f" (fwd + {args.loss.upper()} + bwd + torch clip_grad_norm_ + torch Adam)" if args.torch_optimizer
                                      else f" (fwd + {args.loss.upper()} + bwd + fused clip_grad_norm+Adam)")),
                       "views_per_step": world, "parallelism": f"dp{world} over views",
                       "step_driver": ("module forward + autograd backward + optimizer.step()" if getattr(opt, "_fused_step", None) is None
                                       else "gsx_train_step (one library call per step)" if opt._fused_step.dp is None
                                       else "gsx_train_step_grads -> gradient exchange -> gsx_train_step_apply (one-call step in two halves)"),
                       "step_outputs": "rgb, alpha, mean2d, depth, cov2d written every step (the reference forward's returned tensors)",
                       "view_order": ("balanced: every orbit view once per epoch, views of similar instance count share a step"
                                      if schedule_views is not None else
                                      f"rank r, step s -> orbit view ((s*{world} + r) * {VIEW_STRIDE}) mod {N_VIEWS}"),
                       **({"step_balance": balance} if balance else {}),
                       "views_timed_rank0": {"count": len(timed_views), "distinct": len(set(timed_views)), "first": timed_views[:8]},
                       **({"gradient_exchange": net.schedule} if use_dp else {})},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_c1": cpu_c1, "rccl": rccl, "stage_ms": stages,
            "inference_orbit": inference,
        }
        print(json.dumps(line), file=json_out, flush=True)
    if use_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
