#!/usr/bin/env python3
"""bench.py -- fwd+bwd splat-render throughput of the gfx950 path on BASELINE.json's metric
configuration: 1M Gaussians, 1024x1024, the 60-view orbit of config 3 (SURVEY 8d seed 5).

    python bench.py --gpus 1 --steps 60 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one trainer step on one view per GPU (trainer.py:134-175 sequence): zero_grad ->
render forward -> L1 loss -> render backward -> [N>1: one RCCL all-reduce of the flat 14*N
gradient buffer, or the all-to-all + all-gather mesh schedule if that timed faster at start-up] -> clip_grad_norm_(0.1) -> Adam step.  View of rank r at step s is
(s*N + r) mod 60: views are sharded over GPUs, per-GPU work is fixed (weak scaling).
value = N * K * H * W / max-over-ranks wall time / 1e6 (Mpix/s), inputs resident in HBM.

Rank 0 also reports (same JSON line):
  roofline      dominant kernel's algorithmic HBM bytes per launch / its HIP-event duration vs 8 TB/s
  cpu_baseline  the CPU oracle (tiled restatement, "port") timed on the host for ONE view
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
# stage -> kernel whose PMC traffic (profiles/r1_pmc_hbm_traffic.json, separate rocprofv3 --pmc passes) belongs to it
STAGE_KERNEL = {"composite_bwd": "gsx::chunk_bwd_kernel", "composite_fwd": "gsx::chunk_fwd_kernel<false, true>",
                "preprocess_bwd": "gsx::preprocess_bwd_kernel", "build_records": "gsx::build_records_kernel",
                "project_fwd": "gsx::project_fwd_kernel"}


def pmc_traffic_bytes(stage):
    """HBM bytes per launch of the stage's kernel from the committed PMC summary (None if absent)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.json")) as f:
            k = json.load(f)["kernels"][STAGE_KERNEL[stage]]
        return int(k["hbm_total_MB"] * 1024 * 1024)
    except (OSError, KeyError, ValueError):
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
    ap.add_argument("--dp-schedule", default=None, choices=("allreduce", "mesh", "auto"),
                    help="gradient exchange: one RCCL all-reduce (default), all-to-all + local sum + all-gather (SURVEY 8e's mesh "
                         "schedule), or auto = time both on a scratch buffer at the first exchange and keep the faster")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="use torch clip_grad_norm_ + torch.optim.Adam instead of the fused HIP clip+Adam step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_dp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd import render as R
    from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
    from gaussian_splatting_amd.distributed import DataParallelSplat
    from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss, l1_ssim_loss
    loss_fn = l1_ssim_loss if args.loss == "l1_ssim" else l1_loss

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
    if use_dp and net.schedule == "auto":
        net.tune_schedule(14 * n, dev)
    opt = make_optimizer(net, fused_clip_norm=None if args.torch_optimizer else 0.1)
    n_views = 60
    poses = torch.from_numpy(create_orbit_poses(4.0, n_views, 45.0)).to(dev)      # README.md:129-137
    g = torch.Generator(device="cpu").manual_seed(3)
    target = torch.rand(1, 3, H, W, generator=g).to(dev)

    def step(s):
        v = (s * world + rank) % n_views
        if args.inference:
            raise RuntimeError("inference steps go through orbit_pass()")
        elif args.no_optimizer:
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
        idx = torch.tensor([((first + i) * world + rank) % n_views for i in range(count)], device=dev)
        model.render_frames(poses[idx], (H, W))

    def run(first, count):
        if args.inference:
            done = 0
            while done < count:
                k = min(n_views, count - done)
                orbit_pass(first + done, k); done += k
        else:
            for s in range(count):
                step(first + s)

    run(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel durations (HIP events on the launch stream) over one more orbit pass.  Every rank runs
    # the loop (the gradient all-reduce inside backward is collective); rank 0 records the events. -----------
    roofline = None
    stages = {}
    timer = R.StageTimer() if rank == 0 else None
    R.set_stage_timer(timer)
    d_sum = 0
    k_prof = 0 if args.inference else min(args.steps, n_views)
    for s in range(k_prof):
        v = (s * world + rank) % n_views
        opt.zero_grad()
        out = net(poses[v:v + 1], (H, W))
        d_sum += R.last_instance_count()
        loss_fn(out, target).backward()
    R.set_stage_timer(None)
    fence()
    if rank == 0 and k_prof:
        tot = timer.totals_ms()
        d_avg = d_sum / k_prof
        stages = {k: round(t / c, 4) for k, (t, c) in tot.items()}
        dom = max(("composite_fwd", "composite_bwd", "depth_sort", "bin_instances", "project_fwd", "build_records",
                   "preprocess_bwd"), key=lambda k: stages.get(k, 0.0))
        alg = algorithmic_bytes(dom, n, d_avg, H * W)
        ach = alg / (stages[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic_bytes(dom) if args.config == "c3" else None,
                    "algorithmic_bytes_per_launch": int(alg),
                    "avg_launch_ms": stages[dom], "avg_instances_D": int(d_avg),
                    "note": "traffic = HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                            "(profiles/r1_pmc_hbm_traffic.json; reads x2 as the gfx950 guide prescribes). The kernel is "
                            "VALU-issue bound (85% of the chip's VALU issue slots, SQ counters in profiles/r1_pmc_sq_counters.json, "
                            "DESIGN.md 4): its HBM fraction is small by construction"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # N=1 only: the other ranks would sit in the teardown meanwhile
        from oracle import gsx_oracle as O          # checker / reported baseline only
        O.lib()
        all_poses = create_orbit_poses(4.0, n_views, 45.0)
        d_rgb = np.sign(np.random.default_rng(1).standard_normal((3, H, W))).astype(np.float32) / (3 * H * W)
        sample_views, ds = (10, 30, 50), []
        t1 = time.perf_counter()
        for v in sample_views:                         # ~4 s of 16-thread CPU work per view
            fw = O.forward(params, all_poses[v], H, W, "tiled")
            O.backward(params, all_poses[v], H, W, fw, d_rgb, None, None, None, "tiled")
            ds.append(int(fw["D"]))
        tc = time.perf_counter() - t1
        cpu = {"value": round(len(sample_views) * H * W / tc / 1e6, 4), "unit": "Mpix/s", "cores": O.num_threads(), "kind": "port",
               "sample": f"{len(sample_views)} views (orbit views {sample_views}, D={ds}) of the same workload, tiled CPU restatement "
                         f"fwd+bwd (render only, no loss/optimizer), {tc:.2f} s; the reference's own dense loop cannot run at this "
                         "size (16 ns*N*P, SURVEY 6)"}

    if rank == 0:
        value = world * args.steps * H * W / dt / 1e6
        line = {
            "metric": ("forward-only (inference) " if args.inference else "fwd+bwd ") + "splat-render Mpixels/s "
                      + ("@1M Gaussians 1024^2" if args.config == "c3" else f"@{n} Gaussians {W}x{H} (not the headline configuration)"),
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n} Gaussians, {W}x{H}, 60-view orbit (radius 4, elevation 45), "
                                   f"one trainer step per view per GPU"
                                   + (" (INFERENCE: forward only, not the headline metric)" if args.inference else " (render fwd+bwd only)" if args.no_optimizer else (f" (fwd + {args.loss.upper()} + bwd + torch clip_grad_norm_ + torch Adam)" if args.torch_optimizer
                                      else f" (fwd + {args.loss.upper()} + bwd + fused clip_grad_norm+Adam)")),
                       "views_per_step": world, "parallelism": f"dp{world} over views",
                       **({"gradient_exchange": net.schedule, "exchange_ms_on_scratch": net.tuned_ms} if use_dp else {})},
            "roofline": roofline, "cpu_baseline": cpu, "stage_ms": stages,
        }
        print(json.dumps(line), flush=True)
    if use_dp:
        if rank == 0:
            print(f"[bench] gradient all-reduce: {net.last_allreduce_bytes} B per step, "
                  f"zero-copy flat buffer: {net.last_sync_was_flat}", file=sys.stderr)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
