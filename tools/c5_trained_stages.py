#!/usr/bin/env python3
"""Why do config-5 trainer steps slow down as training proceeds? (dev tool)  Trains K steps over the first K orbit views,
then stage-times view K with the trained parameters and again with the original ones, and prints how many opacities have
left [0, 1].   python tools/c5_trained_stages.py [cfg] [K]"""
import sys, os, json, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd import render as R, _lib as L
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses
from gaussian_splatting_amd.trainer import make_optimizer, train_step, l1_loss
import gaussian_splatting_amd.trainer as _trainer
_trainer.FUSED_STEP = False        # one arena for the stage calls and the steps (two of config 5's do not fit)

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
n = params["positions"].shape[0]
model = GaussianSplat(n); model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); model = model.to(dev)
opt = make_optimizer(model, fused_clip_norm=0.1)
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
target = torch.rand(1, 3, H, W, device=dev)
d_rgb = torch.randn(3, H, W, device=dev)


def stages(tag):
    P = tuple(getattr(model, k).data for k in ("positions", "scales", "rotations", "opacity", "features"))
    grads = [torch.empty_like(p) for p in P]
    for rep in range(2):
        timer = R.StageTimer(); R.set_stage_timer(timer if rep else None)
        outs, vs, _ = R.forward_view(P, poses[K], H, W)
        R.backward_view(P, vs, d_rgb, None, None, None, grads, False)
        torch.cuda.synchronize()
    R.set_stage_timer(None)
    mc = vs.max_chunks
    tail = vs.chunk_state[mc * L.STATE_WORDS * 256:].view(torch.int32)
    nch = int(vs.chunk_base.cpu().numpy().view(np.uint32)[-2])
    meta = tail[:nch].cpu().numpy().view(np.uint32)
    h0 = 2 * mc + 16 + 20 * mc
    hdr = tail[h0:h0 + nch].cpu().numpy().view(np.uint32)
    pay0 = tail[h0 + 2 * mc: h0 + 2 * mc + 64 * nch].view(nch, 64)[:, 0].cpu().numpy().view(np.uint32)
    skipped = int((pay0 == 0).sum())
    why = np.bincount((meta[pay0 != 0] & 0xff).astype(np.int64), minlength=8)[:8]
    dark = tail[mc:mc + nch].cpu().numpy().view(np.uint32).view(np.uint8).reshape(nch, 4)
    print(f"   tiles with growth {int(((hdr & 0x3fffffff) != 0).sum())} (max {int((hdr & 0x3fffffff).max())} quarter-bits), unbounded {int(((hdr & 0x80000000) != 0).sum())}; "
          f"the rule calls {int((dark == 0x81).all(1).sum())} chunks dark in all quadrants, {(dark == 0x81).mean():.1%} of the quadrants")
    print(f"   chunks {nch}: skipped themselves {skipped} ({skipped / nch:.1%}); "
          f"computed chunks by reason code (1 level 0, 2 header missing/unbounded, 3 payload missing, 4 not dark, 5 partly dark): {why.tolist()}")
    cbase = vs.chunk_base.cpu().numpy().view(np.uint32)[:-1].astype(np.int64)
    alld = (dark == 0x81).all(1); sk = pay0 == 0
    lag, depth, lrule = [], [], []
    for t in range(len(cbase) - 1):
        a, b = cbase[t], cbase[t + 1]
        if b - a > 1 and alld[a:b].any():
            Lr = int(np.argmax(alld[a:b])); S = int(np.argmax(sk[a:b])) if sk[a:b].any() else int(b - a)
            lag.append(S - Lr); depth.append(int(b - a)); lrule.append(Lr)
    lag, depth, lrule = np.array(lag), np.array(depth), np.array(lrule)
    if os.environ.get("DUMP_TILES"):
        shown = 0
        for t in range(len(cbase) - 1):
            a, b = cbase[t], cbase[t + 1]
            if b - a > 1 and alld[a:b].any():
                Lr = int(np.argmax(alld[a:b])); S = int(np.argmax(sk[a:b])) if sk[a:b].any() else int(b - a)
                if S - Lr > 300 and shown < 4:
                    shown += 1
                    w = (meta[a:b] & 0xff).astype(int); w[sk[a:b]] = 0
                    runs, prev, cnt = [], None, 0
                    for v in w.tolist() + [None]:
                        if v == prev: cnt += 1
                        else:
                            if prev is not None: runs.append(f"{prev}x{cnt}")
                            prev, cnt = v, 1
                    qd = [(int(np.argmax(dark[a:b, q] == 0x81)) if (dark[a:b, q] == 0x81).any() else -1) for q in range(4)]
                    print(f"   tile {t} ({t % 120},{t // 120}) depth {b - a} rule dark point {Lr} per quadrant {qd} first skip {S}; reason runs (0 = skipped): {' '.join(runs[:60])}")
    print(f"   tiles with a dark point {len(lag)}: its level median {np.median(lrule):.0f} / 90% {np.quantile(lrule, .9):.0f}; tile depth median {np.median(depth):.0f}; "
          f"levels between the dark point and the first chunk that skipped itself: median {np.median(lag):.0f}, 75% {np.quantile(lag, .75):.0f}, 90% {np.quantile(lag, .9):.0f}, max {lag.max()}")
    o = model.opacity.data
    print(tag, "D", R.last_instance_count(), json.dumps({k: round(t / c, 2) for k, (t, c) in timer.totals_ms().items()}),
          "opacity<0:", int((o < 0).sum()), ">1:", int((o > 1).sum()), ">2:", int((o > 2).sum()), flush=True)


stages("untrained")
for s in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_step(model, opt, poses[s:s + 1], target, l1_loss, 0.1)
    torch.cuda.synchronize()
    print(f"step {s} {1e3 * (time.perf_counter() - t0):.1f} ms D={R.last_instance_count()}", flush=True)
stages("trained  ")
with torch.no_grad():
    model.opacity.clamp_(0.0, 1.0)
stages("trained, opacity clamped to [0,1]")
