"""Forward occlusion skipping on the bench orbit: per view, chunks in total / in the dark run (combine pass's rule) / that
skipped themselves in the chunk pass, and the chunk pass's duration.  python tools/skip_stats.py [c3] [views...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R, _lib as L
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
views = [int(v) for v in sys.argv[2:]] or [0, 5, 10, 15, 20, 30]
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
for v in views:
    for rep in range(2):
        timer = R.StageTimer(); R.set_stage_timer(timer)
        outs, vs, _ = R.forward_view(P, poses[v], H, W, want_depth=False)
        R.set_stage_timer(None)
        tot = timer.totals_ms()
    mc = vs.max_chunks
    cs = vs.chunk_state
    tail = cs[mc * L.STATE_WORDS * 256:].view(torch.int32).cpu().numpy().view(np.uint32)
    nch = int(vs.chunk_base.cpu().numpy().view(np.uint32)[-2])
    dark = tail[mc:mc + nch]
    h0 = 2 * mc + 16 + 20 * mc
    hdr = tail[h0:h0 + nch]
    pay = tail[h0 + 2 * mc: h0 + 2 * mc + 64 * nch].reshape(nch, 64)
    skipped = int((pay[:, 0] == 0).sum()); unpublished = 0
    db = dark.view(np.uint8).reshape(nch, 4)
    darkrun = int((db == 0x81).all(1).sum()); darkq = int((db == 0x81).sum())
    print(f"view {v}: D={vs.D} chunks={nch} all-4-quadrants-dark={darkrun} ({darkrun / max(nch, 1):.0%}) dark quadrants={darkq / max(4 * nch, 1):.0%} "
          f"self-skipped={skipped} ({skipped / max(nch, 1):.0%}) unpublished={unpublished} unbounded tiles={int(((hdr & 0x80000000) != 0).sum())} "
          f"composite_fwd={tot['composite_fwd'][0]:.3f} ms", flush=True)
    # where the dark point lies: per tile, the level of its first all-dark chunk
    cbase = vs.chunk_base.cpu().numpy().view(np.uint32)[:-1].astype(np.int64)
    alld = (db == 0x81).all(1)
    js, ns = [], []
    for t in range(len(cbase) - 1):
        a, b = cbase[t], cbase[t + 1]
        if b - a > 1 and alld[a:b].any():
            js.append(int(np.argmax(alld[a:b]))); ns.append(int(b - a))
    if js:
        js = np.array(js); ns = np.array(ns)
        print(f"   tiles with a dark point: {len(js)}; its level: median {np.median(js):.0f}, 90% {np.quantile(js, 0.9):.0f}, max {js.max()}; "
              f"dark chunks behind a point at level <= 8: {int((ns - js)[js <= 8].sum())}, <= 16: {int((ns - js)[js <= 16].sum())}, all: {int((ns - js).sum())}", flush=True)
