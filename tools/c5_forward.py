"""Forward time of config-5 views (6 M splats of up to ~100 px, 1920x1080) and how many chunks the occlusion rule makes dark /
the chunk pass skipped.  python tools/c5_forward.py [views...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R, _lib as L
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

views = [int(v) for v in sys.argv[1:]] or [1, 2, 57]
dev = torch.device("cuda:0")
params, H, W = config_scene("c5")
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
for v in views:
    R.forward_view(P, poses[v], H, W, want_depth=False)           # settles the capacity
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); outs, vs, _ = R.forward_view(P, poses[v], H, W, want_depth=False); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    mc = vs.max_chunks
    tail = vs.chunk_state[mc * L.STATE_WORDS * 256:].view(torch.int32)
    nch = int(vs.chunk_base.cpu().numpy().view(np.uint32)[-2])
    h0 = 2 * mc + 16 + 20 * mc
    hdr = tail[h0:h0 + nch].cpu().numpy().view(np.uint32)
    pay0 = tail[h0 + 2 * mc: h0 + 2 * mc + 64 * nch].view(-1, 64)[:, 0].cpu().numpy().view(np.uint32)
    db = tail[mc:mc + nch].cpu().numpy().view(np.uint8).reshape(nch, 4)
    meta = tail[0:nch].cpu().numpy().view(np.uint32) & 0xff
    print("   why not skipped (1 level 0, 2 unbounded tile, 3 nothing proven and a payload missing, 4 nothing proven, 5 partly dark, 0 skipped or shallow view):",
          {int(k): int(c) for k, c in zip(*np.unique(meta, return_counts=True))})
    print(f"view {v}: D={vs.D} chunks={nch} forward {min(ts):.2f} ms; chunks dark by the rule {int((db == 0x81).all(1).sum()) / nch:.1%}, "
          f"skipped by the chunk pass {int((pay0 == 0).sum()) / nch:.1%}, unbounded tiles {int(((hdr & 0x80000000) != 0).sum())}, tiles with growth {int(((hdr & 0x3FFFFFFF) != 0).sum())}, checksum {float(outs[0].double().sum()):.6f}", flush=True)
    del outs, vs, tail
