#!/usr/bin/env python3
"""How many chunks start fully occluded (max over the tile's pixels of the transmittance entering the chunk below eps)? (dev tool)"""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
views = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 13, 27, 44]
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
for v in views:
    outs, vs, extra = R.forward_view(P, poses[v], H, W)
    torch.cuda.synchronize()
    cb = vs.chunk_base.cpu().numpy().astype(np.int64)[: vs.ranges.shape[0] + 1]
    nch = int(cb[-1])
    st = vs.chunk_state[: nch * 8 * 256].view(nch, 8, 256)
    T_end = torch.ldexp(st[:, 6, :], st[:, 7, :].view(torch.int32))          # (nch, 256)
    first = torch.zeros(nch, dtype=torch.bool, device=dev)
    first[torch.from_numpy(cb[:-1][cb[:-1] < nch]).to(dev)] = True
    T_start = torch.ones_like(T_end); T_start[1:] = T_end[:-1]; T_start[first] = 1.0
    mx = T_start.abs().max(dim=1).values
    rng = vs.ranges.cpu().numpy().astype(np.int64)
    lens = rng[:, 1] - rng[:, 0]
    # instances per chunk
    inst = np.zeros(nch, np.int64)
    for t in np.nonzero(lens)[0]:
        c0, c1 = cb[t], cb[t + 1]
        inst[c0:c1] = 256; inst[c1 - 1] = lens[t] - 256 * (c1 - c0 - 1)
    inst_t = torch.from_numpy(inst).to(dev)
    res = {"view": v, "D": vs.D, "chunks": nch}
    for eps in (2.0 ** -10, 2.0 ** -16, 2.0 ** -24):
        sk = mx < eps
        res[f"skip_chunks<{eps:.1e}"] = round(float(sk.float().mean()), 3)
        res[f"skip_inst<{eps:.1e}"] = round(float(inst_t[sk].sum()) / max(vs.D, 1), 3)
    print(json.dumps(res))
    # forward level schemes: a chunk with local index j >= b is skipped iff its tile is dark by the last boundary b <= j
    dark = (mx < 2.0 ** -24).cpu().numpy()
    tiles = np.nonzero(lens)[0]
    res2 = {}
    for name, bounds in {"4,16": (4, 16), "8,24": (8, 24), "2,6,14,30": (2, 6, 14, 30), "4,12,28": (4, 12, 28), "8": (8,), "16": (16,), "20": (20,), "24": (24,), "28": (28,), "32": (32,), "16,32": (16, 32), "12,24": (12, 24)}.items():
        sk = 0
        for t in tiles:
            c0, c1 = cb[t], cb[t + 1]; nc = c1 - c0
            d = dark[c0:c1]                      # d[j]: tile dark at the START of local chunk j
            for bi, b in enumerate(bounds):
                if b >= nc: break
                hi = bounds[bi + 1] if bi + 1 < len(bounds) else nc
                if d[b]:
                    sk += nc - b; break
        res2[name] = round(sk / nch, 3)
    print(json.dumps({"view": v, "fwd_skippable_by_levels": res2, "ideal": round(float(dark.mean()), 3)}))
