import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_scene, default_scene, pose_tz, orbit_pose, rel_err, PARAMS
import test_gpu_parity as T
from oracle import gsx_oracle as O
from gaussian_splatting_amd import render as R
O.lib(); dev = torch.device("cuda:0")
def make(seed):
    g = np.random.default_rng(seed)
    n = int(g.choice([1, 2, 7, 64, 300, 1000, 3000]))
    H = int(g.choice([1, 2, 15, 16, 17, 33, 48, 100, 131, 256])); W = int(g.choice([1, 3, 16, 31, 64, 97, 160, 200, 333]))
    lo = float(g.choice([0.001, 0.004, 0.02, 0.1])); hi = lo * float(g.choice([1.5, 5.0, 20.0]))
    o_lo, o_hi = [(0.0, 1.0), (0.9, 1.0), (-0.5, 1.5), (0.0, 0.05), (1.0, 3.0)][int(g.integers(0, 5))]
    zq = [None, None, 64.0][int(g.integers(0, 3))]
    kind = int(g.integers(0, 3))
    params = default_scene(n, seed) if kind == 0 else synth_scene(n, seed, lo, hi, zq=zq, o_lo=o_lo, o_hi=o_hi)
    pose = pose_tz(float(g.choice([1.0, 2.0, 3.0, 6.0]))) if g.random() < 0.6 else orbit_pose(int(g.integers(0, 60)))
    fc = int(g.choice([0, 0, 64, 128, 256]))
    return g, n, H, W, params, pose, fc
for seed in [int(a) for a in sys.argv[1:]]:
    g, n, H, W, params, pose, fc = make(seed)
    R._scratch.force_chunk = fc
    gp = T.gpu_pipeline(params, pose, H, W, dev)
    fw = O.forward(params, pose, H, W, "tiled")
    fd = O.forward(params, pose, H, W, "dense")
    print(f"seed {seed} n={n} {W}x{H} D={fw['D']} exit gpu/tiled/dense = {gp['n_composited']}/{fw['n_composited']}/{fd['n_composited']}",
          "alpha min gpu/tiled/dense", float(np.nanmin(gp['alpha'])), float(np.nanmin(fw['alpha'])), float(np.nanmin(fd['alpha'])),
          "rgb nan gpu/tiled/dense", int(np.isnan(gp['rgb']).sum()), int(np.isnan(fw['rgb']).sum()), int(np.isnan(fd['rgb']).sum()))
    d_rgb = g.standard_normal((3, H, W)).astype(np.float32)
    d_m = g.standard_normal((n, 2)).astype(np.float32); d_d = g.standard_normal(n).astype(np.float32)
    d_c = g.standard_normal((n, 2, 2)).astype(np.float32)
    gr = T.gpu_backward(gp, H, W, d_rgb, d_m, d_d, d_c)
    go = O.backward(params, pose, H, W, fw, d_rgb, d_m, d_d, d_c, "tiled")
    gd = O.backward(params, pose, H, W, fd, d_rgb, d_m, d_d, d_c, "dense")
    for k in PARAMS:
        a, b, c = gr[k].astype(np.float64), go[k].astype(np.float64), gd[k].astype(np.float64)
        i = int(np.nanargmax(np.abs(a - b))) if a.size else 0
        print(f"   {k:10s} gpu-vs-tiled {rel_err(gr[k], go[k]):.3e}  gpu-vs-dense {rel_err(gr[k], gd[k]):.3e}  tiled-vs-dense {rel_err(go[k], gd[k]):.3e}"
              f"  max|tiled| {np.nanmax(np.abs(b)):.3e} worst idx {i}: gpu {a.flat[i]:.6e} tiled {b.flat[i]:.6e} dense {c.flat[i]:.6e} o={params['opacity'].flat[i // max(a.shape[-1],1)] if k=='opacity' else ''}")
