#!/usr/bin/env python3
"""Full-size check of the footprint cut (run on the GPU box): the HIP path with the library's GSX_CUT_LOG2 against the CPU
oracle's TILED mode built with the round-1 cut (2^-24, floor 2^-6: within 4e-7 / 1e-5 of the dense restatement on every
scene of tools/cut_sweep.py) at BASELINE sizes, where the dense mode itself would take hours.
    python tools/cut_check_fullsize.py > gpurun_out/cut_check_fullsize.json"""
import ctypes as C, json, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import gsx_oracle as O
from helpers import synth_scene, pose_tz, orbit_pose, PARAMS, rel_err
from gaussian_splatting_amd import render as R

so = os.path.join(ROOT, "oracle", "_build", "libgsx_oracle_cut24.so")
subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-fvisibility=hidden",
                       "-DGSX_CUT_LOG2=24", "-DGSX_CUT_MIN_EXP=(-7)", "-shared", "-o", so, os.path.join(ROOT, "oracle", "gsx_oracle.c"), "-lm"])
lib = C.CDLL(so)
lib.gsxo_render_dense.restype = C.c_int64; lib.gsxo_render_tiled.restype = C.c_int64
lib.gsxo_scan.restype = C.c_int64; lib.gsxo_num_threads.restype = C.c_int
O._lib = lib
dev = torch.device("cuda:0")
CASES = {
    "c3_1M_1024_orbit10": lambda: (synth_scene(1_000_000, 5, 0.001, 0.005), orbit_pose(10), 1024, 1024),
    "c3_1M_1024_orbit30": lambda: (synth_scene(1_000_000, 5, 0.001, 0.005), orbit_pose(30), 1024, 1024),
    "c2_100k_800": lambda: (synth_scene(100_000, 2, 0.002, 0.01), pose_tz(2.0), 800, 800),
    "c5like_60k_1920x1080": lambda: (synth_scene(60_000, 8, 0.01, 0.1, zq=64.0), orbit_pose(1), 1080, 1920),
}
out = {}
for name, make in CASES.items():
    params, pose, H, W = make()
    P = tuple(torch.from_numpy(params[k]).to(dev) for k in PARAMS)
    pose_t = torch.from_numpy(pose).to(dev)
    d_rgb = np.random.default_rng(9).standard_normal((3, H, W)).astype(np.float32)
    outs, vs, _ = R.forward_view(P, pose_t, H, W)
    grads = [torch.empty_like(p) for p in P]
    R.backward_view(P, vs, torch.from_numpy(d_rgb).to(dev), None, None, None, grads, accumulate=False)
    torch.cuda.synchronize()
    O._lib = lib
    fw = O.forward(params, pose, H, W, "tiled")
    go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
    O._lib = None; O.lib()                        # the oracle at the library's own cut: separates the cut from float32 arithmetic
    fn = O.forward(params, pose, H, W, "tiled")
    gn = O.backward(params, pose, H, W, fn, d_rgb, None, None, None, "tiled")
    rgb = outs[0].cpu().numpy()
    out[name] = {"D_hip": int(vs.D), "D_oracle_cut24": int(fw["D"]),
                 "hip_vs_oracle_cut24": {"rgb_max_abs": float(np.abs(rgb - fw["rgb"]).max()),
                                         "grad_rel": {k: rel_err(g.cpu().numpy(), go[k]) for k, g in zip(PARAMS, grads)}},
                 "hip_vs_oracle_same_cut": {"rgb_max_abs": float(np.abs(rgb - fn["rgb"]).max()),
                                            "grad_rel": {k: rel_err(g.cpu().numpy(), gn[k]) for k, g in zip(PARAMS, grads)}},
                 "oracle_same_cut_vs_cut24": {"rgb_max_abs": float(np.abs(fn["rgb"] - fw["rgb"]).max()),
                                              "grad_rel": {k: rel_err(gn[k], go[k]) for k in PARAMS}}}
    print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
