#!/usr/bin/env python3
"""Error-vs-cut curve of the footprint cut: for GSX_CUT_LOG2 in {24, 22, 20, 18, 16, 14} build a variant
of the CPU oracle and compare its TILED mode (the algorithm the HIP kernels implement, bit-exact in its binning) with
  * its DENSE mode (the reference algorithm: every pair, no cut) on down-scaled instances of the BASELINE configs, and
  * the reference's own outputs in tests/golden/*.npz,
reporting RGB max-abs, gradient error (max|d| / max|ref| per tensor, worst tensor), instance count D and pair count.
CPU only.   python tools/cut_sweep.py [--quick] > profiles/r2_cut_sweep.json"""
import ctypes as C, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import gsx_oracle as O
from helpers import golden_names, load_golden, synth_scene, default_scene, pose_tz, orbit_pose, PARAMS, rel_err, rgb_err

# variants: GSX_CUT_LOG2, or "cut:minexp" to move the opacity floor GSX_CUT_MIN_EXP with it
CUTS = (24, 22, 20, 18, 16, 14)
quick = "--quick" in sys.argv
for a in sys.argv[1:]:
    if a.startswith("--variants="):
        CUTS = tuple(a[len("--variants="):].split(","))


def variant(cut):
    log2, _, minexp = str(cut).partition(":")
    so = os.path.join(ROOT, "oracle", "_build", f"libgsx_oracle_cut{log2}_{minexp or 'd'}.so")
    defs = [f"-DGSX_CUT_LOG2={log2}"] + ([f"-DGSX_CUT_MIN_EXP=({minexp})"] if minexp else [])
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-fvisibility=hidden",
                           *defs, "-shared", "-o", so, os.path.join(ROOT, "oracle", "gsx_oracle.c"), "-lm"])
    lib = C.CDLL(so)
    lib.gsxo_render_dense.restype = C.c_int64; lib.gsxo_render_tiled.restype = C.c_int64
    lib.gsxo_scan.restype = C.c_int64; lib.gsxo_num_threads.restype = C.c_int
    return lib


SCENES = {
    "c1_default_3000_200x150": lambda: (default_scene(3000, 1), pose_tz(3.0), 150, 200),
    "c2like_20000_256x256": lambda: (synth_scene(20000, 2, 0.002, 0.01), pose_tz(2.0), 256, 256),
    "c3like_30000_256x256_orbit15": lambda: (synth_scene(30000, 5, 0.001, 0.005), orbit_pose(15), 256, 256),
    "c3like_30000_256x256_orbit10": lambda: (synth_scene(30000, 5, 0.001, 0.005), orbit_pose(10), 256, 256),
    "c5like_3000_480x270_ties": lambda: (synth_scene(3000, 8, 0.01, 0.1, zq=64.0), pose_tz(2.0), 270, 480),
    "faint_2000_128x96": lambda: (synth_scene(2000, 9, 0.005, 0.03, o_lo=0.0, o_hi=0.02), pose_tz(2.0), 96, 128),
}
if quick:
    SCENES = {k: SCENES[k] for k in ("c2like_20000_256x256", "c3like_30000_256x256_orbit15")}

out = {"cuts": [str(c) for c in CUTS], "scenes_vs_dense": {}, "fixtures_vs_reference": {}}
dense_cache = {}
for cut in CUTS:
    O._lib = variant(cut)
    for name, make in SCENES.items():
        params, pose, H, W = make()
        d_rgb = np.random.default_rng(7).standard_normal((3, H, W)).astype(np.float32)
        if name not in dense_cache:                      # the dense mode does not depend on the cut
            t = time.time()
            fd = O.forward(params, pose, H, W, "dense")
            gd = O.backward(params, pose, H, W, fd, d_rgb, None, None, None, "dense")
            dense_cache[name] = (fd["rgb"], fd["n_composited"], {k: gd[k] for k in PARAMS})
            print(f"[dense {name}: {time.time() - t:.1f} s]", file=sys.stderr, flush=True)
        rgb_d, k_d, g_d = dense_cache[name]
        ft = O.forward(params, pose, H, W, "tiled")
        gt = O.backward(params, pose, H, W, ft, d_rgb, None, None, None, "tiled")
        row = {"D": int(ft["D"]), "rgb_max_abs": float(np.abs(ft["rgb"] - rgb_d).max()),
               "exit_index_equal": bool(ft["n_composited"] == k_d),
               "grad_rel_worst": max(rel_err(gt[k], g_d[k]) for k in PARAMS),
               "grad_rel": {k: rel_err(gt[k], g_d[k]) for k in PARAMS}}
        out["scenes_vs_dense"].setdefault(name, {})[str(cut)] = row
        print(f"cut 2^-{cut} {name}: D={row['D']} rgb={row['rgb_max_abs']:.2e} grad={row['grad_rel_worst']:.2e}", file=sys.stderr, flush=True)
    for name in golden_names(include_big=not quick):
        fx = load_golden(name)
        H, W = fx["H"], fx["W"]
        ft = O.forward(fx["params"], fx["poses"][0], H, W, "tiled", order=fx["order"] if fx["has_ties"] else None)
        gt = O.backward(fx["params"], fx["poses"][0], H, W, ft, fx["d_rgb"][0], fx["d_pos2d"][0], fx["d_depths"][0], fx["d_cov2d"][0], "tiled")
        row = {"D": int(ft["D"]), "rgb_max_abs": rgb_err(ft["rgb"], fx), "exit_index_equal": bool(ft["n_composited"] == int(fx["n_composited"])),
               "grad_rel_worst": max(rel_err(gt[k], fx["g_" + k]) - 2.0 * float(fx.get("ref_noise_" + k, 0.0)) for k in PARAMS)}
        out["fixtures_vs_reference"].setdefault(name, {})[str(cut)] = row
summary = {}
for cut in CUTS:
    c = str(cut)
    summary[c] = {
        "rgb_max_abs_vs_dense": max(v[c]["rgb_max_abs"] for v in out["scenes_vs_dense"].values()),
        "grad_rel_vs_dense": max(v[c]["grad_rel_worst"] for v in out["scenes_vs_dense"].values()),
        "rgb_max_abs_vs_reference": max(v[c]["rgb_max_abs"] for v in out["fixtures_vs_reference"].values()),
        "grad_rel_vs_reference_minus_2noise": max(v[c]["grad_rel_worst"] for v in out["fixtures_vs_reference"].values()),
        "exit_index_equal_everywhere": all(v[c]["exit_index_equal"] for d in (out["scenes_vs_dense"], out["fixtures_vs_reference"]) for v in d.values()),
        "D_relative_to_first_variant": {k: round(v[c]["D"] / max(v[str(CUTS[0])]["D"], 1), 4) for k, v in out["scenes_vs_dense"].items()},
    }
out["summary"] = summary
print(json.dumps(out, indent=1))
