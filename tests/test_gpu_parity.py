"""GPU tier (-m gpu): the HIP path, called through the C ABI (ctypes, include/gsx.h), against
the CPU oracle on the same inputs and against the golden vectors of the reference.

Bars (BASELINE.json north_star): integer products (depth order, footprint boxes, tiles_touched,
offsets, (tile<<32|rank) keys, tile ranges, early-exit index) BIT-EXACT; rendered RGB / depth
map within 1e-4 max-abs; parameter gradients within 1e-3 relative (to the tensor's max).
"""
import numpy as np
import pytest
import torch

from helpers import (golden_names, load_golden, rel_err, elem_rel_err, l2_rel_err, grad_tol, rgb_err, RGB_TOL, GRAD_RTOL, PARAMS,
                     synth_scene, default_scene, pose_tz, orbit_pose)

pytestmark = pytest.mark.gpu
from gaussian_splatting_amd._lib import CHUNK as L_CHUNK


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import gsx_oracle
    gsx_oracle.lib()
    return gsx_oracle


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def gpu_pipeline(params, pose, H, W, dev, order=None):
    """Runs the path stage by stage through the C-ABI wrappers; returns numpy views of every product."""
    from gaussian_splatting_amd import render as R
    P = tuple(torch.from_numpy(np.ascontiguousarray(params[k], np.float32)).to(dev) for k in PARAMS)
    pose_t = torch.from_numpy(np.ascontiguousarray(pose, np.float32)).to(dev)
    mean2d, depth, cov2d, keys, pre = R.project(P[0], P[1], P[2], P[3], P[4], pose_t)
    order_t = R.depth_order(keys) if order is None else torch.from_numpy(order.astype(np.int32)).to(dev)
    rec, tt, rank_of = R.build_records(pre, order_t, H, W)
    inv = np.empty(P[0].shape[0], np.uint32); inv[_u32(order_t)] = np.arange(P[0].shape[0], dtype=np.uint32)
    assert np.array_equal(_u32(rank_of), inv), "rank_of is the inverse permutation of order"
    offsets, tiles, ranks, ranges, chunk_base, D = R.bin_instances(rec, tt, H, W)
    n = P[0].shape[0]
    rgb, alpha, dmap, state, chunk_state, max_chunks = R.composite_forward(rec, ranks, ranges, chunk_base, n, H, W, D, True, True)
    torch.cuda.synchronize()
    s = int(_u32(state)[0])
    out = dict(mean2d=mean2d.cpu().numpy(), depth=depth.cpu().numpy(), cov2d=cov2d.cpu().numpy(), keys=_u32(keys),
               order=_u32(order_t), rec=rec.cpu().numpy(), tiles_touched=_u32(tt), offsets=_u32(offsets)[:n], D=D,
               tiles=_u32(tiles)[:D], ranks=_u32(ranks)[:D], ranges=_u32(ranges), rgb=rgb.cpu().numpy(),
               alpha=alpha.cpu().numpy()[0], depth_map=dmap.cpu().numpy()[0],
               n_composited=(s + 1 if (s != 0xFFFFFFFF and s + 1 < n) else n))
    cb = _u32(chunk_base)
    lens = (out["ranges"][:, 1] - out["ranges"][:, 0]).astype(np.int64)
    ch = int(cb[-1])                                   # the chunk size the device picked: a function of D alone
    assert ch == (R._scratch.force_chunk or (64 if D < (1 << 19) else 128 if D < (1 << 20) else L_CHUNK))
    assert np.array_equal(cb[:-1], np.concatenate([[0], np.cumsum((lens + ch - 1) // ch)]).astype(np.uint32)), "chunk table"
    assert int(cb[-2]) <= max_chunks
    out["_t"] = dict(P=P, pose=pose_t, order=order_t, rank_of=rank_of, rec=rec, tt=tt, offsets=offsets, ranks=ranks, ranges=ranges,
                     chunk_base=chunk_base, chunk_state=chunk_state, max_chunks=max_chunks, state=state, D=D)
    return out


def gpu_backward(gp, H, W, d_rgb, d_mean2d=None, d_depth=None, d_cov=None):
    from gaussian_splatting_amd import render as R
    t = gp["_t"]; dev = t["rec"].device
    vs = R.ViewState()
    vs.pose, vs.order, vs.rank_of, vs.rec, vs.tt, vs.offsets = t["pose"], t["order"], t["rank_of"], t["rec"], t["tt"], t["offsets"]
    vs.ranks, vs.ranges, vs.state, vs.D, vs.H, vs.W = t["ranks"], t["ranges"], t["state"], t["D"], H, W
    vs.chunk_base, vs.chunk_state, vs.max_chunks = t["chunk_base"], t["chunk_state"], t["max_chunks"]
    grads = [torch.empty_like(p) for p in t["P"]]
    cv = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    R.backward_view(t["P"], vs, cv(d_rgb), cv(d_mean2d), cv(d_depth), cv(d_cov), grads, accumulate=False)
    torch.cuda.synchronize()
    return {k: g.cpu().numpy() for k, g in zip(PARAMS, grads)}


# Per-element view of the gradient error: the tensor-level figure max|d| / max|ref| lets a mid-sized
# element be far off.  Two more figures per tensor, asserted and written to gpurun_out/r3_grad_errors.json:
#   L2-relative  |g - ref|_2 / |ref|_2                                                   bound ELEM_L2_TOL
#   per-element  max over elements with |ref| > 1e-3 max|ref| of |g - ref| / |ref|        bound ELEM_REL_TOL
# The per-element bound cannot be 1e-3: an element at the floor is 1000x smaller than the sums of +/- terms it is the
# float32 remainder of (the reference's own float32 result has the same property against a float64 evaluation).
# Measured over all fixtures and full-size cases (r2): L2 <= 7e-5 and per-element <= 1.3e-2 (config 1, one position
# gradient 1000x below the tensor's maximum: an absolute error of 1.3e-5 of that maximum) everywhere except exit8 (the
# fixture with a near-singular Gaussian, 8.8e-4 and 1.8e-2) -- and the ORACLE's per-element figure against the reference is
# the same size (up to 8e-3): that is the float32 noise floor on small elements of long sums, the reference's as much as
# this path's.
ELEM_L2_TOL = 2e-4           # + 2 x the reference's recorded float32 noise for the tensor (exit8 only)
ELEM_REL_TOL = 3e-2          # x (1 + 2 x noise / 1e-3)
_GRAD_REPORT = {}


def check_elementwise(case, k, g, ref, oracle=None, fx=None):
    import json, os
    row = {"tensor_rel_vs_ref": rel_err(g, ref), "l2_rel_vs_ref": l2_rel_err(g, ref)}
    row["elem_rel_vs_ref"], row["elements_above_floor"] = elem_rel_err(g, ref)
    if oracle is not None:
        row["l2_rel_vs_oracle"] = l2_rel_err(g, oracle)
        row["elem_rel_vs_oracle"], _ = elem_rel_err(g, oracle)
        row["oracle_elem_rel_vs_ref"], _ = elem_rel_err(oracle, ref)
    if fx is not None and ("ref_noise_" + k) in fx:
        row["ref_float32_noise"] = float(fx["ref_noise_" + k])
    _GRAD_REPORT.setdefault(case, {})[k] = row
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "r3_grad_errors.json"), "w") as f:
            json.dump(_GRAD_REPORT, f, indent=1)
    except OSError:
        pass
    noise = float(fx.get("ref_noise_" + k, 0.0)) if fx is not None else 0.0
    assert row["l2_rel_vs_ref"] < ELEM_L2_TOL + 2.0 * noise, (case, k, row)
    assert row["elem_rel_vs_ref"] < ELEM_REL_TOL * (1.0 + 2.0 * noise / GRAD_RTOL), (case, k, row)


def assert_integer_products_equal(gp, fw):
    """bit-exact binning/sort contract"""
    assert np.array_equal(gp["order"], fw["order"]), "depth order"
    assert np.array_equal(gp["rec"].view(np.uint32)[:, 10:12], fw["rec"].view(np.uint32)[:, 10:12]), "footprint boxes"
    assert np.array_equal(gp["tiles_touched"], fw["tiles_touched"]), "tiles_touched"
    assert gp["D"] == fw["D"], "instance count"
    assert np.array_equal(gp["offsets"].astype(np.uint64), fw["offsets"]), "offsets"
    keys = (gp["tiles"].astype(np.uint64) << np.uint64(32)) | gp["ranks"].astype(np.uint64)
    assert np.array_equal(keys, fw["keys"]), "sorted (tile<<32|rank) keys"
    assert np.array_equal(gp["ranges"], fw["ranges"]), "tile ranges"


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 63, 256, 4096, 4097, 70001, 1 << 20])
@pytest.mark.parametrize("bits", [(0, 32), (0, 12), (4, 9), (0, 1), (0, 25), (3, 32), (0, 24)])
def test_radix_sort_stable(dev, n, bits):
    import ctypes as C
    from gaussian_splatting_amd import _lib as L
    from gaussian_splatting_amd.render import _ptr, _stream
    lib = L.lib()
    rng = np.random.default_rng(n + bits[1])
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if n > 1000:
        keys[::3] = keys[0]                     # heavy duplicates
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    k = torch.from_numpy(keys.view(np.int32)).to(dev); v = torch.from_numpy(vals.view(np.int32)).to(dev)
    ko, vo, kt, vt = (torch.empty_like(k) for _ in range(4))
    wsb = lib.gsx_sort_workspace_bytes(n); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    check_bit = 20 + (n % 9)
    mism = torch.full((1,), -1, dtype=torch.int64, device=dev)
    L.check(lib.gsx_sort_pairs_u32(_ptr(k), _ptr(v), _ptr(ko), _ptr(vo), _ptr(kt), _ptr(vt), n, bits[0], bits[1],
                                   check_bit, _ptr(mism), _ptr(ws), wsb, _stream()), "sort")
    torch.cuda.synchronize()
    digit = (keys >> np.uint32(bits[0])) & np.uint32((1 << (bits[1] - bits[0])) - 1)
    perm = np.argsort(digit, kind="stable")
    assert np.array_equal(_u32(ko), keys[perm])
    assert np.array_equal(_u32(vo), vals[perm])
    # the count of keys whose bits [check_bit, 32) differ from the first key's (what lets a caller skip constant high bits)
    assert int(mism.item()) == int(np.count_nonzero((keys >> np.uint32(check_bit)) != (keys[0] >> np.uint32(check_bit))))


@pytest.mark.parametrize("n", [1, 255, 2048, 2049, 100003, 3_000_000])
def test_scan(dev, n):
    from gaussian_splatting_amd import _lib as L
    from gaussian_splatting_amd.render import _ptr, _stream
    lib = L.lib()
    rng = np.random.default_rng(n)
    a = rng.integers(0, 2000, n, dtype=np.int64).astype(np.uint32)
    t = torch.from_numpy(a.view(np.int32)).to(dev); out = torch.empty_like(t)
    tot = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = lib.gsx_scan_workspace_bytes(n); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L.check(lib.gsx_scan_u32(_ptr(t), _ptr(out), _ptr(tot), n, _ptr(ws), wsb, _stream()), "scan")
    torch.cuda.synchronize()
    ref = np.concatenate([[0], np.cumsum(a.astype(np.uint64))[:-1]]).astype(np.uint32)
    assert np.array_equal(_u32(out), ref)
    assert int(tot.item()) == int(a.astype(np.uint64).sum())


# ---------------------------------------------------------------------------------------------
# golden fixtures of the reference: stage-wise bit-exactness vs the oracle, image/grad tolerance
# vs the reference's own outputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names())
def test_golden_forward_backward(dev, O, name):
    fx = load_golden(name)
    H, W = fx["H"], fx["W"]
    order = fx["order"].astype(np.uint32) if fx["has_ties"] else None
    gp = gpu_pipeline(fx["params"], fx["poses"][0], H, W, dev, order=order)
    fw = O.forward(fx["params"], fx["poses"][0], H, W, "tiled", order=order)
    # projection: bit-exact vs oracle AND vs the reference
    assert np.array_equal(gp["mean2d"], fx["out_pos2d"][0])
    assert np.array_equal(gp["depth"], fx["out_depths"][0])
    assert np.array_equal(gp["cov2d"], fw["cov2d"])
    assert np.array_equal(gp["keys"], O.depth_keys(gp["depth"]))
    if order is None:
        assert np.array_equal(gp["order"].astype(np.int64), fx["order"])
    assert np.array_equal(gp["rec"].view(np.uint32), fw["rec"].view(np.uint32)), "records (conic, colour, box) bit-exact"
    assert_integer_products_equal(gp, fw)
    assert gp["n_composited"] == int(fx["n_composited"]) == fw["n_composited"]
    assert rgb_err(gp["rgb"], fx) < RGB_TOL
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
    assert np.abs(gp["alpha"] - fw["alpha"]).max() < RGB_TOL
    assert np.abs(gp["depth_map"] - fw["depth_map"]).max() < RGB_TOL * max(1.0, np.abs(fw["depth_map"]).max())
    g = gpu_backward(gp, H, W, fx["d_rgb"][0], fx["d_pos2d"][0], fx["d_depths"][0], fx["d_cov2d"][0])
    go = O.backward(fx["params"], fx["poses"][0], H, W, fw, fx["d_rgb"][0], fx["d_pos2d"][0], fx["d_depths"][0],
                    fx["d_cov2d"][0], "tiled")
    for k in PARAMS:
        assert rel_err(g[k], fx["g_" + k]) < grad_tol(fx, k), f"{k} vs reference"
        # vs the oracle (float64 accumulation: within 4e-5 of the float64 yardstick on every fixture) the float32 kernels are
        # allowed the noise the reference's own float32 run shows on the same scene (non-zero for exit8 only, see grad_tol)
        assert rel_err(g[k], go[k]) < GRAD_RTOL + float(fx.get("ref_noise_" + k, 0.0)), f"{k} vs oracle"
        check_elementwise(name, k, g[k], fx["g_" + k], go[k], fx)


# ---------------------------------------------------------------------------------------------
# down-scaled instances of each BASELINE config's distribution, vs the oracle
# ---------------------------------------------------------------------------------------------
CASES = {
    "c1_default_3000_200x150": lambda: (default_scene(3000, 1), pose_tz(3.0), 150, 200),
    "c2_20000_256x256": lambda: (synth_scene(20000, 2, 0.002, 0.01), pose_tz(2.0), 256, 256),
    "c3_50000_320x320_orbit": lambda: (synth_scene(50000, 5, 0.001, 0.005), orbit_pose(15), 320, 320),
    "c5_6000_480x270_ties": lambda: (synth_scene(6000, 8, 0.01, 0.1, zq=64.0), pose_tz(2.0), 270, 480),
    "opacity_out_of_range": lambda: (synth_scene(3000, 9, 0.005, 0.03, o_lo=-0.5, o_hi=1.5), pose_tz(2.0), 96, 128),
    # 66 x 65 = 4290 tiles: beyond the one-pass 12-bit tile sort: the 13-bit one-pass form (512-thread workgroups, round 3)
    "more_than_4096_tiles_1056x1040": lambda: (synth_scene(4000, 12, 0.004, 0.03), pose_tz(2.0), 1040, 1056),
    # 129 x 65 = 8385 tiles: beyond that one too, i.e. the generic two-pass radix path + ranges kernel + chunk-table scan
    "more_than_8192_tiles_2064x1040": lambda: (synth_scene(4000, 13, 0.004, 0.03), pose_tz(2.0), 1040, 2064),
    # every pixel saturates after a few screen-sized opaque splats: the reference's global early exit fires (n_composited < N)
    # on 4096 tiles and > 2048 chunks, i.e. the exit search and the second compositing pass walk their small grids with a stride
    "global_exit_4096_tiles": lambda: (synth_scene(400, 21, 1.0, 2.0, o_lo=0.9, o_hi=1.0), pose_tz(2.0), 1024, 1024),
    "empty_screen": lambda: (synth_scene(500, 10, 0.001, 0.002), _shifted_pose(50.0), 64, 64),   # D == 0
}


def _shifted_pose(tx):
    m = np.eye(4, dtype=np.float32); m[0, 3] = tx; m[2, 3] = 2.0
    return m


@pytest.mark.parametrize("case", list(CASES))
def test_synthetic_vs_oracle(dev, O, case):
    params, pose, H, W = CASES[case]()
    gp = gpu_pipeline(params, pose, H, W, dev)
    fw = O.forward(params, pose, H, W, "tiled")
    assert np.array_equal(gp["mean2d"], fw["mean2d"], equal_nan=True)
    assert np.array_equal(gp["depth"], fw["depth"], equal_nan=True)
    assert np.array_equal(gp["cov2d"], fw["cov2d"], equal_nan=True)
    assert_integer_products_equal(gp, fw)
    assert gp["n_composited"] == fw["n_composited"]
    if case.startswith("global_exit"):
        assert fw["n_composited"] < params["positions"].shape[0] and fw["D"] > 2048 * 256
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
    assert np.abs(gp["alpha"] - fw["alpha"]).max() < RGB_TOL
    rng = np.random.default_rng(7)
    d_rgb = rng.standard_normal((3, H, W)).astype(np.float32)
    n = params["positions"].shape[0]
    d_m = rng.standard_normal((n, 2)).astype(np.float32); d_d = rng.standard_normal(n).astype(np.float32)
    d_c = rng.standard_normal((n, 2, 2)).astype(np.float32)
    tol = GRAD_RTOL      # also for opacity outside [0,1] (alpha crossing 1): the back-to-front backward has no cancellation
    g = gpu_backward(gp, H, W, d_rgb, d_m, d_d, d_c)
    go = O.backward(params, pose, H, W, fw, d_rgb, d_m, d_d, d_c, "tiled")
    for k in PARAMS:
        assert rel_err(g[k], go[k]) < tol, k
    # rgb-only gradients (no upstream on the projected outputs): the compositing backward alone
    g = gpu_backward(gp, H, W, d_rgb)
    go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
    for k in PARAMS:
        assert rel_err(g[k], go[k]) < tol, k + " (rgb only)"


# The exit search's ROUNDS at scale (end of round 3): faint splats with opacities on both sides of zero, so that A goes down as well as
# up (non-monotone: the fixpoint search, not the one-pass rule), every pixel saturates only after hundreds to thousands of splats and
# the exit rank lies deep in the lists -- several batches of 256 per tile (checkpoints between rounds, the below-k loop), 512
# workgroups with two or more tiles each, 256 signal-word copies, and, in the two scenes of mixed scales, the chunk pass's quadrant
# masks.  The reference-run exit fixtures (tests/golden exit*, exitop*) are 16-40 pixels a side: one or two workgroups.
EXIT_ROUND_CASES = {
    "screen_sized_4000_512x512": (4000, 33, 0.3, 0.8, -0.05, 0.15, 512),
    "mixed_scales_8000_512x512": (8000, 34, 0.05, 0.6, -0.05, 0.2, 512),
    "mixed_scales_12000_384x384": (12000, 35, 0.03, 0.4, -0.1, 0.4, 384),
}


@pytest.mark.parametrize("case", list(EXIT_ROUND_CASES))
def test_global_exit_rounds_vs_oracle(dev, O, case):
    n, seed, s_lo, s_hi, o_lo, o_hi, hw = EXIT_ROUND_CASES[case]
    params = synth_scene(n, seed, s_lo, s_hi, o_lo=o_lo, o_hi=o_hi)
    pose = pose_tz(2.0)
    gp = gpu_pipeline(params, pose, hw, hw, dev)
    fw = O.forward(params, pose, hw, hw, "tiled")
    assert (params["opacity"] < 0).any()                                  # a splat that takes A down: the search runs in rounds
    assert 256 < fw["n_composited"] < n and fw["D"] > (1 << 20)          # the exit fires, deep in the lists; 256-instance chunks
    assert_integer_products_equal(gp, fw)
    assert gp["n_composited"] == fw["n_composited"]
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
    assert np.abs(gp["alpha"] - fw["alpha"]).max() < RGB_TOL


@pytest.mark.parametrize("chunk", [64, 128, 256])
def test_every_chunk_size_vs_oracle(dev, O, chunk):
    """The binning stage cuts small views into shorter compositing chunks (64 / 128 / 256 instances by instance count);
    each size against the oracle on the same scene, forward and backward (the automatic choice is covered by every
    other test: small scenes get 64, the full-size test 256)."""
    from gaussian_splatting_amd import render as R
    params, pose, H, W = CASES["c2_20000_256x256"]()
    R._scratch.force_chunk = chunk
    try:
        gp = gpu_pipeline(params, pose, H, W, dev)
        fw = O.forward(params, pose, H, W, "tiled")
        assert_integer_products_equal(gp, fw)
        assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
        d_rgb = np.random.default_rng(5).standard_normal((3, H, W)).astype(np.float32)
        g = gpu_backward(gp, H, W, d_rgb)
        go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
        for k in PARAMS:
            assert rel_err(g[k], go[k]) < GRAD_RTOL, k
    finally:
        R._scratch.force_chunk = 0


def test_module_dropin_matches_reference_outputs(dev):
    """GaussianSplat(nn.Module): same state_dict keys, same output dict, autograd through all
    three gradient routes (rendered_images, scales, opacity: SURVEY fact 8)."""
    from gaussian_splatting_amd import GaussianSplat
    fx = load_golden("n300_32x24")
    m = GaussianSplat(300, chunk_size=10)
    assert list(m.state_dict().keys()) == PARAMS
    m.load_state_dict({k: torch.from_numpy(fx[k]) for k in PARAMS})
    m = m.to(dev)
    out = m(torch.from_numpy(fx["poses"]).to(dev), (fx["H"], fx["W"]))
    for k in ["rendered_images", "positions", "depths", "scales", "features", "opacity"]:
        assert k in out
    assert out["features"] is m.features and out["opacity"] is m.opacity
    assert out["rendered_images"].shape == (1, 3, fx["H"], fx["W"])
    assert out["scales"].shape == (1, 300, 2, 2) and out["positions"].shape == (1, 300, 2) and out["depths"].shape == (1, 300)
    t = lambda a: torch.from_numpy(a).to(dev)
    L = (out["rendered_images"] * t(fx["d_rgb"])).sum() + (out["positions"] * t(fx["d_pos2d"])).sum() \
        + (out["depths"] * t(fx["d_depths"])).sum() + (out["scales"] * t(fx["d_cov2d"])).sum()
    L.backward()
    assert m.last_n_composited() == [int(fx["n_composited"])]
    assert rgb_err(out["rendered_images"][0].detach().cpu().numpy(), fx) < RGB_TOL
    for k in PARAMS:
        assert rel_err(getattr(m, k).grad.cpu().numpy(), fx["g_" + k]) < grad_tol(fx, k), k
    # the depth map is an extra (the reference returns none): off by default, same image with it on
    assert "depth_map" not in out
    m.render_depth_map = True
    out2 = m(torch.from_numpy(fx["poses"]).to(dev), (fx["H"], fx["W"]))
    assert out2["depth_map"].shape == (1, 1, fx["H"], fx["W"])
    assert torch.equal(out2["rendered_images"], out["rendered_images"])


@pytest.mark.parametrize("cap", [1, 777, 5000, 1 << 20])
def test_async_binning_matches_synchronous(dev, cap):
    """gsx_bin_count + gsx_bin_emit_sort (instance count read on the device, launches sized by capacity) against
    gsx_bin_instances (host read-back): bit-equal products when the capacity suffices, truncated but in-bounds
    results plus the true count when it does not; the render path then grows its buffers and redoes the view."""
    from gaussian_splatting_amd import render as R
    params = synth_scene(3000, 21, 0.004, 0.03)
    H, W = 96, 80
    gp = gpu_pipeline(params, pose_tz(2.0), H, W, dev)
    t = gp["_t"]; n = 3000; D = gp["D"]
    assert D > 5000
    offsets, total, (pinned, ev) = R.bin_count(t["tt"])
    tiles, ranks, ranges, chunk_base = R.bin_emit_sort(t["rec"], offsets, total, n, H, W, cap)
    torch.cuda.synchronize()
    assert int(pinned[0]) == D and int(total[0].item()) == D
    assert np.array_equal(_u32(offsets)[:n], gp["offsets"])
    if cap >= D:
        assert np.array_equal(_u32(tiles)[:D], gp["tiles"]) and np.array_equal(_u32(ranks)[:D], gp["ranks"])
        assert np.array_equal(_u32(ranges), gp["ranges"])
        assert np.array_equal(_u32(chunk_base), _u32(t["chunk_base"]))
    else:       # truncated: the first `cap` emitted instances, sorted; ranges partition [0, cap)
        r = _u32(ranges)
        assert r[0, 0] == 0 and r[-1, 1] == cap and np.all(r[1:, 0] == r[:-1, 1])
        tk = _u32(tiles)[:cap]
        assert np.all(tk[:-1] <= tk[1:])
    # the render path recovers from a capacity that is too small
    R._scratch.capacity[dev] = cap
    outs, vs, _ = R.forward_view(t["P"], t["pose"], H, W)
    torch.cuda.synchronize()
    assert vs.D == D and R._scratch.cap(dev, n) >= D
    assert np.array_equal(outs[0].cpu().numpy(), gp["rgb"])
    R._scratch.capacity.pop(dev, None)


def test_tile_sort_beyond_one_workgroup_per_cu(dev, O):
    """More than 256 x 16384 instances on <= 4096 tiles: the one-pass tile sort then runs several workgroups per CU at its
    largest keys-per-workgroup setting (config 4's regime).  Integer products against the oracle, image at tolerance."""
    params = synth_scene(300000, 41, 0.01, 0.04)
    H = W = 1024
    pose = pose_tz(2.0)
    gp = gpu_pipeline(params, pose, H, W, dev)
    assert gp["D"] > 256 * 16384, gp["D"]
    fw = O.forward(params, pose, H, W, "tiled")
    assert_integer_products_equal(gp, fw)
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL


def test_optimistic_depth_sort_recovers(dev, O):
    """The render path sorts the depth keys on 24 bits while their top byte is constant (three radix passes instead of
    four) and learns from the sort's own mismatch count, which travels with the instance count, whether that held.
    Depths in [1.5, 2.5] straddle a binade whose keys differ in the top byte: a view sorted on 24 bits must be redone
    on 32 and come out in the oracle's order; depths in [2.5, 3.5] keep the shortcut, with bit-identical results."""
    from gaussian_splatting_amd import render as R
    params = synth_scene(5000, 31, 0.004, 0.02)
    P = tuple(torch.from_numpy(np.ascontiguousarray(params[k], np.float32)).to(dev) for k in PARAMS)
    H, W = 64, 80
    for tz, constant in ((1.0, False), (2.0, True)):
        pose = pose_tz(tz); pose_t = torch.from_numpy(pose).to(dev)
        fw = O.forward(params, pose, H, W, "tiled")
        imgs = []
        for hint in (24, 32):
            R._scratch.depth_bits[dev] = hint
            outs, vs, _ = R.forward_view(P, pose_t, H, W)
            torch.cuda.synchronize()
            assert np.array_equal(_u32(vs.order), fw["order"]), (tz, hint)
            assert R._scratch.depth_bits[dev] == (24 if constant else 32), (tz, hint)
            imgs.append(outs[0].cpu().numpy())
        assert np.array_equal(imgs[0], imgs[1])
        assert np.abs(imgs[0] - fw["rgb"]).max() < RGB_TOL
    R._scratch.depth_bits.pop(dev, None)


def test_occlusion_cut_backward(dev, O):
    """Tiles buried under thousands of near-opaque splats: most of their chunks start with T < 2^-24 and
    the backward writes their rows as zeros (occlusion cut, gsx.h).  Gradients must still match the oracle, which
    evaluates every pair, at the usual tolerance; the image is untouched by the cut."""
    rng = np.random.default_rng(77)
    n, H, W = 6000, 64, 64                         # no global exit (min alpha 0.96) but the 4 central tiles are buried
    params = synth_scene(n, 78, 0.05, 0.2)
    params["opacity"] = rng.uniform(0.6, 1.0, (n, 1)).astype(np.float32)
    pose = pose_tz(2.0)
    gp = gpu_pipeline(params, pose, H, W, dev)
    fw = O.forward(params, pose, H, W, "tiled")
    assert_integer_products_equal(gp, fw)
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
    # the cut must actually fire here: quadrants of chunks that the combine pass marked dark for the backward
    dark, _, _ = _chunk_tail(gp["_t"])
    assert (dark & 1).mean() > 0.1, (dark & 1).mean()
    d_rgb = rng.standard_normal((3, H, W)).astype(np.float32)
    g = gpu_backward(gp, H, W, d_rgb)
    go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
    for k in PARAMS:
        assert rel_err(g[k], go[k]) < GRAD_RTOL, k
    # opacities in (1, 2] keep |1 - alpha| <= 1; negative ones let |T| grow inside a chunk, which raises the chunk's
    # darkness threshold by the growth bound (gsx_constants.h): still within tolerance
    params["opacity"][::7] = 1.5
    params["opacity"][3::11] = -0.05
    gp = gpu_pipeline(params, pose, H, W, dev)
    fw = O.forward(params, pose, H, W, "tiled")
    g = gpu_backward(gp, H, W, d_rgb)
    go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
    for k in PARAMS:
        assert rel_err(g[k], go[k]) < GRAD_RTOL, k + " (opacity > 1)"


def _chunk_tail(t):
    """(dark bytes [chunks, 4], headers [chunks], first payload word [chunks]) from the tail of a view's chunk-state buffer
    (layout: gsx_constants.h)"""
    mc = t["max_chunks"]
    nch = int(_u32(t["chunk_base"])[-2])
    tail = t["chunk_state"][mc * 8 * 256:].view(torch.int32)
    h0 = 2 * mc + 16 + 20 * mc
    dark = tail[mc:mc + nch].cpu().numpy().view(np.uint8).reshape(nch, 4)
    hdr = tail[h0:h0 + nch].cpu().numpy().view(np.uint32)
    pay0 = tail[h0 + 2 * mc:h0 + 2 * mc + 64 * nch].view(-1, 64)[:, 0].cpu().numpy().view(np.uint32)
    return dark, hdr, pay0


@pytest.mark.parametrize("growers", [False, "sparse", "dense"])
def test_forward_occlusion_skipping_deep_view(dev, O, growers):
    """Forward occlusion skipping (gsx_composite.hip): a view whose tiles are many rounds of the machine deep (> 20480 chunks)
    and buried under near-opaque splats.  The chunk pass must actually skip chunks, and none of it may show: image and
    gradients against the oracle (which evaluates every pair) at the usual bars, bitwise equal between two runs (what is
    skipped depends on timing, what the combine pass ignores does not), module (one-call path) == staged path.  With
    `growers`, opacities below 0 and above 1 are mixed in: |T| can grow inside a chunk, the rule carries the bound --
    "sparse": one splat in a thousand slightly below 0, what a few trainer steps do to a scene (skipping must go on);
    "dense": one in eleven at -0.05 (a tile's growth bound is then hundreds of bits: nothing is ever dark -- a limit)."""
    from gaussian_splatting_amd import render as R, GaussianSplat
    rng = np.random.default_rng(5)
    n, H, W = 100000, 192, 192                       # the centre of the screen is buried, its border stays empty: no global exit
    params = synth_scene(n, 79, 0.02, 0.08)
    params["opacity"] = rng.uniform(0.5, 1.0, (n, 1)).astype(np.float32)
    if growers == "dense":
        params["opacity"][::7] = 1.5
        params["opacity"][3::11] = -0.05
    elif growers == "sparse":
        params["opacity"][::7] = 1.5
        params["opacity"][3::1000] = -0.005
    pose = pose_tz(2.0)
    d_rgb = rng.standard_normal((3, H, W)).astype(np.float32)
    R._scratch.force_chunk = 64                      # 64-instance chunks: 1.5 M instances are > 20480 chunks
    try:
        gp = gpu_pipeline(params, pose, H, W, dev)
        dark, hdr, pay0 = _chunk_tail(gp["_t"])
        nch = hdr.shape[0]
        assert nch > 20480, nch
        assert gp["n_composited"] == n                   # (a global early exit would recompute every chunk in the second pass)
        self_skipped = pay0 == 0                         # a chunk that computes publishes its payload; one that skips itself does not
        # the header pass: the growth of a tile in the word of its first chunk, every other word 0; no growers, no growth
        firsts = np.unique(_u32(gp["_t"]["chunk_base"])[:-2])
        firsts = firsts[firsts < nch]
        rest = np.ones(nch, bool); rest[firsts] = False
        growth = hdr & 0xBFFFFFFF                        # (bit 30: the chunk has published its payload)
        assert not growth[rest].any()
        assert (growth[firsts] != 0).mean() > 0.9 if growers else not growth.any()
        assert np.array_equal((hdr & 0x40000000) != 0, pay0 != 0)
        mc_ = gp["_t"]["max_chunks"]
        why = gp["_t"]["chunk_state"][mc_ * 8 * 256:].view(torch.int32)[:nch].cpu().numpy().view(np.uint32) & 0xff
        # the chunk pass skipped, with a few growers too (the rule is monotone: a tile's growth only raises its threshold)
        if growers != "dense":
            assert self_skipped.mean() > 0.3, (self_skipped.mean(), dict(zip(*np.unique(why, return_counts=True))))
        assert (dark == 0x81).all(1).mean() >= self_skipped.mean()            # ... only what the rule calls dark
        assert np.all((dark[self_skipped] == 0x81))
        g = gpu_backward(gp, H, W, d_rgb)
        gp2 = gpu_pipeline(params, pose, H, W, dev)
        g2 = gpu_backward(gp2, H, W, d_rgb)
        assert np.array_equal(gp["rgb"], gp2["rgb"]) and np.array_equal(gp["alpha"], gp2["alpha"]) and np.array_equal(gp["depth_map"], gp2["depth_map"])
        for k in PARAMS:
            assert np.array_equal(g[k], g2[k]), k
        m = GaussianSplat(n)
        m.load_state_dict({k: torch.from_numpy(params[k]) for k in PARAMS})
        m = m.to(dev)
        out = m(torch.from_numpy(pose).to(dev)[None], (H, W))
        assert np.array_equal(out["rendered_images"][0].detach().cpu().numpy(), gp["rgb"])
        (out["rendered_images"][0] * torch.from_numpy(d_rgb).to(dev)).sum().backward()
        for k in PARAMS:
            assert np.array_equal(getattr(m, k).grad.cpu().numpy(), g[k]), k
    finally:
        R._scratch.force_chunk = 0
    fw = O.forward(params, pose, H, W, "tiled")
    assert_integer_products_equal(gp, fw)
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
    assert np.abs(gp["alpha"] - fw["alpha"]).max() < RGB_TOL
    assert np.abs(gp["depth_map"] - fw["depth_map"]).max() < RGB_TOL * max(1.0, np.abs(fw["depth_map"]).max())
    go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
    for k in PARAMS:
        assert rel_err(g[k], go[k]) < GRAD_RTOL, k


def test_backward_rows_zero_filled_up_front_equal_rows_zeroed_by_chunks(dev):
    """gsx_composite_bwd with the view's instance count (>= GSX_PREZERO_ROWS = 16 Mi: one fill, occluded chunks write
    nothing) against the same call without it (every chunk writes its rows): the five gradients bit for bit.  A view of
    screen-sized splats: every splat in every tile, 24 M instances, most chunks occluded."""
    from gaussian_splatting_amd import render as R
    rng = np.random.default_rng(11)
    n, H, W = 60000, 320, 320
    params = synth_scene(n, 83, 0.5, 1.0, o_lo=0.3, o_hi=1.0)
    d_rgb = rng.standard_normal((3, H, W)).astype(np.float32)
    gp = gpu_pipeline(params, pose_tz(2.0), H, W, dev)
    assert gp["_t"]["D"] >= 16 << 20, gp["_t"]["D"]
    dark, _, _ = _chunk_tail(gp["_t"])
    assert ((dark & 1) != 0).all(1).mean() > 0.3                  # occluded chunks: the ones whose rows are at stake
    assert R.PREZERO_ROWS
    g_fill = gpu_backward(gp, H, W, d_rgb)
    R.PREZERO_ROWS = False
    try:
        g_own = gpu_backward(gp, H, W, d_rgb)
    finally:
        R.PREZERO_ROWS = True
    for k in PARAMS:
        assert np.array_equal(g_fill[k], g_own[k]), k
        assert np.isfinite(g_fill[k]).all() and np.abs(g_fill[k]).max() > 0, k


def test_orbit_frames_equal_per_view_forward(dev, O):
    """render_frames(V poses) == V module forwards (bitwise), and == the oracle within 1e-4."""
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.scenes import create_orbit_poses
    n, H, W = 1500, 48, 40
    params = synth_scene(n, 31, 0.01, 0.06)
    m = GaussianSplat(n)
    m.load_state_dict({k: torch.from_numpy(params[k]) for k in PARAMS})
    m = m.to(dev).eval()
    poses_np = create_orbit_poses(4.0, 6, 45.0)
    poses = torch.from_numpy(poses_np).to(dev)
    frames = m.render_frames(poses, (H, W))
    assert frames.shape == (6, 3, H, W)
    with torch.no_grad():
        for v in range(6):
            one = m(poses[v].unsqueeze(0), (H, W))["rendered_images"][0]
            assert torch.equal(frames[v], one), v
    fw = O.forward(params, poses_np[2], H, W, "tiled")
    assert np.abs(frames[2].cpu().numpy() - fw["rgb"]).max() < RGB_TOL


def test_orbit_frames_into_a_sink(dev, tmp_path):
    """SURVEY 8f N1 with its sink: frames rendered in blocks and written as PNGs through the GPU conversion, the side-stream
    copy and the worker thread are the bytes the reference's conversion gives for the frames of one render_frames call
    (utils/visualization.py:254: (frame * 255).astype(uint8); values clamped to [0, 1] first)."""
    from PIL import Image
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd.frames import FrameSink
    from gaussian_splatting_amd.scenes import create_orbit_poses
    n, H, W = 1500, 48, 40
    params = synth_scene(n, 31, 0.01, 0.06)
    m = GaussianSplat(n)
    m.load_state_dict({k: torch.from_numpy(params[k]) for k in PARAMS})
    m = m.to(dev).eval()
    poses = torch.from_numpy(create_orbit_poses(4.0, 7, 45.0)).to(dev)
    frames = m.render_frames(poses, (H, W))
    with FrameSink(tmp_path / "orbit") as sink:
        assert m.render_frames(poses, (H, W), sink=sink, block=3) == 7
    want = (frames.clamp(0.0, 1.0).cpu().numpy() * 255).astype(np.uint8).transpose(0, 2, 3, 1)
    for v in range(7):
        assert np.array_equal(np.asarray(Image.open(tmp_path / "orbit" / f"frame_{v:04d}.png")), want[v]), v
    with FrameSink(tmp_path / "orbit.gif", fps=12) as sink:
        m.render_frames(poses, (H, W), sink=sink)
    assert Image.open(tmp_path / "orbit.gif").n_frames == 7


def test_cpu_tensors_fail_loudly():
    from gaussian_splatting_amd import GaussianSplat
    m = GaussianSplat(8)
    with pytest.raises(RuntimeError):
        m(torch.eye(4)[None], (8, 8))


def test_host_poses_and_mixed_devices(dev):
    """render_views called directly: poses on the host are moved (a host pointer must never reach a kernel), parameters on
    different devices are refused"""
    from gaussian_splatting_amd import GaussianSplat, render_views
    m = GaussianSplat(50).to(dev)
    pose_host = torch.from_numpy(pose_tz(3.0))[None]
    a = render_views(m.positions, m.scales, m.rotations, m.opacity, m.features, pose_host, (24, 40))[0]
    b = render_views(m.positions, m.scales, m.rotations, m.opacity, m.features, pose_host.to(dev), (24, 40))[0]
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        render_views(m.positions, m.scales.detach().cpu(), m.rotations, m.opacity, m.features, pose_host, (24, 40))


def test_full_size_properties(dev):
    """BASELINE config 3/bench scene (1M Gaussians, 1024^2): size-independent properties."""
    params = synth_scene(1_000_000, 5, 0.001, 0.005)
    H = W = 1024
    gp = gpu_pipeline(params, pose_tz(2.0), H, W, dev)
    n = 1_000_000
    assert np.array_equal(np.sort(gp["order"]), np.arange(n, dtype=np.uint32))           # a permutation
    d = gp["depth"][gp["order"]]
    assert np.all(d[:-1] >= d[1:])                                                        # descending depth
    keys = (gp["tiles"].astype(np.uint64) << np.uint64(32)) | gp["ranks"].astype(np.uint64)
    assert np.all(keys[:-1] < keys[1:])                                                   # strictly sorted, unique
    r = gp["ranges"]
    assert r[0, 0] == 0 and r[-1, 1] == gp["D"] and np.all(r[1:, 0] == r[:-1, 1])         # ranges partition [0, D)
    assert gp["D"] == int(gp["tiles_touched"].astype(np.uint64).sum())
    t_of = np.repeat(np.arange(r.shape[0], dtype=np.uint32), (r[:, 1] - r[:, 0]).astype(np.int64))
    assert np.array_equal(t_of, gp["tiles"])
    assert np.isfinite(gp["rgb"]).all() and gp["alpha"].min() >= 0 and gp["alpha"].max() <= 1 + 1e-5
    # determinism: the path has no float atomics, two runs are bitwise equal (image and gradients)
    rng = np.random.default_rng(3)
    d_rgb = rng.standard_normal((3, H, W)).astype(np.float32)
    g1 = gpu_backward(gp, H, W, d_rgb)
    gp2 = gpu_pipeline(params, pose_tz(2.0), H, W, dev)
    g2 = gpu_backward(gp2, H, W, d_rgb)
    assert np.array_equal(gp["rgb"], gp2["rgb"])
    for k in PARAMS:
        assert np.array_equal(g1[k], g2[k]), k
        assert np.isfinite(g1[k]).all()
    # linearity of the backward in the upstream gradient
    g3 = gpu_backward(gp, H, W, 2.0 * d_rgb)
    for k in PARAMS:
        assert rel_err(g3[k], 2.0 * g1[k]) < 1e-5
    # the module (asynchronous binning, launches sized by capacity, autograd) gives the same bits as the staged pipeline
    from gaussian_splatting_amd import GaussianSplat
    m = GaussianSplat(n)
    m.load_state_dict({k: torch.from_numpy(params[k]) for k in PARAMS})
    m = m.to(dev)
    out = m(torch.from_numpy(pose_tz(2.0)).to(dev)[None], (H, W))
    assert np.array_equal(out["rendered_images"][0].detach().cpu().numpy(), gp["rgb"])
    (out["rendered_images"][0] * torch.from_numpy(d_rgb).to(dev)).sum().backward()
    for k in PARAMS:
        assert np.array_equal(getattr(m, k).grad.cpu().numpy(), g1[k]), k


FULL_SIZE = {
    # config 3 = the bench workload: 1 M Gaussians on 1024^2, orbit view 10 (2.6 M instances, 256-instance chunks, half occluded)
    "c3_orbit_view10": lambda: (synth_scene(1_000_000, 5, 0.001, 0.005), orbit_pose(10), 1024, 1024),
    # config 2 at its own size and camera: 100 k Gaussians on 800^2, M = I with tz = 2
    "c2_100k_800x800": lambda: (synth_scene(100_000, 2, 0.002, 0.01), pose_tz(2.0), 800, 800),
    # config 4's per-GPU share: 3 M Gaussians on 1024^2, orbit view 7 (one of its eight views)
    "c4_3M_orbit_view7": lambda: (synth_scene(3_000_000, 6, 0.001, 0.005), orbit_pose(7), 1024, 1024),
    # config 5's image (1920x1080: 8160 tiles, the 13-bit one-pass tile sort), splat sizes and quantised depths at 1/1000 of
    # its Gaussian count (the oracle's time goes with the instance count: millions here, billions at 6 M)
    "c5_6k_1920x1080": lambda: (synth_scene(6_000, 8, 0.01, 0.1, zq=64.0), pose_tz(2.0), 1080, 1920),
}


@pytest.mark.parametrize("case", list(FULL_SIZE))
def test_full_size_vs_oracle(dev, O, case):
    """BASELINE configs at FULL size directly against the CPU restatement (a few seconds of 16 threads): integer products
    bit-exact, RGB 1e-4 max-abs, gradients 1e-3 relative."""
    params, pose, H, W = FULL_SIZE[case]()
    gp = gpu_pipeline(params, pose, H, W, dev)
    fw = O.forward(params, pose, H, W, "tiled")
    assert_integer_products_equal(gp, fw)
    assert gp["n_composited"] == fw["n_composited"]
    assert np.abs(gp["rgb"] - fw["rgb"]).max() < RGB_TOL
    d_rgb = np.random.default_rng(9).standard_normal((3, H, W)).astype(np.float32)
    g = gpu_backward(gp, H, W, d_rgb)
    go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
    for k in PARAMS:
        assert rel_err(g[k], go[k]) < GRAD_RTOL, k
        check_elementwise("full_size_" + case, k, g[k], go[k])


def test_config5_full_size_properties(dev):
    """BASELINE config 5 AT ITS OWN SIZE: 6 M Gaussians with splats of up to ~100 px and depths quantised to 2^-6 (ties) on
    1920x1080 = 8160 tiles (the 13-bit one-pass tile sort), orbit view 1: ~39 M (tile, Gaussian) instances.  The oracle
    would need minutes here, so the bar is the size-independent properties (as test_full_size_properties for config 3):
    a permutation in descending depth with ties by index, strictly sorted unique 64-bit keys, ranges that partition [0, D),
    instances inside their Gaussian's box, a finite image, bitwise determinism of image and gradients, and the module
    (one-call path, asynchronous count, capacity-sized launches) equal to the staged path bit for bit."""
    from gaussian_splatting_amd.scenes import config_scene
    from gaussian_splatting_amd import GaussianSplat
    params, H, W = config_scene("c5")
    n = params["positions"].shape[0]
    assert (n, H, W) == (6_000_000, 1080, 1920)
    pose = orbit_pose(1)
    gp = gpu_pipeline(params, pose, H, W, dev)
    D = gp["D"]
    assert 30_000_000 < D < 50_000_000, D                      # 38.7 M at the 2^-20 cut (44.7 M at round 1's 2^-24)
    order = gp["order"]
    seen = np.zeros(n, np.uint8); seen[order] = 1
    assert seen.all()                                                                     # a permutation
    d = gp["depth"][order]
    assert np.all(d[:-1] >= d[1:])                                                        # descending depth ...
    tie = d[:-1] == d[1:]
    assert tie.sum() > 1_000_000 and np.all(order[:-1][tie] < order[1:][tie])             # ... ties (2.15 M of them in this view) by index
    keys = (gp["tiles"].astype(np.uint64) << np.uint64(32)) | gp["ranks"].astype(np.uint64)
    assert np.all(keys[:-1] < keys[1:])                                                   # strictly sorted, unique
    del keys
    r = gp["ranges"]
    assert r.shape[0] == 8160 and r[0, 0] == 0 and r[-1, 1] == D and np.all(r[1:, 0] == r[:-1, 1])
    assert D == int(gp["tiles_touched"].astype(np.uint64).sum())
    t_of = np.repeat(np.arange(r.shape[0], dtype=np.uint32), (r[:, 1] - r[:, 0]).astype(np.int64))
    assert np.array_equal(t_of, gp["tiles"])
    # every instance lies inside its Gaussian's footprint box (record words 10, 11: pixel box, rank order)
    box = gp["rec"].view(np.uint32)[:, 10:12][gp["ranks"]]
    tw = (W + 15) // 16
    tx, ty = gp["tiles"] % tw, gp["tiles"] // tw
    assert np.all((tx >= (box[:, 0] & 0xffff) >> 4) & (tx <= (box[:, 0] >> 16) >> 4) & (ty >= (box[:, 1] & 0xffff) >> 4) & (ty <= (box[:, 1] >> 16) >> 4))
    del box, tx, ty, t_of
    assert np.isfinite(gp["rgb"]).all() and gp["alpha"].min() >= 0 and gp["alpha"].max() <= 1 + 1e-5
    d_rgb = np.random.default_rng(3).standard_normal((3, H, W)).astype(np.float32)
    g1 = gpu_backward(gp, H, W, d_rgb)
    rgb1 = gp["rgb"]
    del gp
    torch.cuda.empty_cache()
    gp2 = gpu_pipeline(params, pose, H, W, dev)
    g2 = gpu_backward(gp2, H, W, d_rgb)
    assert np.array_equal(rgb1, gp2["rgb"])
    for k in PARAMS:
        assert np.array_equal(g1[k], g2[k]), k
        assert np.isfinite(g1[k]).all() and np.abs(g1[k]).max() > 0
    del gp2, g2
    torch.cuda.empty_cache()
    m = GaussianSplat(n)
    m.load_state_dict({k: torch.from_numpy(params[k]) for k in PARAMS})
    m = m.to(dev)
    out = m(torch.from_numpy(pose).to(dev)[None], (H, W))
    assert np.array_equal(out["rendered_images"][0].detach().cpu().numpy(), rgb1)
    (out["rendered_images"][0] * torch.from_numpy(d_rgb).to(dev)).sum().backward()
    for k in PARAMS:
        assert np.array_equal(getattr(m, k).grad.cpu().numpy(), g1[k]), k


# ---------------------------------------------------------------------------------------------
# gsx_view_forward: one call over one arena == the stage entry points, bit for bit
# ---------------------------------------------------------------------------------------------
def _both_paths(monkeypatch, fn):
    from gaussian_splatting_amd import render as R
    monkeypatch.setattr(R, "STAGED", True); a = fn()
    monkeypatch.setattr(R, "STAGED", False); b = fn()
    return a, b


@pytest.mark.parametrize("case", ["small_400x300", "c3_like_1024", "wide_1056x1040"])
def test_view_driver_equals_stage_calls(dev, monkeypatch, case):
    from gaussian_splatting_amd import render as R
    if case == "small_400x300":
        params, H, W, pose = default_scene(5000, 0), 300, 400, pose_tz(3.0)
    elif case == "c3_like_1024":
        params, H, W, pose = synth_scene(200000, 5, 0.001, 0.005), 1024, 1024, orbit_pose(10)
    else:                                                    # > 4096 tiles: generic tile sort, no packed tile boxes
        params, H, W, pose = synth_scene(20000, 9, 0.004, 0.03), 1040, 1056 * 4, pose_tz(2.0)
    P = tuple(torch.from_numpy(np.ascontiguousarray(params[k], np.float32)).to(dev) for k in PARAMS)
    pose_t = torch.from_numpy(np.ascontiguousarray(pose, np.float32)).to(dev)
    d_rgb = torch.from_numpy(np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)).to(dev)

    def run():
        R.forward_view(P, pose_t, H, W)                      # settles capacity and depth-sort width for this scene
        outs, vs, extra = R.forward_view(P, pose_t, H, W)
        grads = [torch.empty_like(p) for p in P]
        R.backward_view(P, vs, d_rgb, None, None, None, grads, accumulate=False)
        torch.cuda.synchronize()
        D = vs.D
        saved = dict(order=vs.order, rank_of=vs.rank_of, rec=vs.rec, tt=vs.tt, offsets=vs.offsets, tiles=extra["tiles"][:D],
                     ranks=vs.ranks[:D], ranges=vs.ranges, chunk_base=vs.chunk_base, state=vs.state)
        return outs, saved, grads, D, vs

    (o1, s1, g1, D1, vs1), (o2, s2, g2, D2, vs2) = _both_paths(monkeypatch, run)
    assert vs1.arena is None and vs2.arena is not None, "the second run went through gsx_view_forward"
    assert D1 == D2 and D1 > 0
    for a, b, name in zip(o1, o2, ("rgb", "alpha", "depth_map", "mean2d", "depth", "cov2d")):
        assert torch.equal(a, b), name
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    for a, b, name in zip(g1, g2, PARAMS):
        assert torch.equal(a, b), name
        assert float(a.abs().max()) > 0


def test_view_driver_batch_uses_first_views_order(dev, monkeypatch):
    """B = 2 through the module: view 1 is composited in view 0's depth order (gaussian_splatting.py:206) on both paths"""
    from gaussian_splatting_amd import GaussianSplat
    params = synth_scene(3000, 4, 0.004, 0.03)
    poses = torch.from_numpy(np.stack([pose_tz(2.0), pose_tz(2.5)])).to(dev)

    def run():
        m = GaussianSplat(3000)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        m = m.to(dev)
        out = m(poses, (96, 128))
        out["rendered_images"].square().sum().backward()
        torch.cuda.synchronize()
        return out["rendered_images"].detach().clone(), [getattr(m, k).grad.clone() for k in PARAMS]

    (rgb1, g1), (rgb2, g2) = _both_paths(monkeypatch, run)
    assert torch.equal(rgb1, rgb2)
    for a, b, name in zip(g1, g2, PARAMS):
        assert torch.equal(a, b), name


def test_view_driver_rejects_bad_arguments(dev):
    import ctypes as C
    from gaussian_splatting_amd import _lib as L
    lib = L.lib()
    offs = (C.c_int64 * L.VIEW_SLOTS)(); mc = C.c_int64()
    assert lib.gsx_view_layout(0, 64, 64, 1024, 0, 1, offs, C.byref(mc)) == L.GSX_E_INVALID
    assert lib.gsx_view_layout(10, 70000, 64, 1024, 0, 1, offs, C.byref(mc)) == L.GSX_E_RANGE
    nbytes = lib.gsx_view_layout(10, 64, 64, 1024, 0, 1, offs, C.byref(mc))
    o = dict(zip(L.VIEW_SLOT_NAMES, offs))
    assert nbytes > 0 and o["order"] == 0 and all(v % 256 == 0 for v in offs) and mc.value == lib.gsx_max_chunks(1024, 64, 64, 0)
    assert o["partials"] == o["depth_keys"] > o["total"], "the backward's rows share the region of the forward's scratch"
    assert nbytes >= o["partials"] + 36 * 1024 and nbytes > o["bin_ws"]
    assert lib.gsx_view_layout(10, 64, 64, 1024, 0, 1, None, None) == nbytes
    fwd_only = lib.gsx_view_layout(10, 64, 64, 1024, 0, 0, offs, C.byref(mc))
    assert 0 < fwd_only <= nbytes and offs[L.VIEW_SLOT_NAMES.index("partials")] == -1
    arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    scene = synth_scene(10, 3, 0.01, 0.05)
    p, sc, q, op, c = (torch.from_numpy(scene[k]).to(dev) for k in PARAMS)
    pose = torch.from_numpy(pose_tz(2.0)).to(dev)
    m2, d, cv, rgb, al = f(10, 2), f(10), f(10, 2, 2), f(3, 64, 64), f(1, 64, 64)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    args = lambda ab: [p.data_ptr(), sc.data_ptr(), q.data_ptr(), op.data_ptr(), c.data_ptr(), pose.data_ptr(), 10, 64, 64, 1024, 32,
                       None, 1, 0, arena.data_ptr(), ab, m2.data_ptr(), d.data_ptr(), cv.data_ptr(),
                       rgb.data_ptr(), al.data_ptr(), None, state.data_ptr(), None, None, L.stream_ptr()]
    assert lib.gsx_view_forward(*args(nbytes - 1)) == L.GSX_E_CAPACITY          # block too small: nothing is launched
    a = args(nbytes); a[14] = arena.data_ptr() + 4
    assert lib.gsx_view_forward(*a) == L.GSX_E_INVALID                          # misaligned block
    a = args(nbytes); a[10] = 0
    assert lib.gsx_view_forward(*a) == L.GSX_E_INVALID                          # depth_bits out of range
    assert lib.gsx_view_forward(*args(nbytes)) == L.GSX_OK
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# randomised sweep (the pytest-sized cut of tools/fuzz_parity.py): odd image shapes, scale / opacity ranges, poses, chunk
# sizes; stage calls vs oracle, and the module's one-call path vs the stage calls bit for bit
# ---------------------------------------------------------------------------------------------
def test_random_scenes_vs_oracle(dev, O):
    from gaussian_splatting_amd import render as R
    failures = []
    try:
        for seed in range(7000, 7080):
            g = np.random.default_rng(seed)
            n = int(g.choice([1, 2, 7, 64, 300, 1000, 3000]))
            H = int(g.choice([2, 15, 16, 17, 33, 48, 100, 131])); W = int(g.choice([3, 16, 31, 64, 97, 160, 200]))
            lo = float(g.choice([0.001, 0.004, 0.02, 0.1])); hi = lo * float(g.choice([1.5, 5.0, 20.0]))
            o_lo, o_hi = [(0.0, 1.0), (0.9, 1.0), (0.0, 0.05), (0.0, 1e-4)][int(g.integers(0, 4))]
            zq = [None, None, 64.0][int(g.integers(0, 3))]
            # (not the reference's own init here: its world z crosses 0, the Jacobian's 1/z then makes 2x2 covariances so
            # ill-conditioned that rounding-level differences show as 1e-3..1e-2 in the position gradients -- seed 7067 of
            # tools/fuzz_parity.py's generator; that distribution is held to the reference itself by the golden fixtures)
            params = synth_scene(n, seed, lo, hi, zq=zq, o_lo=o_lo, o_hi=o_hi)
            pose = pose_tz(float(g.choice([1.0, 2.0, 3.0, 6.0]))) if g.random() < 0.6 else orbit_pose(int(g.integers(0, 60)))
            R._scratch.force_chunk = int(g.choice([0, 0, 64, 128, 256]))
            gp = gpu_pipeline(params, pose, H, W, dev)
            fw = O.forward(params, pose, H, W, "tiled")
            assert_integer_products_equal(gp, fw)
            assert gp["n_composited"] == fw["n_composited"], seed
            assert np.array_equal(np.isnan(gp["rgb"]), np.isnan(fw["rgb"])), seed
            scale = max(1.0, float(np.nanmax(np.abs(fw["rgb"]), initial=0.0)))
            assert np.nanmax(np.abs(gp["rgb"] - fw["rgb"]), initial=0.0) < RGB_TOL * scale, seed
            t = gp["_t"]
            (rgb_f, alpha_f, _, _, _, _), vs, _ = R.forward_view(t["P"], t["pose"], H, W)
            assert vs.D == gp["D"] and np.array_equal(rgb_f.cpu().numpy(), gp["rgb"], equal_nan=True), seed
            d_rgb = g.standard_normal((3, H, W)).astype(np.float32)
            gr = gpu_backward(gp, H, W, d_rgb)
            grads_f = [torch.empty_like(p) for p in t["P"]]
            R.backward_view(t["P"], vs, torch.from_numpy(d_rgb).to(dev), None, None, None, grads_f, accumulate=False)
            torch.cuda.synchronize()
            go = O.backward(params, pose, H, W, fw, d_rgb, None, None, None, "tiled")
            floor = 1e-30                       # no floor under the yardstick: the tensor's own maximum
            for k, gf in zip(PARAMS, grads_f):
                assert np.array_equal(gf.cpu().numpy(), gr[k], equal_nan=True), (seed, k)
                if np.all(np.isfinite(go[k])):
                    err = float(np.abs(gr[k].astype(np.float64) - go[k]).max()) / max(float(np.abs(go[k]).max()), floor)
                    if not err < GRAD_RTOL:
                        failures.append((seed, k, err))
    finally:
        R._scratch.force_chunk = 0
    assert not failures, failures


def test_empty_and_tiny_models(dev):
    """N = 0 (every tile list empty: a black image, empty per-Gaussian outputs, empty gradients), 1 and 2 through the module,
    forward + backward + render_frames"""
    from gaussian_splatting_amd import GaussianSplat
    pose = torch.from_numpy(pose_tz(3.0))[None].to(dev)
    for n in (0, 1, 2):
        m = GaussianSplat(n).to(dev)
        out = m(pose, (40, 56))
        img = out["rendered_images"]
        assert tuple(img.shape) == (1, 3, 40, 56) and tuple(out["positions"].shape) == (1, n, 2)
        assert tuple(out["depths"].shape) == (1, n) and tuple(out["scales"].shape) == (1, n, 2, 2)
        assert bool(torch.isfinite(img).all())
        if n == 0:
            assert float(img.detach().abs().sum()) == 0.0 and float(out["alpha"].detach().abs().sum()) == 0.0
        (img.sum() + out["positions"].sum() + out["depths"].sum()).backward()
        for k in PARAMS:
            assert tuple(getattr(m, k).grad.shape) == tuple(getattr(m, k).shape), k
        frames = m.render_frames(pose.repeat(3, 1, 1), (40, 56))
        assert tuple(frames.shape) == (3, 3, 40, 56) and torch.equal(frames[0], img[0].detach())
    torch.cuda.synchronize()
