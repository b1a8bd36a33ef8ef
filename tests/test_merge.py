"""merge_nearby_gaussians (/root/reference/model/gaussian_splatting.py:477-528).

Fixtures tests/golden/merge_*.npz were made by running the reference's own method (make_golden.py --merge-only); they
keep its pair list.  Above 25 rows torch.cdist takes a matmul form whose self-distances round above zero for about a
tenth of the points: the reference then merges such a point with itself and removes it (`self_pairs` in the fixture).
The HIP pair search computes distances directly -- a self-distance is exactly 0 -- so it is compared with the
reference's list minus the diagonal; the greedy walk is compared with the reference bit for bit by replaying the
reference's full list, diagonal included.  Parity of the whole method with the reference is therefore exact at
N <= 25 (direct cdist) and "parity unpinned at cdist's rounding" above that.
"""
import ctypes as C
import os
import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["merge_n24", "merge_n300", "merge_n1500"]
KEYS = ("positions", "scales", "rotations", "opacity", "features")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _model_from(fx, dev="cpu"):
    from gaussian_splatting_amd import GaussianSplat
    n = fx["in_positions"].shape[0]
    m = GaussianSplat(n)
    with torch.no_grad():
        for k in KEYS:
            getattr(m, k).data = torch.from_numpy(fx["in_" + k].copy())
    return m.to(dev)


def _greedy_host(n, pi, pj, pos, fea, opa):
    """gsx_merge_greedy_host on numpy copies; returns (pos, fea, opa, removed)."""
    from gaussian_splatting_amd import _lib as L
    counts = np.bincount(pi, minlength=n).astype(np.uint32)
    off = np.zeros(n + 1, dtype=np.uint32); off[1:] = np.cumsum(counts)
    order = np.lexsort((pj, pi))
    partners = np.ascontiguousarray(pj[order].astype(np.uint32))
    pos, fea, opa = pos.copy(), fea.copy(), opa.copy()
    removed = np.empty(n, dtype=np.uint8)
    p = lambda a: C.c_void_p(a.ctypes.data)
    L.check(L.lib().gsx_merge_greedy_host(n, p(off), p(partners), p(pos), p(fea), p(opa), p(removed)), "gsx_merge_greedy_host")
    return pos, fea, opa, removed


@pytest.mark.parametrize("name", CASES)
def test_greedy_walk_replays_the_reference(name):
    """The reference's pair list through the library's host walk + a compaction = the reference's result, bit for bit."""
    fx = _load(name)
    n = fx["in_positions"].shape[0]
    pos, fea, opa, removed = _greedy_host(n, fx["pairs_i"], fx["pairs_j"], fx["in_positions"], fx["in_features"], fx["in_opacity"])
    keep = removed == 0
    assert int(keep.sum()) == int(fx["n_after"])
    np.testing.assert_array_equal(pos[keep], fx["out_positions"])
    np.testing.assert_array_equal(fea[keep], fx["out_features"])
    np.testing.assert_array_equal(opa[keep], fx["out_opacity"])
    np.testing.assert_array_equal(fx["in_scales"][keep], fx["out_scales"])
    np.testing.assert_array_equal(fx["in_rotations"][keep], fx["out_rotations"])


@pytest.mark.parametrize("name", CASES)
def test_cpu_tensors_follow_the_reference(name):
    """CPU tensors go through torch.cdist in row blocks, the reference's own distance form: identical results,
    self-pairs and all."""
    fx = _load(name)
    m = _model_from(fx)
    m.merge_nearby_gaussians(float(fx["threshold"]))
    assert m.num_gaussians == int(fx["n_after"])
    for k in KEYS:
        np.testing.assert_array_equal(getattr(m, k).data.numpy(), fx["out_" + k], err_msg=k)


def test_greedy_walk_rejects_bad_pairs():
    from gaussian_splatting_amd import _lib as L
    off = np.array([0, 0, 1], dtype=np.uint32); partners = np.array([0], dtype=np.uint32)      # pair (1, 0): j < i
    z3 = np.zeros((2, 3), np.float32); o = np.ones((2, 1), np.float32); r = np.zeros(2, np.uint8)
    p = lambda a: C.c_void_p(a.ctypes.data)
    assert L.lib().gsx_merge_greedy_host(2, p(off), p(partners), p(z3), p(z3.copy()), p(o), p(r)) == L.GSX_E_INVALID
    partners[0] = 5                                                                               # out of range
    assert L.lib().gsx_merge_greedy_host(2, p(off), p(partners), p(z3), p(z3.copy()), p(o), p(r)) == L.GSX_E_INVALID


# ------------------------------------------------------------------------------------------ GPU
def _pairs_direct_fp32(pos, thr):
    """Independent CPU pair search: k-d tree candidates (float64, generous radius), then the exact fp32 test the kernel
    documents: 0 < sqrt((dx*dx + dy*dy) + dz*dz) < thr.  Row-major (i, j), i < j."""
    from scipy.spatial import cKDTree
    cand = cKDTree(pos.astype(np.float64)).query_pairs(float(thr) * 1.01 + 1e-6, output_type="ndarray")
    if cand.shape[0] == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    i, j = cand.min(1), cand.max(1)
    e = pos[i] - pos[j]
    d = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2], dtype=np.float32)
    ok = (d > 0) & (d < np.float32(thr))
    i, j = i[ok], j[ok]
    order = np.lexsort((j, i))
    return i[order], j[order]


def _hip_pairs(m, thr):
    off, partners = m._merge_pairs_hip(float(thr))
    off = off.numpy().astype(np.int64)
    i = np.repeat(np.arange(m.num_gaussians), np.diff(off))
    return i, partners.numpy().astype(np.int64)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_pairs_are_the_reference_pairs_off_the_diagonal(name):
    fx = _load(name)
    assert float(fx["margin"]) > 1e-4             # no pair sits at the threshold: fp32 rounding cannot decide one
    m = _model_from(fx, "cuda:0")
    i, j = _hip_pairs(m, fx["threshold"])
    off_diag = fx["pairs_i"] != fx["pairs_j"]
    np.testing.assert_array_equal(i, fx["pairs_i"][off_diag])
    np.testing.assert_array_equal(j, fx["pairs_j"][off_diag])


@pytest.mark.gpu
def test_hip_merge_equals_the_reference_where_cdist_is_direct():
    """N = 24: torch.cdist computes distances directly, the reference has no self-pairs, and the whole method on the
    GPU gives the reference's tensors bit for bit."""
    fx = _load("merge_n24")
    assert int(fx["self_pairs"]) == 0
    m = _model_from(fx, "cuda:0")
    m.merge_nearby_gaussians(float(fx["threshold"]))
    assert m.num_gaussians == int(fx["n_after"])
    for k in KEYS:
        assert getattr(m, k).is_cuda
        np.testing.assert_array_equal(getattr(m, k).data.cpu().numpy(), fx["out_" + k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("n,thr,spread", [(100_000, 0.02, 1.0), (100_000, 0.004, 0.2), (20_000, 0.5, 40.0)])
def test_hip_merge_vs_cpu_loop(n, thr, spread):
    """N = 100 k (the N x N matrix would be 40 GB): the spatial hash against an independent CPU search + the greedy loop
    written out in Python on float32 scalars.  The third case spreads few points over many cells (threshold 0.5, extent
    +-40) so that distinct cells share hash buckets."""
    from gaussian_splatting_amd import GaussianSplat
    torch.manual_seed(n + int(thr * 1e4))
    a = GaussianSplat(n)
    with torch.no_grad():
        a.positions.data *= spread
        a.positions.data[n // 2] = a.positions.data[n // 3]                  # an exact duplicate: distance 0 is not a pair
        a.positions.data[7] = a.positions.data[11] + torch.tensor([thr * 0.5, 0.0, 0.0])
    pos = a.positions.data.numpy().copy()
    ri, rj = _pairs_direct_fp32(pos, thr)
    assert len(ri) > 50
    b = GaussianSplat(n); b.load_state_dict(a.state_dict()); b = b.to("cuda:0")
    gi, gj = _hip_pairs(b, thr)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gj, rj)
    # the greedy loop of :500-515 on float32 scalars
    P, Fe, O = pos.copy(), a.features.data.numpy().copy(), a.opacity.data.numpy().copy()
    merged = set()
    for i, j in zip(ri.tolist(), rj.tolist()):
        if i in merged or j in merged:
            continue
        w1, w2 = O[i, 0], O[j, 0]
        tw = np.float32(w1 + w2)
        P[i] = (w1 * P[i] + w2 * P[j]) / tw
        Fe[i] = (w1 * Fe[i] + w2 * Fe[j]) / tw
        O[i, 0] = max(w1, w2)
        merged.add(j)
    keep = np.ones(n, bool); keep[list(merged)] = False
    b.merge_nearby_gaussians(thr)
    assert b.num_gaussians == int(keep.sum()) < n
    np.testing.assert_array_equal(b.positions.data.cpu().numpy(), P[keep])
    np.testing.assert_array_equal(b.features.data.cpu().numpy(), Fe[keep])
    np.testing.assert_array_equal(b.opacity.data.cpu().numpy(), O[keep])
    np.testing.assert_array_equal(b.scales.data.cpu().numpy(), a.scales.data.numpy()[keep])
    np.testing.assert_array_equal(b.rotations.data.cpu().numpy(), a.rotations.data.numpy()[keep])


@pytest.mark.gpu
def test_hip_merge_edge_cases():
    from gaussian_splatting_amd import GaussianSplat
    for n in (0, 1, 2):
        m = GaussianSplat(n).to("cuda:0")
        with torch.no_grad():
            m.positions.data.zero_()                       # all coincident: distance 0, nothing merges
        m.merge_nearby_gaussians(0.1)
        assert m.num_gaussians == n
    m = GaussianSplat(64).to("cuda:0")
    with torch.no_grad():
        m.positions.data.zero_()
        m.positions.data[:, 0] = torch.arange(64, device="cuda:0") * 0.01                       # a chain 0.01 apart
        m.positions.data[5] = float("nan"); m.positions.data[9, 1] = float("inf")
        m.opacity.data.fill_(0.5)
    m.merge_nearby_gaussians(0.0101)
    # the walk takes (0,1) (2,3) (6,7) (10,11) (12,13) ... (62,63): the NaN point 5 and the inf point 9 pair with
    # nothing and stay, and so do the neighbours they leave without a partner (4, 8)
    assert m.num_gaussians == 34
    assert torch.isnan(m.positions.data).any() and torch.isinf(m.positions.data).any()
    for thr in (0.0, -1.0, float("nan")):                  # d < threshold never holds (:494): nothing happens
        m2 = GaussianSplat(8).to("cuda:0")
        m2.merge_nearby_gaussians(thr)
        assert m2.num_gaussians == 8
