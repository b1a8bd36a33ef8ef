"""CPU tier: the oracle (oracle/gsx_oracle.c) against golden vectors produced by running the
reference itself (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
from oracle import gsx_oracle as O
from helpers import golden_names, load_golden, rel_err, rgb_err, grad_tol, RGB_TOL, GRAD_RTOL, PARAMS


@pytest.fixture(scope="module", autouse=True)
def _build():
    O.build()


@pytest.mark.parametrize("name", golden_names())
def test_linspace_bitexact(name):
    fx = load_golden(name)
    assert np.array_equal(O.linspace(fx["W"]), fx["grid_x"])
    assert np.array_equal(O.linspace(fx["H"]), fx["grid_y"])


@pytest.mark.parametrize("name", golden_names())
def test_projection_bitexact(name):
    """a5-a8: projected means, depths and 2D covariances equal the reference's bit for bit."""
    fx = load_golden(name)
    m, d, c = O.project(fx["positions"], fx["scales"], fx["rotations"], fx["poses"][0])
    assert np.array_equal(m, fx["out_pos2d"][0])
    assert np.array_equal(d, fx["out_depths"][0])
    assert rel_err(c, fx["out_cov2d"][0]) < 1e-6


@pytest.mark.parametrize("name", golden_names())
def test_depth_order(name):
    fx = load_golden(name)
    order = O.depth_order(fx["out_depths"][0])
    if not fx["has_ties"]:
        assert np.array_equal(order.astype(np.int64), fx["order"])
    d = fx["out_depths"][0][order]
    assert np.all(d[:-1] >= d[1:])
    # ties: index ascending
    same = d[:-1] == d[1:]
    assert np.all(order[:-1][same] < order[1:][same])


@pytest.mark.parametrize("mode", ["dense", "tiled"])
@pytest.mark.parametrize("name", golden_names())
def test_render_and_grads(name, mode):
    fx = load_golden(name)
    if name.startswith("config1") and mode == "dense":
        pytest.skip("dense config-1 backward takes minutes; the tiled mode covers config 1")
    H, W = fx["H"], fx["W"]
    fw = O.forward(fx["params"], fx["poses"][0], H, W, mode, order=fx["order"] if fx["has_ties"] else None)
    assert fw["n_composited"] == int(fx["n_composited"])          # global early exit index
    # dense mode is the reference's algorithm literally: 1e-5 (a tenth of north_star's bar).  With opacities up to 3 (exitop*)
    # every splat multiplies a pixel's (1 - A) by up to |1 - alpha| = 2 and the image reaches 1.7: two float32 evaluations of
    # that product differ by 1.6e-5 at the worst pixel, so those fixtures get 3e-5 -- still a third of the bar
    dense_tol = 1e-5 if float(fx["opacity"].max()) <= 1.5 else 3e-5
    assert rgb_err(fw["rgb"], fx) < (dense_tol if mode == "dense" else RGB_TOL)
    g = O.backward(fx["params"], fx["poses"][0], H, W, fw, fx["d_rgb"][0], fx["d_pos2d"][0],
                   fx["d_depths"][0], fx["d_cov2d"][0], mode)
    for k in PARAMS:
        assert rel_err(g[k], fx["g_" + k]) < grad_tol(fx, k), k


def test_tiled_products_consistent():
    """keys sorted, ranges partition [0,D), every instance inside its Gaussian's rectangle."""
    fx = load_golden("n300_32x24")
    fw = O.forward(fx["params"], fx["poses"][0], fx["H"], fx["W"], "tiled")
    keys = fw["keys"]
    assert np.all(keys[:-1] < keys[1:])
    r = fw["ranges"]
    assert r[0, 0] == 0 and r[-1, 1] == fw["D"] and np.all(r[1:, 0] == r[:-1, 1])
    assert fw["D"] == int(fw["tiles_touched"].sum())
    tiles = (keys >> np.uint64(32)).astype(np.int64); ranks = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    rc = fw["rect"][ranks]
    tx, ty = tiles % fw["tw"], tiles // fw["tw"]
    assert np.all((tx >= rc[:, 0]) & (tx < rc[:, 2]) & (ty >= rc[:, 1]) & (ty < rc[:, 3]))
