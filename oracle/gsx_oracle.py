"""ctypes/numpy front-end of the CPU oracle (oracle/gsx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (gaussian-splatting_amd/).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgsx_oracle.so")
TILE = 16
REC_FLOATS = 12
RANK_NONE = 0xFFFFFFFF


def build(force=False):
    src = os.path.join(_HERE, "gsx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.gsxo_render_dense.restype = C.c_int64
        _lib.gsxo_render_tiled.restype = C.c_int64
        _lib.gsxo_scan.restype = C.c_int64
        _lib.gsxo_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def linspace(steps):
    out = np.empty(steps, np.float32)
    lib().gsxo_linspace(C.c_int(steps), _p(out))
    return out


def project(pos, scl, rot, pose):
    pos, scl, rot, pose = _f32(pos), _f32(scl), _f32(rot), _f32(pose).reshape(16)
    n = pos.shape[0]
    mean2d = np.empty((n, 2), np.float32); depth = np.empty(n, np.float32); cov = np.empty((n, 2, 2), np.float32)
    lib().gsxo_project(_p(pos), _p(scl), _p(rot), _p(pose), C.c_int(n), _p(mean2d), _p(depth), _p(cov))
    return mean2d, depth, cov


def depth_keys(depth):
    depth = _f32(depth); out = np.empty(depth.shape[0], np.uint32)
    lib().gsxo_depth_keys(_p(depth), C.c_int(depth.shape[0]), _p(out))
    return out


def depth_order(depth):
    depth = _f32(depth); out = np.empty(depth.shape[0], np.uint32)
    lib().gsxo_depth_order(_p(depth), C.c_int(depth.shape[0]), _p(out))
    return out


def records(mean2d, depth, cov, opacity, features, order, H, W):
    n = order.shape[0]
    rec = np.zeros((n, REC_FLOATS), np.float32); rect = np.zeros((n, 4), np.int32); tt = np.zeros(n, np.uint32)
    lib().gsxo_records(_p(_f32(mean2d)), _p(_f32(depth)), _p(_f32(cov)), _p(_f32(opacity).reshape(-1)),
                       _p(_f32(features)), _p(np.ascontiguousarray(order, np.uint32)), C.c_int(n), C.c_int(H), C.c_int(W),
                       _p(rec), _p(rect), _p(tt))
    return rec, rect, tt


def render_dense(rec, H, W):
    n = rec.shape[0]
    rgb = np.empty((3, H, W), np.float32); acc = np.empty((H, W), np.float32)
    k = lib().gsxo_render_dense(_p(rec), C.c_int(n), C.c_int(H), C.c_int(W), _p(rgb), _p(acc))
    return rgb, acc, int(k)


def backward_dense(rec, H, W, n_composited, d_rgb):
    n = rec.shape[0]
    d_rec = np.empty((n, 9), np.float64)
    lib().gsxo_backward_dense(_p(rec), C.c_int(n), C.c_int(H), C.c_int(W), C.c_int64(n_composited),
                              _p(_f32(d_rgb).reshape(3, H, W)), _p(d_rec))
    return d_rec


def preprocess_backward(pos, scl, rot, opa, fea, pose, order, d_rec, d_mean2d=None, d_depth=None, d_cov=None, n_composited=-1):
    n = pos.shape[0]
    g = [np.zeros((n, 3), np.float64), np.zeros((n, 3), np.float64), np.zeros((n, 4), np.float64),
         np.zeros((n, 1), np.float64), np.zeros((n, 3), np.float64)]
    dm = _f32(d_mean2d) if d_mean2d is not None else None
    dd = _f32(d_depth) if d_depth is not None else None
    dc = _f32(d_cov) if d_cov is not None else None
    lib().gsxo_preprocess_backward(_p(_f32(pos)), _p(_f32(scl)), _p(_f32(rot)), _p(_f32(opa)), _p(_f32(fea)),
                                   _p(_f32(pose).reshape(16)), _p(np.ascontiguousarray(order, np.uint32)), C.c_int(n),
                                   _p(np.ascontiguousarray(d_rec, np.float64)), _p(dm), _p(dd), _p(dc),
                                   *[_p(x) for x in g], C.c_int64(n_composited))
    return dict(positions=g[0], scales=g[1], rotations=g[2], opacity=g[3], features=g[4])


def bin_tiles(rect, tiles_touched, H, W):
    """tiled-mode integer products: offsets, D, unsorted (tile,rank), sorted keys/ranks, ranges"""
    n = rect.shape[0]
    tw, th = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    offsets = np.zeros(n, np.uint64)
    D = int(lib().gsxo_scan(_p(tiles_touched), C.c_int(n), _p(offsets)))
    tile_of = np.zeros(max(D, 1), np.uint32); rank_of = np.zeros(max(D, 1), np.uint32)
    lib().gsxo_emit(_p(rect), _p(offsets), C.c_int(n), C.c_int(tw), _p(tile_of), _p(rank_of))
    keys = np.zeros(max(D, 1), np.uint64); ranks = np.zeros(max(D, 1), np.uint32)
    ranges = np.zeros((tw * th, 2), np.uint32)
    lib().gsxo_sort_ranges(_p(tile_of), _p(rank_of), C.c_int64(D), C.c_int(tw * th), _p(keys), _p(ranks), _p(ranges))
    return dict(offsets=offsets, D=D, tile_of=tile_of[:D], rank_of=rank_of[:D], keys=keys[:D], ranks=ranks[:D],
                ranges=ranges, tw=tw, th=th)


def render_tiled(rec, H, W, ranks, ranges, rank_limit=None, want_depth=False):
    n = rec.shape[0]
    rgb = np.zeros((3, H, W), np.float32); acc = np.zeros((H, W), np.float32)
    dm = np.zeros((H, W), np.float32) if want_depth else None
    lim = n if rank_limit is None else rank_limit
    ranks = np.ascontiguousarray(ranks, np.uint32)
    if ranks.size == 0:
        ranks = np.zeros(1, np.uint32)
    k = lib().gsxo_render_tiled(_p(rec), C.c_int(n), C.c_int(H), C.c_int(W), _p(ranks),
                                _p(np.ascontiguousarray(ranges, np.uint32)), C.c_uint32(lim), _p(rgb), _p(acc), _p(dm))
    return rgb, acc, int(k), dm


def backward_tiled(rec, H, W, ranks, ranges, rank_limit, d_rgb):
    n = rec.shape[0]
    d_rec = np.empty((n, 9), np.float64)
    ranks = np.ascontiguousarray(ranks, np.uint32)
    if ranks.size == 0:
        ranks = np.zeros(1, np.uint32)
    lib().gsxo_backward_tiled(_p(rec), C.c_int(n), C.c_int(H), C.c_int(W), _p(ranks),
                              _p(np.ascontiguousarray(ranges, np.uint32)), C.c_uint32(rank_limit),
                              _p(_f32(d_rgb).reshape(3, H, W)), _p(d_rec))
    return d_rec


def num_threads():
    return int(lib().gsxo_num_threads())


# ---------------------------------------------------------------------------------------------
# whole-path conveniences used by the tests
# ---------------------------------------------------------------------------------------------
def forward(params, pose, H, W, mode="tiled", order=None):
    """params: dict positions/scales/rotations/opacity/features.  Returns a dict of everything.
    `order` overrides the depth order (used where the reference's unstable argsort broke ties
    differently from this build's index-ascending rule)."""
    mean2d, depth, cov = project(params["positions"], params["scales"], params["rotations"], pose)
    order = depth_order(depth) if order is None else np.ascontiguousarray(order, np.uint32)
    rec, rect, tt = records(mean2d, depth, cov, params["opacity"], params["features"], order, H, W)
    out = dict(mean2d=mean2d, depth=depth, cov2d=cov, order=order, rec=rec, rect=rect, tiles_touched=tt)
    if mode == "dense":
        rgb, acc, k = render_dense(rec, H, W)
        out.update(rgb=rgb, alpha=acc, n_composited=k)
    else:
        b = bin_tiles(rect, tt, H, W)
        out.update(b)
        rgb, acc, k, dm = render_tiled(rec, H, W, b["ranks"], b["ranges"], None, want_depth=True)
        if k < rec.shape[0]:            # global early exit fired: recomposite with the rank limit
            rgb, acc, k2, dm = render_tiled(rec, H, W, b["ranks"], b["ranges"], k, want_depth=True)
        out.update(rgb=rgb, alpha=acc, n_composited=k, depth_map=dm)
    return out


def backward(params, pose, H, W, fwd, d_rgb, d_mean2d=None, d_depth=None, d_cov=None, mode="tiled"):
    if mode == "dense":
        d_rec = backward_dense(fwd["rec"], H, W, fwd["n_composited"], d_rgb)
    else:
        d_rec = backward_tiled(fwd["rec"], H, W, fwd["ranks"], fwd["ranges"], fwd["n_composited"], d_rgb)
    g = preprocess_backward(params["positions"], params["scales"], params["rotations"], params["opacity"],
                            params["features"], pose, fwd["order"], d_rec, d_mean2d, d_depth, d_cov, int(fwd["n_composited"]))
    g["d_rec"] = d_rec
    return g
