"""Host side of the gfx950 render path: orchestrates the C-ABI kernels for one view and wraps
them in a torch.autograd.Function.  torch is used for device memory, streams and autograd
plumbing only; all arithmetic of the path runs in csrc/libgsplat_gfx950.so.

Replaces GaussianSplat._forward_impl / _render_chunk and the autograd graph recorded over them
(/root/reference/model/gaussian_splatting.py:70-154, 188-344).
"""
import ctypes as C
import os
import torch
from . import _lib as L


def _ptr(t):
    # a plain int: ctypes converts it per argtypes; building a c_void_p object per argument was ~10 % of a step's host time
    return t if (t is None or isinstance(t, int)) else t.data_ptr()


_stream = L.stream_ptr


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class StageTimer:
    """Optional per-stage HIP-event timing (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream)."""
    def __init__(self):
        self.events = []          # (name, start, end)

    def totals_ms(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.events:
            t, n = out.get(name, (0.0, 0))
            out[name] = (t + a.elapsed_time(b), n + 1)
        return out


_timer = None


def set_stage_timer(t):
    global _timer
    _timer = t


class _stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _timer is not None:
            self.a = torch.cuda.Event(enable_timing=True); self.a.record()

    def __exit__(self, *exc):
        if _timer is not None:
            b = torch.cuda.Event(enable_timing=True); b.record()
            _timer.events.append((self.name, self.a, b))
        return False


class _Scratch:
    """Per-device grow-only scratch for binning (instance buffers are sized by the data)."""
    def __init__(self):
        self.capacity = {}
        self.pinned = {}
        self.raw = {}
        self.layouts = {}
        self.depth_bits = {}      # per device: key bits the next depth sort covers (24 while the top byte is constant)
        self.force_chunk = 0      # 64 / 128 / 256 pins the compositing chunk size (tests); 0 = the device picks it from D

    def cap(self, dev, n):
        return self.capacity.get(dev, max(4 * n, 1 << 16))

    def grow(self, dev, d):
        self.capacity[dev] = int(d) + max(int(d) >> 3, 4096)
        # every per-view buffer changes size with the capacity: hand the cached blocks of the old size back first, or
        # at config-5 scale (tens of GB per buffer) the allocator's reserved-but-unusable blocks exhaust the device.
        # (~100 B of buffers per instance of capacity; below a couple of GB the stale blocks do no harm and emptying the
        # cache would only make a small scene's first orbit stall on hipFree / hipMalloc)
        if self.capacity[dev] * 100 > (2 << 30):
            torch.cuda.empty_cache()

    def count_slot(self, dev):
        """(pinned int64[2], event) for the asynchronous read-back of the instance count and the depth sort's mismatch
        count; one per stream, so that views in flight on different streams (render_orbit) do not share it"""
        key = (dev, L.stream_ptr())
        if key not in self.pinned:
            self.pinned[key] = (torch.zeros(4, dtype=torch.int64).pin_memory(), torch.cuda.Event())    # (D, mismatch count; gsx_train_step: + empty tiles)
        return self.pinned[key]

    def count_slot_raw(self, dev):
        """(pinned, event, pinned address, raw hipEvent_t) -- gsx_view_forward records the event itself, in the middle of
        its launches; torch creates the handle at the event's first record()"""
        key = (dev, L.stream_ptr())
        hit = self.raw.get(key)
        if hit is None:
            pinned, ev = self.count_slot(dev)
            ev.record()
            handle = ev.cuda_event
            hit = self.raw[key] = (pinned, ev, pinned.data_ptr(), handle.value if hasattr(handle, "value") else int(handle))
        return hit


_scratch = _Scratch()


DEPTH_CHECK_BIT = 24      # the depth sort skips its last pass (key bits 24..31) while those bits are the same for every Gaussian


def depth_order(depth_keys, end_bit=32, mismatch=None):
    """order[r] = index of the Gaussian composited r-th (descending depth, ties by index).
    end_bit < 32 sorts on the low bits only -- valid iff the bits above are constant, which `mismatch` (device int64[1],
    the number of keys whose bits [DEPTH_CHECK_BIT, 32) differ from the first key's) lets the caller verify afterwards."""
    lib = L.lib()
    n = depth_keys.numel()
    dev = depth_keys.device
    order = torch.empty(n, dtype=torch.int32, device=dev)
    k_out = torch.empty(n, dtype=torch.int32, device=dev)
    k_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    v_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    wsb = lib.gsx_sort_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with _stage("depth_sort"):
        L.check(lib.gsx_sort_pairs_u32(_ptr(depth_keys), None, _ptr(k_out), _ptr(order), _ptr(k_tmp), _ptr(v_tmp),
                                       n, 0, end_bit, DEPTH_CHECK_BIT, _ptr(mismatch), _ptr(ws), wsb, _stream()), "gsx_sort_pairs_u32")
    return order


def project(positions, scales, rotations, opacity, features, pose):
    lib = L.lib()
    n = positions.shape[0]; dev = positions.device
    mean2d = torch.empty(n, 2, dtype=torch.float32, device=dev)
    depth = torch.empty(n, dtype=torch.float32, device=dev)
    cov2d = torch.empty(n, 2, 2, dtype=torch.float32, device=dev)
    keys = torch.empty(n, dtype=torch.int32, device=dev)
    pre = torch.empty(n, L.REC_FLOATS, dtype=torch.float32, device=dev)
    with _stage("project_fwd"):
        L.check(lib.gsx_project_fwd(_ptr(positions), _ptr(scales), _ptr(rotations), _ptr(opacity), _ptr(features), _ptr(pose), n,
                                    _ptr(mean2d), _ptr(depth), _ptr(cov2d), _ptr(keys), _ptr(pre), _stream()), "gsx_project_fwd")
    return mean2d, depth, cov2d, keys, pre


def build_records(pre, order, H, W):
    lib = L.lib()
    n = order.numel(); dev = order.device
    rec = torch.empty(n, L.REC_FLOATS, dtype=torch.float32, device=dev)
    tt = torch.empty(n, dtype=torch.int32, device=dev)
    rank_of = torch.empty(n, dtype=torch.int32, device=dev)
    tt_sums = torch.empty((n + 255) // 256, dtype=torch.int64, device=dev)
    tile_box = torch.empty(n, dtype=torch.int32, device=dev) if max(H, W) <= 256 * L.TILE else None
    with _stage("build_records"):
        L.check(lib.gsx_build_records(_ptr(pre), _ptr(order), n, H, W, _ptr(rec), _ptr(tt), _ptr(rank_of), _ptr(tt_sums), _ptr(tile_box),
                                      _stream()), "gsx_build_records")
    tt._gsx_block_sums = tt_sums               # travels with the tensor: bin_count(tt) then scans in one launch
    rec._gsx_tile_box = tile_box               # likewise: bin_emit_sort(rec, ...) emits from the packed footprints
    return rec, tt, rank_of


def bin_instances(rec, tt, H, W):
    """-> offsets, tiles_sorted[:D], ranks_sorted[:D], ranges (n_tiles,2), D"""
    lib = L.lib()
    n = tt.numel(); dev = tt.device
    tw, th = (W + L.TILE - 1) // L.TILE, (H + L.TILE - 1) // L.TILE
    offsets = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ranges = torch.empty(tw * th, 2, dtype=torch.int32, device=dev)
    chunk_base = torch.empty(tw * th + 2, dtype=torch.int32, device=dev)      # + the chunk size the device picked
    D = C.c_int64(0)
    for _attempt in range(2):
        cap = _scratch.cap(dev, n)
        bufs = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(4)]
        wsb = lib.gsx_bin_workspace_bytes(n, cap)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        with _stage("bin_instances"):
            st = lib.gsx_bin_instances(_ptr(rec), _ptr(tt), n, H, W, cap, _ptr(offsets), _ptr(bufs[0]), _ptr(bufs[1]),
                                       _ptr(bufs[2]), _ptr(bufs[3]), _ptr(ranges), _ptr(chunk_base), _scratch.force_chunk, C.byref(D), _ptr(ws), wsb, _stream())
        if st == L.GSX_E_CAPACITY:
            _scratch.grow(dev, D.value)
            continue
        L.check(st, "gsx_bin_instances")
        break
    else:
        raise L.GsxError(L.GSX_E_CAPACITY, "gsx_bin_instances")
    global _last_D
    _last_D = int(D.value)
    return offsets, bufs[0], bufs[1], ranges, chunk_base, int(D.value)


def bin_count(tt, total=None):
    """Enqueue scan(tiles_touched) and the asynchronous copy of the instance count to pinned host memory.
    -> offsets, total (device int64[2]: the count, and a word of the caller's that is copied along), (pinned, event):
    the pinned copy is valid after event.synchronize()."""
    lib = L.lib()
    n = tt.numel(); dev = tt.device
    tt_sums = getattr(tt, "_gsx_block_sums", None)      # left by build_records on exactly this tensor, if any
    offsets = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    if total is None:
        total = torch.zeros(2, dtype=torch.int64, device=dev)
    wsb = lib.gsx_scan_workspace_bytes(max(n, 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    pinned, ev = _scratch.count_slot(dev)
    with _stage("bin_count"):
        L.check(lib.gsx_bin_count(_ptr(tt), _ptr(tt_sums), n, _ptr(offsets), _ptr(total), pinned.data_ptr(), _ptr(ws), wsb, _stream()),
                "gsx_bin_count")
    ev.record()
    return offsets, total, (pinned, ev)


def bin_emit_sort(rec, offsets, total, n, H, W, cap):
    """Enqueue emit + sort + ranges + chunk table for at most `cap` instances (count read on the device).
    -> tiles_sorted, ranks_sorted, ranges, chunk_base"""
    lib = L.lib()
    dev = rec.device
    tw, th = (W + L.TILE - 1) // L.TILE, (H + L.TILE - 1) // L.TILE
    ranges = torch.empty(tw * th, 2, dtype=torch.int32, device=dev)
    chunk_base = torch.empty(tw * th + 2, dtype=torch.int32, device=dev)      # + the chunk size the device picked
    bufs = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(4)]
    wsb = lib.gsx_bin_workspace_bytes(n, cap)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with _stage("bin_emit_sort"):
        L.check(lib.gsx_bin_emit_sort(_ptr(rec), _ptr(getattr(rec, "_gsx_tile_box", None)), _ptr(offsets), n, H, W, cap, _ptr(total), _ptr(bufs[0]), _ptr(bufs[1]),
                                      _ptr(bufs[2]), _ptr(bufs[3]), _ptr(ranges), _ptr(chunk_base), _scratch.force_chunk, _ptr(ws), wsb, _stream()),
                "gsx_bin_emit_sort")
    return bufs[0], bufs[1], ranges, chunk_base


_last_D = 0


def last_instance_count():
    """(tile, Gaussian) instance count D of the most recent binning call."""
    return _last_D


def composite_forward(rec, ranks_sorted, ranges, chunk_base, n, H, W, D, want_depth=True, need_bwd=True):
    lib = L.lib()
    dev = rec.device
    rgb = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    dmap = torch.empty(1, H, W, dtype=torch.float32, device=dev) if want_depth else None
    state = torch.empty(4, dtype=torch.int32, device=dev)
    max_chunks = int(lib.gsx_max_chunks(D, H, W, _scratch.force_chunk))
    chunk_state = torch.empty(int(L.lib().gsx_chunk_state_floats(max(max_chunks, 1))), dtype=torch.float32, device=dev)
    with _stage("composite_fwd"):
        L.check(lib.gsx_composite_fwd(_ptr(rec), _ptr(ranks_sorted), _ptr(ranges), _ptr(chunk_base), n, H, W,
                                      max_chunks, _ptr(chunk_state), 1 if need_bwd else 0, _ptr(state),
                                      _ptr(rgb), _ptr(alpha), _ptr(dmap), _stream()), "gsx_composite_fwd")
    return rgb, alpha, dmap, state, chunk_state, max_chunks


def _arena_shapes(n, H, W, cap, max_chunks):
    tiles = ((W + L.TILE - 1) // L.TILE) * ((H + L.TILE - 1) // L.TILE)
    i32, f32 = torch.int32, torch.float32
    return {"order": (i32, (n,)), "rank_of": (i32, (n,)), "rec": (f32, (n, L.REC_FLOATS)), "tt": (i32, (n,)), "offsets": (i32, (n,)),
            "tiles": (i32, (cap,)), "ranks": (i32, (cap,)), "ranges": (i32, (tiles, 2)), "chunk_base": (i32, (tiles + 2,)),
            "chunk_state": (f32, (int(L.lib().gsx_chunk_state_floats(max(max_chunks, 1))),))}


class ViewState:
    """Everything one view's backward needs (kept alive by the autograd node).  Either the buffers are separate tensors
    (the stage-by-stage path) or they live in ONE block, `arena`, at the byte offsets `offs` gsx_view_layout gave: then a
    field becomes a tensor view only when somebody asks for it (tests, tools), and the backward takes addresses."""
    __slots__ = ("pose", "order", "rank_of", "rec", "tt", "offsets", "ranks", "ranges", "chunk_base", "chunk_state", "max_chunks",
                 "state", "D", "H", "W", "cap", "tiles", "arena", "offs", "n")

    def __init__(self):
        self.arena = None

    def __getattr__(self, name):                      # reached only for a slot that holds nothing yet
        arena = object.__getattribute__(self, "arena")
        if arena is None or name not in _ARENA_FIELDS:
            raise AttributeError(name)
        dtype, shape = _arena_shapes(self.n, self.H, self.W, self.cap, self.max_chunks)[name]
        count = 1
        for d in shape:
            count *= d
        off = self.offs[name]
        t = arena[off:off + 4 * count].view(dtype).view(shape)
        setattr(self, name, t)
        return t

    def ptr(self, name):
        """device address of a buffer, without materialising a tensor view for it"""
        arena = self.arena
        if arena is not None and name in _ARENA_FIELDS:
            return arena.data_ptr() + self.offs[name]
        return getattr(self, name).data_ptr()


_ARENA_FIELDS = frozenset(("order", "rank_of", "rec", "tt", "offsets", "tiles", "ranks", "ranges", "chunk_base", "chunk_state"))

# GSX_STAGED=1: render through the stage entry points (one allocation per buffer, one FFI call per stage) instead of
# gsx_view_forward; a stage timer (bench.py's per-kernel HIP events) selects that path too.  Same launches either way.
STAGED = os.environ.get("GSX_STAGED", "0") == "1"
# backward_view tells gsx_composite_bwd the view's instance count, so that a very large view (GSX_PREZERO_ROWS and more) gets
# its partial rows zero-filled in one pass instead of by its occluded chunks; False: never (tests compare the two)
PREZERO_ROWS = True


def forward_view(params, pose, H, W, order=None, want_depth=True, need_bwd=True):
    """One view. params = (positions, scales, rotations, opacity, features), fp32 contiguous CUDA."""
    # Binning and compositing are enqueued against the buffers' CAPACITY; the instance count D reaches the host
    # through pinned memory while the GPU is already sorting, so the stream never drains in the middle of a view
    # (gsx_bin_instances, the synchronising form, idles the GPU for ~0.1 ms per view).  If D exceeds the capacity
    # -- first view of a scene, or a sudden jump -- the truncated results are discarded and the view is redone.
    # The depth sort is optimistic in the same way: while the top byte of the depth keys is the same for every Gaussian
    # (depths within a binade or two, as in an orbit around a scene) it sorts on 24 bits, three passes instead of four;
    # the sort itself counts the keys that break the assumption, the count travels with D, and a view sorted on too few
    # bits is redone on all 32.
    n = params[0].shape[0]
    with L.on_device(params[0].device):
        att = _enqueue_any(params, pose, n, H, W, order, want_depth, need_bwd)
        while not _check_view(att):
            att = _enqueue_any(params, pose, n, H, W, order, want_depth, need_bwd)
    return _finish_view(att, pose, att.mean2d, att.depth, att.cov2d)


def _enqueue_any(params, pose, n, H, W, order, want_depth, need_bwd):
    """One view enqueued, projection to compositing, instance count not yet looked at: through gsx_view_forward, or stage
    by stage (GSX_STAGED=1, a stage timer, or an empty model)."""
    if n > 0 and _timer is None and not STAGED:
        return _enqueue_view_fused(params, pose, n, H, W, order, want_depth, need_bwd)
    mean2d, depth, cov2d, keys, pre = project(*params, pose)
    a = _enqueue_view(pre, keys, n, H, W, order, want_depth, need_bwd)
    a.mean2d, a.depth, a.cov2d = mean2d, depth, cov2d
    return a


class _Attempt:
    """One enqueued view (everything after the projection) whose instance count the host has not looked at yet."""
    __slots__ = ("order", "rec", "tt", "rank_of", "offsets", "tiles", "ranks", "ranges", "chunk_base", "rgb", "alpha", "dmap", "state",
                 "chunk_state", "max_chunks", "cap", "bits", "sorted_here", "pinned", "ev", "dev", "n", "H", "W", "D",
                 "arena", "offs", "mean2d", "depth", "cov2d")

    def __init__(self):
        self.arena = None


def _view_layout(n, H, W, cap, force, need_bwd):
    key = (n, H, W, cap, force, need_bwd)
    hit = _scratch.layouts.get(key)
    if hit is None:
        offs = (C.c_int64 * L.VIEW_SLOTS)(); mc = C.c_int64(0)
        nbytes = int(L.lib().gsx_view_layout(n, H, W, cap, force, 1 if need_bwd else 0, offs, C.byref(mc)))
        if nbytes < 0:
            raise L.GsxError(nbytes, "gsx_view_layout")
        if len(_scratch.layouts) >= 64:
            _scratch.layouts.clear()
        hit = _scratch.layouts[key] = (nbytes, dict(zip(L.VIEW_SLOT_NAMES, (int(o) for o in offs))), int(mc.value))
    return hit


def _enqueue_view_fused(params, pose, n, H, W, given_order, want_depth, need_bwd):
    """The whole view -- projection to compositing -- in ONE call over ONE block (gsx_view_forward): the same launches as
    project() + _enqueue_view(), without their ~30 allocations and 7 FFI calls."""
    a = _Attempt()
    positions, scales, rotations, opacity, features = params
    dev = a.dev = positions.device
    a.n, a.H, a.W = n, H, W
    a.sorted_here = given_order is None
    a.bits = _scratch.depth_bits.get(dev, 32) if given_order is None else 32
    if given_order is not None:
        a.order = given_order
    a.cap = _scratch.cap(dev, n)
    force = _scratch.force_chunk
    nbytes, a.offs, a.max_chunks = _view_layout(n, H, W, a.cap, force, bool(need_bwd))
    a.arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    f32 = torch.float32
    a.mean2d = torch.empty(n, 2, dtype=f32, device=dev)
    a.depth = torch.empty(n, dtype=f32, device=dev)
    a.cov2d = torch.empty(n, 2, 2, dtype=f32, device=dev)
    a.rgb = torch.empty(3, H, W, dtype=f32, device=dev)
    a.alpha = torch.empty(1, H, W, dtype=f32, device=dev)
    a.dmap = torch.empty(1, H, W, dtype=f32, device=dev) if want_depth else None
    # not in the block: callers keep it (the exit rank is one of forward()'s outputs) and must not pin the block with it
    a.state = torch.empty(4, dtype=torch.int32, device=dev)
    a.pinned, a.ev, pinned_ptr, ev_handle = _scratch.count_slot_raw(dev)
    L.check(L.lib().gsx_view_forward(positions.data_ptr(), scales.data_ptr(), rotations.data_ptr(), opacity.data_ptr(), features.data_ptr(),
                                     pose.data_ptr(), n, H, W, a.cap, a.bits, _ptr(given_order), 1 if need_bwd else 0, force,
                                     a.arena.data_ptr(), nbytes, a.mean2d.data_ptr(), a.depth.data_ptr(), a.cov2d.data_ptr(),
                                     a.rgb.data_ptr(), a.alpha.data_ptr(), _ptr(a.dmap), a.state.data_ptr(), pinned_ptr, ev_handle, _stream()),
            "gsx_view_forward")
    return a


def _enqueue_view(pre, keys, n, H, W, given_order, want_depth, need_bwd):
    a = _Attempt()
    dev = a.dev = pre.device
    a.n, a.H, a.W = n, H, W
    total = torch.empty(2, dtype=torch.int64, device=dev)
    a.bits, a.sorted_here = 32, given_order is None
    if given_order is None:
        a.bits = _scratch.depth_bits.get(dev, 32)
        a.order = depth_order(keys, a.bits, total[1:])
    else:
        a.order = given_order
    a.rec, a.tt, a.rank_of = build_records(pre, a.order, H, W)
    a.offsets, total, (a.pinned, a.ev) = bin_count(a.tt, total)
    a.cap = _scratch.cap(dev, n)
    a.tiles, a.ranks, a.ranges, a.chunk_base = bin_emit_sort(a.rec, a.offsets, total, n, H, W, a.cap)
    a.rgb, a.alpha, a.dmap, a.state, a.chunk_state, a.max_chunks = composite_forward(a.rec, a.ranks, a.ranges, a.chunk_base, n, H, W, a.cap,
                                                                                     want_depth, need_bwd)
    return a


def _check_view(a):
    """Waits for the attempt's instance count (the GPU is still compositing) -> True if its results stand; otherwise the
    scratch capacity / depth-sort width have been corrected and the view must be enqueued again."""
    a.ev.synchronize()
    a.D = D = int(a.pinned[0])
    redo = False
    if a.sorted_here:
        constant_top = int(a.pinned[1]) == 0
        _scratch.depth_bits[a.dev] = DEPTH_CHECK_BIT if constant_top else 32
        redo = a.bits < 32 and not constant_top
    if D >= 0xFFFFFFFF:
        raise L.GsxError(L.GSX_E_RANGE, "gsx_bin_emit_sort")
    if D > a.cap:
        redo = True
        for k in ("tiles", "ranks", "ranges", "chunk_base", "rgb", "alpha", "dmap", "state", "chunk_state", "arena"):
            setattr(a, k, None)                   # truncated results: release them before the buffers grow
        _scratch.grow(a.dev, D)
    return not redo


class _LazyExtra:
    """forward_view's third result: buffers only tests and tools look at ("tiles": the sorted tile ids)"""
    def __init__(self, vs):
        self._vs = vs

    def __getitem__(self, name):
        return getattr(self._vs, name)


def _finish_view(a, pose, mean2d, depth, cov2d):
    global _last_D
    _last_D = a.D
    vs = ViewState()
    vs.pose, vs.max_chunks, vs.D, vs.H, vs.W, vs.cap, vs.n = pose, a.max_chunks, a.D, a.H, a.W, a.cap, a.n
    if a.arena is not None:
        vs.arena, vs.offs, vs.state = a.arena, a.offs, a.state
        if not a.sorted_here:
            vs.order = a.order
    else:
        vs.order, vs.rank_of, vs.rec, vs.tt, vs.offsets = a.order, a.rank_of, a.rec, a.tt, a.offsets
        vs.ranks, vs.ranges, vs.chunk_base, vs.chunk_state, vs.state, vs.tiles = a.ranks, a.ranges, a.chunk_base, a.chunk_state, a.state, a.tiles
    return (a.rgb, a.alpha, a.dmap, mean2d, depth, cov2d), vs, _LazyExtra(vs)


def backward_view(params, vs, d_rgb, d_mean2d, d_depth, d_cov2d, grads, accumulate):
    """Adds (or writes) this view's parameter gradients into `grads` (5 tensors)."""
    lib = L.lib()
    positions, scales, rotations, opacity, features = params
    n = positions.shape[0]; dev = positions.device
    partials = None
    if d_rgb is not None and vs.D > 0:
        # sized by the buffers' capacity, not by this view's count: the same size every step, so torch's caching allocator
        # hands the block back instead of freeing and re-allocating tens of GB per view at config-5 scale
        if vs.arena is not None and vs.offs.get("partials", -1) >= 0:
            partials = vs.arena.data_ptr() + vs.offs["partials"]        # `cap` rows, over the forward's dead scratch
        else:
            partials = torch.empty(max(vs.D, getattr(vs, "cap", 0) or 0), 9, dtype=torch.float32, device=dev)
        with _stage("composite_bwd"):
            L.check(lib.gsx_composite_bwd(vs.ptr("rec"), vs.ptr("ranks"), vs.ptr("ranges"), vs.ptr("chunk_base"), vs.ptr("offsets"),
                                          n, vs.H, vs.W, vs.max_chunks, vs.ptr("chunk_state"), vs.ptr("state"),
                                          _ptr(d_rgb), _ptr(partials), int(vs.D) if PREZERO_ROWS else 0, _stream()), "gsx_composite_bwd")
    with _stage("preprocess_bwd"):
      L.check(lib.gsx_preprocess_bwd(_ptr(positions), _ptr(scales), _ptr(rotations),
                                   _ptr(vs.pose), vs.ptr("rank_of"), vs.ptr("offsets"), vs.ptr("tt"),
                                   _ptr(partials), vs.ptr("state"), _ptr(d_mean2d), _ptr(d_depth), _ptr(d_cov2d),
                                   n, 1 if accumulate else 0, *[_ptr(g) for g in grads], _stream()), "gsx_preprocess_bwd")


GRAD_TAIL = 8


class RenderFunction(torch.autograd.Function):
    """(positions, scales, rotations, opacity, features, poses (B,4,4)) ->
       rendered_images (B,3,H,W), positions2d (B,N,2), depths (B,N), cov2d (B,N,2,2),
       alpha (B,1,H,W), depth_map (B,1,H,W), n_composited-state (B,) [last three non-differentiable]"""

    @staticmethod
    def forward(ctx, positions, scales, rotations, opacity, features, poses, H, W, want_depth=True):
        if not positions.is_cuda:
            raise RuntimeError("gaussian-splatting_amd renders on an MI355X only: parameters must live on a CUDA(HIP) device; "
                               "there is no CPU fallback")
        ctx.set_materialize_grads(False)
        dev = positions.device
        if not (scales.device == rotations.device == opacity.device == features.device == dev):
            raise RuntimeError("gaussian-splatting_amd: the five parameter tensors must live on the same device")
        params = tuple(_f32c(t.detach()) for t in (positions, scales, rotations, opacity, features))
        poses = _f32c(poses.detach().to(dev))          # a host pointer must never reach a kernel
        B = poses.shape[0]
        need_bwd = any(ctx.needs_input_grad[:5])
        outs, views = [], []
        order = None
        # the kernels go to the stream of the PARAMETERS' device, whatever the caller's current device is (the reference's
        # launcher never calls set_device: scripts/train.py:76,132)
        with L.on_device(dev):
            for b in range(B):
                o, vs, _ = forward_view(params, poses[b], H, W, order=order, want_depth=want_depth, need_bwd=need_bwd)
                if order is None and B > 1:
                    order = vs.order       # every view is composited in view 0's order (gaussian_splatting.py:206)
                outs.append(o); views.append(vs)
        # detach() shares the version counter: an in-place update of a parameter between forward and backward (the backward
        # recomputes the projection from the live parameter memory) raises autograd's version error instead of mixing old
        # records with new projections
        ctx.save_for_backward(*params)
        ctx.views, ctx.hw = views, (H, W)
        # B == 1 (the reference's only working batch size): views, no copies
        cat = (lambda i: outs[0][i].unsqueeze(0)) if B == 1 else (lambda i: torch.stack([o[i] for o in outs]))
        rgb, alpha, dmap, mean2d, depth, cov2d = (cat(i) if outs[0][i] is not None else rgb_placeholder(outs[0][0])
                                                  for i in range(6))
        state = views[0].state[0:1] if B == 1 else torch.stack([vs.state[0] for vs in views])
        ctx.mark_non_differentiable(alpha, dmap, state)
        return rgb, mean2d, depth, cov2d, alpha, dmap, state

    @staticmethod
    def backward(ctx, d_rgb, d_mean2d, d_depth, d_cov2d, _a, _d, _s):
        params = ctx.saved_tensors
        n = params[0].shape[0]
        # one flat 14*N gradient buffer (rotations first: its float4 stores need 16-byte alignment);
        # the data-parallel path all-reduces this buffer in ONE collective (distributed.py)
        # GRAD_TAIL spare floats behind the gradients: the data-parallel wrapper parks its per-tensor presence flags there,
        # so that gradients + flags are one contiguous run and still ONE collective without a copy
        flat = torch.empty(14 * n + GRAD_TAIL, dtype=torch.float32, device=params[0].device)
        g_rot = flat[0:4 * n].view(n, 4); g_pos = flat[4 * n:7 * n].view(n, 3); g_scl = flat[7 * n:10 * n].view(n, 3)
        g_opa = flat[10 * n:11 * n].view(n, 1); g_fea = flat[11 * n:14 * n].view(n, 3)
        grads = [g_pos, g_scl, g_rot, g_opa, g_fea]
        with L.on_device(params[0].device):
            for b, vs in enumerate(ctx.views):
                backward_view(params, vs,
                              None if d_rgb is None else _f32c(d_rgb[b]),
                              None if d_mean2d is None else _f32c(d_mean2d[b]),
                              None if d_depth is None else _f32c(d_depth[b]),
                              None if d_cov2d is None else _f32c(d_cov2d[b]),
                              grads, accumulate=(b > 0))
        return grads[0], grads[1], grads[2], grads[3], grads[4], None, None, None, None


def rgb_placeholder(like):
    """Empty stand-in for an output that was not requested (autograd Functions must return tensors)."""
    return like.new_empty(0)


def render_views(positions, scales, rotations, opacity, features, poses, image_size, want_depth=True):
    """want_depth=False skips the alpha-composited depth map (the reference has none): one accumulator and one
    chunk-state word less per pixel in the compositing kernels; the depth_map output is then an empty tensor."""
    H, W = int(image_size[0]), int(image_size[1])
    return RenderFunction.apply(positions, scales, rotations, opacity, features, poses, H, W, want_depth)


@torch.no_grad()
def render_orbit(positions, scales, rotations, opacity, features, poses, image_size, out=None, streams=2):
    """the inference caller of the path (/root/reference/utils/visualization.py:343-347 renders an
    orbit one `model(pose.unsqueeze(0), size)` call at a time).  Renders V poses (V,4,4) into ONE (V,3,H,W) tensor: no
    autograd node, no backward state (the combine pass skips its second scan and the chunk-end transmittances), each
    view composited in its own depth order exactly as V separate forward calls would.  The views are independent, so
    they are enqueued round-robin on `streams` HIP streams: one view's launch-bound stretch (sorts, scans, binning: a
    third of its time) overlaps another's compositing, and the host looks at a view's instance count only after it has
    enqueued the next one."""
    if not positions.is_cuda:
        raise RuntimeError("gaussian-splatting_amd renders on an MI355X only: parameters must live on a CUDA(HIP) device; "
                           "there is no CPU fallback")
    with L.on_device(positions.device):               # streams, events and launches all belong to the parameters' device
        return _render_orbit(positions, scales, rotations, opacity, features, poses, image_size, out, streams)


def _render_orbit(positions, scales, rotations, opacity, features, poses, image_size, out, streams):
    H, W = int(image_size[0]), int(image_size[1])
    params = tuple(_f32c(t.detach()) for t in (positions, scales, rotations, opacity, features))
    poses = _f32c(poses.detach().to(positions.device))
    V = poses.shape[0]
    n = params[0].shape[0]
    frames = out if out is not None else torch.empty(V, 3, H, W, dtype=torch.float32, device=positions.device)
    dev = positions.device
    S = max(1, min(int(streams), V))
    if S == 1:
        for v in range(V):
            o, _, _ = forward_view(params, poses[v], H, W, want_depth=False, need_bwd=False)
            frames[v].copy_(o[0])
        return frames
    main = torch.cuda.current_stream(dev)
    pool = _orbit_streams(dev, S)
    start = torch.cuda.Event(); start.record(main)
    in_flight = []

    def retire(item):
        v, s, att = item
        with torch.cuda.stream(s):
            while not _check_view(att):                       # rare: capacity or depth-sort width had to be corrected
                att = _enqueue_any(params, poses[v], n, H, W, None, False, False)
            global _last_D
            _last_D = att.D
            frames[v].copy_(att.rgb)

    for v in range(V):
        s = pool[v % S]
        with torch.cuda.stream(s):
            if v < S:
                s.wait_event(start)                           # parameters and poses were produced on the caller's stream
            att = _enqueue_any(params, poses[v], n, H, W, None, False, False)
        in_flight.append((v, s, att))
        if len(in_flight) == S:
            retire(in_flight.pop(0))
    while in_flight:
        retire(in_flight.pop(0))
    for s in pool:
        done = torch.cuda.Event(); done.record(s); main.wait_event(done)
    return frames


_orbit_pool = {}


def _orbit_streams(dev, count):
    pool = _orbit_pool.setdefault(dev, [])
    while len(pool) < count:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:count]
