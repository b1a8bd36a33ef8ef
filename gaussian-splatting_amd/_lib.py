"""ctypes binding of csrc/libgsplat_gfx950.so (C ABI declared in include/gsx.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgsplat_gfx950.so")

# every symbol include/gsx.h declares
SYMBOLS = [
    "gsx_strerror", "gsx_last_hip_error", "gsx_abi_version",
    "gsx_project_fwd", "gsx_sort_workspace_bytes", "gsx_sort_pairs_u32", "gsx_sort_pairs_u32_static",
    "gsx_scan_workspace_bytes", "gsx_scan_u32", "gsx_build_records",
    "gsx_bin_workspace_bytes", "gsx_bin_instances", "gsx_bin_count", "gsx_bin_emit_sort", "gsx_max_chunks", "gsx_chunk_state_floats",
    "gsx_view_layout", "gsx_view_forward",
    "gsx_composite_fwd", "gsx_composite_bwd", "gsx_preprocess_bwd",
    "gsx_clip_adam_workspace_bytes", "gsx_clip_adam_step",
    "gsx_image_loss_workspace_bytes", "gsx_image_loss_fwd", "gsx_image_loss_bwd", "gsx_image_loss_fwd_grad",
    "gsx_l1_ssim_workspace_bytes", "gsx_l1_ssim_fwd", "gsx_l1_ssim_bwd",
    "gsx_keep_flags", "gsx_compact_rows", "gsx_clamp_inplace",
    "gsx_merge_workspace_bytes", "gsx_merge_count", "gsx_merge_fill", "gsx_merge_greedy_host", "gsx_train_step", "gsx_train_step_grads", "gsx_train_step_apply",
    "gsx_comm_available", "gsx_comm_unique_id", "gsx_comm_init", "gsx_allreduce_avg", "gsx_bcast", "gsx_comm_destroy",
]

ABI_VERSION = 20            # GSX_ABI_VERSION of include/gsx.h (tests check that the two agree)
GSX_OK, GSX_E_INVALID, GSX_E_CAPACITY, GSX_E_HIP, GSX_E_RANGE, GSX_E_AGAIN, GSX_E_COMM = 0, -1, -2, -3, -4, -5, -6
RANK_NONE = 0xFFFFFFFF
REC_FLOATS = int(os.environ.get("GSX_REC_FLOATS", "12"))      # (16 only with a library built for the 64-byte-stride experiment)
TILE = 16
CHUNK = 256
STATE_WORDS = 8
DEPTH_CHECK_BIT = 24        # GSX_DEPTH_CHECK_BIT
STATIC_SORT_MAX_N = 1 << 20 # GSX_STATIC_SORT_MAX_N
# GSX_VIEW_* slots of gsx_constants.h (buffers of one view inside the arena of gsx_view_forward)
VIEW_SLOT_NAMES = ("order", "rank_of", "rec", "tt", "offsets", "tiles", "ranks", "ranges", "chunk_base", "chunk_state",
                   "total", "partials", "depth_keys", "pre", "keys_out", "keys_tmp", "values_tmp", "sort_ws", "tt_block_sums", "tile_box",
                   "scan_ws", "tiles_tmp", "ranks_tmp", "bin_ws")
VIEW_SLOTS = len(VIEW_SLOT_NAMES)


class GsxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.gsx_strerror(status).decode() if _lib is not None else str(status)
        extra = f" (hipError {_lib.gsx_last_hip_error()})" if (_lib is not None and status == GSX_E_HIP) else ""
        super().__init__(f"{where}: {msg}{extra}")


_lib = None
_p, _i32, _i64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t


class GsxStep(C.Structure):
    """struct gsx_step of include/gsx.h (gsx_train_step); tests check the size against the header's field list"""
    _fields_ = [("param", _p * 5), ("grad", _p * 5), ("exp_avg", _p * 5), ("exp_avg_sq", _p * 5), ("lr", C.c_float * 5),
                ("adam_order", _i32 * 5), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("max_grad_norm", C.c_float),
                ("step", _i64), ("write_clipped_grad", _i32), ("N", _i32), ("H", _i32), ("W", _i32), ("depth_bits", _i32),
                ("chunk_force", _i32), ("loss_kind", _i32), ("lambda_dssim", C.c_float), ("capacity", _i64),
                ("pose", _p), ("target", _p), ("arena", _p), ("arena_bytes", _i64),
                ("mean2d", _p), ("depth", _p), ("cov2d", _p), ("rgb", _p), ("alpha", _p), ("state", _p), ("d_rgb", _p), ("loss", _p),
                ("loss_ws", _p), ("loss_ws_bytes", _sz), ("adam_ws", _p), ("adam_ws_bytes", _sz),
                ("counts_pinned_host", _p), ("count_event", _p)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.dirname(LIB_PATH)}`; there is no CPU/torch fallback for the render path")
    L = C.CDLL(LIB_PATH)
    L.gsx_abi_version.restype = C.c_int
    if L.gsx_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI {L.gsx_abi_version()}, this package binds ABI {ABI_VERSION}: rebuild it (make -C {os.path.dirname(LIB_PATH)})")
    L.gsx_strerror.restype = C.c_char_p; L.gsx_strerror.argtypes = [C.c_int]
    L.gsx_last_hip_error.restype = C.c_int
    L.gsx_abi_version.restype = C.c_int
    L.gsx_project_fwd.argtypes = [_p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p]
    L.gsx_sort_workspace_bytes.restype = _sz; L.gsx_sort_workspace_bytes.argtypes = [_i64]
    L.gsx_sort_pairs_u32.argtypes = [_p, _p, _p, _p, _p, _p, _i64, C.c_int, C.c_int, C.c_int, _p, _p, _sz, _p]
    L.gsx_sort_pairs_u32_static.argtypes = [_p, _p, _p, _p, _i64, C.c_int, C.c_int, _p]
    L.gsx_scan_workspace_bytes.restype = _sz; L.gsx_scan_workspace_bytes.argtypes = [_i64]
    L.gsx_scan_u32.argtypes = [_p, _p, _p, _i64, _p, _sz, _p]
    L.gsx_build_records.argtypes = [_p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]
    L.gsx_bin_workspace_bytes.restype = _sz; L.gsx_bin_workspace_bytes.argtypes = [_i32, _i64]
    L.gsx_bin_instances.argtypes = [_p, _p, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _i32, C.POINTER(_i64), _p, _sz, _p]
    L.gsx_bin_count.argtypes = [_p, _p, _i32, _p, _p, _p, _p, _sz, _p]
    L.gsx_bin_emit_sort.argtypes = [_p, _p, _p, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _sz, _p]
    L.gsx_max_chunks.restype = _i64; L.gsx_max_chunks.argtypes = [_i64, _i32, _i32, _i32]
    L.gsx_chunk_state_floats.restype = _i64; L.gsx_chunk_state_floats.argtypes = [_i64]
    L.gsx_view_layout.restype = _i64; L.gsx_view_layout.argtypes = [_i32, _i32, _i32, _i64, _i32, _i32, _p, _p]
    L.gsx_view_forward.argtypes = [_p] * 6 + [_i32, _i32, _i32, _i64, _i32, _p, _i32, _i32, _p, _i64] + [_p] * 7 + [_p, _p, _p]
    L.gsx_composite_fwd.argtypes = [_p, _p, _p, _p, _i32, _i32, _i32, _i64, _p, _i32, _p, _p, _p, _p, _p]
    L.gsx_composite_bwd.argtypes = [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _i64, _p]
    L.gsx_preprocess_bwd.argtypes = [_p] * 12 + [_i32, _i32] + [_p] * 6
    L.gsx_clip_adam_workspace_bytes.restype = _sz
    L.gsx_clip_adam_step.argtypes = [_i32, _p, _p, _p, _p, _p, _p, C.c_float, C.c_float, C.c_float, _i64, C.c_float,
                                     _i32, _p, _sz, _p]
    L.gsx_keep_flags.argtypes = [_p, _i32, C.c_float, _p, _p]
    L.gsx_compact_rows.argtypes = [_p, _p, _p, _i32, _i32, _p, _p]
    L.gsx_clamp_inplace.argtypes = [_p, _i64, C.c_float, C.c_float, _p]
    L.gsx_merge_workspace_bytes.argtypes = [_i32]; L.gsx_merge_workspace_bytes.restype = C.c_size_t
    L.gsx_merge_count.argtypes = [_p, _i32, C.c_float, _p, _p, C.c_size_t, _p]
    L.gsx_merge_fill.argtypes = [_p, _i32, C.c_float, _p, _p, _p, C.c_size_t, _p]
    L.gsx_merge_greedy_host.argtypes = [_i32, _p, _p, _p, _p, _p, _p]
    L.gsx_train_step.argtypes = [C.POINTER(GsxStep), _p]
    L.gsx_train_step_grads.argtypes = [C.POINTER(GsxStep), _p]
    L.gsx_train_step_apply.argtypes = [C.POINTER(GsxStep), _p]
    L.gsx_comm_unique_id.argtypes = [_p]
    L.gsx_comm_init.argtypes = [_i32, _i32, _p, C.POINTER(_p)]
    L.gsx_allreduce_avg.argtypes = [_p, _p, _i64, _p]
    L.gsx_bcast.argtypes = [_p, _p, _i64, _i32, _p]
    L.gsx_comm_destroy.argtypes = [_p]
    L.gsx_image_loss_workspace_bytes.restype = _sz
    L.gsx_image_loss_fwd.argtypes = [_i32, _p, _p, _i64, _p, _p, _sz, _p]
    L.gsx_image_loss_bwd.argtypes = [_i32, _p, _p, _i64, _p, _p, _p]
    L.gsx_image_loss_fwd_grad.argtypes = [_i32, _p, _p, _i64, _p, _p, _p, _sz, _p]
    L.gsx_l1_ssim_workspace_bytes.restype = _sz; L.gsx_l1_ssim_workspace_bytes.argtypes = [_i32, _i32, _i32]
    L.gsx_l1_ssim_fwd.argtypes = [_p, _p, _i32, _i32, _i32, C.c_float, _p, _p, _sz, _p]
    L.gsx_l1_ssim_bwd.argtypes = [_p, _p, _i32, _i32, _i32, C.c_float, _p, _p, _sz, _p, _p]
    for name in SYMBOLS:
        getattr(L, name)
        if name not in ("gsx_strerror", "gsx_last_hip_error", "gsx_abi_version") and not name.endswith("_bytes") and name not in ("gsx_max_chunks", "gsx_view_layout", "gsx_chunk_state_floats"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


# The three torch hooks the binding needs, behind module-level names so that host-side tests can replace them (no GPU
# in the CPU suite).  All are raw-handle getters: ~20x cheaper than torch.cuda.current_stream() / torch.cuda.device().
def _cur_device():
    import torch
    return torch._C._cuda_getDevice()


def _exchange_device(index):
    """make `index` the current device -> the previous one"""
    import torch
    return torch._C._cuda_exchangeDevice(index)


def _raw_stream(index):
    import torch
    return torch._C._cuda_getCurrentRawStream(index)


def stream_ptr(device=None):
    """torch's current HIP stream ON `device` (a torch.device or an index; default: the current device) as an integer
    handle."""
    if device is None:
        index = _cur_device()
    else:
        index = device if isinstance(device, int) else device.index
        if index is None:
            index = _cur_device()
    return _raw_stream(index)


class on_device:
    """`with on_device(t.device):` -- the library launches on the CURRENT device's stream and never calls hipSetDevice, and
    callers of the reference never call torch.cuda.set_device (scripts/train.py:76,132 only build torch.device(f'cuda:{rank}')
    and `.to(device)`).  Every entry point of this package that reaches the library therefore derives the device from its
    tensors and makes it current for the duration of the call; nothing happens (one getter call) when it already is."""
    __slots__ = ("index", "prev")

    def __init__(self, device):
        self.index = device if isinstance(device, int) else device.index
        self.prev = -1

    def __enter__(self):
        if self.index is not None and _cur_device() != self.index:
            self.prev = _exchange_device(self.index)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            _exchange_device(self.prev)
            self.prev = -1
        return False


def check(status, where):
    if status != GSX_OK:
        raise GsxError(status, where)
