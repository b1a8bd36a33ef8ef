// gsx_view.hip -- one view's forward as ONE host call over ONE caller-owned arena.
//
// The stage entry points (gsx_project_fwd ... gsx_composite_fwd) each take their buffers separately; a Python caller
// then pays an allocator call per buffer and an FFI call per stage -- about thirty of the former and seven of the
// latter per view, ~0.2 ms of host time, which is what bounds a small scene (config 2: 0.31 ms of kernels).  Here the
// caller allocates one block, gsx_view_layout says where each buffer lives in it, and gsx_view_forward enqueues the
// same launches the stage entry points would, in the same order, on the same stream.  The block starts with what the
// backward reads later; behind that, the forward's scratch (sort ping-pong buffers, packed rows, workspaces) and the
// backward's partial rows SHARE one region -- the scratch is dead once the forward's launches have run, the rows are
// written only by the backward -- so a view costs one allocation of a size that repeats from step to step and nothing
// is held twice (at config-5 scale the rows are ~90 GB and the scratch ~30 GB).
// Host code only: no kernels of its own, so results are the stage entry points' results bit for bit.
#include <hip/hip_runtime.h>
#include "gsx_common.h"

namespace {

inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

struct Layout {
    int64_t off[GSX_VIEW_SLOTS];
    int64_t max_chunks;
    int64_t bytes;
};

int make_layout(int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t chunk_force, int32_t need_bwd, Layout& L)
{
    if (N <= 0 || H <= 0 || W <= 0 || capacity <= 0) return GSX_E_INVALID;
    if (H > 65535 || W > 65535 || capacity >= 0xFFFFFFFFll) return GSX_E_RANGE;
    const int64_t n = N, tiles = (int64_t)((W + GSX_TILE - 1) / GSX_TILE) * ((H + GSX_TILE - 1) / GSX_TILE);
    L.max_chunks = gsx_max_chunks(capacity, H, W, chunk_force);
    if (L.max_chunks < 0) return (int)L.max_chunks;
    const bool boxed = (H > W ? H : W) <= 256 * GSX_TILE;
    int64_t at = 0;
    auto put = [&](int slot, int64_t bytes) { L.off[slot] = at; at += align256(bytes); };
    // kept until the backward
    put(GSX_VIEW_ORDER, 4 * n);
    put(GSX_VIEW_RANK_OF, 4 * n);
    put(GSX_VIEW_REC, 4 * GSX_REC_FLOATS * n);
    put(GSX_VIEW_TILES_TOUCHED, 4 * n);
    put(GSX_VIEW_OFFSETS, 4 * n);
    put(GSX_VIEW_TILES_SORTED, 4 * capacity);
    put(GSX_VIEW_RANKS_SORTED, 4 * capacity);
    put(GSX_VIEW_RANGES, 8 * tiles);
    put(GSX_VIEW_CHUNK_BASE, 4 * (tiles + 2));
    put(GSX_VIEW_CHUNK_STATE, 4 * (int64_t)GSX_CHUNK_STATE_FLOATS(L.max_chunks > 0 ? L.max_chunks : 1));
    put(GSX_VIEW_TOTAL, 16);
    // the shared tail: the backward's partial rows (one row of 9 floats per instance of capacity) ...
    const int64_t tail = at;
    L.off[GSX_VIEW_PARTIALS] = need_bwd ? tail : -1;
    const int64_t partial_bytes = need_bwd ? align256(36 * capacity) : 0;
    // ... over the forward's scratch, dead once its launches have run
    put(GSX_VIEW_DEPTH_KEYS, 4 * n);
    put(GSX_VIEW_PRE, 4 * GSX_REC_FLOATS * n);
    put(GSX_VIEW_KEYS_OUT, 4 * n);
    put(GSX_VIEW_KEYS_TMP, 4 * n);
    put(GSX_VIEW_VALUES_TMP, 4 * n);
    put(GSX_VIEW_SORT_WS, (int64_t)gsx_sort_workspace_bytes(n));
    put(GSX_VIEW_TT_BLOCK_SUMS, 8 * ((n + 255) / 256));
    put(GSX_VIEW_TILE_BOX, boxed ? 4 * n : 0);
    put(GSX_VIEW_SCAN_WS, (int64_t)gsx_scan_workspace_bytes(n));
    put(GSX_VIEW_TILES_TMP, 4 * capacity);
    put(GSX_VIEW_RANKS_TMP, 4 * capacity);
    put(GSX_VIEW_BIN_WS, (int64_t)gsx_bin_workspace_bytes(N, capacity));
    if (!boxed) L.off[GSX_VIEW_TILE_BOX] = -1;
    L.bytes = tail + (at - tail > partial_bytes ? at - tail : partial_bytes);
    return GSX_OK;
}

}  // namespace

GSX_API int64_t gsx_view_layout(int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t chunk_force, int32_t need_bwd,
                                int64_t* offsets_host, int64_t* max_chunks_host)
{
    Layout L;
    const int st = make_layout(N, H, W, capacity, chunk_force, need_bwd, L);
    if (st != GSX_OK) return st;
    if (offsets_host) for (int i = 0; i < GSX_VIEW_SLOTS; i++) offsets_host[i] = L.off[i];
    if (max_chunks_host) *max_chunks_host = L.max_chunks;
    return L.bytes;
}

GSX_API int gsx_view_forward(const float* positions, const float* scales, const float* rotations,
                             const float* opacity, const float* features, const float* pose,
                             int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t depth_bits,
                             const uint32_t* given_order, int32_t need_bwd, int32_t chunk_force,
                             void* arena, int64_t arena_bytes,
                             float* mean2d, float* depth, float* cov2d, float* rgb, float* alpha, float* depth_map,
                             uint32_t* state, int64_t* D_pinned_host, void* count_event, void* stream)
{
    if (!positions || !scales || !rotations || !opacity || !features || !pose || !arena || !mean2d || !depth || !cov2d || !rgb || !alpha || !state)
        return GSX_E_INVALID;
    if (depth_bits <= 0 || depth_bits > 32 || ((uintptr_t)arena & 255u)) return GSX_E_INVALID;
    Layout L;
    int st = make_layout(N, H, W, capacity, chunk_force, need_bwd, L);
    if (st != GSX_OK) return st;
    if (arena_bytes < L.bytes) return GSX_E_CAPACITY;
    auto at = [&](int slot) { return static_cast<char*>(arena) + L.off[slot]; };
    auto u32 = [&](int slot) { return L.off[slot] < 0 ? nullptr : reinterpret_cast<uint32_t*>(at(slot)); };
    auto f32 = [&](int slot) { return reinterpret_cast<float*>(at(slot)); };
    auto u64 = [&](int slot) { return reinterpret_cast<uint64_t*>(at(slot)); };
    auto raw = [&](int slot) { return static_cast<void*>(at(slot)); };
    uint64_t* total = u64(GSX_VIEW_TOTAL);

    st = gsx_project_fwd(positions, scales, rotations, opacity, features, pose, N, mean2d, depth, cov2d,
                         u32(GSX_VIEW_DEPTH_KEYS), f32(GSX_VIEW_PRE), stream);
    if (st != GSX_OK) return st;
    const uint32_t* order = given_order;
    if (!order) {
        // descending depth, ties by index (torch.argsort(depths, descending=True), gaussian_splatting.py:198); the count of
        // keys whose bits [24, 32) differ from the first key's lands in total[1] and travels to the host with D
        st = gsx_sort_pairs_u32(u32(GSX_VIEW_DEPTH_KEYS), nullptr, u32(GSX_VIEW_KEYS_OUT), u32(GSX_VIEW_ORDER),
                                u32(GSX_VIEW_KEYS_TMP), u32(GSX_VIEW_VALUES_TMP), N, 0, depth_bits, GSX_DEPTH_CHECK_BIT, total + 1,
                                raw(GSX_VIEW_SORT_WS), gsx_sort_workspace_bytes(N), stream);
        if (st != GSX_OK) return st;
        order = u32(GSX_VIEW_ORDER);
    }
    st = gsx_build_records(f32(GSX_VIEW_PRE), order, N, H, W, f32(GSX_VIEW_REC), u32(GSX_VIEW_TILES_TOUCHED), u32(GSX_VIEW_RANK_OF),
                           u64(GSX_VIEW_TT_BLOCK_SUMS), u32(GSX_VIEW_TILE_BOX), stream);
    if (st != GSX_OK) return st;
    st = gsx_bin_count(u32(GSX_VIEW_TILES_TOUCHED), u64(GSX_VIEW_TT_BLOCK_SUMS), N, u32(GSX_VIEW_OFFSETS), total, D_pinned_host,
                       raw(GSX_VIEW_SCAN_WS), gsx_scan_workspace_bytes(N), stream);
    if (st != GSX_OK) return st;
    // the host may read D (and the sort's mismatch count) as soon as THIS point of the stream is reached, long before the
    // compositing enqueued below has run
    if (count_event) GSX_HIP(hipEventRecord(static_cast<hipEvent_t>(count_event), static_cast<hipStream_t>(stream)));
    st = gsx_bin_emit_sort(f32(GSX_VIEW_REC), u32(GSX_VIEW_TILE_BOX), u32(GSX_VIEW_OFFSETS), N, H, W, capacity, total,
                           u32(GSX_VIEW_TILES_SORTED), u32(GSX_VIEW_RANKS_SORTED), u32(GSX_VIEW_TILES_TMP), u32(GSX_VIEW_RANKS_TMP),
                           u32(GSX_VIEW_RANGES), u32(GSX_VIEW_CHUNK_BASE), chunk_force,
                           raw(GSX_VIEW_BIN_WS), gsx_bin_workspace_bytes(N, capacity), stream);
    if (st != GSX_OK) return st;
    return gsx_composite_fwd(f32(GSX_VIEW_REC), u32(GSX_VIEW_RANKS_SORTED), u32(GSX_VIEW_RANGES), u32(GSX_VIEW_CHUNK_BASE), N, H, W,
                             L.max_chunks, f32(GSX_VIEW_CHUNK_STATE), need_bwd ? 1 : 0, state,
                             rgb, alpha, depth_map, stream);
}
