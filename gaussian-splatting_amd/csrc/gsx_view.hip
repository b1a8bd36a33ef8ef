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

// late = 0: the public call.  late = 1 (the one-call trainer step): D_pinned_host has FOUR words, the third receives 1 + the number of
// empty tiles from the tile sort (0 stays for "not known": images of more than 4096 tiles), the event is recorded behind the tile
// sort instead of behind the count, and the global-exit path's three launches are left to the caller (composite_fwd_impl, which = 2),
// who knows by then whether the view can saturate at all.
static int view_forward_impl(const float* positions, const float* scales, const float* rotations,
                             const float* opacity, const float* features, const float* pose,
                             int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t depth_bits,
                             const uint32_t* given_order, int32_t need_bwd, int32_t chunk_force,
                             void* arena, int64_t arena_bytes,
                             float* mean2d, float* depth, float* cov2d, float* rgb, float* alpha, float* depth_map,
                             uint32_t* state, int64_t* D_pinned_host, void* count_event, int late, void* stream);

GSX_API int gsx_view_forward(const float* positions, const float* scales, const float* rotations,
                             const float* opacity, const float* features, const float* pose,
                             int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t depth_bits,
                             const uint32_t* given_order, int32_t need_bwd, int32_t chunk_force,
                             void* arena, int64_t arena_bytes,
                             float* mean2d, float* depth, float* cov2d, float* rgb, float* alpha, float* depth_map,
                             uint32_t* state, int64_t* D_pinned_host, void* count_event, void* stream)
{
    return view_forward_impl(positions, scales, rotations, opacity, features, pose, N, H, W, capacity, depth_bits, given_order, need_bwd,
                             chunk_force, arena, arena_bytes, mean2d, depth, cov2d, rgb, alpha, depth_map, state, D_pinned_host, count_event,
                             0, stream);
}

// The one-call trainer step marks the Gaussians whose footprint misses the image while it projects them (one bit each): the records
// stage then neither gathers their packed rows nor writes their records nor their rank_of entries (a random 4-byte store each), and
// the backward asks the bitmap before it looks up a rank.  Nothing reads the record of a splat without instances; the public stage
// calls and gsx_view_forward keep writing every rank's.  The bitmap lives where the sorted tile ids would (which the one-call step
// does not write, gsx.h): images of up to 8192 tiles (the one-pass tile sorts; they also have the packed tile boxes, so that the
// emit pass does not need the records' boxes), and a block whose capacity leaves room for N bits there.
static uint64_t* step_vis(const Layout& L, char* arena, int32_t N, int32_t H, int32_t W, int64_t capacity)
{
    const int64_t tiles = (int64_t)((W + GSX_TILE - 1) / GSX_TILE) * ((H + GSX_TILE - 1) / GSX_TILE);
    if (N <= 0 || tiles > 8192 || L.off[GSX_VIEW_TILE_BOX] < 0 || 4 * capacity < 8 * (((int64_t)N + 63) / 64)) return nullptr;
    return reinterpret_cast<uint64_t*>(arena + L.off[GSX_VIEW_TILES_SORTED]);
}

static int view_forward_impl(const float* positions, const float* scales, const float* rotations,
                             const float* opacity, const float* features, const float* pose,
                             int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t depth_bits,
                             const uint32_t* given_order, int32_t need_bwd, int32_t chunk_force,
                             void* arena, int64_t arena_bytes,
                             float* mean2d, float* depth, float* cov2d, float* rgb, float* alpha, float* depth_map,
                             uint32_t* state, int64_t* D_pinned_host, void* count_event, int late, void* stream)
{
    if (!positions || !scales || !rotations || !opacity || !features || !pose || !arena || !rgb || !alpha || !state)
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

    uint64_t* vis = late ? step_vis(L, static_cast<char*>(arena), N, H, W, capacity) : nullptr;
    st = gsx::project_fwd_impl(positions, scales, rotations, opacity, features, pose, N, mean2d, depth, cov2d,
                               u32(GSX_VIEW_DEPTH_KEYS), f32(GSX_VIEW_PRE), H, W, vis, stream);
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
    st = gsx::build_records_impl(f32(GSX_VIEW_PRE), order, N, H, W, f32(GSX_VIEW_REC), u32(GSX_VIEW_TILES_TOUCHED), u32(GSX_VIEW_RANK_OF),
                                 u64(GSX_VIEW_TT_BLOCK_SUMS), u32(GSX_VIEW_TILE_BOX), vis, stream);
    if (st != GSX_OK) return st;
    // The scan of the tile counts rides in the emit kernel (one dependent launch less) when the instance count can reach the host
    // from that kernel: pinned words the device can address (what torch's pin_memory() gives), or nobody asking for it.
    uint64_t* pinned_dev = nullptr;
    if (D_pinned_host && N > 0) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, D_pinned_host, 0) == hipSuccess) pinned_dev = static_cast<uint64_t*>(dp);
        else (void)hipGetLastError();
    }
    const bool scan_in_emit = N > 0 && (pinned_dev || !D_pinned_host);
    if (!scan_in_emit) {
        st = gsx_bin_count(u32(GSX_VIEW_TILES_TOUCHED), u64(GSX_VIEW_TT_BLOCK_SUMS), N, u32(GSX_VIEW_OFFSETS), total, D_pinned_host,
                           raw(GSX_VIEW_SCAN_WS), gsx_scan_workspace_bytes(N), stream);
        if (st != GSX_OK) return st;
        // the host may read D (and the sort's mismatch count) as soon as THIS point of the stream is reached, long before the
        // compositing enqueued below has run
        if (count_event && !late) GSX_HIP(hipEventRecord(static_cast<hipEvent_t>(count_event), static_cast<hipStream_t>(stream)));
    }
    uint64_t* info = (late && pinned_dev) ? pinned_dev + 2 : nullptr;
    const gsx::EmitScan scan = { u64(GSX_VIEW_TT_BLOCK_SUMS), u32(GSX_VIEW_OFFSETS), total, pinned_dev, late ? 1 : 0 };      // (the one-call step: nobody looks at the sorted tile ids; half the scatter's stores)
    st = gsx::bin_emit_sort_impl(f32(GSX_VIEW_REC), u32(GSX_VIEW_TILE_BOX), u32(GSX_VIEW_OFFSETS), N, H, W, capacity, total,
                                 u32(GSX_VIEW_TILES_SORTED), u32(GSX_VIEW_RANKS_SORTED), u32(GSX_VIEW_TILES_TMP), u32(GSX_VIEW_RANKS_TMP),
                                 u32(GSX_VIEW_RANGES), u32(GSX_VIEW_CHUNK_BASE), chunk_force,
                                 raw(GSX_VIEW_BIN_WS), gsx_bin_workspace_bytes(N, capacity), info, stream, scan_in_emit ? &scan : nullptr);
    if (st != GSX_OK) return st;
    // (with the scan in the emit kernel the count is on its way from the binning stage's first kernel: the event stands behind the
    // stage, three short kernels later than it used to)
    if (count_event && (late || scan_in_emit)) GSX_HIP(hipEventRecord(static_cast<hipEvent_t>(count_event), static_cast<hipStream_t>(stream)));
    return gsx::composite_fwd_impl(f32(GSX_VIEW_REC), u32(GSX_VIEW_RANKS_SORTED), u32(GSX_VIEW_RANGES), u32(GSX_VIEW_CHUNK_BASE), N, H, W,
                                   L.max_chunks, f32(GSX_VIEW_CHUNK_STATE), need_bwd ? 1 : 0, state,
                                   rgb, alpha, depth_map, late ? 1 : 3, stream);
}

// One trainer step of one view: the view's forward above, the criterion, the two backward stages and the fused
// clip + Adam, enqueued back to back.  The one place the host looks at the GPU is the count event, between the criterion
// and the backward: the GPU is busy with the forward's second half meanwhile, and a step whose buffers were too small (or
// whose depth sort was too narrow) reaches neither the backward nor the parameters.
static int check_step(const gsx_step* s, bool grads, bool apply)
{
    if (!s) return GSX_E_INVALID;
    if (grads) {
        if (!s->pose || !s->target || !s->d_rgb || !s->loss || !s->loss_ws || !s->counts_pinned_host || !s->count_event) return GSX_E_INVALID;
        if (s->loss_kind < GSX_LOSS_L1 || s->loss_kind > GSX_LOSS_L1_SSIM) return GSX_E_INVALID;
    }
    if (apply && s->step < 1) return GSX_E_INVALID;
    uint32_t seen = 0;
    for (int k = 0; k < 5; k++) {
        if (!s->param[k] || !s->grad[k]) return GSX_E_INVALID;
        if (apply && (!s->exp_avg[k] || !s->exp_avg_sq[k])) return GSX_E_INVALID;
        if (s->adam_order[k] < 0 || s->adam_order[k] > 4) return GSX_E_INVALID;
        seen |= 1u << s->adam_order[k];
    }
    return seen == 31u ? GSX_OK : GSX_E_INVALID;
}

// The step in its two halves, for data-parallel ranks: everything up to the gradients (the flat buffer the caller hands
// in through grad[]), then -- after the caller's one collective on the same stream -- clip + Adam.  gsx_train_step is
// the two back to back.
// where the one-call step keeps the clip norm's partial sums (one double per workgroup of preprocess_bwd) when the caller's Adam
// workspace has room for them behind what gsx_clip_adam_step uses; nullptr: the optimizer half computes the norm itself
static double* folded_norm_partials(const gsx_step* s)
{
    const size_t base = gsx_clip_adam_workspace_bytes();
    const size_t need = base + (size_t)gsx::ceil_div(s->N, gsx::kBlock) * sizeof(double);
    if (!(s->max_grad_norm > 0.0f) || !s->adam_ws || s->adam_ws_bytes < need) return nullptr;
    return reinterpret_cast<double*>(static_cast<char*>(s->adam_ws) + base);
}

static int train_step_grads_impl(const gsx_step* s, double* norm_partials, void* stream)
{
    int st = check_step(s, true, false);
    if (st != GSX_OK) return st;
    Layout L;
    st = make_layout(s->N, s->H, s->W, s->capacity, s->chunk_force, 1, L);
    if (st != GSX_OK) return st;
    s->counts_pinned_host[2] = 0;                 // "number of empty tiles not known" until the tile sort says otherwise
    st = view_forward_impl(s->param[0], s->param[1], s->param[2], s->param[3], s->param[4], s->pose, s->N, s->H, s->W, s->capacity,
                           s->depth_bits, nullptr, 1, s->chunk_force, s->arena, s->arena_bytes, s->mean2d, s->depth, s->cov2d,
                           s->rgb, s->alpha, nullptr, s->state, s->counts_pinned_host, s->count_event, 1, stream);
    if (st != GSX_OK) return st;
    // The verdict on this step's optimism, BEFORE anything behind the view's compositing is enqueued: a forward whose buffers were
    // too small is truncated but in bounds, while the backward would follow offsets[] (which count every instance) past the end of
    // the partial rows.  The event sits behind the tile sort: the GPU is compositing the view when the host gets here, and has
    // the chunk pass and the combine pass in front of it while the host enqueues the rest of the step.
    GSX_HIP(hipEventSynchronize(static_cast<hipEvent_t>(s->count_event)));
    const int64_t D = s->counts_pinned_host[0], mismatch = s->counts_pinned_host[1], tiles_info = s->counts_pinned_host[2];
    if (D >= 0xFFFFFFFFll) return GSX_E_RANGE;
    if (D > s->capacity) return GSX_E_CAPACITY;
    if (s->depth_bits < 32 && mismatch != 0) return GSX_E_AGAIN;
    char* arena = static_cast<char*>(s->arena);
    auto u32 = [&](int slot) { return reinterpret_cast<const uint32_t*>(arena + L.off[slot]); };
    const float* rec = reinterpret_cast<const float*>(arena + L.off[GSX_VIEW_REC]);
    // the reference's global early exit (all pixels saturated at once) cannot fire on a view that leaves a tile empty: the exit
    // path's three launches -- no-ops then, 4.5 us of launch floor each -- are only enqueued when no tile is known to be empty
    // (shallow views only: theirs is the combine pass that writes "no exit rank" next to the empty tiles' veto)
    const int64_t n_tiles = (int64_t)((s->W + GSX_TILE - 1) / GSX_TILE) * ((s->H + GSX_TILE - 1) / GSX_TILE);
    const bool shallow = D / (int64_t)gsx::chunk_size_for((uint64_t)D, s->chunk_force) + n_tiles + 1 < (int64_t)GSX_DEEP_CHUNKS;
    if (tiles_info <= 1 || !shallow) {
        st = gsx::composite_fwd_impl(rec, u32(GSX_VIEW_RANKS_SORTED), u32(GSX_VIEW_RANGES), u32(GSX_VIEW_CHUNK_BASE), s->N, s->H, s->W,
                                     L.max_chunks, reinterpret_cast<float*>(arena + L.off[GSX_VIEW_CHUNK_STATE]), 1, s->state,
                                     s->rgb, s->alpha, nullptr, 2, stream);
        if (st != GSX_OK) return st;
    }
    const int64_t px = 3ll * s->H * s->W;
    const bool loss_in_bwd = s->loss_kind != GSX_LOSS_L1_SSIM && s->N > 0 && L.max_chunks > 0 &&
                             s->loss_ws_bytes >= (size_t)gsx::kLossBlocks * 8;
    if (s->loss_kind == GSX_LOSS_L1_SSIM) {
        st = gsx_l1_ssim_fwd(s->rgb, s->target, 3, s->H, s->W, s->lambda_dssim, s->loss, s->loss_ws, s->loss_ws_bytes, stream);
        if (st != GSX_OK) return st;
        st = gsx_l1_ssim_bwd(s->rgb, s->target, 3, s->H, s->W, s->lambda_dssim, nullptr, s->loss_ws, s->loss_ws_bytes, s->d_rgb, stream);
    } else if (!loss_in_bwd) {
        // (the loss VALUE is reduced by the last kernel of this half: preprocess_bwd_with_loss)
        st = gsx::image_loss_fwd_grad_partials(s->loss_kind, s->rgb, s->target, px, s->d_rgb, s->loss_ws, s->loss_ws_bytes, stream);
    }
    if (st != GSX_OK) return st;
    float* partials = reinterpret_cast<float*>(arena + L.off[GSX_VIEW_PARTIALS]);
    // L1 / clamped MSE: no launch of its own for the criterion -- the backward forms dL/d(rgb) from the image and the target itself and
    // its grid's last workgroups leave the loss's partial sums (the same slices, the same bits: gsx_composite.hip, loss_slice_wave)
    const gsx::LossInBwd lw = { s->rgb, s->target, static_cast<double*>(s->loss_ws), (long long)px, s->loss_kind,
                                ((((uintptr_t)s->rgb | (uintptr_t)s->target) & 15u) == 0) ? 1 : 0, 1.0f / (float)px };
    st = gsx::composite_bwd_impl(rec, u32(GSX_VIEW_RANKS_SORTED), u32(GSX_VIEW_RANGES), u32(GSX_VIEW_CHUNK_BASE), u32(GSX_VIEW_OFFSETS),
                                 s->N, s->H, s->W, L.max_chunks, reinterpret_cast<const float*>(arena + L.off[GSX_VIEW_CHUNK_STATE]), s->state,
                                 s->d_rgb, partials, D, loss_in_bwd ? &lw : nullptr, stream);
    if (st != GSX_OK) return st;
    const bool fold_loss = s->loss_kind != GSX_LOSS_L1_SSIM;
    return gsx::preprocess_bwd_with_loss(s->param[0], s->param[1], s->param[2], s->pose, u32(GSX_VIEW_RANK_OF), u32(GSX_VIEW_OFFSETS),
                                    u32(GSX_VIEW_TILES_TOUCHED), partials, s->state, s->N,
                                    s->grad[0], s->grad[1], s->grad[2], s->grad[3], s->grad[4],
                                    fold_loss ? static_cast<const double*>(s->loss_ws) : nullptr, px, fold_loss ? s->loss : nullptr,
                                    norm_partials, step_vis(L, arena, s->N, s->H, s->W, s->capacity), stream);
}

GSX_API int gsx_train_step_grads(const gsx_step* s, void* stream) { return train_step_grads_impl(s, nullptr, stream); }

static int train_step_apply_impl(const gsx_step* s, const double* norm_partials, void* stream);

GSX_API int gsx_train_step_apply(const gsx_step* s, void* stream) { return train_step_apply_impl(s, nullptr, stream); }

static int train_step_apply_impl(const gsx_step* s, const double* norm_partials, void* stream)
{
    const int st = check_step(s, false, true);
    if (st != GSX_OK) return st;
    if (s->N <= 0) return GSX_E_INVALID;
    float* p[5]; float* g[5]; float* m[5]; float* v[5]; int64_t numel[5]; float lr[5];
    static const int64_t per_row[5] = {3, 3, 4, 1, 3};
    for (int k = 0; k < 5; k++) {
        const int t = s->adam_order[k];
        p[k] = s->param[t]; g[k] = s->grad[t]; m[k] = s->exp_avg[t]; v[k] = s->exp_avg_sq[t]; lr[k] = s->lr[t];
        numel[k] = per_row[t] * s->N;
    }
    return gsx::clip_adam_step_impl(5, p, g, m, v, numel, lr, s->beta1, s->beta2, s->eps, s->step, s->max_grad_norm, s->write_clipped_grad,
                                    s->adam_ws, s->adam_ws_bytes, norm_partials, norm_partials ? (int32_t)gsx::ceil_div(s->N, gsx::kBlock) : 0, stream);
}

GSX_API int gsx_train_step(const gsx_step* s, void* stream)
{
    int st = check_step(s, true, true);
    if (st != GSX_OK) return st;
    // nothing stands between the two halves here (no exchange): the clip norm's sums of squares ride on the last backward kernel
    double* norm_partials = folded_norm_partials(s);
    st = train_step_grads_impl(s, norm_partials, stream);
    if (st != GSX_OK) return st;
    return train_step_apply_impl(s, norm_partials, stream);
}
