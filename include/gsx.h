/* gsx.h -- C ABI of libgsplat_gfx950.so: the MI355X (gfx950) Gaussian-splat render path.
 *
 * Drop-in boundary for ONE hot path of ChristianLin0420/gaussian-splatting:
 * GaussianSplat.forward and its (implicit, autograd) backward.  The reference has no FFI for
 * this path -- it is ~25 ATen calls per Gaussian issued from Python
 * (/root/reference/model/gaussian_splatting.py:70-344) -- so each entry point below cites the
 * reference lines whose work it replaces.  The Python side that binds these with ctypes is
 * gaussian-splatting_amd/_lib.py; INTEGRATION.md shows the stub a maintainer of the reference adds.
 *
 * Conventions (all entry points unless an entry point's own comment says otherwise):
 *   - extern "C", plain pointers and sizes, no torch types; return 0 (GSX_OK) or a negative GSX_E_*.
 *   - every pointer is a DEVICE pointer owned by the caller unless its name ends in _host;
 *     nothing here allocates or frees device memory.
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and is asynchronous.  Three entry points make the HOST
 *     wait: gsx_bin_instances synchronises the stream once (8-byte read-back of the instance count); gsx_train_step and
 *     gsx_train_step_grads block on the caller's count_event (recorded in the middle of the view's forward, behind the
 *     tile sort: the host learns the instance count and the number of empty tiles while the GPU composites) before they
 *     enqueue the criterion and the backward; the gsx_comm_* calls are
 *     RCCL's and block as RCCL does (gsx_comm_init until every rank has joined).
 *   - fp32 arithmetic; integer products (sort keys, tile rectangles, tile ranges) are bit-exact
 *     against oracle/gsx_oracle.c.
 *   - re-entrant, one host thread per stream.  Global state, all of it: the thread-local last HIP error
 *     (gsx_last_hip_error), the handle of librccl once gsx_comm_* has opened it, and the device array of
 *     gsx_sort_pairs_u32_static with its per-device address table (calls of THAT entry point that overlap on one GPU race).
 */
#ifndef GSX_H
#define GSX_H
#include <stddef.h>
#include <stdint.h>
#include "gsx_constants.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GSX_OK 0
#define GSX_E_INVALID (-1)     /* bad argument (null pointer, non-positive size, misaligned) */
#define GSX_E_CAPACITY (-2)    /* caller's instance buffers / workspace too small; see out parameter */
#define GSX_E_HIP (-3)         /* a HIP runtime call failed; gsx_last_hip_error() has the code */
#define GSX_E_RANGE (-4)       /* problem exceeds 32-bit instance indexing */
#define GSX_E_COMM (-6)        /* RCCL could not be loaded, or one of its calls failed */
#define GSX_E_AGAIN (-5)       /* an optimistic assumption did not hold (gsx_train_step: depth keys not constant above bit 24);
                                  nothing was applied, call again as the entry point's comment says */

const char* gsx_strerror(int status);
int gsx_last_hip_error(void);
/* ABI version; bumped on any signature change */
#define GSX_ABI_VERSION 20
int gsx_abi_version(void);

/* ---- a5-a8 + sort key: replaces _project_positions, _project_scales,
 * _quaternion_to_rotation_matrix, _compute_jacobian (gaussian_splatting.py:221-344).
 * One thread per Gaussian.  Outputs are the reference's returned tensors for ONE view:
 *   mean2d (N,2) = 'positions', depth (N) = 'depths', cov2d (N,2,2) = 'scales',
 * plus depth_key (N) u32: ascending-unsigned order == descending depth (the order
 * torch.argsort(depths, descending=True) produces, :198), -0 folded onto +0.
 * mean2d / depth / cov2d may be NULL, all three together, for a caller that does not need the returned tensors (a trainer step
 * whose criterion looks at the image only): 28 of the kernel's 120 bytes per Gaussian less.
 * pre (N x 12 floats): packed copy (mean2d, cov2d, depth, opacity, colour) that gsx_build_records gathers
 * as one 48-byte row per Gaussian. */
int gsx_project_fwd(const float* positions, const float* scales, const float* rotations,
                    const float* opacity, const float* features,
                    const float* pose /* 16 floats, row-major 4x4 */, int32_t N,
                    float* mean2d, float* depth, float* cov2d, uint32_t* depth_key, float* pre, void* stream);

/* ---- device LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit),
 * stable.  values_in == NULL means values are 0..n-1.  Result lands in keys_out/values_out;
 * keys_tmp/values_tmp are ping-pong buffers of n elements.  ws: gsx_sort_workspace_bytes(n).
 * Replaces torch.argsort (:198) [keys = depth_key, 32 bits] and is reused to order the
 * (tile, rank) instances by tile [keys = tile id].
 * mismatch_dev (nullable, device u64): receives the number of keys whose bits [check_bit, 32) differ from keys_in[0]'s
 * (0 for check_bit == 32).  A caller that sorted on fewer than 32 bits because the high bits were constant the last
 * time (depth keys of a scene whose depths stay inside one binade or two: a quarter of the passes) reads it later to
 * learn whether that still held, and sorts again on all bits if it did not; a full sort uses it to learn whether the
 * next one may skip them. */
size_t gsx_sort_workspace_bytes(int64_t n);
int gsx_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* values_in,
                       uint32_t* keys_out, uint32_t* values_out,
                       uint32_t* keys_tmp, uint32_t* values_tmp,
                       int64_t n, int begin_bit, int end_bit, int check_bit, uint64_t* mismatch_dev,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- gsx_sort_pairs_u32 for callers that cannot allocate or size a workspace -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  NO counterpart in the reference (nor in its FFI: it has none).  Same
 * kernels, same result bit for bit; the ping-pong buffers and the histogram workspace live in a device array inside the
 * library (about 8.6 MB per GPU the library runs on, held until the process ends).  Limits and caveats:
 *   - n <= GSX_STATIC_SORT_MAX_N, else GSX_E_CAPACITY (use gsx_sort_pairs_u32 with a workspace);
 *   - the device that `stream` belongs to must be the CURRENT device of the calling thread;
 *   - the first call on a device looks the array's address up (hipGetSymbolAddress: no allocation, but make that call
 *     outside a graph capture); no call allocates or frees device memory;
 *   - all calls on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race. */
#define GSX_STATIC_SORT_MAX_N (1 << 20)
int gsx_sort_pairs_u32_static(const uint32_t* keys_in, const uint32_t* values_in,
                              uint32_t* keys_out, uint32_t* values_out,
                              int64_t n, int begin_bit, int end_bit, void* stream);

/* ---- exclusive prefix sum of n u32; total (u64) written to *total_dev.  ws: gsx_scan_workspace_bytes(n). */
size_t gsx_scan_workspace_bytes(int64_t n);
int gsx_scan_u32(const uint32_t* in, uint32_t* out_exclusive, uint64_t* total_dev, int64_t n,
                 void* ws, size_t ws_bytes, void* stream);

/* ---- rank-ordered splat records: replaces the per-Gaussian gathers and torch.inverse of
 * _render_chunk (:110-115, :128-130) and adds the footprint rectangle (no counterpart in the
 * reference: it evaluates every Gaussian at every pixel).  order[r] = Gaussian composited r-th.
 * pre: from gsx_project_fwd.  rec: N x GSX_REC_FLOATS floats (layout: gsx_constants.h);
 * tiles_touched: N (rank order); rank_of: N, inverse of order (rank_of[order[r]] = r).
 * tt_block_sums (nullable): ceil(N / 256) u64, sums of tiles_touched over consecutive groups of 256 ranks; handed to
 * gsx_bin_count they save its reduction launch.
 * tile_box (nullable): N u32, the footprint in tile units (x0 | y0 << 8 | x1 << 16 | y1 << 24, inclusive; x0 > x1 = none),
 * written for images of up to 256 tiles a side; handed to gsx_bin_emit_sort it replaces the strided read of the records. */
int gsx_build_records(const float* pre, const uint32_t* order, int32_t N, int32_t H, int32_t W,
                      float* rec, uint32_t* tiles_touched, uint32_t* rank_of, uint64_t* tt_block_sums, uint32_t* tile_box,
                      void* stream);

/* ---- binning: scan tiles_touched -> offsets, emit (tile, rank) instances, sort them by tile
 * (stable, so each tile's list stays in rank order == key (tile<<32 | rank) order), cut tile
 * ranges and the chunk table.  Synchronises `stream` once to read the instance count D back.
 *   capacity     : elements available in each of the four instance buffers
 *   offsets      : N u32 (exclusive scan of tiles_touched, rank order)
 *   tiles_sorted / ranks_sorted : results (D valid elements)
 *   tiles_tmp / ranks_tmp       : scratch of the same capacity
 *   ranges       : n_tiles x 2 u32, ranges[t] = [lower_bound(t), lower_bound(t+1)) into the sorted arrays
 *   chunk_base   : n_tiles + 2 u32: exclusive scan of ceil(len(t) / chunk) (entry n_tiles = chunk count), then the chunk
 *                  size itself (64 for D < 2^19, 128 for D < 2^20, else GSX_CHUNK = 256: a small view cut into long chunks
 *                  would leave most of the chip idle); chunk_force = 64 / 128 / 256 overrides the choice, 0 = automatic
 *   *D_host      : instance count (also on GSX_E_CAPACITY, so the caller can grow and retry)
 * ws: gsx_bin_workspace_bytes(N, capacity). */
size_t gsx_bin_workspace_bytes(int32_t N, int64_t capacity);
int gsx_bin_instances(const float* rec, const uint32_t* tiles_touched, int32_t N, int32_t H, int32_t W,
                      int64_t capacity, uint32_t* offsets,
                      uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                      uint32_t* ranges, uint32_t* chunk_base, int32_t chunk_force, int64_t* D_host,
                      void* ws, size_t ws_bytes, void* stream);
/* The same binning in two asynchronous halves, for callers that must not stall the stream in the middle of a
 * forward (the synchronising read-back above idles the GPU for the length of a host round trip).
 * gsx_bin_count: offsets = exclusive scan of tiles_touched, total_dev[0] (device, u64) = instance count D;
 *   total_dev has TWO u64: [1] belongs to the caller (the render path parks the depth sort's mismatch count there).
 *   If D_pinned_host != NULL both words also reach it asynchronously (16 bytes of pinned host memory; valid
 *   once the caller has synchronised on an event recorded after this call): written by the scan kernel itself when the
 *   device can address that memory (hipHostMalloc'ed or registered), copied with hipMemcpyAsync otherwise.  ws: gsx_scan_workspace_bytes(N).
 *   tt_block_sums: from gsx_build_records (the scan is then one launch), or NULL.
 * gsx_bin_emit_sort: everything after the count, with D read from *total_dev ON THE DEVICE and clamped to
 *   `capacity` (which sizes the launches): no host synchronisation.  If D turns out to exceed capacity the
 *   results are truncated (never out of bounds); the caller detects that from the count and calls again with
 *   larger buffers.  ws: gsx_bin_workspace_bytes(N, capacity).  Size chunk_state / partials / launch grids from
 *   capacity (gsx_max_chunks(capacity, H, W, chunk_force)), not from D. */
int gsx_bin_count(const uint32_t* tiles_touched, const uint64_t* tt_block_sums, int32_t N, uint32_t* offsets, uint64_t* total_dev,
                  int64_t* D_pinned_host, void* ws, size_t ws_bytes, void* stream);
int gsx_bin_emit_sort(const float* rec, const uint32_t* tile_box /* from gsx_build_records, or NULL */,
                      const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                      int64_t capacity, const uint64_t* total_dev,
                      uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                      uint32_t* ranges, uint32_t* chunk_base, int32_t chunk_force, void* ws, size_t ws_bytes, void* stream);
/* upper bound on the chunk count of any view of at most D instances (max over the chunk sizes the device may pick, or
 * D / chunk_force) + min(D, n_tiles).  The caller sizes
 * chunk_state with it: GSX_CHUNK_STATE_FLOATS(max_chunks) floats (the tail holds the per-chunk
 * meta words, the backward work lists and a shallow view's mid-chunk states, gsx_constants.h). */
int64_t gsx_max_chunks(int64_t D, int32_t H, int32_t W, int32_t chunk_force);
/* floats of the chunk_state buffer for that bound (GSX_CHUNK_STATE_FLOATS of gsx_constants.h, for callers that cannot
 * evaluate a C macro); negative GSX_E_* when max_chunks is out of range. */
int64_t gsx_chunk_state_floats(int64_t max_chunks);

/* ---- compositing: replaces the inner loop of _render_chunk (:118-148).  A tile's depth-ordered
 * list is cut into chunks of 64 / 128 / GSX_CHUNK instances (chunk_base[n_tiles + 1]); a workgroup of four wave64 composites
 * one chunk for the whole 16x16 tile, a quarter of the chunk per wave, merged in order through LDS (one wave per chunk in a
 * deep view; 4 pixels per lane, splat records staged through LDS); a per-tile pass then combines
 * the chunks (compositing is associative), writes the image and -- with need_bwd -- each chunk's dark-quadrant
 * bytes and the backward's work lists.  Enqueues, in order: (only when max_chunks could hold a deep view, 20480
 * chunks and more: the level-order table and the header pass of the forward occlusion skipping -- launches that return at
 * once for a shallow view --) chunk pass,
 * combine pass, exact early-exit search (only does work when every pixel saturated: the
 * reference's global exit, :147), and a second chunk + combine pass that only does work when the
 * exit fired before the last Gaussian.
 *   state (4 x u32, out): state[0] = exit rank k* (GSX_RANK_NONE: no exit); state[1] bit 0 = some pixel can never have
 *       been saturated (no exit), bit 1 = some tile holds a splat with alpha outside [0, 1] (the exit search then iterates:
 *       the accumulated alpha is not monotone); state[2..3] are the search's scratch.  Ranks 0..k* are composited.
 *   chunk_state: GSX_CHUNK_STATE_FLOATS(max_chunks) floats; with need_bwd = 1 it is left holding
 *       what gsx_composite_bwd needs (per chunk and pixel: colour behind the chunk, transmittance
 *       at its end).
 *   rgb (3,H,W), alpha (H,W) accumulated alpha, depth_map (H,W) or NULL. */
int gsx_composite_fwd(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                      const uint32_t* chunk_base, int32_t N, int32_t H, int32_t W,
                      int64_t max_chunks, float* chunk_state, int32_t need_bwd, uint32_t* state,
                      float* rgb, float* alpha, float* depth_map, void* stream);

/* ---- one view's whole forward in ONE call over ONE caller-owned block: gsx_project_fwd -> gsx_sort_pairs_u32 (depth
 * order) -> gsx_build_records -> gsx_bin_count -> gsx_bin_emit_sort -> gsx_composite_fwd, the same launches in the same
 * order on `stream` (GaussianSplat._forward_impl for one camera, gaussian_splatting.py:188-219).  Host-side convenience
 * only -- no kernels of its own, results identical to the stage calls -- for callers whose per-call overhead matters (a
 * Python caller otherwise pays ~30 allocations and 7 FFI calls per view).
 * gsx_view_layout: bytes of the block (or a negative GSX_E_*) for N Gaussians, an H x W image and instance buffers of
 *   `capacity`; offsets_host (nullable, GSX_VIEW_SLOTS entries, gsx_constants.h) receives each buffer's byte offset (-1:
 *   not present), *max_chunks_host (nullable) the chunk bound compositing is launched with (gsx_max_chunks(capacity, ...)).
 *   With need_bwd the block also holds the backward's partial rows (GSX_VIEW_PARTIALS, `capacity` rows of 9 floats, what
 *   gsx_composite_bwd writes and gsx_preprocess_bwd sums) in the SAME region as the forward's scratch, which is dead by then.
 * gsx_view_forward: arena = that block (256-byte aligned, arena_bytes >= the layout's size for the same arguments).
 *   depth_bits = 32, or GSX_DEPTH_CHECK_BIT if the previous view's mismatch count was 0 (see gsx_sort_pairs_u32);
 *   given_order (nullable) = a depth order to use instead of sorting (views after the first of a batch, :206).  The
 *   instance count D and the mismatch count land in arena[GSX_VIEW_TOTAL] (2 x u64) and, if D_pinned_host != NULL, in
 *   16 bytes of pinned host memory; count_event (nullable hipEvent_t) is recorded behind the binning stage, BEFORE
 *   compositing is enqueued, so the host can learn D while the GPU is still busy with the view (the scan of the tile counts
 *   rides in the binning stage's first kernel, which writes the pinned words itself when the device can address them).  As with
 *   gsx_bin_emit_sort, D > capacity truncates the results (never out of bounds): the caller checks D and calls again
 *   with a larger block.  Buffers GSX_VIEW_ORDER .. GSX_VIEW_TOTAL are what gsx_composite_bwd / gsx_preprocess_bwd read;
 *   `state` (gsx_composite_fwd's, 16 bytes) stays outside the block because callers keep it (the exit rank is an output). */
int64_t gsx_view_layout(int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t chunk_force, int32_t need_bwd,
                        int64_t* offsets_host, int64_t* max_chunks_host);
int gsx_view_forward(const float* positions, const float* scales, const float* rotations,
                     const float* opacity, const float* features, const float* pose,
                     int32_t N, int32_t H, int32_t W, int64_t capacity, int32_t depth_bits,
                     const uint32_t* given_order, int32_t need_bwd, int32_t chunk_force,
                     void* arena, int64_t arena_bytes,
                     float* mean2d, float* depth, float* cov2d, float* rgb, float* alpha, float* depth_map,
                     uint32_t* state /* 4 x u32, as gsx_composite_fwd */, int64_t* D_pinned_host, void* count_event, void* stream);

/* ---- backward of compositing (the reference relies on autograd over :124-144).  One wave64 per
 * chunk walks it back to front from the chunk-end state the forward left, adds up the nine
 * per-splat partial sums over the wavefront (through LDS: every lane parks its nine values, 2 x 9 x 2 lanes
 * add up half a row each) and stores ONE row per (tile, rank) instance at
 * partials[offsets[rank] + tile index inside the splat's rectangle] -- no atomics; the
 * per-Gaussian sum happens in gsx_preprocess_bwd in a fixed order (bitwise reproducible).
 * A row: S(dq dx), S(dq dy) (raw first moments about the splat's mean; dq = alpha dL/dalpha: gsx_preprocess_bwd takes the
 * chain rule to the mean once per Gaussian), d k00, d kb, d k11 (the conic's entries), d opacity, d colour (3).
 * Occlusion cut: an 8x8 quadrant whose pixels all enter a chunk with |T| < 2^-(GSX_CUT_LOG2 + g), g = the chunk's
 * bound on the growth of |T| in bits (gsx_constants.h; 0 unless an opacity lies outside [0, 2]), is masked out for
 * that chunk: all its weights T*alpha stay below 2^-23, like the pairs the footprint cut drops.  A chunk with four
 * dark quadrants gets zero rows without evaluating a pair.  The work items are taken from the lists the forward's combine
 * pass left in chunk_state (costliest first): a chunk, or -- in a shallow view, whose chunk pass also left each chunk's
 * mid-chunk state -- half a chunk.
 *   partials: >= D rows of 9 floats.
 *   n_rows: the view's instance count D as the caller knows it (rows 0 .. n_rows - 1 exist and are this view's), or 0.
 *       From GSX_PREZERO_ROWS on the call zero-fills those rows in one pass up front and occluded chunks write nothing
 *       (a view that large is mostly occluded chunks); below it, and with 0, every chunk writes its own rows.  Same
 *       results either way. */
#define GSX_PREZERO_ROWS (16ll << 20)
int gsx_composite_bwd(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                      const uint32_t* chunk_base, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                      int64_t max_chunks, const float* chunk_state, const uint32_t* state,
                      const float* d_rgb, float* partials, int64_t n_rows, void* stream);

/* ---- backward of records + projection: sums each splat's partial rows, then chains through
 * the 2x2 inverse, J*Sigma3*J^T, R(q)diag(s^2)R(q)^T, quaternion normalisation and the
 * perspective division (autograd of :221-344).  d_mean2d/d_depth/d_cov2d are upstream
 * gradients on the returned 'positions'/'depths'/'scales' (may be NULL).  Gradients of the
 * five parameters are written (accumulate = 0) or added (accumulate = 1) at Gaussian order. */
int gsx_preprocess_bwd(const float* positions, const float* scales, const float* rotations,
                       const float* pose, const uint32_t* rank_of, const uint32_t* offsets,
                       const uint32_t* tiles_touched, const float* partials, const uint32_t* state,
                       const float* d_mean2d, const float* d_depth, const float* d_cov2d,
                       int32_t N, int32_t accumulate,
                       float* g_positions, float* g_scales, float* g_rotations, float* g_opacity,
                       float* g_features, void* stream);

/* ---- SURVEY 8f row N4 (the step right after the gradient exchange): fused
 * torch.nn.utils.clip_grad_norm_(params, max_grad_norm) + torch.optim.Adam.step() over up to 8
 * parameter tensors with per-tensor learning rates (training/trainer.py:49-61,169-175).
 * *_host arguments are HOST arrays (length n_tensors) of DEVICE pointers / sizes / learning rates.
 * step >= 1 is the Adam step count AFTER this update (bias corrections use it).  With
 * max_grad_norm <= 0 no clipping is done and ws may be NULL.  ws (device): gsx_clip_adam_workspace_bytes();
 * after the call ((float*)((char*)ws + 8192))[0] holds the total gradient norm, [1] the clip coefficient.
 * write_clipped_grad = 1 also scales the gradients in place, as clip_grad_norm_ does. */
size_t gsx_clip_adam_workspace_bytes(void);
int gsx_clip_adam_step(int32_t n_tensors, float* const* params_host, float* const* grads_host,
                       float* const* exp_avg_host, float* const* exp_avg_sq_host,
                       const int64_t* numel_host, const float* lr_host,
                       float beta1, float beta2, float eps, int64_t step, float max_grad_norm,
                       int32_t write_clipped_grad, void* ws, size_t ws_bytes, void* stream);

/* ---- the image loss between the render forward and backward of a trainer step (training/trainer.py:144-149;
 * RGB term = clamped MSE, model/loss.py:126-137).  kind: GSX_LOSS_L1 = mean |pred - target|;
 * GSX_LOSS_CLAMPED_MSE = mean (clamp(pred, 1e-8, 1 - 1e-8) - target)^2.  n = element count.
 * fwd writes the scalar loss to *loss (device); ws (device): gsx_image_loss_workspace_bytes().
 * bwd writes d_pred = dLoss/dpred * upstream[0] (upstream: device scalar, NULL means 1). */
#define GSX_LOSS_L1 0
#define GSX_LOSS_CLAMPED_MSE 1
size_t gsx_image_loss_workspace_bytes(void);
int gsx_image_loss_fwd(int32_t kind, const float* pred, const float* target, int64_t n,
                       float* loss, void* ws, size_t ws_bytes, void* stream);
int gsx_image_loss_bwd(int32_t kind, const float* pred, const float* target, int64_t n,
                       const float* upstream, float* d_pred, void* stream);
/* forward and gradient in one pass for a loss that is the root of the backward: *loss as gsx_image_loss_fwd (same bits),
 * d_pred_unit = dLoss/dpred for upstream 1 (what gsx_image_loss_bwd writes for upstream[0] == 1).  The trainer step
 * (training/trainer.py:144-166: loss.backward() on the criterion's value) then needs no backward launch for the loss. */
int gsx_image_loss_fwd_grad(int32_t kind, const float* pred, const float* target, int64_t n,
                            float* loss, float* d_pred_unit, void* ws, size_t ws_bytes, void* stream);

/* ---- L1 + D-SSIM: (1 - lambda) * mean |pred - target| + lambda * (1 - mean SSIM(pred, target)), the criterion
 * BASELINE config 2 names ("render + L1/SSIM backward").  No counterpart in the reference (its SSIM is
 * a scikit-image metric, model/evaluation.py:97-105; fact 9 of SURVEY 0): the usual splatting form -- per-plane 11x11
 * Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2.  pred/target: planes x H x W (planes = B*3).
 * fwd writes loss3[0] = the loss, loss3[1] = mean |pred - target|, loss3[2] = mean SSIM (device) and leaves the three
 * derivative maps the backward needs in ws (device, gsx_l1_ssim_workspace_bytes: 12 B per element + partial sums);
 * bwd must get the same ws, untouched, and writes d_pred = dLoss/dpred * upstream[0] (NULL means 1). */
size_t gsx_l1_ssim_workspace_bytes(int32_t planes, int32_t H, int32_t W);
int gsx_l1_ssim_fwd(const float* pred, const float* target, int32_t planes, int32_t H, int32_t W, float lambda_dssim,
                    float* loss3, void* ws, size_t ws_bytes, void* stream);
int gsx_l1_ssim_bwd(const float* pred, const float* target, int32_t planes, int32_t H, int32_t W, float lambda_dssim,
                    const float* upstream, const void* ws, size_t ws_bytes, float* d_pred, void* stream);

/* ---- one trainer step of one view in ONE call (training/trainer.py:134-175: forward, criterion, backward,
 * clip_grad_norm_, Adam): gsx_view_forward -> image loss (+ its gradient) -> gsx_composite_bwd -> gsx_preprocess_bwd ->
 * gsx_clip_adam_step, the same launches in the same order on `stream` as those entry points called one by one.  Host-side
 * convenience only, for callers whose per-step host time matters: a small scene (BASELINE config 2: 0.31 ms of kernels)
 * is otherwise bound by its ~0.4 ms of interpreter, autograd and allocator work per step.
 * Tensors are indexed 0..4 = positions, scales, rotations, opacity, features.  adam_order[k] = which tensor
 * gsx_clip_adam_step sees k-th (the clip norm's partial sums follow that order; the reference's optimizer groups are
 * positions, features, scales, rotations, opacity = {0, 4, 1, 2, 3}, trainer.py:49-61).
 * Behind the view's compositing launches the call waits for count_event (recorded behind the tile sort: the GPU has the chunk
 * pass and the combine pass in front of it meanwhile) and reads the pinned words: D > capacity returns GSX_E_CAPACITY,
 * depth_bits < 32 with a non-zero mismatch count returns GSX_E_AGAIN (retry with depth_bits = 32) -- in both cases
 * neither the criterion, the backward nor the optimizer was enqueued: gradients, parameters and Adam state are untouched (rgb
 * is that of the truncated / mis-sorted view); the caller grows the block / widens the sort and calls again with the
 * same `step`.  GSX_OK: the step is enqueued whole.  The third word is why the event sits where it does: the tile sort writes
 * 1 + the number of EMPTY tiles there (images of up to 4096 tiles), and a view that leaves a tile empty cannot take the
 * reference's global early exit (that tile's pixels stay at alpha 0) -- gsx_composite_fwd's exit path, three launches that
 * would find nothing to do, is then not enqueued at all.  Same results either way.  For images of up to 8192 tiles three buffers of the
 * block differ from gsx_view_forward's: GSX_VIEW_TILES_SORTED does not hold the sorted tile ids (nothing behind the tile sort reads them
 * -- compositing and the backward go by the ranges and the sorted ranks) but, in its first N / 8 bytes, one bit per GAUSSIAN: set when
 * its footprint touches the image (written by the projection kernel); and GSX_VIEW_REC / GSX_VIEW_RANK_OF are written for those
 * Gaussians only (a splat without instances has no reader for its record, and the backward asks the bitmap before it looks up a rank).
 * gsx_view_forward itself always writes all of them.
 * gsx_train_step_grads / gsx_train_step_apply are the step's two halves, for data-parallel ranks (the reference wraps the
 * model in DistributedDataParallel, scripts/train.py:134-139: the gradient average sits between backward and
 * clip_grad_norm_, training/trainer.py:166-175): _grads enqueues everything up to the gradients (forward, criterion,
 * the count check described above, both backward stages) and leaves them in grad[] -- five views of ONE flat buffer when
 * the caller lays them out that way -- and _apply enqueues gsx_clip_adam_step; the caller issues its one collective over
 * the gradient buffer on the same stream between the two.  _grads ignores the Adam fields (exp_avg, exp_avg_sq, lr,
 * betas, step), _apply reads only those, param, grad, adam_order, N and the workspace.  gsx_train_step == _grads; _apply --
 * with one launch less when adam_ws has room for one double per 256 Gaussians behind gsx_clip_adam_workspace_bytes() (adam_ws_bytes
 * >= that + 8 * ceil(N / 256)): the clip norm's sums of squares are then left there by the last backward kernel instead of a
 * pass of their own over the gradients.  Both ways sum exact squares in double, so the coefficient is the same float. */
#define GSX_LOSS_L1_SSIM 2
typedef struct gsx_step {
    float* param[5];               /* updated in place */
    float* grad[5];                /* written */
    float* exp_avg[5];
    float* exp_avg_sq[5];
    float lr[5];
    int32_t adam_order[5];
    float beta1, beta2, eps, max_grad_norm;
    int64_t step;                  /* Adam step count AFTER this update (>= 1) */
    int32_t write_clipped_grad;
    int32_t N, H, W;
    int32_t depth_bits;            /* 32, or GSX_DEPTH_CHECK_BIT after a view whose mismatch count was 0 */
    int32_t chunk_force;
    int32_t loss_kind;             /* GSX_LOSS_L1, GSX_LOSS_CLAMPED_MSE, GSX_LOSS_L1_SSIM */
    float lambda_dssim;
    int64_t capacity;
    const float* pose;             /* 16 floats */
    const float* target;           /* 3 x H x W */
    void* arena; int64_t arena_bytes;      /* gsx_view_layout(N, H, W, capacity, chunk_force, need_bwd = 1) */
    float* mean2d; float* depth; float* cov2d;     /* the reference's returned tensors; all three NULL = not wanted */
    float* rgb; float* alpha;
    uint32_t* state;               /* 4 x u32, as gsx_composite_fwd */
    float* d_rgb;                  /* 3 x H x W floats of scratch: dLoss/drgb (written for GSX_LOSS_L1_SSIM only: the other criteria's
                                    * gradient is formed inside the compositing backward) */
    float* loss;                   /* 3 floats: [0] the criterion; GSX_LOSS_L1_SSIM also writes [1] = L1, [2] = SSIM */
    void* loss_ws; size_t loss_ws_bytes;   /* gsx_image_loss_workspace_bytes() or gsx_l1_ssim_workspace_bytes(3, H, W) */
    void* adam_ws; size_t adam_ws_bytes;   /* gsx_clip_adam_workspace_bytes() (+ 8 * ceil(N / 256), see above) */
    int64_t* counts_pinned_host;   /* 4 x i64 of pinned host memory: D, mismatch count, 1 + number of empty tiles (0: not known), spare */
    void* count_event;             /* hipEvent_t */
} gsx_step;
int gsx_train_step(const gsx_step* s, void* stream);
int gsx_train_step_grads(const gsx_step* s, void* stream);
int gsx_train_step_apply(const gsx_step* s, void* stream);

/* ---- the gradient exchange for hosts without torch.distributed: RCCL behind five calls.  Replaces what the
 * reference gets from dist.init_process_group("nccl") + DistributedDataParallel (scripts/train.py:65-73,134-139: the
 * gradient average after backward) and the trainer's parameter broadcast (training/trainer.py:83-87).  The Python mirror
 * of this package uses torch.distributed instead and never calls these.  librccl is opened on first use (by soname),
 * not linked: gsx_comm_available() says whether that worked.
 *   gsx_comm_unique_id : rank 0 fills GSX_COMM_ID_BYTES bytes and hands them to the other ranks out of band (a file, a socket);
 *   gsx_comm_init      : every rank, with its GPU current; collective (returns when all `world` ranks have joined);
 *   gsx_allreduce_avg  : buf[i] = mean over the ranks of buf[i], in place, on `stream` -- one call over the flat
 *                        14 * N gradient block is the whole exchange of a data-parallel step;
 *   gsx_bcast          : buf on every rank = root's buf (parameters at start, trainer.py:83-87). */
#define GSX_COMM_ID_BYTES 128
int gsx_comm_available(void);
int gsx_comm_unique_id(void* id_out);
int gsx_comm_init(int32_t rank, int32_t world, const void* unique_id, void** comm_out);
int gsx_allreduce_avg(void* comm, float* buf, int64_t n, void* stream);
int gsx_bcast(void* comm, float* buf, int64_t n, int32_t root, void* stream);
int gsx_comm_destroy(void* comm);

/* ---- maintenance of the parameter buffers (model/gaussian_splatting.py:346-369, :404-418).
 * prune_gaussians = gsx_keep_flags (keep[i] = opacity[i] > threshold) -> gsx_scan_u32 (offsets, kept count) ->
 * gsx_compact_rows once per parameter tensor (order-preserving: dst[offsets[i]] = src[i] for kept rows of
 * row_floats floats; dst needs kept-count rows).  optimize_scales = gsx_clamp_inplace(scales, 3N, 1e-4, 1.0). */
int gsx_keep_flags(const float* opacity, int32_t N, float threshold, uint32_t* keep, void* stream);
int gsx_compact_rows(const float* src, const uint32_t* keep, const uint32_t* offsets, int32_t N, int32_t row_floats,
                     float* dst, void* stream);
int gsx_clamp_inplace(float* x, int64_t n, float lo, float hi, void* stream);

/* merge_nearby_gaussians (model/gaussian_splatting.py:477-528) without the N x N matrix of torch.cdist (:491).
 * gsx_merge_count hashes the positions (N x 3, device) into a grid of cells just wider than the threshold, sorts the
 * points by bucket and writes counts[i] = number of partners j > i with 0 < |p_i - p_j| < threshold (:494-497: the
 * upper triangle); the caller scans the counts (gsx_scan_u32; the total is the number of pairs) and gsx_merge_fill, given
 * the SAME ws untouched, writes i's partners at partners[offsets[i] ..], ascending -- together the pairs in the
 * reference's row-major order.  Distances are sqrt((dx*dx + dy*dy) + dz*dz) in fp32; torch.cdist's matmul form rounds
 * differently by ~1e-4 relative, so pairs that close to the threshold (or to zero) may be decided differently.
 * gsx_merge_greedy_host is the order-dependent walk over those pairs (:500-515) on HOST copies: skips a pair when
 * either end is already removed, otherwise writes the opacity-weighted mean of the current positions / colours and the
 * larger opacity into i and sets removed[j]; the caller compacts with gsx_compact_rows (:518-528).  It accepts j == i
 * (the reference's list holds such pairs where cdist rounds a self-distance above zero; gsx_merge_fill never emits one). */
size_t gsx_merge_workspace_bytes(int32_t N);
int gsx_merge_count(const float* positions, int32_t N, float threshold, uint32_t* counts, void* ws, size_t ws_bytes, void* stream);
int gsx_merge_fill(const float* positions, int32_t N, float threshold, const uint32_t* offsets, uint32_t* partners,
                   const void* ws, size_t ws_bytes, void* stream);
int gsx_merge_greedy_host(int32_t N, const uint32_t* offsets_host, const uint32_t* partners_host,
                          float* positions_host, float* features_host, float* opacity_host, uint8_t* removed_host);

#ifdef __cplusplus
}
#endif
#endif
