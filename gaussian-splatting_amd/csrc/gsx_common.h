// gsx_common.h -- shared host/device helpers of libgsplat_gfx950.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gsx.h"

#define GSX_API extern "C" __attribute__((visibility("default")))

namespace gsx {

extern thread_local int g_last_hip_error;

inline int hip_check(hipError_t e)
{
    if (e == hipSuccess) return GSX_OK;
    g_last_hip_error = (int)e;
    return GSX_E_HIP;
}
#define GSX_HIP(expr) do { int _s = gsx::hip_check(expr); if (_s != GSX_OK) return _s; } while (0)
#define GSX_LAUNCH_CHECK() GSX_HIP(hipGetLastError())

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

constexpr int kBlock = 256;        // 4 wave64
constexpr int kRecVec = GSX_REC_FLOATS / 4;      // float4s per rank-ordered record (its stride; the first three hold the record)
constexpr int kWave = 64;

// Instances per compositing chunk for a view of D instances (gsx_constants.h): one wave composites one chunk, so a small
// view cut into 256-instance chunks is a few hundred waves on 4096 wave slots, each running its 256 splats alone -- the
// kernels then take one chunk's latency whatever the size of the scene.  Shorter chunks for smaller views; the choice is
// a function of D alone, made on the device, and written behind the chunk table for the compositing kernels.
__host__ __device__ inline uint32_t chunk_size_for(uint64_t D, int force)
{
    if (force == 64 || force == 128 || force == 256) return (uint32_t)force;
    return D < (1ull << 19) ? 64u : D < (1ull << 20) ? 128u : (uint32_t)GSX_CHUNK;
}

// -0.5 * log2(e): conic pre-multiplier so that alpha = o * exp2(q')
#define GSX_NEG_HALF_LOG2E (-0.72134752f)
#define GSX_LN2 0.69314718f

// ---- wave64 DPP reductions (row = 16 lanes) -------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row_sum16(float v)
{
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}

// ---- wave64 match-any on the low BITS bits of d: rank = number of lower valid lanes with the same digit (stable
// order), count = number of valid lanes with it.  Per bit: one v_bfe_i32 (m = all ones where the lane's bit is 0), one
// compare for the ballot, and p &= ballot ^ m on both halves (lanes whose bit is 1 keep the ballot's lanes, the others its
// complement) -- no 64-bit selects, which cost two v_cndmask plus two moves per bit on a one-SGPR-per-instruction ISA.
template <int BITS>
__device__ __forceinline__ void wave_match_rank(uint32_t d, bool valid, uint32_t& rank, uint32_t& count)
{
    const unsigned long long vm = __ballot(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
    const uint32_t nd = ~d;
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)nd, b, 1);
        const unsigned long long bal = __ballot(m == 0u);
        plo &= (uint32_t)bal ^ m;
        phi &= (uint32_t)(bal >> 32) ^ m;
    }
    rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    count = __popc(plo) + __popc(phi);
}

// ---- inside the library only (the one-call trainer step, gsx_view.hip): the image loss with its final reduction left to
// the step's last kernel -- the loss value is a report, nothing on the GPU waits for it, and a launch of its own for one
// workgroup costs as much of the stream as a thousand (4.6 us of a 0.77 ms step)
constexpr int kLossBlocks = 1024;      // partial sums of the image losses (one per workgroup, fixed slices: deterministic)
int image_loss_fwd_grad_partials(int32_t kind, const float* pred, const float* target, int64_t n, float* d_pred_unit,
                                 void* ws, size_t ws_bytes, void* stream);
// norm_partials (nullable): one double per workgroup of the kernel, ceil(N / kBlock) of them -- the sum of the squares of the
// gradients that workgroup wrote (what sumsq_kernel of gsx_optim.hip would have to read back from HBM)
int preprocess_bwd_with_loss(const float* positions, const float* scales, const float* rotations,
                             const float* pose, const uint32_t* rank_of, const uint32_t* offsets,
                             const uint32_t* tiles_touched, const float* partials, const uint32_t* state,
                             int32_t N, float* g_positions, float* g_scales, float* g_rotations, float* g_opacity,
                             float* g_features, const double* loss_partials, int64_t loss_n, float* loss, double* norm_partials,
                             const uint64_t* vis /* nullable: project_fwd_impl's bitmap; rank_of then holds visible Gaussians only */, void* stream);
#ifdef __HIPCC__
// (gsx_loss.hip's terms; the backward of the one-call trainer step evaluates them itself)
constexpr float kClampLo = 1e-8f, kClampHi = 1.0f - 1e-8f;     // the reference's fp32 constants (1 - 1e-8 == 1.0f)

template <int KIND>
__device__ __forceinline__ float loss_term(float p, float t)
{
    if (KIND == GSX_LOSS_L1) return fabsf(p - t);
    const float c = fminf(fmaxf(p, kClampLo), kClampHi);
    const float d = c - t;
    return d * d;
}

template <int KIND>
__device__ __forceinline__ float loss_grad(float p, float t, float g_over_n)
{
    if (KIND == GSX_LOSS_L1) {
        const float d = p - t;                              // abs backward: grad * sign(d), sign(0) = 0
        return d > 0.0f ? g_over_n : (d < 0.0f ? -g_over_n : 0.0f);
    }
    // clamp passes the gradient on [lo, hi] inclusive; mse backward = 2 (c - t) g / n
    const float c = fminf(fmaxf(p, kClampLo), kClampHi);
    const bool pass = p >= kClampLo && p <= kClampHi;
    return pass ? (2.0f * g_over_n) * (c - t) : 0.0f;
}

// (gsx_sort.hip's scans and the emit kernel's)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(v);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
    uint32_t base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + incl - v;
}
#endif

// gsx_composite_bwd with the one-call trainer step's criterion inside its launch: kind = GSX_LOSS_L1 or GSX_LOSS_CLAMPED_MSE of pred
// (the view's rgb) against target, n elements, unit upstream gradient (g_over_n = 1 / n); partials = kLossBlocks doubles, the slices
// and bits of gsx_loss.hip's loss_partial_kernel; vec = both images 16-byte aligned
struct LossInBwd { const float* pred; const float* target; double* partials; long long n; int kind; int vec; float g_over_n; };
int composite_bwd_impl(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                       const uint32_t* chunk_base, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                       int64_t max_chunks, const float* chunk_state, const uint32_t* state,
                       const float* d_rgb, float* partials, int64_t n_rows, const LossInBwd* lw, void* stream);
// gsx_project_fwd / gsx_build_records with the visibility bitmap of the one-call trainer step (vis: one bit per Gaussian, N / 64 words
// rounded up; H, W: the image the footprints are cut against).  vis == nullptr: the public entry points' behaviour
int project_fwd_impl(const float* positions, const float* scales, const float* rotations,
                     const float* opacity, const float* features, const float* pose, int32_t N,
                     float* mean2d, float* depth, float* cov2d, uint32_t* depth_key, float* pre,
                     int32_t H, int32_t W, uint64_t* vis, void* stream);
int build_records_impl(const float* pre, const uint32_t* order, int32_t N, int32_t H, int32_t W,
                       float* rec, uint32_t* tiles_touched, uint32_t* rank_of, uint64_t* tt_block_sums, uint32_t* tile_box,
                       const uint64_t* vis, void* stream);
// gsx_bin_emit_sort with a pinned host word for the number of empty tiles; gsx_composite_fwd in two parts (which: 1 = the view's
// compositing, 2 = the global-exit path behind it -- search, second chunk pass, second combine pass --, 3 = both)
// scan (nullable; N > 0): the emit kernel scans the tile counts itself instead of reading offsets[] (no gsx_bin_count in front of this
// call): sums = gsx_build_records' tt_block_sums, offsets_out = the `offsets` argument (written here), total = the `total_dev`
// argument (written here), mirror = device view of two pinned host words for the total and the word behind it, or nullptr
// no_sorted_keys: the one-pass tile sorts leave tiles_sorted unwritten (compositing reads ranges and ranks_sorted only)
struct EmitScan { const uint64_t* sums; uint32_t* offsets_out; uint64_t* total; uint64_t* mirror; int no_sorted_keys; };
int bin_emit_sort_impl(const float* rec, const uint32_t* tile_box, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                       int64_t capacity, const uint64_t* total_dev,
                       uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                       uint32_t* ranges, uint32_t* chunk_base, int32_t chunk_force, void* ws, size_t ws_bytes, uint64_t* info, void* stream,
                       const EmitScan* scan = nullptr);
int composite_fwd_impl(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                       const uint32_t* chunk_base, int32_t N, int32_t H, int32_t W,
                       int64_t max_chunks, float* chunk_state, int32_t need_bwd, uint32_t* state,
                       float* rgb, float* alpha, float* depth_map, int which, void* stream);
int clip_adam_step_impl(int32_t n_tensors, float* const* params_host, float* const* grads_host,
                        float* const* exp_avg_host, float* const* exp_avg_sq_host,
                        const int64_t* numel_host, const float* lr_host,
                        float beta1, float beta2, float eps, int64_t step, float max_grad_norm,
                        int32_t write_clipped_grad, void* ws, size_t ws_bytes, const double* norm_partials, int32_t n_norm_partials,
                        void* stream);

}  // namespace gsx
