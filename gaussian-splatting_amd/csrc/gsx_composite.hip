// gsx_composite.hip -- per-tile alpha compositing, forward and backward (gfx950, wave64).
// Replaces the inner loop of _render_chunk (/root/reference/model/gaussian_splatting.py:118-148)
// and the autograd graph recorded over it.  One 256-thread workgroup (4 wave64) per 16x16 tile;
// the tile's depth-ordered splat records are staged through LDS in batches of 256 and every
// lane (= pixel) walks the batch with broadcast LDS reads.
//
// Bound: LDS-broadcast + VALU (about 20 VALU + 1 v_exp_f32 per pixel-splat pair); HBM traffic is
// 52 B per instance forward (4 B rank + 48 B record) plus 24 B per pixel.  No MFMA: there is no
// contraction on this path.
#include "gsx_common.h"

namespace gsx {

__device__ __forceinline__ float pix_coord(int i, int steps, float step)
{
    // torch.linspace(-1, 1, steps) in fp32 (:100-102): symmetric fill, one fma per element
    if (steps == 1) return -1.0f;
    return (i < (steps >> 1)) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

// state[0] -> number of ranks to composite (N when the global early exit does not fire)
__device__ __forceinline__ uint32_t rank_limit_from_state(const uint32_t* state, uint32_t N)
{
    const uint32_t s = state[0];
    return (s != GSX_RANK_NONE && s + 1u < N) ? s + 1u : N;
}

struct TileGeom { int tx, ty, x, y; bool inside; float gx, gy; };

__device__ __forceinline__ TileGeom tile_geom(int tile, int tw, int H, int W, float stepx, float stepy)
{
    TileGeom t;
    t.tx = tile % tw; t.ty = tile / tw;
    t.x = t.tx * GSX_TILE + (threadIdx.x & 15);
    t.y = t.ty * GSX_TILE + (threadIdx.x >> 4);
    t.inside = t.x < W && t.y < H;
    t.gx = pix_coord(t.x < W ? t.x : W - 1, W, stepx);
    t.gy = pix_coord(t.y < H ? t.y : H - 1, H, stepy);
    return t;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <bool PASS1, bool DEPTH>
__global__ __launch_bounds__(kBlock) void composite_fwd_kernel(
    const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
    uint32_t N, int H, int W, int tw, float stepx, float stepy, uint32_t* __restrict__ state,
    float* __restrict__ rgb, float* __restrict__ alpha, float* __restrict__ depth_map)
{
    __shared__ float4 s_a[kBlock], s_b[kBlock], s_c[kBlock];
    __shared__ uint32_t s_rank[kBlock];
    __shared__ uint32_t s_red[4];
    const int tile = blockIdx.x;
    const uint32_t begin = ranges[2*tile], end = ranges[2*tile+1];
    uint32_t limit = N;
    if (PASS1) {
        limit = rank_limit_from_state(state, N);
        if (limit >= N) return;                              // early exit did not fire: pass 0 image stands
        if (end == begin || ranks[end-1] < limit) return;    // this tile holds nothing beyond the exit rank
    }
    const TileGeom t = tile_geom(tile, tw, H, W, stepx, stepy);
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f, Dm = 0.f;
    uint32_t sat = GSX_RANK_NONE;
    for (uint32_t start = begin; start < end; start += kBlock) {
        const uint32_t i = start + threadIdx.x;
        if (i < end) {
            const uint32_t r = ranks[i];
            s_rank[threadIdx.x] = r;
            const float4* R = rec + (size_t)3 * r;
            s_a[threadIdx.x] = R[0]; s_b[threadIdx.x] = R[1]; s_c[threadIdx.x] = R[2];
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)kBlock, end - start);
        bool stop = false;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t r = s_rank[j];
            if (PASS1 && r >= limit) { stop = true; break; }
            const float4 a = s_a[j], b = s_b[j], c = s_c[j];
            const float dx = t.gx - a.x, dy = t.gy - a.y;
            const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
            const float am = b.y * exp2f(q);
            const float T = 1.0f - A;
            C0 = C0 + T * (b.z * am); C1 = C1 + T * (b.w * am); C2 = C2 + T * (c.x * am);
            if (DEPTH) Dm = Dm + T * (c.y * am);
            A = A + T * am;
            if (!PASS1) { if (sat == GSX_RANK_NONE && A > GSX_SAT_ALPHA) sat = r; }
        }
        if (stop) break;
        __syncthreads();
    }
    if (t.inside) {
        const size_t P = (size_t)H * W, p = (size_t)t.y * W + t.x;
        rgb[p] = C0; rgb[P + p] = C1; rgb[2*P + p] = C2;
        alpha[p] = A;
        if (DEPTH) depth_map[p] = Dm;
    }
    if (!PASS1) {
        // max over the image of the per-pixel saturation rank; pixels outside the image do not vote
        uint32_t v = t.inside ? sat : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(state, max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
    }
}

// ------------------------------------------------------------------------------------------
// backward.  Front-to-back again (same order, same arithmetic as the forward, so T_k and the
// colour prefix are reproduced exactly and never divided out of a product that may underflow):
//   dL/dalpha_k = T_k (c_k . G) - S_k / (1 - alpha_k),   S_k = G . (C_final - C_prefix_k)
// Nine partial sums per (tile, splat) are reduced over the 16-lane rows with DPP, over the
// workgroup with LDS float adds, and stored as ONE row per instance (no global atomics).
// ------------------------------------------------------------------------------------------
constexpr int kBwdBatch = 128;     // splats staged per batch in the backward (LDS: 6 KB records + 18 KB partial sums)

// wave64 total of v, valid in lane 63: 16-lane row sums, then row_bcast15 / row_bcast31 (fixed order -> deterministic)
__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v = row_sum16(v);
    v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v += dpp_mov<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

__global__ __launch_bounds__(kBlock) void composite_bwd_kernel(
    const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
    const uint32_t* __restrict__ offsets, uint32_t N, int H, int W, int tw, float stepx, float stepy,
    const uint32_t* __restrict__ state, const float* __restrict__ rgb, const float* __restrict__ d_rgb,
    float* __restrict__ partials)
{
    __shared__ float4 s_a[kBwdBatch], s_b[kBwdBatch], s_c[kBwdBatch];
    __shared__ uint32_t s_rank[kBwdBatch], s_slot[kBwdBatch];
    __shared__ float s_part[4][kBwdBatch][9];          // one slab per wave: no LDS atomics, fixed summation order
    const int tile = blockIdx.x;
    const uint32_t begin = ranges[2*tile], end = ranges[2*tile+1];
    if (begin == end) return;
    const uint32_t limit = rank_limit_from_state(state, N);
    const TileGeom t = tile_geom(tile, tw, H, W, stepx, stepy);
    const size_t P = (size_t)H * W, p = (size_t)t.y * W + t.x;
    float G0 = 0.f, G1 = 0.f, G2 = 0.f, S = 0.f;
    if (t.inside) {
        G0 = d_rgb[p]; G1 = d_rgb[P + p]; G2 = d_rgb[2*P + p];
        S = G0 * rgb[p] + G1 * rgb[P + p] + G2 * rgb[2*P + p];
    }
    float A = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t start = begin; start < end; start += kBwdBatch) {
        const uint32_t i = start + threadIdx.x;
        if (threadIdx.x < kBwdBatch && i < end) {
            const uint32_t r = ranks[i];
            s_rank[threadIdx.x] = r;
            const float4* R = rec + (size_t)3 * r;
            const float4 c = R[2];
            s_a[threadIdx.x] = R[0]; s_b[threadIdx.x] = R[1]; s_c[threadIdx.x] = c;
            const uint32_t bx = __float_as_uint(c.z), by = __float_as_uint(c.w);
            const int x0 = (int)(bx & 0xffff) >> GSX_TILE_SHIFT, x1 = ((int)(bx >> 16) >> GSX_TILE_SHIFT) + 1;
            const int y0 = (int)(by & 0xffff) >> GSX_TILE_SHIFT;
            s_slot[threadIdx.x] = offsets[r] + (uint32_t)((t.ty - y0) * (x1 - x0) + (t.tx - x0));
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)kBwdBatch, end - start);
        uint32_t done = cnt;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t r = s_rank[j];
            if (r >= limit) { done = j; break; }
            const float4 a = s_a[j], b = s_b[j], c = s_c[j];
            const float dx = t.gx - a.x, dy = t.gy - a.y;
            const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
            const float am = b.y * exp2f(q);
            const float T = 1.0f - A;
            const float w = T * am;
            A = A + w;
            const float cG = b.z * G0 + b.w * G1 + c.x * G2;
            S = S - w * cG;
            const float om = 1.0f - am;
            const float inv = (om != 0.0f) ? __frcp_rn(om) : 0.0f;
            float dq = (T * cG - S * inv) * am * GSX_LN2;      // dL/dq'   (alpha = o * exp2(q'))
            float wv = w;
            if (!t.inside) { dq = 0.f; wv = 0.f; }
            const float v0 = wave_sum_lane63(dq);
            const float v1 = wave_sum_lane63(dq * dx);
            const float v2 = wave_sum_lane63(dq * dy);
            const float v3 = wave_sum_lane63(dq * dx * dx);
            const float v4 = wave_sum_lane63(dq * dx * dy);
            const float v5 = wave_sum_lane63(dq * dy * dy);
            const float v6 = wave_sum_lane63(wv * G0);
            const float v7 = wave_sum_lane63(wv * G1);
            const float v8 = wave_sum_lane63(wv * G2);
            if (lane == 63) {
                float* dst = s_part[wave][j];
                dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; dst[4] = v4;
                dst[5] = v5; dst[6] = v6; dst[7] = v7; dst[8] = v8;
            }
        }
        __syncthreads();
        // one row per instance: (dmu_x, dmu_y, dk00, dkb, dk11, d opacity, dr, dg, db)
        for (uint32_t idx = threadIdx.x; idx < done * 9u; idx += kBlock) {
            const uint32_t j = idx / 9u, c = idx - j * 9u;
            float sp[9];
#pragma unroll
            for (int k = 0; k < 9; k++) sp[k] = 0.f;
            // only the sums this component needs, waves added in a fixed order
            auto tot = [&](int k) { return ((s_part[0][j][k] + s_part[1][j][k]) + s_part[2][j][k]) + s_part[3][j][k]; };
            float out;
            if (c == 0)      out = -(2.0f * s_a[j].z * tot(1) + s_a[j].w * tot(2));
            else if (c == 1) out = -(s_a[j].w * tot(1) + 2.0f * s_b[j].x * tot(2));
            else if (c <= 4) out = tot((int)c + 1);
            else if (c == 5) out = tot(0) / (s_b[j].y * GSX_LN2);
            else             out = tot((int)c);
            (void)sp;
            partials[(size_t)s_slot[j] * 9u + c] = out;
        }
        if (done < cnt) break;
        __syncthreads();
    }
}

}  // namespace gsx

using namespace gsx;

static inline float lin_step(int steps) { return steps > 1 ? (1.0f - (-1.0f)) / (float)(steps - 1) : 0.0f; }

GSX_API int gsx_composite_fwd(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                              int32_t N, int32_t H, int32_t W, int32_t pass, uint32_t* state,
                              float* rgb, float* alpha, float* depth_map, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || (pass != 0 && pass != 1)) return GSX_E_INVALID;
    if (!ranges || !state || !rgb || !alpha) return GSX_E_INVALID;
    if (N > 0 && (!rec || !ranks_sorted)) return GSX_E_INVALID;
    if ((uintptr_t)rec & 15u) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE;
    const float sx = lin_step(W), sy = lin_step(H);
    const dim3 grid((unsigned)(tw * th)), block(kBlock);
    const float4* r4 = (const float4*)rec;
    if (pass == 0) {
        GSX_HIP(hipMemsetAsync(state, 0, 4, st));
        if (depth_map) composite_fwd_kernel<false, true><<<grid, block, 0, st>>>(r4, ranks_sorted, ranges, (uint32_t)N, H, W, tw, sx, sy, state, rgb, alpha, depth_map);
        else           composite_fwd_kernel<false, false><<<grid, block, 0, st>>>(r4, ranks_sorted, ranges, (uint32_t)N, H, W, tw, sx, sy, state, rgb, alpha, depth_map);
    } else {
        if (depth_map) composite_fwd_kernel<true, true><<<grid, block, 0, st>>>(r4, ranks_sorted, ranges, (uint32_t)N, H, W, tw, sx, sy, state, rgb, alpha, depth_map);
        else           composite_fwd_kernel<true, false><<<grid, block, 0, st>>>(r4, ranks_sorted, ranges, (uint32_t)N, H, W, tw, sx, sy, state, rgb, alpha, depth_map);
    }
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_composite_bwd(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                              const uint32_t* offsets, int32_t N, int32_t H, int32_t W, const uint32_t* state,
                              const float* rgb, const float* d_rgb, float* partials, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!rec || !ranks_sorted || !ranges || !offsets || !state || !rgb || !d_rgb || !partials) return GSX_E_INVALID;
    if ((uintptr_t)rec & 15u) return GSX_E_INVALID;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE;
    composite_bwd_kernel<<<dim3((unsigned)(tw * th)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        (const float4*)rec, ranks_sorted, ranges, offsets, (uint32_t)N, H, W, tw, lin_step(W), lin_step(H),
        state, rgb, d_rgb, partials);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API const char* gsx_strerror(int status)
{
    switch (status) {
        case GSX_OK: return "ok";
        case GSX_E_INVALID: return "invalid argument";
        case GSX_E_CAPACITY: return "instance buffers or workspace too small";
        case GSX_E_HIP: return "HIP runtime error";
        case GSX_E_RANGE: return "problem exceeds 32-bit instance indexing";
        default: return "unknown status";
    }
}
GSX_API int gsx_last_hip_error(void) { return gsx::g_last_hip_error; }
GSX_API int gsx_abi_version(void) { return 1; }
