// gsx_loss.hip -- the image loss between the render forward and the render backward of a trainer step
// (/root/reference/training/trainer.py:144-149 calls the criterion on the rendered image; its RGB term is
// the clamped MSE of /root/reference/model/loss.py:126-137).  Two kinds:
//   GSX_LOSS_L1          mean |pred - target|                       (the bench's loss)
//   GSX_LOSS_CLAMPED_MSE mean (clamp(pred, 1e-8, 1 - 1e-8) - target)^2   (loss.py:135-137)
// torch spends 6-7 elementwise/reduction launches and ~5 image-sized temporaries on either; here it is
// one read of (pred, target) forward and one read + one write backward.  HBM-bound: 8 B/px-channel
// forward, 12 B backward.  The reduction is a fixed two-level tree in double (deterministic).
#include "gsx_common.h"

namespace gsx {

constexpr int kLossBlocks = 1024;
constexpr float kClampLo = 1e-8f, kClampHi = 1.0f - 1e-8f;     // the reference's fp32 constants (1 - 1e-8 == 1.0f)

template <int KIND>
__device__ __forceinline__ float loss_term(float p, float t)
{
    if (KIND == GSX_LOSS_L1) return fabsf(p - t);
    const float c = fminf(fmaxf(p, kClampLo), kClampHi);
    const float d = c - t;
    return d * d;
}

template <int KIND, bool VEC>
__global__ __launch_bounds__(kBlock) void loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                              long long n, double* __restrict__ partials)
{
    __shared__ double s_w[4];
    const long long n4 = n >> 2;
    const long long per = (n4 + kLossBlocks - 1) / kLossBlocks;
    const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, n4);
    float s = 0.0f;
    const float4* p4 = reinterpret_cast<const float4*>(pred);
    const float4* t4 = reinterpret_cast<const float4*>(target);
    for (long long q = lo + threadIdx.x; q < hi; q += kBlock) {
        float4 p, t;
        if (VEC) { p = p4[q]; t = t4[q]; }
        else {   // misaligned views: same grouping, scalar loads (same result bit for bit)
            p = make_float4(pred[4*q], pred[4*q+1], pred[4*q+2], pred[4*q+3]);
            t = make_float4(target[4*q], target[4*q+1], target[4*q+2], target[4*q+3]);
        }
        s += (loss_term<KIND>(p.x, t.x) + loss_term<KIND>(p.y, t.y)) + (loss_term<KIND>(p.z, t.z) + loss_term<KIND>(p.w, t.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += loss_term<KIND>(pred[n4 * 4 + threadIdx.x], target[n4 * 4 + threadIdx.x]);
    double sd = (double)s;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sd += __shfl_down(sd, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sd;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(kBlock) void loss_final_kernel(const double* __restrict__ partials, long long n, float* __restrict__ loss)
{
    __shared__ double s_v[kBlock];
    double s = 0.0;
    for (int i = threadIdx.x; i < kLossBlocks; i += kBlock) s += partials[i];
    s_v[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) { if (threadIdx.x < d) s_v[threadIdx.x] += s_v[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = (float)(s_v[0] / (double)n);
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

template <int KIND, bool VEC>
__global__ __launch_bounds__(kBlock) void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, long long n,
                                                          const float* __restrict__ upstream, float* __restrict__ d_pred)
{
    const float g = (upstream ? upstream[0] : 1.0f) / (float)n;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (VEC) {
        if (i >= (n >> 2)) return;
        const float4 p = reinterpret_cast<const float4*>(pred)[i], t = reinterpret_cast<const float4*>(target)[i];
        float4 o;
        o.x = loss_grad<KIND>(p.x, t.x, g); o.y = loss_grad<KIND>(p.y, t.y, g);
        o.z = loss_grad<KIND>(p.z, t.z, g); o.w = loss_grad<KIND>(p.w, t.w, g);
        reinterpret_cast<float4*>(d_pred)[i] = o;
    } else {
        if (i < n) d_pred[i] = loss_grad<KIND>(pred[i], target[i], g);
    }
}

}  // namespace gsx

using namespace gsx;

GSX_API size_t gsx_image_loss_workspace_bytes(void) { return (size_t)kLossBlocks * 8; }

GSX_API int gsx_image_loss_fwd(int32_t kind, const float* pred, const float* target, int64_t n,
                               float* loss, void* ws, size_t ws_bytes, void* stream)
{
    if ((kind != GSX_LOSS_L1 && kind != GSX_LOSS_CLAMPED_MSE) || n <= 0 || !pred || !target || !loss) return GSX_E_INVALID;
    if (!ws || ws_bytes < gsx_image_loss_workspace_bytes()) return GSX_E_INVALID;
    const bool vec = (((uintptr_t)pred | (uintptr_t)target) & 15u) == 0;
    hipStream_t st = (hipStream_t)stream;
    double* partials = (double*)ws;
    const dim3 grid(kLossBlocks), block(kBlock);
    if (kind == GSX_LOSS_L1) {
        if (vec) loss_partial_kernel<GSX_LOSS_L1, true><<<grid, block, 0, st>>>(pred, target, n, partials);
        else loss_partial_kernel<GSX_LOSS_L1, false><<<grid, block, 0, st>>>(pred, target, n, partials);
    } else {
        if (vec) loss_partial_kernel<GSX_LOSS_CLAMPED_MSE, true><<<grid, block, 0, st>>>(pred, target, n, partials);
        else loss_partial_kernel<GSX_LOSS_CLAMPED_MSE, false><<<grid, block, 0, st>>>(pred, target, n, partials);
    }
    loss_final_kernel<<<dim3(1), dim3(kBlock), 0, st>>>(partials, n, loss);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_image_loss_bwd(int32_t kind, const float* pred, const float* target, int64_t n,
                               const float* upstream, float* d_pred, void* stream)
{
    if ((kind != GSX_LOSS_L1 && kind != GSX_LOSS_CLAMPED_MSE) || n <= 0 || !pred || !target || !d_pred) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)d_pred) & 15u) == 0) && (n & 3) == 0;
    const long long items = vec ? (n >> 2) : n;
    const dim3 grid((unsigned)ceil_div(items, kBlock));
    if (kind == GSX_LOSS_L1) {
        if (vec) loss_bwd_kernel<GSX_LOSS_L1, true><<<grid, dim3(kBlock), 0, st>>>(pred, target, n, upstream, d_pred);
        else loss_bwd_kernel<GSX_LOSS_L1, false><<<grid, dim3(kBlock), 0, st>>>(pred, target, n, upstream, d_pred);
    } else {
        if (vec) loss_bwd_kernel<GSX_LOSS_CLAMPED_MSE, true><<<grid, dim3(kBlock), 0, st>>>(pred, target, n, upstream, d_pred);
        else loss_bwd_kernel<GSX_LOSS_CLAMPED_MSE, false><<<grid, dim3(kBlock), 0, st>>>(pred, target, n, upstream, d_pred);
    }
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}
