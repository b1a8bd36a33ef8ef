// gsx_maint.hip -- per-Gaussian maintenance of the parameter buffers the render kernels read:
// prune_gaussians (/root/reference/model/gaussian_splatting.py:346-369: boolean-mask indexing of five tensors ->
// here one flag pass, one scan, one order-preserving compaction per tensor) and optimize_scales (:404-418: two
// full-tensor maximum/minimum passes with temporaries -> one in-place clamp).  HBM-bound byte moving; no MFMA.
#include "gsx_common.h"

namespace gsx {

int scan_internal(const uint32_t*, uint32_t*, uint64_t*, int64_t, void*, size_t, hipStream_t, const uint64_t* block_sums = nullptr, int sums_tile = 0,
                  uint64_t* mirror = nullptr);

__global__ __launch_bounds__(kBlock) void keep_flags_kernel(const float* __restrict__ opacity, int N, float threshold,
                                                            uint32_t* __restrict__ keep)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < N) keep[i] = opacity[i] > threshold ? 1u : 0u;          // NaN > t is false, as in torch
}

// dst[offsets[i]] = src[i] for kept rows; rows of row_floats floats; one thread per float
__global__ __launch_bounds__(kBlock) void compact_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ keep,
                                                              const uint32_t* __restrict__ offsets, long long total, int row_floats,
                                                              float* __restrict__ dst)
{
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total) return;
    const long long i = t / row_floats; const int c = (int)(t - i * row_floats);
    if (keep[i]) dst[(size_t)offsets[i] * row_floats + c] = src[t];
}

__global__ __launch_bounds__(kBlock) void clamp_kernel(float* __restrict__ x, long long n, float lo, float hi)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    x[i] = (v != v) ? v : fminf(fmaxf(v, lo), hi);                   // torch.maximum / minimum propagate NaN
}

}  // namespace gsx

using namespace gsx;

GSX_API int gsx_keep_flags(const float* opacity, int32_t N, float threshold, uint32_t* keep, void* stream)
{
    if (N < 0) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!opacity || !keep) return GSX_E_INVALID;
    keep_flags_kernel<<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(opacity, N, threshold, keep);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_compact_rows(const float* src, const uint32_t* keep, const uint32_t* offsets, int32_t N, int32_t row_floats,
                             float* dst, void* stream)
{
    if (N < 0 || row_floats <= 0) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!src || !keep || !offsets || !dst) return GSX_E_INVALID;
    const long long total = (long long)N * row_floats;
    compact_rows_kernel<<<dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(src, keep, offsets, total, row_floats, dst);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_clamp_inplace(float* x, int64_t n, float lo, float hi, void* stream)
{
    if (n < 0 || !(lo <= hi)) return GSX_E_INVALID;
    if (n == 0) return GSX_OK;
    if (!x) return GSX_E_INVALID;
    clamp_kernel<<<dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(x, n, lo, hi);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}
