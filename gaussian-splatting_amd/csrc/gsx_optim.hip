// gsx_optim.hip -- fused clip_grad_norm_ + Adam step over the (up to 8) parameter tensors of the
// splat model: the step right after the gradient all-reduce in the reference's trainer
// (/root/reference/training/trainer.py:169-175: torch.nn.utils.clip_grad_norm_(params, 0.1) then
// torch.optim.Adam.step() with five per-tensor learning rates, :49-61).  SURVEY 8f row N4.
// Two launches instead of ~25 foreach kernels; HBM-bound: 4 B read for the norm + 28 B
// read/written per element for the update.  Arithmetic follows torch's foreach Adam operation by
// operation (lerp, mul+addcmul, sqrt/div/add, addcdiv); compiled with -ffp-contract=off.
#include "gsx_common.h"

namespace gsx {

constexpr int kMaxTensors = 8;
constexpr int kNormBlocks = 1024;

struct AdamArgs {
    float* p[kMaxTensors]; float* g[kMaxTensors]; float* m[kMaxTensors]; float* v[kMaxTensors];
    long long begin[kMaxTensors + 1];      // prefix offsets in the virtual concatenation
    float neg_step_size[kMaxTensors];      // -lr / (1 - beta1^t)
    long long qbegin[kMaxTensors + 1];     // prefix offsets in units of 4-element groups (per tensor: ceil(n/4))
    int vec[kMaxTensors];                  // 1: all four pointers of the tensor are 16-byte aligned
    int n_tensors;
};

__device__ __forceinline__ int find_tensor_q(const AdamArgs& a, long long q)
{
    int t = 0;
#pragma unroll
    for (int k = 1; k < kMaxTensors; k++) if (k < a.n_tensors && q >= a.qbegin[k]) t = k;
    return t;
}

__device__ __forceinline__ int find_tensor(const AdamArgs& a, long long i)
{
    int t = 0;
#pragma unroll
    for (int k = 1; k < kMaxTensors; k++) if (k < a.n_tensors && i >= a.begin[k]) t = k;
    return t;
}

__device__ __forceinline__ double sq4(const float4& x)
{
    return ((double)x.x * (double)x.x + (double)x.y * (double)x.y) + ((double)x.z * (double)x.z + (double)x.w * (double)x.w);
}

// partial sums of squares over contiguous slices of the 4-element groups (fixed slicing -> deterministic)
__global__ __launch_bounds__(kBlock) void sumsq_kernel(AdamArgs a, double* __restrict__ partials)
{
    __shared__ double s_w[4];
    const long long total = a.qbegin[a.n_tensors];
    const long long per = (total + kNormBlocks - 1) / kNormBlocks;
    const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, total);
    // squares and sums in DOUBLE (a float's square is exact in a double; the sums then differ between two ways of cutting the
    // buffer by ~1e-16 relative, far below the float the norm is rounded to): the one-call trainer step takes its partial sums
    // from preprocess_bwd's workgroups instead of this kernel (gsx_view.hip) and must arrive at the same coefficient
    double s = 0.0;
    // the slice usually lies inside one tensor: walk the tensors it meets, so that the inner loop is nothing but
    // independent 16-byte loads (4 in flight per thread) -- with the tensor lookup inside it the loop ran one load at a time
    for (int t = 0; t < a.n_tensors; t++) {
        const long long qlo = lo > a.qbegin[t] ? lo : a.qbegin[t], qhi = hi < a.qbegin[t + 1] ? hi : a.qbegin[t + 1];
        if (qlo >= qhi) continue;
        const long long n = a.begin[t + 1] - a.begin[t];
        const float* __restrict__ g = a.g[t];
        const long long full = a.vec[t] ? (a.qbegin[t] + (n >> 2)) : a.qbegin[t];     // groups below `full` are whole float4s
        const long long vhi = qhi < full ? qhi : full;
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g) - a.qbegin[t];
        long long q = qlo + threadIdx.x;
        for (; q + 3 * kBlock < vhi; q += 4 * kBlock) {
            const float4 x0 = g4[q], x1 = g4[q + kBlock], x2 = g4[q + 2 * kBlock], x3 = g4[q + 3 * kBlock];
            s += sq4(x0); s += sq4(x1); s += sq4(x2); s += sq4(x3);
        }
        for (; q < vhi; q += kBlock) s += sq4(g4[q]);
        for (q = (qlo > full ? qlo : full) + threadIdx.x; q < qhi; q += kBlock) {      // unaligned tensors and the ragged last group
            const long long j = (q - a.qbegin[t]) * 4;
            for (int k = 0; k < 4 && j + k < n; k++) s += (double)g[j + k] * (double)g[j + k];
        }
    }
    double sd = s;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sd += __shfl_down(sd, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sd;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// total_norm and clip coefficient = min(1, max_norm / (total_norm + 1e-6)) from the kNormBlocks partial sums.  Every
// workgroup of the update kernel runs it itself (same values, same order -> the same bits everywhere): 8 KB of
// L2 reads per workgroup instead of a separate one-workgroup launch between the two passes.
__device__ __forceinline__ float clip_coefficient(const double* __restrict__ partials, int n_partials, float max_norm, float* norm_out)
{
    __shared__ double s_v[kBlock];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += kBlock) s += partials[i];
    s_v[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) { if (threadIdx.x < d) s_v[threadIdx.x] += s_v[threadIdx.x + d]; __syncthreads(); }
    const float norm = (float)sqrt(s_v[0]);
    const float coef = max_norm / (norm + 1e-6f);
    *norm_out = norm;
    return coef < 1.0f ? coef : 1.0f;
}

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, bool clip, float coef, float w1, float w2,
                                         float beta2, float eps, float bc2_sqrt, float neg_step)
{
    if (clip) g = g * coef;
    m = m + w1 * (g - m);                    // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2;                           // exp_avg_sq.mul_(beta2)
    v = v + (w2 * g) * g;                    //           .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p + neg_step * (m / denom);          // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ __launch_bounds__(kBlock) void adam_kernel(AdamArgs a, const double* __restrict__ partials, int n_partials, float max_norm,
                                                      float* __restrict__ norm_coef_out, int write_clipped_grad,
                                                      float beta1, float beta2, float eps, float bc2_sqrt)
{
    const long long total = a.qbegin[a.n_tensors];
    const bool clip = partials != nullptr;
    float coef = 1.0f;
    if (clip) {
        float norm;
        coef = clip_coefficient(partials, n_partials, max_norm, &norm);
        if (blockIdx.x == 0 && threadIdx.x == 0) { norm_coef_out[0] = norm; norm_coef_out[1] = coef; }
    }
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < total; q += (long long)gridDim.x * kBlock) {
        const int t = find_tensor_q(a, q);
        const long long j = (q - a.qbegin[t]) * 4, n = a.begin[t + 1] - a.begin[t];
        const float ns = a.neg_step_size[t];
        if (a.vec[t] && j + 4 <= n) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            // Streaming loads and stores for what nothing reads again before the next step's optimizer -- the two moments, and the gradient
            // unless it is written back: 280 of the kernel's 390 MB no longer push the parameters (the projection reads them next), the
            // packed rows and the records out of the caches.  One A/B on one box: step 0.686 -> 0.668 ms (this kernel 72 -> 68.5 us,
            // the records stage behind it 44 -> 40.5).
            float4 p = *reinterpret_cast<float4*>(a.p[t] + j), g;
            if (!(clip && write_clipped_grad)) { const f4 g_ = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.g[t] + j)); g = make_float4(g_.x, g_.y, g_.z, g_.w); }
            else g = *reinterpret_cast<float4*>(a.g[t] + j);
            const f4 m_ = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.m[t] + j)), v_ = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.v[t] + j));
            float4 m = make_float4(m_.x, m_.y, m_.z, m_.w), v = make_float4(v_.x, v_.y, v_.z, v_.w);
            adam_one(p.x, g.x, m.x, v.x, clip, coef, w1, w2, beta2, eps, bc2_sqrt, ns);
            adam_one(p.y, g.y, m.y, v.y, clip, coef, w1, w2, beta2, eps, bc2_sqrt, ns);
            adam_one(p.z, g.z, m.z, v.z, clip, coef, w1, w2, beta2, eps, bc2_sqrt, ns);
            adam_one(p.w, g.w, m.w, v.w, clip, coef, w1, w2, beta2, eps, bc2_sqrt, ns);
            *reinterpret_cast<float4*>(a.p[t] + j) = p;
            __builtin_nontemporal_store(f4{m.x, m.y, m.z, m.w}, reinterpret_cast<f4*>(a.m[t] + j));
            __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(a.v[t] + j));
            if (clip && write_clipped_grad) *reinterpret_cast<float4*>(a.g[t] + j) = g;
        } else {
            for (int k = 0; k < 4 && j + k < n; k++) {
                float p = a.p[t][j + k], g = a.g[t][j + k], m = a.m[t][j + k], v = a.v[t][j + k];
                adam_one(p, g, m, v, clip, coef, w1, w2, beta2, eps, bc2_sqrt, ns);
                a.p[t][j + k] = p; a.m[t][j + k] = m; a.v[t][j + k] = v;
                if (clip && write_clipped_grad) a.g[t][j + k] = g;
            }
        }
    }
}

}  // namespace gsx

using namespace gsx;

GSX_API size_t gsx_clip_adam_workspace_bytes(void) { return (size_t)kNormBlocks * 8 + 256; }

// n_norm_partials > 0: the sums of squares are already there (norm_partials, that many doubles: the one-call trainer step's
// preprocess_bwd left one per workgroup) and the first launch is left out
int gsx::clip_adam_step_impl(int32_t n_tensors, float* const* params_host, float* const* grads_host,
                             float* const* exp_avg_host, float* const* exp_avg_sq_host,
                             const int64_t* numel_host, const float* lr_host,
                             float beta1, float beta2, float eps, int64_t step, float max_grad_norm,
                             int32_t write_clipped_grad, void* ws, size_t ws_bytes, const double* norm_partials, int32_t n_norm_partials,
                             void* stream)
{
    if (n_tensors <= 0 || n_tensors > kMaxTensors || step <= 0) return GSX_E_INVALID;
    if (!params_host || !grads_host || !exp_avg_host || !exp_avg_sq_host || !numel_host || !lr_host) return GSX_E_INVALID;
    if (max_grad_norm > 0.0f && (!ws || ws_bytes < gsx_clip_adam_workspace_bytes())) return GSX_E_INVALID;
    AdamArgs a;
    a.n_tensors = n_tensors; a.begin[0] = 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int t = 0; t < kMaxTensors; t++) {
        const bool on = t < n_tensors;
        if (on && (!params_host[t] || !grads_host[t] || !exp_avg_host[t] || !exp_avg_sq_host[t] || numel_host[t] < 0)) return GSX_E_INVALID;
        a.p[t] = on ? params_host[t] : nullptr; a.g[t] = on ? grads_host[t] : nullptr;
        a.m[t] = on ? exp_avg_host[t] : nullptr; a.v[t] = on ? exp_avg_sq_host[t] : nullptr;
        a.begin[t + 1] = a.begin[t] + (on ? numel_host[t] : 0);
        if (t == 0) a.qbegin[0] = 0;
        a.qbegin[t + 1] = a.qbegin[t] + (on ? (numel_host[t] + 3) / 4 : 0);
        a.vec[t] = on && ((((uintptr_t)a.p[t] | (uintptr_t)a.g[t] | (uintptr_t)a.m[t] | (uintptr_t)a.v[t]) & 15u) == 0);
        a.neg_step_size[t] = on ? (float)(-(double)lr_host[t] / bc1) : 0.0f;
    }
    const long long total = a.begin[n_tensors];
    if (total == 0) return GSX_OK;
    hipStream_t st = (hipStream_t)stream;
    const double* partials = nullptr;
    int n_partials = 0;
    float* norm_coef = nullptr;
    if (max_grad_norm > 0.0f) {
        norm_coef = (float*)((char*)ws + (size_t)kNormBlocks * 8);
        if (n_norm_partials > 0 && norm_partials) { partials = norm_partials; n_partials = n_norm_partials; }
        else {
            sumsq_kernel<<<dim3(kNormBlocks), dim3(kBlock), 0, st>>>(a, (double*)ws);
            partials = (const double*)ws; n_partials = kNormBlocks;
        }
    }
    const long long blocks = ceil_div(a.qbegin[n_tensors], kBlock);
    adam_kernel<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0, st>>>(
        a, partials, n_partials, max_grad_norm, norm_coef, write_clipped_grad, beta1, beta2, eps, (float)sqrt(bc2));
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_clip_adam_step(int32_t n_tensors, float* const* params_host, float* const* grads_host,
                               float* const* exp_avg_host, float* const* exp_avg_sq_host,
                               const int64_t* numel_host, const float* lr_host,
                               float beta1, float beta2, float eps, int64_t step, float max_grad_norm,
                               int32_t write_clipped_grad, void* ws, size_t ws_bytes, void* stream)
{
    return gsx::clip_adam_step_impl(n_tensors, params_host, grads_host, exp_avg_host, exp_avg_sq_host, numel_host, lr_host, beta1, beta2, eps,
                                    step, max_grad_norm, write_clipped_grad, ws, ws_bytes, nullptr, 0, stream);
}
