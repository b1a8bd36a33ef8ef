// What rocprofv3's FETCH_SIZE / WRITE_SIZE count for the access shapes of the compositing kernels: a coalesced stream of
// float4s, a gather of 48-byte records at random indices (one record per lane, three float4 loads), the same at a 64-byte
// stride, and 36-byte rows written at random indices.  Every kernel touches a known number of bytes once (arrays far larger than
// the L2s and the Infinity Cache are not needed: each byte is read once, cold); run it under
//   rocprofv3 --pmc FETCH_SIZE -- tools/fetch_size_probe.bin      and      rocprofv3 --pmc WRITE_SIZE -- tools/fetch_size_probe.bin
// and compare the counters (KB) with the bytes printed here.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/fetch_size_probe.bin tools/fetch_size_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream_read(const float4* __restrict__ a, size_t n, float* __restrict__ out)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
template <int STRIDE4>   // record stride in float4s (3 = 48 bytes, 4 = 64 bytes); 3 float4s of each record are read
__global__ void gather_records(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, size_t n, float* __restrict__ out)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* r = rec + (size_t)STRIDE4 * idx[i];
        const float4 a = r[0], b = r[1], c = r[2];
        s += a.x + b.y + c.z;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void scatter_rows(float* __restrict__ rows, const uint32_t* __restrict__ idx, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n * 9; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / 9, c = i - r * 9;
        rows[(size_t)idx[r] * 9 + c] = (float)c;          // nine consecutive lanes write one 36-byte row
    }
}

int main()
{
    const size_t n = 4u << 20;                                  // 4 Mi records / rows
    std::vector<uint32_t> h(n); std::iota(h.begin(), h.end(), 0u);
    std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
    uint32_t* idx; float4 *rec48, *rec64, *big; float *rows, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&rec48, n * 48)); CK(hipMalloc(&rec64, n * 64)); CK(hipMalloc(&big, n * 64));
    CK(hipMalloc(&rows, n * 36)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(rec48, 0, n * 48)); CK(hipMemset(rec64, 0, n * 64)); CK(hipMemset(big, 0, n * 64)); CK(hipMemset(rows, 0, n * 36));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        stream_read<<<2048, 256>>>(big, n * 4, out);
        gather_records<3><<<2048, 256>>>(rec48, idx, n, out);
        gather_records<4><<<2048, 256>>>(rec64, idx, n, out);
        scatter_rows<<<2048, 256>>>(rows, idx, n);
    }
    CK(hipDeviceSynchronize());
    printf("bytes touched per launch: stream_read %zu KB | gather_records<3> %zu KB of records (+ %zu KB of indices) | gather_records<4> %zu KB "
           "(48 of every 64 bytes) + indices | scatter_rows %zu KB of rows (+ %zu KB of indices read)\n",
           n * 64 / 1024, n * 48 / 1024, n * 4 / 1024, n * 48 / 1024, n * 36 / 1024, n * 4 / 1024);
    return 0;
}
