// Host-side cost of enqueueing K dependent small kernels: K hipLaunchKernel calls vs one hipGraphLaunch of the same K nodes
// (captured from the stream).  Build: hipcc --offload-arch=gfx950 -O2 -o graph_launch_cost graph_launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void bump(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
int main()
{
    const int n = 1 << 16, K = 20, reps = 200;
    float* d; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemset(d, 0, n * sizeof(float)));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int k = 0; k < K; k++) bump<<<n / 256, 256, 0, st>>>(d, n);
    CK(hipStreamSynchronize(st));
    double host_direct = 0, wall_direct = 0;
    for (int r = 0; r < reps; r++) {
        auto t0 = now();
        for (int k = 0; k < K; k++) bump<<<n / 256, 256, 0, st>>>(d, n);
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        host_direct += us(t0, t1); wall_direct += us(t0, t2);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; k++) bump<<<n / 256, 256, 0, st>>>(d, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    double host_graph = 0, wall_graph = 0;
    for (int r = 0; r < reps; r++) {
        auto t0 = now();
        CK(hipGraphLaunch(ge, st));
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        host_graph += us(t0, t1); wall_graph += us(t0, t2);
    }
    printf("K=%d kernels: direct host %.1f us, wall %.1f us | graph host %.1f us, wall %.1f us\n", K,
           host_direct / reps, wall_direct / reps, host_graph / reps, wall_graph / reps);
    return 0;
}
