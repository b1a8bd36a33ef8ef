// VALU issue rate of gfx950 as the compositing kernels see it (VERDICT r2 item 6 / "next" 1a): how many wave64 vector
// instructions per cycle one SIMD issues with 1, 2, 4, 8 waves resident, for the instruction kinds of chunk_fwd / chunk_bwd
// (v_fma_f32, the transcendentals, DPP adds, permlane swaps, s_nop, VALU beside SALU and beside LDS broadcast reads).
// Every CU gets exactly W workgroups of 256 threads (one wave per SIMD each): the dynamic-LDS request is sized so that W fit
// and W + 1 do not, and the grid is 256 * W.  Cycles come from s_memtime (shader clock) around the loop of lane 0 of each
// wave; the table prints cycles per wave-instruction per SIMD = (wave cycles) / (instructions of one wave) / W ... i.e. the
// SIMD's issue interval, next to the wall-clock rate (hipEvents) as a cross-check of the clock.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue_rate.bin tools/valu_issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIters = 2000;

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)

// 32 instructions per iteration in every body (independent unless the name says otherwise)
#define FMA16 \
    "v_fma_f32 %0, %0, %16, %0\n\t v_fma_f32 %1, %1, %16, %1\n\t v_fma_f32 %2, %2, %16, %2\n\t v_fma_f32 %3, %3, %16, %3\n\t" \
    "v_fma_f32 %4, %4, %16, %4\n\t v_fma_f32 %5, %5, %16, %5\n\t v_fma_f32 %6, %6, %16, %6\n\t v_fma_f32 %7, %7, %16, %7\n\t" \
    "v_fma_f32 %8, %8, %16, %8\n\t v_fma_f32 %9, %9, %16, %9\n\t v_fma_f32 %10, %10, %16, %10\n\t v_fma_f32 %11, %11, %16, %11\n\t" \
    "v_fma_f32 %12, %12, %16, %12\n\t v_fma_f32 %13, %13, %16, %13\n\t v_fma_f32 %14, %14, %16, %14\n\t v_fma_f32 %15, %15, %16, %15\n\t"
#define FMA14 \
    "v_fma_f32 %0, %0, %16, %0\n\t v_fma_f32 %1, %1, %16, %1\n\t v_fma_f32 %2, %2, %16, %2\n\t v_fma_f32 %3, %3, %16, %3\n\t" \
    "v_fma_f32 %4, %4, %16, %4\n\t v_fma_f32 %5, %5, %16, %5\n\t v_fma_f32 %6, %6, %16, %6\n\t v_fma_f32 %7, %7, %16, %7\n\t" \
    "v_fma_f32 %8, %8, %16, %8\n\t v_fma_f32 %9, %9, %16, %9\n\t v_fma_f32 %10, %10, %16, %10\n\t v_fma_f32 %11, %11, %16, %11\n\t" \
    "v_fma_f32 %12, %12, %16, %12\n\t v_fma_f32 %13, %13, %16, %13\n\t"
#define FMA8 \
    "v_fma_f32 %0, %0, %16, %0\n\t v_fma_f32 %1, %1, %16, %1\n\t v_fma_f32 %2, %2, %16, %2\n\t v_fma_f32 %3, %3, %16, %3\n\t" \
    "v_fma_f32 %4, %4, %16, %4\n\t v_fma_f32 %5, %5, %16, %5\n\t v_fma_f32 %6, %6, %16, %6\n\t v_fma_f32 %7, %7, %16, %7\n\t"
#define FMA4 "v_fma_f32 %0, %0, %16, %0\n\t v_fma_f32 %1, %1, %16, %1\n\t v_fma_f32 %2, %2, %16, %2\n\t v_fma_f32 %3, %3, %16, %3\n\t"
#define FMA4X "v_fma_f32 %0, %0, %17, %0\n\t v_fma_f32 %1, %1, %17, %1\n\t v_fma_f32 %2, %2, %17, %2\n\t v_fma_f32 %3, %3, %17, %3\n\t"
#define FMA4B "v_fma_f32 %4, %4, %16, %4\n\t v_fma_f32 %5, %5, %16, %5\n\t v_fma_f32 %6, %6, %16, %6\n\t v_fma_f32 %7, %7, %16, %7\n\t"
#define EXP16 \
    "v_exp_f32 %0, %0\n\t v_exp_f32 %1, %1\n\t v_exp_f32 %2, %2\n\t v_exp_f32 %3, %3\n\t v_exp_f32 %4, %4\n\t v_exp_f32 %5, %5\n\t v_exp_f32 %6, %6\n\t v_exp_f32 %7, %7\n\t" \
    "v_exp_f32 %8, %8\n\t v_exp_f32 %9, %9\n\t v_exp_f32 %10, %10\n\t v_exp_f32 %11, %11\n\t v_exp_f32 %12, %12\n\t v_exp_f32 %13, %13\n\t v_exp_f32 %14, %14\n\t v_exp_f32 %15, %15\n\t"
#define DPP16 \
    "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %10, %10, %10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t v_add_f32_dpp %11, %11, %11 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %12, %12, %12 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t v_add_f32_dpp %13, %13, %13 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
    "v_add_f32_dpp %14, %14, %14 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t v_add_f32_dpp %15, %15, %15 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
#define SWAP8 \
    "v_permlane32_swap_b32 %0, %1\n\t v_permlane32_swap_b32 %2, %3\n\t v_permlane32_swap_b32 %4, %5\n\t v_permlane32_swap_b32 %6, %7\n\t" \
    "v_permlane16_swap_b32 %8, %9\n\t v_permlane16_swap_b32 %10, %11\n\t v_permlane16_swap_b32 %12, %13\n\t v_permlane16_swap_b32 %14, %15\n\t"
#define PKFMA8 \
    "v_pk_fma_f32 %0, %0, %1, %0\n\t v_pk_fma_f32 %2, %2, %3, %2\n\t v_pk_fma_f32 %4, %4, %5, %4\n\t v_pk_fma_f32 %6, %6, %7, %6\n\t"
#define MISC16 \
    "v_ldexp_f32 %0, %0, 0\n\t v_frexp_mant_f32 %1, %1\n\t v_frexp_exp_i32_f32 %2, %3\n\t v_max_u32 %4, %4, %16\n\t" \
    "v_ldexp_f32 %5, %5, 0\n\t v_frexp_mant_f32 %6, %6\n\t v_frexp_exp_i32_f32 %7, %8\n\t v_max_u32 %9, %9, %16\n\t" \
    "v_rcp_f32 %10, %10\n\t v_cndmask_b32 %11, %11, %12, vcc\n\t v_mul_f32 %13, %13, %16\n\t v_sub_f32 %14, %14, %16\n\t" \
    "v_ldexp_f32 %15, %15, 0\n\t v_mul_f32 %0, %0, %16\n\t v_mul_f32 %1, %1, %16\n\t v_add_f32 %3, %3, %16\n\t"

enum Body { B_FMA, B_FMA_DEP, B_EXP, B_MIX_1EXP_IN_16, B_MIX_BWD, B_DPP, B_SWAP, B_FMA_NOP, B_FMA_SALU, B_FMA_BRANCH, B_FMA_LDS, B_MISC, B_PK_ADD, B_COUNT };
static const char* kNames[B_COUNT] = {
    "32 v_fma_f32 independent", "32 v_fma_f32 one dependent chain", "32 v_exp_f32 independent", "30 v_fma + 2 v_exp (1 in 16)",
    "26 v_fma + 2 v_exp + 2 v_rcp + 2 s_nop (bwd-like)", "32 v_add_f32_dpp independent", "16 fma + 16 v_permlane{32,16}_swap", "28 v_fma + 4 s_nop 0",
    "32 v_fma + 16 s_and_b64 (48 instr)", "32 v_fma + 8 (s_bitcmp + s_cbranch not taken) (48 instr)", "32 v_fma + 3 ds_read_b128 bcast + wait (36 instr)",
    "16 ldexp/frexp/max_u32/rcp/cndmask/mul + 16 fma", "32 v_pk_add_f32 independent (64 adds)" };
static const int kInstr[B_COUNT] = { 32, 32, 32, 32, 32, 32, 32, 32, 48, 48, 36, 32, 32 };

template <int BODY>
__global__ __launch_bounds__(256) void issue_kernel(float* out, uint64_t* cycles, float seed)
{
    extern __shared__ float4 lds[];
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = seed * (float)(threadIdx.x + i) * 1e-6f;
    float s = 1.0f - seed * 1e-7f;
    if (BODY == B_FMA_LDS) { lds[threadIdx.x & 63] = make_float4(seed, seed, seed, seed); __syncthreads(); }
    unsigned long long sm = 0x5555555555555555ull + (unsigned long long)(seed > 2.0f);
    float4 l0 = {0, 0, 0, 0}, l1 = l0, l2 = l0;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pk[8], pks = { s * 1e-3f, s * 2e-3f };
#pragma unroll
    for (int i = 0; i < 8; i++) pk[i] = v2f{ v[2 * i], v[2 * i + 1] };
    const uint32_t ldsaddr = 0;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < kIters; it++) {
        if (BODY == B_FMA) asm volatile(FMA16 FMA16 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_FMA_DEP) asm volatile(R16("v_fma_f32 %0, %0, %16, %0\n\t") R16("v_fma_f32 %0, %0, %16, %0\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_EXP) asm volatile(EXP16 EXP16 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_MIX_1EXP_IN_16) asm volatile(FMA14 "v_exp_f32 %14, %14\n\t v_fma_f32 %15, %15, %16, %15\n\t" FMA14 "v_exp_f32 %15, %15\n\t v_fma_f32 %14, %14, %16, %14\n\t" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_MIX_BWD) asm volatile(FMA8 "v_exp_f32 %14, %14\n\t" FMA4 "v_rcp_f32 %15, %15\n\t s_nop 0\n\t" FMA8 "v_exp_f32 %13, %13\n\t" FMA4B "v_rcp_f32 %12, %12\n\t s_nop 0\n\t v_fma_f32 %8, %8, %16, %8\n\t v_fma_f32 %9, %9, %16, %9\n\t" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_DPP) asm volatile(DPP16 DPP16 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_SWAP) asm volatile(SWAP8 FMA8 SWAP8 FMA8 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_FMA_NOP) asm volatile(FMA14 "s_nop 0\n\t s_nop 0\n\t" FMA14 "s_nop 0\n\t s_nop 0\n\t" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s));
        if (BODY == B_FMA_SALU) asm volatile(
            R8(FMA4X "s_and_b64 %16, %16, %16\n\t s_and_b64 %16, %16, %16\n\t")
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+s"(sm) : "v"(s) : "scc");
        if (BODY == B_FMA_BRANCH) asm volatile(
            "s_bitcmp1_b64 %16, 0\n\t s_cbranch_scc0 1f\n\t" FMA4X "1:\n\t"
            "s_bitcmp1_b64 %16, 2\n\t s_cbranch_scc0 2f\n\t" FMA4X "2:\n\t"
            "s_bitcmp1_b64 %16, 4\n\t s_cbranch_scc0 3f\n\t" FMA4X "3:\n\t"
            "s_bitcmp1_b64 %16, 6\n\t s_cbranch_scc0 4f\n\t" FMA4X "4:\n\t"
            "s_bitcmp1_b64 %16, 8\n\t s_cbranch_scc0 5f\n\t" FMA4X "5:\n\t"
            "s_bitcmp1_b64 %16, 10\n\t s_cbranch_scc0 6f\n\t" FMA4X "6:\n\t"
            "s_bitcmp1_b64 %16, 12\n\t s_cbranch_scc0 7f\n\t" FMA4X "7:\n\t"
            "s_bitcmp1_b64 %16, 14\n\t s_cbranch_scc0 8f\n\t" FMA4X "8:\n\t"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+s"(sm) : "v"(s) : "scc");
        if (BODY == B_FMA_LDS) asm volatile(
            "ds_read_b128 %17, %20\n\t ds_read_b128 %18, %20 offset:16\n\t ds_read_b128 %19, %20 offset:32\n\t" FMA16 FMA16 "s_waitcnt lgkmcnt(0)\n\t"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(s), "=&v"(l0), "=&v"(l1), "=&v"(l2) : "v"(ldsaddr) : "memory");
        if (BODY == B_PK_ADD) asm volatile(R4("v_pk_add_f32 %0, %0, %8\n\t v_pk_add_f32 %1, %1, %8\n\t v_pk_add_f32 %2, %2, %8\n\t v_pk_add_f32 %3, %3, %8\n\t"
                                              "v_pk_add_f32 %4, %4, %8\n\t v_pk_add_f32 %5, %5, %8\n\t v_pk_add_f32 %6, %6, %8\n\t v_pk_add_f32 %7, %7, %8\n\t")
                                           : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4]), "+v"(pk[5]), "+v"(pk[6]), "+v"(pk[7]) : "v"(pks));
        if (BODY == B_MISC) asm volatile(MISC16 FMA16 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(s) : "vcc");
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float acc = l0.x + l1.y + l2.z + (float)(sm & 1ull);
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i];
    if (BODY == B_PK_ADD) { for (int i = 0; i < 8; i++) acc += pk[i].x + pk[i].y; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int BODY>
static int run_body(float* out, uint64_t* cyc_d, int n_cu, double* clock_mhz_out)
{
    const int waves[4] = { 1, 2, 4, 8 };
    const int lds_bytes[4] = { 100 * 1024, 60 * 1024, 36 * 1024, 18 * 1024 };        // W fit in 160 KB, W + 1 do not
    printf("%-62s", kNames[BODY]);
    for (int wi = 0; wi < 4; wi++) {
        const int W = waves[wi];
        CK(hipFuncSetAttribute((const void*)issue_kernel<BODY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes[wi]));
        const int grid = n_cu * W;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        issue_kernel<BODY><<<grid, 256, lds_bytes[wi]>>>(out, cyc_d, 1.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        issue_kernel<BODY><<<grid, 256, lds_bytes[wi]>>>(out, cyc_d, 1.5f);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> cyc((size_t)grid * 4);
        CK(hipMemcpy(cyc.data(), cyc_d, cyc.size() * 8, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        const double med = (double)cyc[cyc.size() / 2];
        // the SIMD's issue interval: W waves share it, each issuing kInstr * kIters instructions in `med` cycles
        const double per_instr = med / ((double)kInstr[BODY] * kIters * W);
        // wall-clock cross-check: instructions per SIMD / wall time -> instructions per microsecond per SIMD
        const double wall_rate = (double)kInstr[BODY] * kIters * W / (ms * 1e3);
        printf(" | W=%d: %5.2f cyc/instr (%6.0f instr/us/SIMD)", W, per_instr, wall_rate);
        if (BODY == B_FMA && W == 4) *clock_mhz_out = med / (ms * 1e3);
    }
    printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz; cyc = s_memtime ticks (__builtin_readcyclecounter) per wave-instruction per SIMD\n", prop.name, n_cu, prop.clockRate / 1000);
    float* out; uint64_t* cyc;
    CK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * 4)); CK(hipMalloc(&cyc, (size_t)n_cu * 8 * 4 * 8));
    double mhz = 0;
    if (run_body<B_FMA>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_FMA_DEP>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_EXP>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_MIX_1EXP_IN_16>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_MIX_BWD>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_DPP>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_SWAP>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_FMA_NOP>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_FMA_SALU>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_FMA_BRANCH>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_FMA_LDS>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_MISC>(out, cyc, n_cu, &mhz)) return 1;
    if (run_body<B_PK_ADD>(out, cyc, n_cu, &mhz)) return 1;
    printf("counter ticks per microsecond of wall time (fma body, W=4): %.0f\n", mhz);
    return 0;
}
