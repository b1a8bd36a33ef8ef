// gsx_preprocess.hip -- per-Gaussian kernels of the render path (one thread per Gaussian):
//   project_fwd     a5-a8 of SURVEY 8a: camera transform, quaternion -> R, Sigma3, Jacobian, Sigma2
//                   (/root/reference/model/gaussian_splatting.py:221-344) + the depth sort key
//   build_records   rank-ordered splat records: 2x2 inverse (:128-130), conic, footprint box
//   preprocess_bwd  autograd of all of the above, written by hand
// HBM-bound: every thread touches ~150-250 B once; no reuse, no LDS, no MFMA.
// Compiled with -ffp-contract=off: the float expressions below must round exactly like
// oracle/gsx_oracle.c (tile boxes are compared bit-exactly), so every fused multiply-add is
// spelled fmaf() and nothing else may be contracted.
#include "gsx_common.h"

namespace gsx {

struct Proj {
    float mx, my, depth;
    float cov[4];
    float cam[3];
    float R[9], S3[9], J[4];
    float qn;
};

__device__ __forceinline__ void project_one(const float p[3], const float s[3], const float q[4], const float* __restrict__ M, Proj& o)
{
    // cam = [p,1] @ M^T as ATen's bmm accumulates it: one rounded product, two fmas, + M[.][3]
    float cx = fmaf(p[2], M[2],  fmaf(p[1], M[1], p[0] * M[0])) + M[3];
    float cy = fmaf(p[2], M[6],  fmaf(p[1], M[5], p[0] * M[4])) + M[7];
    float cz = fmaf(p[2], M[10], fmaf(p[1], M[9], p[0] * M[8])) + M[11];
    o.cam[0] = cx; o.cam[1] = cy; o.cam[2] = cz;
    o.depth = cz;
    o.mx = cx / cz; o.my = cy / cz;
    float n = sqrtf(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    if (n < 1e-12f) n = 1e-12f;
    o.qn = n;
    float w = q[0]/n, x = q[1]/n, y = q[2]/n, z = q[3]/n;
    float* R = o.R;
    R[0] = 1 - 2*y*y - 2*z*z; R[1] = 2*x*y - 2*w*z;     R[2] = 2*x*z + 2*w*y;
    R[3] = 2*x*y + 2*w*z;     R[4] = 1 - 2*x*x - 2*z*z; R[5] = 2*y*z - 2*w*x;
    R[6] = 2*x*z - 2*w*y;     R[7] = 2*y*z + 2*w*x;     R[8] = 1 - 2*x*x - 2*y*y;
    float RS[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) RS[i*3+j] = R[i*3+j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            o.S3[i*3+k] = RS[i*3+0]*RS[k*3+0] + RS[i*3+1]*RS[k*3+1] + RS[i*3+2]*RS[k*3+2];
    float fx = M[0], fy = M[5], zw = p[2];
    float j00 = fx / zw, j11 = fy / zw;
    float j02 = -fx * p[0] / (zw * zw), j12 = -fy * p[1] / (zw * zw);
    o.J[0] = j00; o.J[1] = j11; o.J[2] = j02; o.J[3] = j12;
    const float* C = o.S3;
    float a0[3], a1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { a0[k] = j00*C[0*3+k] + j02*C[2*3+k]; a1[k] = j11*C[1*3+k] + j12*C[2*3+k]; }
    o.cov[0] = a0[0]*j00 + a0[2]*j02;
    o.cov[1] = a0[1]*j11 + a0[2]*j12;
    o.cov[2] = a1[0]*j00 + a1[2]*j02;
    o.cov[3] = a1[1]*j11 + a1[2]*j12;
}

__device__ __forceinline__ uint32_t depth_key(float d)
{
    d = d + 0.0f;
    uint32_t u = __float_as_uint(d);
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

// Gaussian elimination with partial pivoting, same operation order as the oracle
__device__ __forceinline__ float inv2x2(float a, float b, float c, float d, float m[4])
{
    if (fabsf(c) > fabsf(a)) {
        float l = a / c, u22 = b - l * d;
        float x2 = 1.0f / u22;
        m[2] = x2; m[0] = (0.0f - d * x2) / c;
        x2 = -l / u22;
        m[3] = x2; m[1] = (1.0f - d * x2) / c;
        return -(c * u22);
    } else {
        float l = c / a, u22 = d - l * b;
        float x2 = -l / u22;
        m[2] = x2; m[0] = (1.0f - b * x2) / a;
        x2 = 1.0f / u22;
        m[3] = x2; m[1] = (0.0f - b * x2) / a;
        return a * u22;
    }
}

// footprint box in pixels (bit-exact contract with oracle footprint()); empty: xlo > xhi.  a, d: the covariance's diagonal WITH
// GSX_COV_EPS, det: inv2x2's determinant of it
__device__ __forceinline__ void footprint_box(float2 mu, float a, float d, float det, float op, int H, int W,
                                              int& xlo, int& ylo, int& xhi, int& yhi)
{
    uint32_t ob = __float_as_uint(op) & 0x7fffffffu;
    const int e = max((int)(ob >> 23) - 127, GSX_CUT_MIN_EXP_FOR(H, W));
    const int mm = e + 1 + GSX_CUT_LOG2_FOR(H, W);
    xlo = 1; ylo = 1; xhi = 0; yhi = 0;
    if (!(ob < 0x7f800000u && mm <= 0)) {
        bool full = !(ob < 0x7f800000u);
        const float kap2 = GSX_TWO_LN2 * (float)mm;
        const float ex = sqrtf(kap2 * a), ey = sqrtf(kap2 * d);
        if (!(a > 0.0f) || !(d > 0.0f) || !(det > 0.0f)) full = true;
        if (!(fabsf(mu.x) <= 3.0e38f) || !(fabsf(mu.y) <= 3.0e38f) || !(ex <= 3.0e38f) || !(ey <= 3.0e38f)) full = true;
        if (full) { xlo = 0; ylo = 0; xhi = W - 1; yhi = H - 1; }
        else {
            const float hx = 0.5f * (float)(W - 1), hy = 0.5f * (float)(H - 1);
            float fxlo = ceilf((mu.x - ex + 1.0f) * hx), fxhi = floorf((mu.x + ex + 1.0f) * hx);
            float fylo = ceilf((mu.y - ey + 1.0f) * hy), fyhi = floorf((mu.y + ey + 1.0f) * hy);
            fxlo = fmaxf(fxlo, 0.0f); fylo = fmaxf(fylo, 0.0f);
            fxhi = fminf(fxhi, (float)(W - 1)); fyhi = fminf(fyhi, (float)(H - 1));
            if ((fxlo <= fxhi) && (fylo <= fyhi)) { xlo = (int)fxlo; ylo = (int)fylo; xhi = (int)fxhi; yhi = (int)fyhi; }
        }
    }
}

__global__ __launch_bounds__(kBlock) void project_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ scl,
                                                             const float* __restrict__ rot, const float* __restrict__ opacity,
                                                             const float* __restrict__ features, const float* __restrict__ pose, int N,
                                                             float* __restrict__ mean2d, float* __restrict__ depth,
                                                             float* __restrict__ cov2d, uint32_t* __restrict__ key,
                                                             float4* __restrict__ pre, int H, int W, uint64_t* __restrict__ vis)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < N;
    if (!live && !vis) return;
    bool visible = false;
    if (live) {
    float M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = pose[k];
    const float p[3] = { pos[3*i], pos[3*i+1], pos[3*i+2] };
    const float s[3] = { scl[3*i], scl[3*i+1], scl[3*i+2] };
    const float4 qv = reinterpret_cast<const float4*>(rot)[i];
    const float q[4] = { qv.x, qv.y, qv.z, qv.w };
    Proj o; project_one(p, s, q, M, o);
    if (mean2d) {       // the reference's returned tensors: a caller that never looks at them (a trainer step with an image-only criterion) passes none
        // (streaming stores: nothing in the view reads these 28 bytes per Gaussian back)
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f2{o.mx, o.my}, reinterpret_cast<f2*>(mean2d) + i);
        __builtin_nontemporal_store(o.depth, depth + i);
        __builtin_nontemporal_store(f4{o.cov[0], o.cov[1], o.cov[2], o.cov[3]}, reinterpret_cast<f4*>(cov2d) + i);
    }
    key[i] = depth_key(o.depth);
    const float op = opacity[i];
    visible = true;
    if (vis) {
        // does the splat touch the image at all?  build_records' own test (footprint_box on the numbers the packed row would carry)
        const float a = o.cov[0] + GSX_COV_EPS, b = o.cov[1], c = o.cov[2], d = o.cov[3] + GSX_COV_EPS;
        float m[4]; const float det = inv2x2(a, b, c, d, m);
        int xlo, ylo, xhi, yhi;
        footprint_box(make_float2(o.mx, o.my), a, d, det, op, H, W, xlo, ylo, xhi, yhi);
        visible = xlo <= xhi;
    }
    // packed copy of everything build_records needs: ONE 48-byte row to gather per Gaussian instead of
    // five scattered lines (mean2d, depth, cov2d, opacity, features); with `vis`, only for those it will gather
    if (visible) {
        float4* P = pre + (size_t)3 * i;
        P[0] = make_float4(o.mx, o.my, o.cov[0], o.cov[1]);
        P[1] = make_float4(o.cov[2], o.cov[3], o.depth, op);
        P[2] = make_float4(features[3*i], features[3*i+1], features[3*i+2], 0.0f);
    }
    }
    if (vis) {
        const unsigned long long bits = __ballot(visible);               // (whole waves reach this: i is wave-aligned)
        if ((threadIdx.x & 63) == 0 && (i >> 6) < (N + 63) / 64) vis[i >> 6] = bits;
    }
}

// tt_sums (nullable): per-workgroup sum of tiles_touched, so that the scan that follows is one launch (gsx_bin_count)
__device__ __forceinline__ void block_sum_tiles(uint32_t tt, uint64_t* __restrict__ tt_sums)
{
    __shared__ uint32_t s_tt[kBlock / 64];
    uint32_t s = tt;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);          // <= 64 * (4096^2 tiles) would wrap: summed per wave in
    if ((threadIdx.x & 63) == 0) s_tt[threadIdx.x >> 6] = s;              // u32 only for images up to 8192 tiles (checked on the host)
    __syncthreads();
    if (threadIdx.x == 0) tt_sums[blockIdx.x] = ((uint64_t)s_tt[0] + s_tt[1]) + ((uint64_t)s_tt[2] + s_tt[3]);
}

// one rank: gathers the Gaussian's packed row, writes its record, footprint and inverse-order entry; returns tiles_touched.
// vis (nullable; the one-call trainer step): one bit per GAUSSIAN, set by the projection kernel for those whose footprint box is not
// empty (the same function on the same numbers) -- a rank whose bit is clear gets its empty footprint without the 48-byte gather and
// no record (nothing reads the record of a splat without instances); on the bench orbit that is 45 % of the ranks
__device__ __forceinline__ uint32_t build_record_one(const float4* __restrict__ pre, const uint32_t* __restrict__ order, int r,
                                                     int H, int W, float* __restrict__ rec, uint32_t* __restrict__ tiles_touched,
                                                     uint32_t* __restrict__ rank_of, uint32_t* __restrict__ tile_box,
                                                     const uint64_t* __restrict__ vis)
{
    const uint32_t g = order[r];
    if (vis && !((vis[g >> 6] >> (g & 63u)) & 1ull)) {       // (not even its rank_of entry: the backward asks the bitmap first)
        tiles_touched[r] = 0u;
        if (tile_box) tile_box[r] = 0x00000001u;
        return 0u;
    }
    rank_of[g] = (uint32_t)r;                                // (a 4-byte scatter: the most expensive store of this kernel)
    const float4 p0 = pre[(size_t)3 * g], p1 = pre[(size_t)3 * g + 1], p2 = pre[(size_t)3 * g + 2];
    const float2 mu = make_float2(p0.x, p0.y);
    const float a = p0.z + GSX_COV_EPS, b = p0.w, c = p1.x, d = p1.y + GSX_COV_EPS;
    float m[4]; const float det = inv2x2(a, b, c, d, m);
    const float op = p1.w;
    int xlo, ylo, xhi, yhi;
    footprint_box(mu, a, d, det, op, H, W, xlo, ylo, xhi, yhi);
    uint32_t tt = 0;
    if (xlo <= xhi) tt = (uint32_t)(((xhi >> GSX_TILE_SHIFT) + 1 - (xlo >> GSX_TILE_SHIFT)) * ((yhi >> GSX_TILE_SHIFT) + 1 - (ylo >> GSX_TILE_SHIFT)));
    tiles_touched[r] = tt;
    // the footprint in TILE units, one word per rank (images of up to 256 tiles a side): what the emit pass reads instead of
    // pulling every 48-byte record through the cache for 8 bytes of it.  x0 | y0 << 8 | (x1 - 1) << 16 | (y1 - 1) << 24; empty: x0 > x1 - 1
    if (tile_box) tile_box[r] = tt ? ((uint32_t)(xlo >> GSX_TILE_SHIFT) | ((uint32_t)(ylo >> GSX_TILE_SHIFT) << 8) |
                                      ((uint32_t)(xhi >> GSX_TILE_SHIFT) << 16) | ((uint32_t)(yhi >> GSX_TILE_SHIFT) << 24)) : 0x00000001u;
    const uint32_t bx = (uint32_t)xlo | ((uint32_t)xhi << 16), by = (uint32_t)ylo | ((uint32_t)yhi << 16);
    float4* R = reinterpret_cast<float4*>(rec) + (size_t)kRecVec * r;
    R[0] = make_float4(mu.x, mu.y, m[0] * GSX_NEG_HALF_LOG2E, (m[1] + m[2]) * GSX_NEG_HALF_LOG2E);
    R[1] = make_float4(m[3] * GSX_NEG_HALF_LOG2E, op, p2.x, p2.y);
    R[2] = make_float4(p2.z, p1.z, __uint_as_float(bx), __uint_as_float(by));
    return tt;
}

__global__ __launch_bounds__(kBlock) void build_records_kernel(const float4* __restrict__ pre, const uint32_t* __restrict__ order,
                                                               int N, int H, int W, float* __restrict__ rec,
                                                               uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ rank_of,
                                                               uint64_t* __restrict__ tt_sums, uint32_t* __restrict__ tile_box,
                                                               const uint64_t* __restrict__ vis)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t tt = (r < N) ? build_record_one(pre, order, r, H, W, rec, tiles_touched, rank_of, tile_box, vis) : 0u;
    if (tt_sums) block_sum_tiles(tt, tt_sums);          // every thread of the workgroup reaches the barrier inside
}

// ------------------------------------------------------------------------------------------
// backward: thread per GAUSSIAN (parameters, upstream gradients and outputs are coalesced; only the
// splat's offset/count and its partial rows are gathered through rank_of).  Sums the partial rows in
// a fixed order, then chain rule.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void preprocess_bwd_kernel(
    const float* __restrict__ pos, const float* __restrict__ scl, const float* __restrict__ rot,
    const float* __restrict__ pose, const uint32_t* __restrict__ rank_of,
    const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ tiles_touched,
    const float* __restrict__ partials, const uint32_t* __restrict__ state,
    const float* __restrict__ d_mean2d, const float* __restrict__ d_depth, const float* __restrict__ d_cov2d,
    int N, int accumulate,
    float* __restrict__ g_pos, float* __restrict__ g_scl, float* __restrict__ g_rot, float* __restrict__ g_opa, float* __restrict__ g_fea,
    const double* __restrict__ loss_partials, long long loss_n, float* __restrict__ loss, double* __restrict__ norm_partials,
    const uint64_t* __restrict__ vis)
{
    if (loss && blockIdx.x == gridDim.x - 1) {
        // the one-call trainer step's loss value: gsx_loss.hip's loss_final_kernel, same order of summation (same bits), by
        // the last workgroup of the step's last backward kernel instead of a launch of its own
        __shared__ double s_v[kBlock];
        double s = 0.0;
        for (int i = threadIdx.x; i < kLossBlocks; i += kBlock) s += loss_partials[i];
        s_v[threadIdx.x] = s;
        __syncthreads();
        for (int d = kBlock / 2; d > 0; d >>= 1) { if (threadIdx.x < d) s_v[threadIdx.x] += s_v[threadIdx.x + d]; __syncthreads(); }
        if (threadIdx.x == 0) loss[0] = (float)(s_v[0] / (double)loss_n);
    }
    int g = blockIdx.x * kBlock + threadIdx.x;              // N is rounded up to whole waves below: no early return before the shuffles
    const int lane = threadIdx.x & 63;
    const bool live = g < N;
    // vis (the one-call trainer step): the projection kernel's bitmap of the Gaussians whose footprint touches the image; the others
    // have no rows, and the records stage left no rank for them
    const bool seen = live && (!vis || ((vis[g >> 6] >> (g & 63)) & 1ull));
    const uint32_t r = seen ? rank_of[g] : 0u;
    float D[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t limit = (uint32_t)N;
    if (state) { const uint32_t s = state[0]; if (s != GSX_RANK_NONE && s + 1u < (uint32_t)N) limit = s + 1u; }
    uint32_t o0 = 0, tt = 0;
    // the splat's row count is the next offset minus its own (same 64-byte line 15 times in 16): one random gather, not two
    if (seen && partials && r < limit) { o0 = offsets[r]; tt = (r + 1u < (uint32_t)N) ? offsets[r + 1u] - o0 : tiles_touched[r]; }
    // rows of a splat: summed by its own lane in tile order when there are few; a splat with more than kRowsSolo rows
    // is summed by the whole wavefront (lane l takes rows l, l+64, ... in order, then a fixed butterfly): a
    // screen-sized splat would otherwise serialise thousands of loads in one lane.  Both orders are fixed, so the
    // result stays bitwise reproducible.
    constexpr uint32_t kRowsSolo = 32;
    if (tt <= kRowsSolo) {
        for (uint32_t t = 0; t < tt; t++) {
            const float* row = partials + (size_t)9 * (o0 + t);
#pragma unroll
            for (int c = 0; c < 9; c++) D[c] += row[c];
        }
    }
    unsigned long long big = __ballot(tt > kRowsSolo);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t bo = (uint32_t)__shfl((int)o0, src, 64), bt = (uint32_t)__shfl((int)tt, src, 64);
        float S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t t = (uint32_t)lane; t < bt; t += 64) {
            const float* row = partials + (size_t)9 * (bo + t);
#pragma unroll
            for (int c = 0; c < 9; c++) S[c] += row[c];
        }
#pragma unroll
        for (int c = 0; c < 9; c++) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) S[c] += __shfl_xor(S[c], d, 64);
            if (lane == src) D[c] = S[c];
        }
    }
    if (!live && !norm_partials) return;
    const int gl = g;                          // (stores)
    g = live ? g : N - 1;                      // a thread past the end only takes part in the workgroup's sum of squares below
    float M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = pose[k];
    const float p[3] = { pos[3*g], pos[3*g+1], pos[3*g+2] };
    const float s[3] = { scl[3*g], scl[3*g+1], scl[3*g+2] };
    const float4 qv = reinterpret_cast<const float4*>(rot)[g];
    const float q[4] = { qv.x, qv.y, qv.z, qv.w };
    Proj o; project_one(p, s, q, M, o);
    // D: 0 S(dq dx) 1 S(dq dy) (raw moments, chain rule below) 2 dk00 3 dkb 4 dk11 5 d(opacity) 6..8 d(colour)
    // conic -> covariance: dSigma = -M^T dM M^T, dM = c0 * [[D2, D3],[D3, D4]]
    const float a = o.cov[0] + GSX_COV_EPS, b = o.cov[1], c = o.cov[2], d = o.cov[3] + GSX_COV_EPS;
    float mi[4]; inv2x2(a, b, c, d, mi);
    const float c0 = GSX_NEG_HALF_LOG2E;
    const float dM[4] = { c0*D[2], c0*D[3], c0*D[3], c0*D[4] };
    const float Mt[4] = { mi[0], mi[2], mi[1], mi[3] };
    const float t4[4] = { Mt[0]*dM[0] + Mt[1]*dM[2], Mt[0]*dM[1] + Mt[1]*dM[3],
                          Mt[2]*dM[0] + Mt[3]*dM[2], Mt[2]*dM[1] + Mt[3]*dM[3] };
    float dC[4] = { -(t4[0]*Mt[0] + t4[1]*Mt[2]), -(t4[0]*Mt[1] + t4[1]*Mt[3]),
                    -(t4[2]*Mt[0] + t4[3]*Mt[2]), -(t4[2]*Mt[1] + t4[3]*Mt[3]) };
    // a Gaussian behind the exit rank never reached the reference's loop (gaussian_splatting.py:118-148): its 2x2 inverse was
    // never taken and no gradient flows through it -- not 0 * inf either, when its covariance is singular (world z ~ 0)
    if (r >= limit) { dC[0] = 0.0f; dC[1] = 0.0f; dC[2] = 0.0f; dC[3] = 0.0f; }
    if (d_cov2d) {
        const float4 u = reinterpret_cast<const float4*>(d_cov2d)[g];
        dC[0] += u.x; dC[1] += u.y; dC[2] += u.z; dC[3] += u.w;
    }
    // cov2d = J S3 J^T, J = [[j00,0,j02],[0,j11,j12]]
    const float J[6] = { o.J[0], 0.0f, o.J[2], 0.0f, o.J[1], o.J[3] };
    const float* S3 = o.S3;
    float dS3[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float acc = 0;
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
                for (int b2 = 0; b2 < 2; b2++) acc += J[a2*3+i] * dC[a2*2+b2] * J[b2*3+k];
            dS3[i*3+k] = acc;
        }
    float dJ[6];
#pragma unroll
    for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            float acc = 0;
#pragma unroll
            for (int b2 = 0; b2 < 2; b2++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    acc += dC[a2*2+b2] * J[b2*3+k] * S3[i*3+k] + dC[b2*2+a2] * J[b2*3+k] * S3[k*3+i];
            dJ[a2*3+i] = acc;
        }
    const float fx = M[0], fy = M[5], X = p[0], Y = p[1], Z = p[2];
    const float iz2 = 1.0f / (Z * Z);
    float gp[3];
    gp[0] = dJ[2] * (-fx * iz2);
    gp[1] = dJ[5] * (-fy * iz2);
    gp[2] = dJ[0] * (-fx * iz2) + dJ[4] * (-fy * iz2) + dJ[2] * (2.0f * fx * X * iz2 / Z) + dJ[5] * (2.0f * fy * Y * iz2 / Z);
    // S3 = (R S)(R S)^T
    const float* R = o.R;
    float dR[9], gs[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float acc = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) acc += (dS3[i*3+k] + dS3[k*3+i]) * R[k*3+j] * s[j];
            dR[i*3+j] = acc * s[j];
            gs[j] += acc * R[i*3+j];
        }
    const float n = o.qn, w = q[0]/n, x = q[1]/n, y = q[2]/n, z = q[3]/n;
    const float dw = 2*(-z*dR[1] + y*dR[2] + z*dR[3] - x*dR[5] - y*dR[6] + x*dR[7]);
    const float dx = 2*( y*dR[1] + z*dR[2] + y*dR[3] - 2*x*dR[4] - w*dR[5] + z*dR[6] + w*dR[7] - 2*x*dR[8]);
    const float dy = 2*(-2*y*dR[0] + x*dR[1] + w*dR[2] + x*dR[3] + z*dR[5] - w*dR[6] + z*dR[7] - 2*y*dR[8]);
    const float dz = 2*(-2*z*dR[0] - w*dR[1] + x*dR[2] + w*dR[3] - 2*z*dR[4] + y*dR[5] + x*dR[6] + y*dR[7]);
    const float dot = w*dw + x*dx + y*dy + z*dz;
    float gq[4] = { (dw - w*dot) / n, (dx - x*dot) / n, (dy - y*dot) / n, (dz - z*dot) / n };
    // mu = cam.xy / cam.z ; depth = cam.z
    // rows 0, 1 of the backward's partial rows are the raw moments S(dq dx), S(dq dy); the conic is a per-splat constant, so
    // the chain rule to the mean is taken once per Gaussian here, not once per (tile, splat) row there
    float dmx = 0.0f, dmy = 0.0f;
    if (tt > 0u) {
        const float k00 = mi[0] * c0, kb = (mi[1] + mi[2]) * c0, k11 = mi[3] * c0;       // the record's conic (build_record_one)
        dmx = -GSX_LN2 * (2.0f * k00 * D[0] + kb * D[1]);
        dmy = -GSX_LN2 * (kb * D[0] + 2.0f * k11 * D[1]);
    }
    if (d_mean2d) { const float2 u = reinterpret_cast<const float2*>(d_mean2d)[g]; dmx += u.x; dmy += u.y; }
    const float cz = o.cam[2];
    const float dcam[3] = { dmx / cz, dmy / cz, -(dmx * o.cam[0] + dmy * o.cam[1]) / (cz * cz) + (d_depth ? d_depth[g] : 0.0f) };
#pragma unroll
    for (int j = 0; j < 3; j++) gp[j] += dcam[0]*M[0*4+j] + dcam[1]*M[1*4+j] + dcam[2]*M[2*4+j];
    if (live) {
        if (accumulate) {
#pragma unroll
            for (int j = 0; j < 3; j++) { g_pos[3*gl+j] += gp[j]; g_scl[3*gl+j] += gs[j]; g_fea[3*gl+j] += D[6+j]; }
#pragma unroll
            for (int j = 0; j < 4; j++) g_rot[4*gl+j] += gq[j];
            g_opa[gl] += D[5];
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++) { g_pos[3*gl+j] = gp[j]; g_scl[3*gl+j] = gs[j]; g_fea[3*gl+j] = D[6+j]; }
            reinterpret_cast<float4*>(g_rot)[gl] = make_float4(gq[0], gq[1], gq[2], gq[3]);
            g_opa[gl] = D[5];
        }
    }
    if (norm_partials) {
        // the clip norm's sum of squares over what this workgroup just wrote (accumulate == 0), in double like sumsq_kernel:
        // the one-call trainer step's optimizer reads these instead of the 56 N bytes of gradients
        __shared__ double s_nw[kBlock / 64];
        double sq = 0.0;
        if (live) {
#pragma unroll
            for (int j = 0; j < 3; j++) sq += (double)gp[j] * (double)gp[j] + (double)gs[j] * (double)gs[j] + (double)D[6+j] * (double)D[6+j];
#pragma unroll
            for (int j = 0; j < 4; j++) sq += (double)gq[j] * (double)gq[j];
            sq += (double)D[5] * (double)D[5];
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sq += __shfl_down(sq, d, 64);
        if (lane == 0) s_nw[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) norm_partials[blockIdx.x] = (s_nw[0] + s_nw[1]) + (s_nw[2] + s_nw[3]);
    }
}

}  // namespace gsx

using namespace gsx;

GSX_API int gsx_project_fwd(const float* positions, const float* scales, const float* rotations,
                            const float* opacity, const float* features, const float* pose, int32_t N,
                            float* mean2d, float* depth, float* cov2d, uint32_t* depth_key, float* pre, void* stream)
{
    return gsx::project_fwd_impl(positions, scales, rotations, opacity, features, pose, N, mean2d, depth, cov2d, depth_key, pre, 0, 0, nullptr, stream);
}

int gsx::project_fwd_impl(const float* positions, const float* scales, const float* rotations,
                          const float* opacity, const float* features, const float* pose, int32_t N,
                          float* mean2d, float* depth, float* cov2d, uint32_t* depth_key, float* pre,
                          int32_t H, int32_t W, uint64_t* vis, void* stream)
{
    if (N < 0) return GSX_E_INVALID;
    if (vis && (H <= 0 || W <= 0 || ((uintptr_t)vis & 7u))) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!positions || !scales || !rotations || !opacity || !features || !pose || !depth_key || !pre) return GSX_E_INVALID;
    if ((mean2d || depth || cov2d) && !(mean2d && depth && cov2d)) return GSX_E_INVALID;      // all three outputs, or none
    if (((uintptr_t)rotations | (uintptr_t)cov2d | (uintptr_t)pre) & 15u) return GSX_E_INVALID;
    if ((uintptr_t)mean2d & 7u) return GSX_E_INVALID;
    project_fwd_kernel<<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        positions, scales, rotations, opacity, features, pose, N, mean2d, depth, cov2d, depth_key, (float4*)pre, H, W, vis);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_build_records(const float* pre, const uint32_t* order, int32_t N, int32_t H, int32_t W,
                              float* rec, uint32_t* tiles_touched, uint32_t* rank_of, uint64_t* tt_block_sums, uint32_t* tile_box,
                              void* stream)
{
    return gsx::build_records_impl(pre, order, N, H, W, rec, tiles_touched, rank_of, tt_block_sums, tile_box, nullptr, stream);
}

int gsx::build_records_impl(const float* pre, const uint32_t* order, int32_t N, int32_t H, int32_t W,
                            float* rec, uint32_t* tiles_touched, uint32_t* rank_of, uint64_t* tt_block_sums, uint32_t* tile_box,
                            const uint64_t* vis, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!pre || !order || !rec || !tiles_touched || !rank_of) return GSX_E_INVALID;
    if (((uintptr_t)rec | (uintptr_t)pre) & 15u) return GSX_E_INVALID;
    // without the packed tile boxes the emit pass reads the footprint from the RECORD: every rank needs one
    if (vis && !(tile_box && W <= 256 * GSX_TILE && H <= 256 * GSX_TILE)) return GSX_E_INVALID;
    // the per-wave partial sums are u32: 64 splats x n_tiles must fit
    if (tt_block_sums && (int64_t)((W + GSX_TILE - 1) / GSX_TILE) * ((H + GSX_TILE - 1) / GSX_TILE) > (int64_t)(0xFFFFFFFFu / 64u)) return GSX_E_RANGE;
    build_records_kernel<<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        (const float4*)pre, order, N, H, W, rec, tiles_touched, rank_of, tt_block_sums,
        (W <= 256 * GSX_TILE && H <= 256 * GSX_TILE) ? tile_box : nullptr, vis);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_preprocess_bwd(const float* positions, const float* scales, const float* rotations,
                               const float* pose, const uint32_t* rank_of, const uint32_t* offsets,
                               const uint32_t* tiles_touched, const float* partials, const uint32_t* state,
                               const float* d_mean2d, const float* d_depth, const float* d_cov2d,
                               int32_t N, int32_t accumulate,
                               float* g_positions, float* g_scales, float* g_rotations, float* g_opacity,
                               float* g_features, void* stream)
{
    if (N < 0) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!positions || !scales || !rotations || !pose || !rank_of) return GSX_E_INVALID;
    if (partials && (!offsets || !tiles_touched)) return GSX_E_INVALID;
    if (!g_positions || !g_scales || !g_rotations || !g_opacity || !g_features) return GSX_E_INVALID;
    if (((uintptr_t)rotations | (uintptr_t)g_rotations) & 15u) return GSX_E_INVALID;
    if (d_cov2d && ((uintptr_t)d_cov2d & 15u)) return GSX_E_INVALID;
    preprocess_bwd_kernel<<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        positions, scales, rotations, pose, rank_of, offsets, tiles_touched, partials, state,
        d_mean2d, d_depth, d_cov2d, N, accumulate, g_positions, g_scales, g_rotations, g_opacity, g_features, nullptr, 0, nullptr, nullptr, nullptr);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

// the same for the one-call trainer step (no upstream gradients besides the image's, no accumulation), with the image
// loss's final reduction riding on the last workgroup (gsx_common.h)
int gsx::preprocess_bwd_with_loss(const float* positions, const float* scales, const float* rotations,
                                  const float* pose, const uint32_t* rank_of, const uint32_t* offsets,
                                  const uint32_t* tiles_touched, const float* partials, const uint32_t* state,
                                  int32_t N, float* g_positions, float* g_scales, float* g_rotations, float* g_opacity,
                                  float* g_features, const double* loss_partials, int64_t loss_n, float* loss, double* norm_partials,
                                  const uint64_t* vis, void* stream)
{
    if (N <= 0) return GSX_E_INVALID;
    if (!positions || !scales || !rotations || !pose || !rank_of || !offsets || !tiles_touched || !partials) return GSX_E_INVALID;
    if (!g_positions || !g_scales || !g_rotations || !g_opacity || !g_features) return GSX_E_INVALID;
    if (((uintptr_t)rotations | (uintptr_t)g_rotations) & 15u) return GSX_E_INVALID;
    if (loss && (!loss_partials || loss_n <= 0)) return GSX_E_INVALID;
    preprocess_bwd_kernel<<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        positions, scales, rotations, pose, rank_of, offsets, tiles_touched, partials, state,
        nullptr, nullptr, nullptr, N, 0, g_positions, g_scales, g_rotations, g_opacity, g_features, loss_partials, (long long)loss_n, loss,
        norm_partials, vis);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}
