/* gsx_oracle.c -- CPU restatement of the reference Gaussian-splat render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gaussian-splatting_amd/ may import, link or call
 * this file; it is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker / reported baseline.  Plain C11 + OpenMP, fp32 like the reference, no torch.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors produced by
 * running the reference itself (tests/golden/make_golden.py imports
 * /root/reference/model/gaussian_splatting.py); see tests/test_oracle_golden.py.
 *
 * Two modes:
 *   dense  -- the reference algorithm literally: every Gaussian at every pixel, in descending
 *             depth order, global early exit (gaussian_splatting.py:70-154, 188-219).
 *   tiled  -- the algorithm the HIP kernels implement: footprint rectangles, (tile, rank)
 *             instances, sort, tile ranges, per-tile compositing with the same global early
 *             exit.  Its integer products (rects, tiles_touched, offsets, keys, ranges, k*)
 *             are what the GPU path is compared against BIT-EXACTLY; build with
 *             -ffp-contract=off so float expressions round exactly as written.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/gsx_constants.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * a5-a8: projection of one Gaussian (gaussian_splatting.py:221-344)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float mx, my, depth;       /* :256-257 */
    float cov[4];              /* :292  (row-major 2x2, NOT symmetrised) */
    float cam[3];
    float R[9], S3[9], J[4];   /* J = (j00, j11, j02, j12) :339-342 */
    float qn;                  /* max(|q|,1e-12) */
} proj_t;

static void project_one(const float* p, const float* s, const float* q, const float* M, proj_t* o)
{
    /* :250-253  cam = [p,1] @ M^T.  ATen's CPU bmm accumulates k = 0..3 as one FMA chain
     * (first product rounded, then fma, fma, + M[.][3]); bit-exact vs the golden vectors. */
    float cx = fmaf(p[2], M[2],  fmaf(p[1], M[1], p[0] * M[0])) + M[3];
    float cy = fmaf(p[2], M[6],  fmaf(p[1], M[5], p[0] * M[4])) + M[7];
    float cz = fmaf(p[2], M[10], fmaf(p[1], M[9], p[0] * M[8])) + M[11];
    o->cam[0] = cx; o->cam[1] = cy; o->cam[2] = cz;
    o->depth = cz;                       /* :256 */
    o->mx = cx / cz; o->my = cy / cz;    /* :257 */
    /* :307 F.normalize(q, dim=1): q / max(|q|, 1e-12) */
    float n = sqrtf(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    if (n < 1e-12f) n = 1e-12f;
    o->qn = n;
    float w = q[0]/n, x = q[1]/n, y = q[2]/n, z = q[3]/n;
    /* :310-314 */
    float* R = o->R;
    R[0] = 1 - 2*y*y - 2*z*z; R[1] = 2*x*y - 2*w*z;     R[2] = 2*x*z + 2*w*y;
    R[3] = 2*x*y + 2*w*z;     R[4] = 1 - 2*x*x - 2*z*z; R[5] = 2*y*z - 2*w*x;
    R[6] = 2*x*z - 2*w*y;     R[7] = 2*y*z + 2*w*x;     R[8] = 1 - 2*x*x - 2*y*y;
    /* :284-288  cov3d = (R S)(S R^T) */
    float RS[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RS[i*3+j] = R[i*3+j] * s[j];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++)
        o->S3[i*3+k] = RS[i*3+0]*RS[k*3+0] + RS[i*3+1]*RS[k*3+1] + RS[i*3+2]*RS[k*3+2];
    /* :332-342  J from WORLD xyz, fx = M[0][0], fy = M[1][1] */
    float fx = M[0], fy = M[5], zw = p[2];
    float j00 = fx / zw, j11 = fy / zw;
    float j02 = -fx * p[0] / (zw * zw), j12 = -fy * p[1] / (zw * zw);
    o->J[0] = j00; o->J[1] = j11; o->J[2] = j02; o->J[3] = j12;
    /* :292  cov2d = (J cov3d) J^T ; J rows: (j00, 0, j02), (0, j11, j12) */
    const float* C = o->S3;
    float a0[3], a1[3];
    for (int k = 0; k < 3; k++) { a0[k] = j00*C[0*3+k] + j02*C[2*3+k]; a1[k] = j11*C[1*3+k] + j12*C[2*3+k]; }
    o->cov[0] = a0[0]*j00 + a0[2]*j02;
    o->cov[1] = a0[1]*j11 + a0[2]*j12;
    o->cov[2] = a1[0]*j00 + a1[2]*j02;
    o->cov[3] = a1[1]*j11 + a1[2]*j12;
}

API void gsxo_project(const float* pos, const float* scl, const float* rot, const float* pose, int N,
                      float* mean2d, float* depth, float* cov2d)
{
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        proj_t o; project_one(pos + 3*i, scl + 3*i, rot + 4*i, pose, &o);
        mean2d[2*i] = o.mx; mean2d[2*i+1] = o.my; depth[i] = o.depth;
        memcpy(cov2d + 4*i, o.cov, 16);
    }
}

/* ------------------------------------------------------------------------------------------
 * a9: depth order.  torch.argsort(depths, descending=True) (:198) is not stable, so its tie
 * order is unspecified; this build defines: descending depth, ties by ascending index.
 * Key = bit pattern that sorts ascending (unsigned) in that order; -0.0 is folded onto +0.0.
 * ------------------------------------------------------------------------------------------ */
static inline uint32_t depth_key(float d)
{
    d = d + 0.0f;                         /* -0 -> +0 */
    uint32_t u; memcpy(&u, &d, 4);
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   /* ascending in float order */
    return ~asc;                                                   /* descending */
}
API void gsxo_depth_keys(const float* depth, int N, uint32_t* keys)
{ for (int i = 0; i < N; i++) keys[i] = depth_key(depth[i]); }

typedef struct { uint32_t k, i; } kv_t;
static int kv_cmp(const void* a, const void* b)
{
    const kv_t* x = a; const kv_t* y = b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i);
}
/* order[r] = index of the Gaussian composited r-th */
API void gsxo_depth_order(const float* depth, int N, uint32_t* order)
{
    kv_t* t = malloc(sizeof(kv_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { t[i].k = depth_key(depth[i]); t[i].i = (uint32_t)i; }
    qsort(t, N, sizeof(kv_t), kv_cmp);
    for (int i = 0; i < N; i++) order[i] = t[i].i;
    free(t);
}

/* ------------------------------------------------------------------------------------------
 * pixel grid: torch.linspace(-1, 1, steps) in fp32 (:100-102).  ATen's CPU kernel computes
 * step = (end-start)/(steps-1) and fills symmetrically: start + step*i for i < steps/2,
 * end - step*(steps-1-i) otherwise, each as ONE fused multiply-add (verified bit-exact against
 * torch 2.10 for every size in tests/golden and 1024/1080/1920).
 * ------------------------------------------------------------------------------------------ */
API void gsxo_linspace(int steps, float* out)
{
    if (steps == 1) { out[0] = -1.0f; return; }
    float step = (1.0f - (-1.0f)) / (float)(steps - 1);
    int half = steps / 2;
    for (int i = 0; i < steps; i++)
        out[i] = (i < half) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

/* ------------------------------------------------------------------------------------------
 * 2x2 inverse of cov2d + eps*I (:128-130).  torch.inverse on CPU is LAPACK getrf/getrs, i.e.
 * Gaussian elimination with partial pivoting; restated here in that form (closed-form adjugate
 * differs from it by cond(A)*eps on the ill-conditioned splats that z_world -> 0 produces).
 * Returns the product of the two pivots (sign-adjusted determinant).
 * ------------------------------------------------------------------------------------------ */
static float inv2x2(float a, float b, float c, float d, float* m)
{
    if (fabsf(c) > fabsf(a)) {            /* rows swapped: [c d; a b] */
        float l = a / c, u22 = b - l * d;
        float x2 = 1.0f / u22;            /* column 0 of P*I = (0,1) */
        m[2] = x2;       m[0] = (0.0f - d * x2) / c;
        x2 = -l / u22;                    /* column 1 of P*I = (1,0) */
        m[3] = x2;       m[1] = (1.0f - d * x2) / c;
        return -(c * u22);
    } else {
        float l = c / a, u22 = d - l * b;
        float x2 = -l / u22;              /* column 0 = (1,0) */
        m[2] = x2;       m[0] = (1.0f - b * x2) / a;
        x2 = 1.0f / u22;                  /* column 1 = (0,1) */
        m[3] = x2;       m[1] = (0.0f - b * x2) / a;
        return a * u22;
    }
}

/* ------------------------------------------------------------------------------------------
 * Rank-ordered splat records + footprint rectangles (tile units).  Shared by dense and tiled
 * mode; the rectangle arithmetic is the bit-exact contract with the HIP preprocess kernel.
 * rec layout: include/gsx_constants.h.  rect[r] = {x0,y0,x1,y1}, x1/y1 exclusive.
 * ------------------------------------------------------------------------------------------ */
#define NEG_HALF_LOG2E (-0.72134752f)

static void footprint(float mx, float my, float a, float d, float det, float op, int W, int H,
                      int32_t* rc /* tile rect x0,y0,x1,y1 (exclusive) */, int32_t* px /* pixel box xlo,ylo,xhi,yhi inclusive */)
{
    uint32_t ob; memcpy(&ob, &op, 4); ob &= 0x7fffffffu;
    int e = (int)(ob >> 23) - 127;                 /* |o| < 2^(e+1) */
    if (e < GSX_CUT_MIN_EXP_FOR(H, W)) e = GSX_CUT_MIN_EXP_FOR(H, W);  /* sized as if |o| were at least the floor: keeps d alpha / d opacity (gsx_constants.h) */
    int m = e + 1 + GSX_CUT_LOG2_FOR(H, W);
    rc[0] = rc[1] = rc[2] = rc[3] = 0;
    px[0] = px[1] = 1; px[2] = px[3] = 0;          /* empty box: lo > hi */
    if (ob < 0x7f800000u && m <= 0) return;                              /* |alpha| < 2^-CUT everywhere */
    int full = !(ob < 0x7f800000u);                                      /* inf / nan opacity */
    float kap2 = GSX_TWO_LN2 * (float)m;
    float ex = sqrtf(kap2 * a), ey = sqrtf(kap2 * d);
    if (!(a > 0.0f) || !(d > 0.0f) || !(det > 0.0f)) full = 1;          /* not positive definite */
    if (!(fabsf(mx) <= 3.0e38f) || !(fabsf(my) <= 3.0e38f) || !(ex <= 3.0e38f) || !(ey <= 3.0e38f)) full = 1;
    float xlo, xhi, ylo, yhi;
    if (full) { xlo = 0.0f; ylo = 0.0f; xhi = (float)(W - 1); yhi = (float)(H - 1); }
    else {
        float hx = 0.5f * (float)(W - 1), hy = 0.5f * (float)(H - 1);
        xlo = ceilf((mx - ex + 1.0f) * hx); xhi = floorf((mx + ex + 1.0f) * hx);
        ylo = ceilf((my - ey + 1.0f) * hy); yhi = floorf((my + ey + 1.0f) * hy);
        xlo = fmaxf(xlo, 0.0f); ylo = fmaxf(ylo, 0.0f);
        xhi = fminf(xhi, (float)(W - 1)); yhi = fminf(yhi, (float)(H - 1));
        if (!(xlo <= xhi) || !(ylo <= yhi)) return;                      /* off screen */
    }
    px[0] = (int)xlo; px[1] = (int)ylo; px[2] = (int)xhi; px[3] = (int)yhi;
    rc[0] = px[0] >> GSX_TILE_SHIFT; rc[1] = px[1] >> GSX_TILE_SHIFT;
    rc[2] = (px[2] >> GSX_TILE_SHIFT) + 1; rc[3] = (px[3] >> GSX_TILE_SHIFT) + 1;
}

API void gsxo_records(const float* mean2d, const float* depth, const float* cov2d,
                      const float* opacity, const float* features, const uint32_t* order,
                      int N, int H, int W, float* rec, int32_t* rect, uint32_t* tiles_touched)
{
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < N; r++) {
        uint32_t g = order[r];
        float a = cov2d[4*g] + GSX_COV_EPS, b = cov2d[4*g+1], c = cov2d[4*g+2], d = cov2d[4*g+3] + GSX_COV_EPS;
        float m[4]; float det = inv2x2(a, b, c, d, m);
        float* R = rec + (size_t)GSX_REC_FLOATS * r;
        R[0] = mean2d[2*g]; R[1] = mean2d[2*g+1];
        R[2] = m[0] * NEG_HALF_LOG2E; R[3] = (m[1] + m[2]) * NEG_HALF_LOG2E; R[4] = m[3] * NEG_HALF_LOG2E;
        R[5] = opacity[g]; R[6] = features[3*g]; R[7] = features[3*g+1]; R[8] = features[3*g+2];
        R[9] = depth[g];
        int32_t* rc = rect + 4*r; int32_t px[4];
        footprint(R[0], R[1], a, d, det, R[5], W, H, rc, px);
        uint32_t bx = (uint32_t)px[0] | ((uint32_t)px[2] << 16), by = (uint32_t)px[1] | ((uint32_t)px[3] << 16);
        memcpy(R + 10, &bx, 4); memcpy(R + 11, &by, 4);
        tiles_touched[r] = (uint32_t)((rc[2] - rc[0]) * (rc[3] - rc[1]));
    }
}

/* ------------------------------------------------------------------------------------------
 * a10 dense: the reference loop (:108-148).  Returns the number of Gaussians composited
 * (== number of early-exit tests the reference performs).
 * rgb planar (3,H,W), acc (H,W) = accumulated alpha.
 * ------------------------------------------------------------------------------------------ */
static inline float splat_alpha(const float* R, float gx, float gy)
{
    float dx = gx - R[0], dy = gy - R[1];
    float q = R[2]*dx*dx + R[3]*dx*dy + R[4]*dy*dy;     /* = -0.5*log2(e) * mahalanobis  (:131-134) */
    return R[5] * exp2f(q);                              /* :137,:140 */
}

API int64_t gsxo_render_dense(const float* rec, int N, int H, int W, float* rgb, float* acc)
{
    size_t P = (size_t)H * W;
    float* gx = malloc(sizeof(float) * W); float* gy = malloc(sizeof(float) * H);
    gsxo_linspace(W, gx); gsxo_linspace(H, gy);
    memset(rgb, 0, sizeof(float) * 3 * P); memset(acc, 0, sizeof(float) * P);
    int64_t done = 0;
    for (int r = 0; r < N; r++) {
        const float* R = rec + (size_t)GSX_REC_FLOATS * r;
        int all_sat = 1;
        #pragma omp parallel for schedule(static) reduction(&&:all_sat)
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
            size_t p = (size_t)y * W + x;
            float am = splat_alpha(R, gx[x], gy[y]);
            float T = 1.0f - acc[p];
            rgb[p]       = rgb[p]       + T * (R[6] * am);      /* :141-143 */
            rgb[P + p]   = rgb[P + p]   + T * (R[7] * am);
            rgb[2*P + p] = rgb[2*P + p] + T * (R[8] * am);
            acc[p] = acc[p] + T * am;                           /* :144 */
            all_sat = all_sat && (acc[p] > GSX_SAT_ALPHA);
        }
        done = r + 1;
        if (all_sat) break;                                     /* :147-148 */
    }
    free(gx); free(gy);
    return done;
}

/* ------------------------------------------------------------------------------------------
 * Backward of the compositing for ONE pixel over a list of ranks (derived; the reference has
 * no hand-written backward -- torch autograd over :124-144 is what the golden gradients pin).
 * list[0..K) are ranks in compositing order, all < rank_limit.  Accumulates, in double, the
 * per-record gradient d_rec[r][9] = (dmu_x, dmu_y, dk00, dkb, dk11, do, dr, dg, db) where
 * k** are the PRE-MULTIPLIED conic entries of the record.
 * ------------------------------------------------------------------------------------------ */
static void pixel_backward(const float* rec, const uint32_t* list, int K, float gx, float gy,
                           const float* Gbar /*3*/, double* d_rec /*[N][9]*/, float* tmpT, float* tmpA)
{
    /* forward in float exactly as rendered, remembering T_k and alpha_k */
    float A = 0.0f;
    for (int k = 0; k < K; k++) {
        const float* R = rec + (size_t)GSX_REC_FLOATS * list[k];
        float am = splat_alpha(R, gx, gy);
        float T = 1.0f - A;
        tmpT[k] = T; tmpA[k] = am;
        A = A + T * am;
    }
    /* exact reverse recurrence in double:  Wb = colour.Gbar accumulated behind k, per unit T_{k+1} */
    double Wb = 0.0;
    for (int k = K - 1; k >= 0; k--) {
        uint32_t r = list[k];
        const float* R = rec + (size_t)GSX_REC_FLOATS * r;
        double T = tmpT[k], am = tmpA[k];
        double cG = (double)R[6]*Gbar[0] + (double)R[7]*Gbar[1] + (double)R[8]*Gbar[2];
        double dalpha = T * (cG - Wb);
        Wb = am * cG + (1.0 - am) * Wb;
        double* D = d_rec + (size_t)9 * r;
        double w = T * am;
        double dx = (double)gx - R[0], dy = (double)gy - R[1];
        double e = (R[5] != 0.0f) ? am / R[5] : exp2((double)R[2]*dx*dx + (double)R[3]*dx*dy + (double)R[4]*dy*dy);
        double dq = dalpha * am * 0.6931471805599453;    /* d alpha / d q' = alpha * ln2 (q' is the exp2 argument) */
        #pragma omp atomic
        D[6] += w * Gbar[0];
        #pragma omp atomic
        D[7] += w * Gbar[1];
        #pragma omp atomic
        D[8] += w * Gbar[2];
        #pragma omp atomic
        D[5] += e * dalpha;
        #pragma omp atomic
        D[2] += dq * dx * dx;
        #pragma omp atomic
        D[3] += dq * dx * dy;
        #pragma omp atomic
        D[4] += dq * dy * dy;
        #pragma omp atomic
        D[0] += -dq * (2.0 * R[2] * dx + R[3] * dy);
        #pragma omp atomic
        D[1] += -dq * (R[3] * dx + 2.0 * R[4] * dy);
    }
}

/* dense backward: every pixel sees ranks 0..n_composited-1 */
API void gsxo_backward_dense(const float* rec, int N, int H, int W, int64_t n_composited,
                             const float* d_rgb /*3,H,W*/, double* d_rec /*N*9, zeroed here*/)
{
    size_t P = (size_t)H * W; int K = (int)n_composited;
    float* gx = malloc(sizeof(float) * W); float* gy = malloc(sizeof(float) * H);
    gsxo_linspace(W, gx); gsxo_linspace(H, gy);
    memset(d_rec, 0, sizeof(double) * 9 * (size_t)N);
    uint32_t* list = malloc(sizeof(uint32_t) * (size_t)(K > 0 ? K : 1));
    for (int k = 0; k < K; k++) list[k] = (uint32_t)k;
    #pragma omp parallel
    {
        float* tT = malloc(sizeof(float) * (size_t)(K > 0 ? K : 1)); float* tA = malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
        #pragma omp for schedule(dynamic, 4)
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
            size_t p = (size_t)y * W + x;
            float G[3] = { d_rgb[p], d_rgb[P + p], d_rgb[2*P + p] };
            pixel_backward(rec, list, K, gx[x], gy[y], G, d_rec, tT, tA);
        }
        free(tT); free(tA);
    }
    free(list); free(gx); free(gy);
}

/* ------------------------------------------------------------------------------------------
 * a11 preprocess backward (derived; autograd of :221-344 + the 2x2 inverse is the pin).
 * Inputs per rank r: d_rec[r][9] from compositing; per Gaussian: upstream gradients on the
 * returned positions (N,2), depths (N), scales=cov2d (N,2,2).  Output: gradients of the five
 * parameters (accumulated into zeroed arrays).  All in double.
 * ------------------------------------------------------------------------------------------ */
API void gsxo_preprocess_backward(const float* pos, const float* scl, const float* rot,
                                  const float* opacity, const float* features, const float* pose,
                                  const uint32_t* order, int N,
                                  const double* d_rec, const float* d_mean2d, const float* d_depth,
                                  const float* d_cov2d,
                                  double* g_pos, double* g_scl, double* g_rot, double* g_opa, double* g_fea,
                                  int64_t n_composited /* ranks >= this were never composited (global exit); < 0: all were */)
{
    (void)opacity; (void)features;
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < N; r++) {
        uint32_t g = order[r];
        const float* p = pos + 3*g; const float* s = scl + 3*g; const float* q = rot + 4*g; const float* M = pose;
        proj_t o; project_one(p, s, q, M, &o);
        const double* D = d_rec + (size_t)9 * r;
        g_opa[g] = D[5]; g_fea[3*g] = D[6]; g_fea[3*g+1] = D[7]; g_fea[3*g+2] = D[8];
        /* conic: k = c0 * (M00, M01+M10, M11), Minv = inverse(cov + eps I) */
        float a = o.cov[0] + GSX_COV_EPS, b = o.cov[1], c = o.cov[2], d = o.cov[3] + GSX_COV_EPS;
        float mi[4]; inv2x2(a, b, c, d, mi);
        double c0 = NEG_HALF_LOG2E;
        double dM[4] = { c0*D[2], c0*D[3], c0*D[3], c0*D[4] };
        /* dSigma = -M^T dM M^T */
        double Mt[4] = { mi[0], mi[2], mi[1], mi[3] };
        double t[4] = { Mt[0]*dM[0] + Mt[1]*dM[2], Mt[0]*dM[1] + Mt[1]*dM[3],
                        Mt[2]*dM[0] + Mt[3]*dM[2], Mt[2]*dM[1] + Mt[3]*dM[3] };
        double dC[4] = { -(t[0]*Mt[0] + t[1]*Mt[2]), -(t[0]*Mt[1] + t[1]*Mt[3]),
                         -(t[2]*Mt[0] + t[3]*Mt[2]), -(t[2]*Mt[1] + t[3]*Mt[3]) };
        /* a Gaussian behind the exit rank never reached the reference's loop (:118-148): its 2x2 inverse was never taken,
         * and no gradient flows through it -- also not 0 * inf when its covariance is singular (world z ~ 0) */
        if (n_composited >= 0 && r >= n_composited) dC[0] = dC[1] = dC[2] = dC[3] = 0.0;
        for (int i = 0; i < 4; i++) dC[i] += d_cov2d ? (double)d_cov2d[4*g+i] : 0.0;
        /* cov2d = J S3 J^T, J = [[j00,0,j02],[0,j11,j12]] */
        double J[6] = { o.J[0], 0, o.J[2], 0, o.J[1], o.J[3] };
        const float* S3 = o.S3;
        /* dS3 = J^T dC J */
        double dS3[9];
        for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) {
            double acc = 0;
            for (int a2 = 0; a2 < 2; a2++) for (int b2 = 0; b2 < 2; b2++) acc += J[a2*3+i] * dC[a2*2+b2] * J[b2*3+k];
            dS3[i*3+k] = acc;
        }
        /* dJ = dC J S3^T + dC^T J S3 */
        double dJ[6];
        for (int a2 = 0; a2 < 2; a2++) for (int i = 0; i < 3; i++) {
            double acc = 0;
            for (int b2 = 0; b2 < 2; b2++) for (int k = 0; k < 3; k++)
                acc += dC[a2*2+b2] * J[b2*3+k] * S3[i*3+k] + dC[b2*2+a2] * J[b2*3+k] * S3[k*3+i];
            dJ[a2*3+i] = acc;
        }
        double fx = M[0], fy = M[5], X = p[0], Y = p[1], Z = p[2];
        double gp[3] = {0, 0, 0};
        gp[0] += dJ[2] * (-fx / (Z*Z));
        gp[1] += dJ[5] * (-fy / (Z*Z));
        gp[2] += dJ[0] * (-fx / (Z*Z)) + dJ[4] * (-fy / (Z*Z)) + dJ[2] * (2*fx*X / (Z*Z*Z)) + dJ[5] * (2*fy*Y / (Z*Z*Z));
        /* S3 = (R S)(R S)^T : dRS = (dS3 + dS3^T) RS ; ds_j = sum_i dRS_ij R_ij ; dR_ij = dRS_ij s_j */
        const float* R = o.R;
        double dR[9];
        for (int j = 0; j < 3; j++) g_scl[3*g+j] = 0;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (dS3[i*3+k] + dS3[k*3+i]) * (double)R[k*3+j] * s[j];
            dR[i*3+j] = acc * s[j];
            g_scl[3*g+j] += acc * R[i*3+j];
        }
        /* R(q^): :310-314 */
        double n = o.qn, w = q[0]/n, x = q[1]/n, y = q[2]/n, z = q[3]/n;
        double dw = 2*(-z*dR[1] + y*dR[2] + z*dR[3] - x*dR[5] - y*dR[6] + x*dR[7]);
        double dx = 2*( y*dR[1] + z*dR[2] + y*dR[3] - 2*x*dR[4] - w*dR[5] + z*dR[6] + w*dR[7] - 2*x*dR[8]);
        double dy = 2*(-2*y*dR[0] + x*dR[1] + w*dR[2] + x*dR[3] + z*dR[5] - w*dR[6] + z*dR[7] - 2*y*dR[8]);
        double dz = 2*(-2*z*dR[0] - w*dR[1] + x*dR[2] + w*dR[3] - 2*z*dR[4] + y*dR[5] + x*dR[6] + y*dR[7]);
        /* q^ = q/n : dq = (dq^ - q^ (q^ . dq^)) / n */
        double dot = w*dw + x*dx + y*dy + z*dz;
        g_rot[4*g+0] = (dw - w*dot) / n; g_rot[4*g+1] = (dx - x*dot) / n;
        g_rot[4*g+2] = (dy - y*dot) / n; g_rot[4*g+3] = (dz - z*dot) / n;
        /* mu = cam.xy / cam.z ; depth = cam.z */
        double dmx = D[0] + (d_mean2d ? d_mean2d[2*g] : 0.0), dmy = D[1] + (d_mean2d ? d_mean2d[2*g+1] : 0.0);
        double cz = o.cam[2];
        double dcam[3] = { dmx / cz, dmy / cz,
                           -(dmx * o.cam[0] + dmy * o.cam[1]) / (cz*cz) + (d_depth ? d_depth[g] : 0.0) };
        for (int j = 0; j < 3; j++) gp[j] += dcam[0]*M[0*4+j] + dcam[1]*M[1*4+j] + dcam[2]*M[2*4+j];
        g_pos[3*g] = gp[0]; g_pos[3*g+1] = gp[1]; g_pos[3*g+2] = gp[2];
    }
}

/* ------------------------------------------------------------------------------------------
 * tiled mode: binning products (bit-exact contract with the HIP path)
 * ------------------------------------------------------------------------------------------ */
/* exclusive scan of tiles_touched (rank order); returns D */
API int64_t gsxo_scan(const uint32_t* tiles_touched, int N, uint64_t* offsets)
{
    uint64_t s = 0;
    for (int r = 0; r < N; r++) { offsets[r] = s; s += tiles_touched[r]; }
    return (int64_t)s;
}
/* emit instances in rank-major, row-major-tile order: tile_of[i], rank_of[i] */
API void gsxo_emit(const int32_t* rect, const uint64_t* offsets, int N, int tw, uint32_t* tile_of, uint32_t* rank_of)
{
    #pragma omp parallel for schedule(dynamic, 256)
    for (int r = 0; r < N; r++) {
        const int32_t* rc = rect + 4*r; uint64_t o = offsets[r];
        for (int ty = rc[1]; ty < rc[3]; ty++) for (int tx = rc[0]; tx < rc[2]; tx++) {
            tile_of[o] = (uint32_t)(ty * tw + tx); rank_of[o] = (uint32_t)r; o++;
        }
    }
}
/* stable counting sort by tile == sort of key64 = tile<<32 | rank.  Writes sorted keys,
 * sorted ranks and ranges[t] = {begin, end}. */
API void gsxo_sort_ranges(const uint32_t* tile_of, const uint32_t* rank_of, int64_t D, int n_tiles,
                          uint64_t* keys_sorted, uint32_t* ranks_sorted, uint32_t* ranges /*2*n_tiles*/)
{
    uint64_t* cnt = calloc((size_t)n_tiles + 1, sizeof(uint64_t));
    for (int64_t i = 0; i < D; i++) cnt[tile_of[i] + 1]++;
    for (int t = 0; t < n_tiles; t++) cnt[t+1] += cnt[t];
    for (int t = 0; t < n_tiles; t++) { ranges[2*t] = (uint32_t)cnt[t]; ranges[2*t+1] = (uint32_t)cnt[t+1]; }
    for (int64_t i = 0; i < D; i++) {
        uint64_t dst = cnt[tile_of[i]]++;
        ranks_sorted[dst] = rank_of[i];
        if (keys_sorted) keys_sorted[dst] = ((uint64_t)tile_of[i] << 32) | rank_of[i];
    }
    free(cnt);
}

/* max over the pixels of f_p(k) = the first rank k' >= k after which pixel p's accumulated alpha is > 0.99 (between the
 * instances of its tile a pixel keeps its value: "after rank k'" = after the tile's last instance of rank <= k');
 * GSX_RANK_NONE if some pixel has no such rank. */
static uint32_t next_all_saturated(const float* rec, int H, int W, const uint32_t* ranks_sorted, const uint32_t* ranges,
                                   uint32_t rank_limit, uint32_t k, const float* gx, const float* gy)
{
    int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE;
    uint32_t kmax = 0; int never = 0;
    #pragma omp parallel for schedule(dynamic, 1) reduction(max:kmax) reduction(||:never)
    for (int t = 0; t < tw * th; t++) {
        int tx = t % tw, ty = t / tw;
        for (int y = ty*GSX_TILE; y < (ty+1)*GSX_TILE && y < H; y++)
        for (int x = tx*GSX_TILE; x < (tx+1)*GSX_TILE && x < W; x++) {
            float A = 0; uint32_t f = GSX_RANK_NONE;
            uint32_t i = ranges[2*t];
            for (; i < ranges[2*t+1]; i++) {                    /* the state at rank k: every instance of rank <= k */
                uint32_t r = ranks_sorted[i];
                if (r >= rank_limit || r > k) break;
                A = A + (1.0f - A) * splat_alpha(rec + (size_t)GSX_REC_FLOATS * r, gx[x], gy[y]);
            }
            if (A > GSX_SAT_ALPHA) f = k;
            for (; f == GSX_RANK_NONE && i < ranges[2*t+1]; i++) {
                uint32_t r = ranks_sorted[i];
                if (r >= rank_limit) break;
                A = A + (1.0f - A) * splat_alpha(rec + (size_t)GSX_REC_FLOATS * r, gx[x], gy[y]);
                if (A > GSX_SAT_ALPHA) f = r;
            }
            if (f == GSX_RANK_NONE) never = 1; else if (f > kmax) kmax = f;
        }
    }
    return never ? GSX_RANK_NONE : kmax;
}

/* tiled forward: composite every tile's list up to rank_limit (exclusive; pass N for none).
 * Returns k* + 1 = number of ranks the reference would composite = the first k* after which EVERY pixel's accumulated
 * alpha is > 0.99 at once (gaussian_splatting.py:147-148), N if there is none.  While alpha stays in [0, 1] the
 * accumulated alpha only grows and k* is the max over pixels of the first rank after which acc > 0.99 (k0 below); an
 * opacity outside [0, 1] can take a pixel back under the threshold, and then k* is the fixpoint of
 * k <- max over pixels of the first rank >= k at which the pixel is saturated (non-decreasing, bounded by k*, and a
 * fixpoint satisfies the reference's test).  Round 2 returned k0 here and disagreed with dense mode in that regime. */
API int64_t gsxo_render_tiled(const float* rec, int N, int H, int W, const uint32_t* ranks_sorted,
                              const uint32_t* ranges, uint32_t rank_limit, float* rgb, float* acc, float* depth_map)
{
    size_t P = (size_t)H * W;
    int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE;
    float* gx = malloc(sizeof(float) * W); float* gy = malloc(sizeof(float) * H);
    gsxo_linspace(W, gx); gsxo_linspace(H, gy);
    uint32_t kmax = 0; int unsat = 0;
    #pragma omp parallel for schedule(dynamic, 1) reduction(max:kmax) reduction(||:unsat)
    for (int t = 0; t < tw * th; t++) {
        int tx = t % tw, ty = t / tw;
        for (int y = ty*GSX_TILE; y < (ty+1)*GSX_TILE && y < H; y++)
        for (int x = tx*GSX_TILE; x < (tx+1)*GSX_TILE && x < W; x++) {
            float C0 = 0, C1 = 0, C2 = 0, A = 0, Dm = 0; uint32_t sat = GSX_RANK_NONE;
            for (uint32_t i = ranges[2*t]; i < ranges[2*t+1]; i++) {
                uint32_t r = ranks_sorted[i];
                if (r >= rank_limit) break;
                const float* R = rec + (size_t)GSX_REC_FLOATS * r;
                float am = splat_alpha(R, gx[x], gy[y]);
                float T = 1.0f - A;
                C0 = C0 + T * (R[6] * am); C1 = C1 + T * (R[7] * am); C2 = C2 + T * (R[8] * am);
                Dm = Dm + T * (R[9] * am);
                A = A + T * am;
                if (sat == GSX_RANK_NONE && A > GSX_SAT_ALPHA) sat = r;
            }
            size_t p = (size_t)y * W + x;
            rgb[p] = C0; rgb[P+p] = C1; rgb[2*P+p] = C2; acc[p] = A;
            if (depth_map) depth_map[p] = Dm;
            if (sat == GSX_RANK_NONE) unsat = 1; else if (sat > kmax) kmax = sat;
        }
    }
    if (!unsat) {
        for (;;) {
            uint32_t k2 = next_all_saturated(rec, H, W, ranks_sorted, ranges, rank_limit, kmax, gx, gy);
            if (k2 == GSX_RANK_NONE) { unsat = 1; break; }
            if (k2 == kmax) break;
            kmax = k2;
        }
    }
    free(gx); free(gy);
    (void)N;
    return unsat ? (int64_t)N : (int64_t)kmax + 1;
}

API void gsxo_backward_tiled(const float* rec, int N, int H, int W, const uint32_t* ranks_sorted,
                             const uint32_t* ranges, uint32_t rank_limit, const float* d_rgb, double* d_rec)
{
    size_t P = (size_t)H * W;
    int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE;
    float* gx = malloc(sizeof(float) * W); float* gy = malloc(sizeof(float) * H);
    gsxo_linspace(W, gx); gsxo_linspace(H, gy);
    memset(d_rec, 0, sizeof(double) * 9 * (size_t)N);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < tw * th; t++) {
        int tx = t % tw, ty = t / tw;
        uint32_t b = ranges[2*t], e = ranges[2*t+1];
        int K = 0; while (b + K < e && ranks_sorted[b + K] < rank_limit) K++;
        if (K == 0) continue;
        float* tT = malloc(sizeof(float) * K); float* tA = malloc(sizeof(float) * K);
        for (int y = ty*GSX_TILE; y < (ty+1)*GSX_TILE && y < H; y++)
        for (int x = tx*GSX_TILE; x < (tx+1)*GSX_TILE && x < W; x++) {
            size_t p = (size_t)y * W + x;
            float G[3] = { d_rgb[p], d_rgb[P + p], d_rgb[2*P + p] };
            pixel_backward(rec, ranks_sorted + b, K, gx[x], gy[y], G, d_rec, tT, tA);
        }
        free(tT); free(tA);
    }
    free(gx); free(gy);
}

API int gsxo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
