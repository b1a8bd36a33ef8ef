/* Sanitizer run of the CPU oracle (test infrastructure): drives every exported function of gsx_oracle.c once over a
 * small synthetic scene with exactly-sized heap buffers, built with -fsanitize=address,undefined (oracle/Makefile,
 * target `san`; tests/test_sanitizers.py runs it).  Prints a checksum; any sanitizer report makes the run fail. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gsx_oracle.c"

static uint32_t lcg_state = 12345u;
static float urand(float lo, float hi)
{
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return lo + (hi - lo) * (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
}

static int run(int N, int H, int W, float s_lo, float s_hi, float o_lo, float o_hi)
{
    size_t n = (size_t)(N > 0 ? N : 1), P = (size_t)H * W;
    float *pos = malloc(12 * n), *scl = malloc(12 * n), *rot = malloc(16 * n), *opa = malloc(4 * n), *fea = malloc(12 * n);
    for (int i = 0; i < N; i++) {
        for (int c = 0; c < 3; c++) { pos[3*i+c] = urand(-1, 1); scl[3*i+c] = urand(s_lo, s_hi); fea[3*i+c] = urand(0, 1); }
        float q[4] = { 1.0f, urand(-0.3f, 0.3f), urand(-0.3f, 0.3f), urand(-0.3f, 0.3f) };
        float nq = sqrtf(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
        for (int c = 0; c < 4; c++) rot[4*i+c] = q[c] / nq;
        opa[i] = urand(o_lo, o_hi);
    }
    float pose[16] = { 1,0,0,0, 0,1,0,0, 0,0,1,3, 0,0,0,1 };
    float *mean2d = malloc(8 * n), *depth = malloc(4 * n), *cov2d = malloc(16 * n);
    gsxo_project(pos, scl, rot, pose, N, mean2d, depth, cov2d);
    uint32_t *keys = malloc(4 * n), *order = malloc(4 * n);
    gsxo_depth_keys(depth, N, keys); gsxo_depth_order(depth, N, order);
    float* rec = malloc(sizeof(float) * GSX_REC_FLOATS * n); int32_t* rect = malloc(16 * n); uint32_t* tt = malloc(4 * n);
    gsxo_records(mean2d, depth, cov2d, opa, fea, order, N, H, W, rec, rect, tt);
    uint64_t* offsets = malloc(8 * n);
    int64_t D = gsxo_scan(tt, N, offsets);
    int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE, n_tiles = tw * th;
    size_t d = (size_t)(D > 0 ? D : 1);
    uint32_t *tile_of = malloc(4 * d), *rank_of = malloc(4 * d), *ranks_sorted = malloc(4 * d), *ranges = malloc(8 * (size_t)n_tiles);
    uint64_t* keys_sorted = malloc(8 * d);
    gsxo_emit(rect, offsets, N, tw, tile_of, rank_of);
    gsxo_sort_ranges(tile_of, rank_of, D, n_tiles, keys_sorted, ranks_sorted, ranges);
    float *rgb = malloc(12 * P), *acc = malloc(4 * P), *rgb_t = malloc(12 * P), *acc_t = malloc(4 * P), *dm = malloc(4 * P);
    int64_t kd = gsxo_render_dense(rec, N, H, W, rgb, acc);
    int64_t kt = gsxo_render_tiled(rec, N, H, W, ranks_sorted, ranges, (uint32_t)N, rgb_t, acc_t, dm);
    if (kt < N) gsxo_render_tiled(rec, N, H, W, ranks_sorted, ranges, (uint32_t)kt, rgb_t, acc_t, dm);   /* second pass after an exit */
    float* d_rgb = malloc(12 * P);
    for (size_t i = 0; i < 3 * P; i++) d_rgb[i] = urand(-1, 1);
    double *dr_d = malloc(72 * n), *dr_t = malloc(72 * n);
    gsxo_backward_dense(rec, N, H, W, kd, d_rgb, dr_d);
    gsxo_backward_tiled(rec, N, H, W, ranks_sorted, ranges, (uint32_t)(kt < N ? kt : N), d_rgb, dr_t);
    double *g_pos = calloc(3 * n, 8), *g_scl = calloc(3 * n, 8), *g_rot = calloc(4 * n, 8), *g_opa = calloc(n, 8), *g_fea = calloc(3 * n, 8);
    float *u_m = calloc(2 * n, 4), *u_d = calloc(n, 4), *u_c = calloc(4 * n, 4);
    gsxo_preprocess_backward(pos, scl, rot, opa, fea, pose, order, N, dr_t, u_m, u_d, u_c, g_pos, g_scl, g_rot, g_opa, g_fea, -1);
    float* ls = malloc(4 * (size_t)W); gsxo_linspace(W, ls);
    double sum = 0, diff = 0;
    for (size_t i = 0; i < 3 * P; i++) { sum += rgb[i]; diff += fabs((double)rgb[i] - rgb_t[i]); }
    for (int i = 0; i < 3 * N; i++) sum += g_pos[i];
    printf("N %d %dx%d: D %lld, composited dense %lld tiled %lld, checksum %.6e, |dense - tiled| %.3e, threads %d\n",
           N, W, H, (long long)D, (long long)kd, (long long)kt, sum, diff, gsxo_num_threads());
    free(pos); free(scl); free(rot); free(opa); free(fea); free(mean2d); free(depth); free(cov2d); free(keys); free(order);
    free(rec); free(rect); free(tt); free(offsets); free(tile_of); free(rank_of); free(ranks_sorted); free(ranges);
    free(keys_sorted); free(rgb); free(acc); free(rgb_t); free(acc_t); free(dm); free(d_rgb); free(dr_d); free(dr_t);
    free(g_pos); free(g_scl); free(g_rot); free(g_opa); free(g_fea); free(u_m); free(u_d); free(u_c); free(ls);
    return sum == sum ? 0 : 1;
}

int main(void)
{
    int bad = 0;
    bad |= run(300, 24, 32, 0.001f, 0.1f, 0.0f, 1.0f);          /* default-init-like: early exit fires */
    bad |= run(500, 33, 17, 0.005f, 0.03f, -0.3f, 1.4f);        /* odd image, out-of-range opacities */
    bad |= run(1, 1, 5, 0.05f, 0.2f, 0.5f, 1.0f);               /* one Gaussian, one row */
    bad |= run(64, 16, 16, 0.5f, 2.0f, 0.9f, 1.0f);             /* screen-filling splats */
    return bad;
}
