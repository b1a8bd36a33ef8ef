// gsx_merge.hip -- pair search of merge_nearby_gaussians (SURVEY 8f row N3;
// /root/reference/model/gaussian_splatting.py:491-497: torch.cdist(positions, positions) and
// where(triu(d > 0) & triu(d < threshold)) -- an N x N matrix, 4 TB at 1 M Gaussians) as a spatial hash on the GPU:
// cell edge just above the threshold, every point hashed into one of M >= 2N buckets, points sorted by bucket (the radix
// sort of gsx_sort.hip), and each point probes the 27 cells around its own for partners with a larger index.
// Two passes over the grid (count, then fill at the scanned offsets): pairs come out grouped by their first index i in
// ascending order -- the reference's row-major order once each group is sorted by j -- and the greedy merge that follows
// (order-dependent by definition, :500-515) walks them on the host (gsx_merge_greedy_host below).
// Distances are the direct form sqrt((dx*dx + dy*dy) + dz*dz) in fp32.  torch.cdist switches to a matmul form
// (|a|^2 + |b|^2 - 2ab) above 25 rows, whose rounding (~3e-4 at |a| ~ 1) decides pairs within that distance of the
// threshold or of zero differently -- including "pairs" of a point with itself; parity at that rounding is unpinned.
// HBM/latency-bound integer and gather work; no MFMA.
#include <cmath>
#include "gsx_common.h"

namespace gsx {

int sort_pairs_internal(const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int64_t, const uint64_t*, int, int,
                        void*, size_t, hipStream_t, int check_bit = 32, uint64_t* mismatch_dev = nullptr);
size_t sort_ws_bytes_internal(int64_t n);

struct MergeGrid { int bits; uint32_t n_buckets; float inv_cell; };

static MergeGrid merge_grid(int32_t N, float threshold)
{
    MergeGrid g;
    g.bits = 10;
    while (g.bits < 24 && (1ll << g.bits) < 2ll * (N > 0 ? N : 1)) g.bits++;
    g.n_buckets = 1u << g.bits;
    g.inv_cell = 1.0f / (threshold * 1.001f);          // a little wider than the threshold: partners are at most one cell away
    return g;
}

__device__ __forceinline__ int cell_of(float x, float inv_cell)
{
    const float c = floorf(x * inv_cell);
    return (c >= -1.0e9f && c <= 1.0e9f) ? (int)c : 0;       // non-finite or absurd: anywhere (its distances fail the test)
}
__device__ __forceinline__ uint32_t bucket_of(int cx, int cy, int cz, uint32_t mask)
{
    return (((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u) ^ ((uint32_t)cz * 83492791u)) & mask;
}

__global__ __launch_bounds__(kBlock) void merge_keys_kernel(const float* __restrict__ pos, int N, float inv_cell, uint32_t mask,
                                                            uint32_t* __restrict__ keys)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    keys[i] = bucket_of(cell_of(pos[3 * i], inv_cell), cell_of(pos[3 * i + 1], inv_cell), cell_of(pos[3 * i + 2], inv_cell), mask);
}

// start[b] = first sorted position whose bucket is >= b (b = 0 .. n_buckets): a binary search per bucket
__global__ __launch_bounds__(kBlock) void merge_bucket_start_kernel(const uint32_t* __restrict__ keys_sorted, int N, uint32_t n_buckets,
                                                                    uint32_t* __restrict__ start)
{
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
    if (b > n_buckets) return;
    int lo = 0, hi = N;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys_sorted[mid] < b) lo = mid + 1; else hi = mid; }
    start[b] = (uint32_t)lo;
}

// one thread per point i: partners j > i with 0 < d(i, j) < threshold; FILL = false counts them, true writes them at
// offsets[i] and sorts them
template <bool FILL>
__global__ __launch_bounds__(kBlock) void merge_probe_kernel(const float* __restrict__ pos, int N, float threshold, float inv_cell, uint32_t mask,
                                                             const uint32_t* __restrict__ order, const uint32_t* __restrict__ start,
                                                             uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ partners)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    const int cx = cell_of(x, inv_cell), cy = cell_of(y, inv_cell), cz = cell_of(z, inv_cell);
    uint32_t n = 0;
    uint32_t* out = FILL ? partners + offsets[i] : nullptr;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                const uint32_t b = bucket_of(nx, ny, nz, mask);
                for (uint32_t s = start[b], e = start[b + 1]; s < e; s++) {
                    const uint32_t j = order[s];
                    if (j <= (uint32_t)i) continue;
                    const float px = pos[3 * j], py = pos[3 * j + 1], pz = pos[3 * j + 2];
                    // several of the 27 cells may share a bucket: a partner counts in the cell it lives in, once
                    if (cell_of(px, inv_cell) != nx || cell_of(py, inv_cell) != ny || cell_of(pz, inv_cell) != nz) continue;
                    const float ex = x - px, ey = y - py, ez = z - pz;
                    const float d = sqrtf((ex * ex + ey * ey) + ez * ez);
                    if (d > 0.0f && d < threshold) { if (FILL) out[n] = j; n++; }
                }
            }
    if (!FILL) { counts[i] = n; return; }
    // ascending partners: the 27 cells are visited in hash order, the reference's pairs run row-major.  A handful per
    // point in any scene worth merging (a scene with thousands of partners per point has N^2 pairs to store anyway).
    for (uint32_t a = 1; a < n; a++) {
        const uint32_t v = out[a];
        uint32_t b = a;
        for (; b > 0 && out[b - 1] > v; b--) out[b] = out[b - 1];
        out[b] = v;
    }
}

}  // namespace gsx

using namespace gsx;

// ws: keys | keys_sorted | order | keys_tmp | values_tmp (N u32 each) | sort workspace | bucket starts (n_buckets + 1)
static size_t merge_ws_layout(int32_t N, float threshold, size_t off[7])
{
    const MergeGrid g = merge_grid(N, threshold);
    const size_t n = (size_t)(N > 0 ? N : 1);
    size_t at = 0;
    for (int k = 0; k < 5; k++) { off[k] = at; at += align_up(n * 4); }
    off[5] = at; at += align_up(sort_ws_bytes_internal((int64_t)n));
    off[6] = at; at += align_up(((size_t)g.n_buckets + 1) * 4);
    return at;
}

GSX_API size_t gsx_merge_workspace_bytes(int32_t N)
{
    size_t off[7];
    return merge_ws_layout(N, 1.0f, off);
}

GSX_API int gsx_merge_count(const float* positions, int32_t N, float threshold, uint32_t* counts, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || !(threshold > 0.0f) || !std::isfinite(threshold)) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!positions || !counts || !ws) return GSX_E_INVALID;
    size_t off[7];
    if (ws_bytes < merge_ws_layout(N, threshold, off)) return GSX_E_CAPACITY;
    const MergeGrid g = merge_grid(N, threshold);
    char* w = static_cast<char*>(ws);
    uint32_t* keys = (uint32_t*)(w + off[0]); uint32_t* keys_sorted = (uint32_t*)(w + off[1]); uint32_t* order = (uint32_t*)(w + off[2]);
    uint32_t* k_tmp = (uint32_t*)(w + off[3]); uint32_t* v_tmp = (uint32_t*)(w + off[4]);
    uint32_t* start = (uint32_t*)(w + off[6]);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(N, kBlock));
    merge_keys_kernel<<<grid, dim3(kBlock), 0, st>>>(positions, N, g.inv_cell, g.n_buckets - 1u, keys);
    // stable: inside a bucket the points stay in index order
    int s = sort_pairs_internal(keys, nullptr, keys_sorted, order, k_tmp, v_tmp, N, nullptr, 0, g.bits, w + off[5], sort_ws_bytes_internal(N), st);
    if (s != GSX_OK) return s;
    merge_bucket_start_kernel<<<dim3((unsigned)ceil_div((int64_t)g.n_buckets + 1, kBlock)), dim3(kBlock), 0, st>>>(keys_sorted, N, g.n_buckets, start);
    merge_probe_kernel<false><<<grid, dim3(kBlock), 0, st>>>(positions, N, threshold, g.inv_cell, g.n_buckets - 1u, order, start, counts, nullptr, nullptr);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_merge_fill(const float* positions, int32_t N, float threshold, const uint32_t* offsets, uint32_t* partners,
                           const void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || !(threshold > 0.0f) || !std::isfinite(threshold)) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!positions || !offsets || !partners || !ws) return GSX_E_INVALID;
    size_t off[7];
    if (ws_bytes < merge_ws_layout(N, threshold, off)) return GSX_E_CAPACITY;
    const MergeGrid g = merge_grid(N, threshold);
    const char* w = static_cast<const char*>(ws);
    merge_probe_kernel<true><<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream>>>(
        positions, N, threshold, g.inv_cell, g.n_buckets - 1u, (const uint32_t*)(w + off[2]), (const uint32_t*)(w + off[6]), nullptr, offsets, partners);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

// The greedy merge itself (:500-515), on HOST arrays: pairs (i, j), i <= j, in row-major order -- offsets_host[i] ..
// offsets_host[i + 1] delimit i's partners, which must be sorted ascending.  A pair is skipped when either end has been
// removed; otherwise i takes the opacity-weighted mean of position and colour of the CURRENT values and the larger
// opacity, and j is marked removed.  Same fp32 operations in the same order as the reference's tensor expressions.
// j == i is accepted because the reference's own pair list holds such pairs (a self-distance that torch.cdist rounds
// above zero; the point is then averaged with itself and removed): gsx_merge_fill never produces one, a caller
// replaying the reference's list can.
GSX_API int gsx_merge_greedy_host(int32_t N, const uint32_t* offsets_host, const uint32_t* partners_host,
                                  float* positions_host, float* features_host, float* opacity_host, uint8_t* removed_host)
{
    if (N < 0) return GSX_E_INVALID;
    if (N == 0) return GSX_OK;
    if (!offsets_host || !positions_host || !features_host || !opacity_host || !removed_host) return GSX_E_INVALID;
    for (int32_t i = 0; i < N; i++) removed_host[i] = 0;
    for (int32_t i = 0; i < N; i++) {
        for (uint32_t s = offsets_host[i]; s < offsets_host[i + 1]; s++) {
            const uint32_t j = partners_host[s];
            if (j >= (uint32_t)N || j < (uint32_t)i) return GSX_E_INVALID;
            if (removed_host[i] || removed_host[j]) continue;
            const float w1 = opacity_host[i], w2 = opacity_host[j], total = w1 + w2;
            for (int c = 0; c < 3; c++) {
                positions_host[3 * i + c] = (w1 * positions_host[3 * i + c] + w2 * positions_host[3 * j + c]) / total;
                features_host[3 * i + c] = (w1 * features_host[3 * i + c] + w2 * features_host[3 * j + c]) / total;
            }
            opacity_host[i] = w1 > w2 ? w1 : (w2 > w1 ? w2 : (w1 != w1 ? w1 : w2));      // torch.maximum: NaN wins
            removed_host[j] = 1;
        }
    }
    return GSX_OK;
}
