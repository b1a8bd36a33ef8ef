// gsx_bin.hip -- tile binning: scan tiles_touched, emit (tile, rank) instances in rank order,
// stable radix sort by tile id, cut per-tile ranges.  No counterpart in the reference, which
// evaluates every Gaussian at every pixel (/root/reference/model/gaussian_splatting.py:118-144);
// the integer products are bit-exact against oracle/gsx_oracle.c (gsxo_scan/emit/sort_ranges).
#include "gsx_common.h"

namespace gsx {

int sort_pairs_internal(const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int64_t, const uint64_t*, int, int,
                        void*, size_t, hipStream_t);
int scan_internal(const uint32_t*, uint32_t*, uint64_t*, int64_t, void*, size_t, hipStream_t, const uint64_t* block_sums = nullptr, int sums_tile = 0);
size_t sort_ws_bytes_internal(int64_t n);

// one thread per rank: write its tiles row-major at offsets[r]..  Writes stop at `capacity` (the caller finds
// out that the buffers were too small when it reads the instance count, and retries).  A splat that covers more than
// kEmitSolo tiles is written by its whole wavefront (lane l takes tiles l, l+64, ...): one screen-sized splat would
// otherwise keep a single lane busy for thousands of iterations while the rest of the chip waits.
constexpr int kEmitSolo = 32;
__global__ __launch_bounds__(kBlock) void emit_kernel(const float* __restrict__ rec, const uint32_t* __restrict__ offsets, int N, int tw,
                                                      uint32_t capacity, uint32_t* __restrict__ tiles, uint32_t* __restrict__ ranks)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0; uint32_t o = 0;
    if (r < N) {
        const float2 bb = *reinterpret_cast<const float2*>(rec + (size_t)GSX_REC_FLOATS * r + 10);
        const uint32_t bx = __float_as_uint(bb.x), by = __float_as_uint(bb.y);
        const int xlo = bx & 0xffff, xhi = bx >> 16, ylo = by & 0xffff, yhi = by >> 16;
        if (xlo <= xhi) {
            x0 = xlo >> GSX_TILE_SHIFT; x1 = (xhi >> GSX_TILE_SHIFT) + 1;
            y0 = ylo >> GSX_TILE_SHIFT; y1 = (yhi >> GSX_TILE_SHIFT) + 1;
            o = offsets[r];
        }
    }
    const int cnt = (x1 - x0) * (y1 - y0);
    if (cnt > 0 && cnt <= kEmitSolo) {
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) {
                if (o >= capacity) break;
                tiles[o] = (uint32_t)(ty * tw + tx); ranks[o] = (uint32_t)r; o++;
            }
    }
    unsigned long long big = __ballot(cnt > kEmitSolo);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = __shfl(x0, src, 64), bx1 = __shfl(x1, src, 64), by0 = __shfl(y0, src, 64), bcnt = __shfl(cnt, src, 64);
        const uint32_t bo = (uint32_t)__shfl((int)o, src, 64);
        const int br = __shfl(r, src, 64), bw = bx1 - bx0;
        for (int i = lane; i < bcnt; i += 64) {
            const uint32_t dst = bo + (uint32_t)i;
            if (dst < capacity) { tiles[dst] = (uint32_t)((by0 + i / bw) * tw + bx0 + i % bw); ranks[dst] = (uint32_t)br; }
        }
    }
}

// ranges[t] = [lower_bound(tiles, t), lower_bound(tiles, t+1))
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
// also counts the tile's compositing chunks (GSX_CHUNK instances each); entry n_tiles is 0 so that
// the exclusive scan of chunk_cnt[0..n_tiles] ends with the total
__global__ __launch_bounds__(kBlock) void ranges_kernel(const uint32_t* __restrict__ tiles_sorted, const uint64_t* __restrict__ total_dev,
                                                        uint32_t capacity, int n_tiles,
                                                        uint32_t* __restrict__ ranges, uint32_t* __restrict__ chunk_cnt)
{
    const uint64_t total = *total_dev;
    const uint32_t D = total < capacity ? (uint32_t)total : capacity;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { chunk_cnt[t] = 0; return; }
    const uint32_t lb = lower_bound_u32(tiles_sorted, D, (uint32_t)t);
    const uint32_t ub = lower_bound_u32(tiles_sorted, D, (uint32_t)t + 1u);
    ranges[2*t] = lb; ranges[2*t+1] = ub;
    chunk_cnt[t] = (ub - lb + GSX_CHUNK - 1) / GSX_CHUNK;
}

static int tile_bits(int n_tiles) { int b = 1; while ((1 << b) < n_tiles) b++; return b; }

}  // namespace gsx

using namespace gsx;

// scan workspace that covers the chunk table of the largest image (65535/16)^2 tiles
static size_t chunk_scan_ws_bytes() { return gsx_scan_workspace_bytes((int64_t)4096 * 4096 + 1); }

GSX_API size_t gsx_bin_workspace_bytes(int32_t N, int64_t capacity)
{
    size_t a = gsx_scan_workspace_bytes(N > 0 ? N : 1);
    size_t b = sort_ws_bytes_internal(capacity > 0 ? capacity : 1);
    size_t c = chunk_scan_ws_bytes();
    size_t m = a > b ? a : b;
    return 256 + (m > c ? m : c);
}

GSX_API int64_t gsx_max_chunks(int64_t D, int32_t H, int32_t W)
{
    const int64_t n_tiles = (int64_t)((W + GSX_TILE - 1) / GSX_TILE) * ((H + GSX_TILE - 1) / GSX_TILE);
    return D / GSX_CHUNK + (D < n_tiles ? D : n_tiles);
}

GSX_API int gsx_bin_count(const uint32_t* tiles_touched, const uint64_t* tt_block_sums, int32_t N, uint32_t* offsets, uint64_t* total_dev,
                          int64_t* D_pinned_host, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || !total_dev) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (N > 0 && (!tiles_touched || !offsets || !ws || ws_bytes < gsx_scan_workspace_bytes(N))) return GSX_E_INVALID;
    int s = scan_internal(tiles_touched, offsets, total_dev, N, ws, ws_bytes, st, tt_block_sums, kBlock);     // N == 0: total = 0
    if (s != GSX_OK) return s;
    if (D_pinned_host) GSX_HIP(hipMemcpyAsync(D_pinned_host, total_dev, 8, hipMemcpyDeviceToHost, st));
    return GSX_OK;
}

GSX_API int gsx_bin_emit_sort(const float* rec, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                              int64_t capacity, const uint64_t* total_dev,
                              uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                              uint32_t* ranges, uint32_t* chunk_base, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || capacity <= 0) return GSX_E_INVALID;
    if (capacity >= 0xFFFFFFFFll) return GSX_E_RANGE;
    if (!ranges || !chunk_base || !total_dev || !ws || ws_bytes < gsx_bin_workspace_bytes(N, capacity)) return GSX_E_INVALID;
    if (!tiles_sorted || !ranks_sorted || !tiles_tmp || !ranks_tmp) return GSX_E_INVALID;
    if (N > 0 && (!rec || !offsets)) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE, n_tiles = tw * th;
    void* sub_ws = (char*)ws + 256; size_t sub_bytes = ws_bytes - 256;
    const int bits = tile_bits(n_tiles);
    const int passes = (bits + 7) / 8;
    // emit into the buffer the sort does NOT write first, so that the last pass lands in *_sorted
    uint32_t* e_t = (passes & 1) ? tiles_tmp : tiles_sorted;
    uint32_t* e_r = (passes & 1) ? ranks_tmp : ranks_sorted;
    if (N > 0) {
        emit_kernel<<<dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, st>>>(rec, offsets, N, tw, (uint32_t)capacity, e_t, e_r);
        GSX_LAUNCH_CHECK();
        int s = sort_pairs_internal(e_t, e_r, tiles_sorted, ranks_sorted, tiles_tmp, ranks_tmp, capacity, total_dev, 0, bits, sub_ws, sub_bytes, st);
        if (s != GSX_OK) return s;
    }
    ranges_kernel<<<dim3((unsigned)ceil_div(n_tiles + 1, kBlock)), dim3(kBlock), 0, st>>>(tiles_sorted, total_dev, (uint32_t)capacity, n_tiles, ranges, chunk_base);
    GSX_LAUNCH_CHECK();
    return scan_internal(chunk_base, chunk_base, nullptr, (int64_t)n_tiles + 1, sub_ws, sub_bytes, st);
}

GSX_API int gsx_bin_instances(const float* rec, const uint32_t* tiles_touched, int32_t N, int32_t H, int32_t W,
                              int64_t capacity, uint32_t* offsets,
                              uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                              uint32_t* ranges, uint32_t* chunk_base, int64_t* D_host, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || capacity < 0) return GSX_E_INVALID;
    if (!ranges || !chunk_base || !D_host || !ws || ws_bytes < gsx_bin_workspace_bytes(N, capacity)) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE, n_tiles = tw * th;
    *D_host = 0;
    auto empty = [&]() -> int {
        GSX_HIP(hipMemsetAsync(ranges, 0, (size_t)n_tiles * 8, st));
        GSX_HIP(hipMemsetAsync(chunk_base, 0, (size_t)(n_tiles + 1) * 4, st));
        return GSX_OK;
    };
    if (N == 0) return empty();
    if (!rec || !tiles_touched || !offsets) return GSX_E_INVALID;
    uint64_t* total_dev = (uint64_t*)ws;
    void* sub_ws = (char*)ws + 256; size_t sub_bytes = ws_bytes - 256;
    int s = gsx_bin_count(tiles_touched, nullptr, N, offsets, total_dev, nullptr, sub_ws, sub_bytes, stream);
    if (s != GSX_OK) return s;
    uint64_t D = 0;
    GSX_HIP(hipMemcpyAsync(&D, total_dev, 8, hipMemcpyDeviceToHost, st));
    GSX_HIP(hipStreamSynchronize(st));
    *D_host = (int64_t)D;
    if (D >= 0xFFFFFFFFull) return GSX_E_RANGE;
    if ((int64_t)D > capacity) return GSX_E_CAPACITY;
    if (D == 0) return empty();
    return gsx_bin_emit_sort(rec, offsets, N, H, W, capacity, total_dev, tiles_sorted, ranks_sorted, tiles_tmp, ranks_tmp,
                             ranges, chunk_base, ws, ws_bytes, stream);
}
