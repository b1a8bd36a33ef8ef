// gsx_bin.hip -- tile binning: scan tiles_touched, emit (tile, rank) instances in rank order,
// stable radix sort by tile id, cut per-tile ranges.  No counterpart in the reference, which
// evaluates every Gaussian at every pixel (/root/reference/model/gaussian_splatting.py:118-144);
// the integer products are bit-exact against oracle/gsx_oracle.c (gsxo_scan/emit/sort_ranges).
#include "gsx_common.h"

namespace gsx {

int sort_pairs_internal(const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int64_t, const uint64_t*, int, int,
                        void*, size_t, hipStream_t, int check_bit = 32, uint64_t* mismatch_dev = nullptr);
int scan_internal(const uint32_t*, uint32_t*, uint64_t*, int64_t, void*, size_t, hipStream_t, const uint64_t* block_sums = nullptr, int sums_tile = 0,
                  uint64_t* mirror = nullptr);
size_t sort_ws_bytes_internal(int64_t n);

// one thread per rank: write its tiles row-major at offsets[r]..  Writes stop at `capacity` (the caller finds
// out that the buffers were too small when it reads the instance count, and retries).  A splat that covers more than
// kEmitSolo tiles is written by its whole wavefront (lane l takes tiles l, l+64, ...): one screen-sized splat would
// otherwise keep a single lane busy for thousands of iterations while the rest of the chip waits.
// SCAN (the view driver): the exclusive scan of tiles_touched happens here instead of in a launch of its own (gsx_bin_count) -- a
// rank's tile count IS the area of its box, gsx_build_records left the sums per kBlock ranks (`sums`), so a workgroup adds up the
// sums in front of it (as scan_apply_kernel does) and scans its own 256 counts; it writes offsets_out[] for the backward, and the
// last workgroup the total (and its mirror in pinned host memory, with the word behind it: gsx_bin_count).
constexpr int kEmitSolo = 32;
template <bool SCAN>
__global__ __launch_bounds__(kBlock) void emit_kernel(const float* __restrict__ rec, const uint32_t* __restrict__ tile_box,
                                                      const uint32_t* __restrict__ offsets, int N, int tw,
                                                      uint32_t capacity, uint32_t* __restrict__ tiles, uint32_t* __restrict__ ranks,
                                                      const uint64_t* __restrict__ sums, uint32_t* __restrict__ offsets_out,
                                                      uint64_t* __restrict__ total, uint64_t* __restrict__ mirror)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0; uint32_t o = 0;
    if (r < N) {
        if (tile_box) {                               // packed footprint in tile units (gsx_build_records): 4 B per rank, coalesced
            const uint32_t tb = tile_box[r];
            const int a = tb & 0xff, b = (tb >> 8) & 0xff, c = (tb >> 16) & 0xff, d = tb >> 24;
            if (a <= c) { x0 = a; x1 = c + 1; y0 = b; y1 = d + 1; if (!SCAN) o = offsets[r]; }
        } else {
            const float2 bb = *reinterpret_cast<const float2*>(rec + (size_t)GSX_REC_FLOATS * r + 10);
            const uint32_t bx = __float_as_uint(bb.x), by = __float_as_uint(bb.y);
            const int xlo = bx & 0xffff, xhi = bx >> 16, ylo = by & 0xffff, yhi = by >> 16;
            if (xlo <= xhi) {
                x0 = xlo >> GSX_TILE_SHIFT; x1 = (xhi >> GSX_TILE_SHIFT) + 1;
                y0 = ylo >> GSX_TILE_SHIFT; y1 = (yhi >> GSX_TILE_SHIFT) + 1;
                if (!SCAN) o = offsets[r];
            }
        }
    }
    const int cnt = (x1 - x0) * (y1 - y0);
    if (SCAN) {
        __shared__ uint32_t s_wave[4];
        __shared__ uint64_t s_pre[4];
        uint64_t acc = 0;
        for (int j = threadIdx.x; j < (int)blockIdx.x; j += kBlock) acc += sums[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
        if (lane == 0) s_pre[threadIdx.x >> 6] = acc;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32((uint32_t)cnt, s_wave, &tot);      // barriers inside: s_pre is visible after it
        const uint64_t pre = (s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]);
        o = (uint32_t)pre + ex;
        if (r < N) offsets_out[r] = o;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
            total[0] = pre + (uint64_t)tot;
            if (mirror) { mirror[0] = pre + (uint64_t)tot; mirror[1] = total[1]; }
        }
    }
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
// also counts the tile's compositing chunks (chunk_size_for(D) instances each); entry n_tiles is 0 so that
// the exclusive scan of chunk_cnt[0..n_tiles] ends with the total
__global__ __launch_bounds__(kBlock) void ranges_kernel(const uint32_t* __restrict__ tiles_sorted, const uint64_t* __restrict__ total_dev,
                                                        uint32_t capacity, int n_tiles, int chunk_force,
                                                        uint32_t* __restrict__ ranges, uint32_t* __restrict__ chunk_cnt)
{
    const uint64_t total = *total_dev;
    const uint32_t D = total < capacity ? (uint32_t)total : capacity;
    const uint32_t ch = chunk_size_for(D, chunk_force);
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { chunk_cnt[t] = 0; chunk_cnt[t + 1] = ch; return; }      // entry n_tiles + 1 (outside the scan): the chunk size
    const uint32_t lb = lower_bound_u32(tiles_sorted, D, (uint32_t)t);
    const uint32_t ub = lower_bound_u32(tiles_sorted, D, (uint32_t)t + 1u);
    ranges[2*t] = lb; ranges[2*t+1] = ub;
    chunk_cnt[t] = (ub - lb + ch - 1) / ch;
}

static int tile_bits(int n_tiles) { int b = 1; while ((1 << b) < n_tiles) b++; return b; }

// ------------------------------------------------------------------------------------------
// One-pass stable counting sort of the instances by tile id, for images of up to 4096 tiles (config 3: 64 x 64).
// The generic 8-bit LSD sort needs two passes of three launches here, plus a launch of binary searches for the tile
// ranges and one for the chunk table; with the whole tile id as ONE 12-bit digit the per-tile totals ARE the range
// lengths, so the pass is: histogram -> scan down the table's columns -> scatter (whose workgroup 0 also writes the
// ranges and the chunk table).  3 launches instead of 8 and one trip of the instances through HBM instead of two.
// Stability (each tile's list stays in rank order) comes from wave-private digit counters, as in gsx_sort.hip: 16 waves
// x 4096 u16 counters = 128 KB of the CU's 160 KB LDS, so one workgroup of 1024 threads per CU.
// ------------------------------------------------------------------------------------------
constexpr int kTsBits = 12, kTsDigits = 1 << kTsBits;
constexpr int kTsThreads = 1024, kTsWaves = kTsThreads / 64;
constexpr int kTsItems = 16;                                  // most keys per thread (the device picks 2..16 from the count: ts_items)
constexpr int kTsSegDigits = 64;                              // column-scan workgroup: 64 digits x 16 row segments

__device__ __forceinline__ int64_t ts_count(const uint64_t* __restrict__ total_dev, int64_t capacity)
{
    const uint64_t d = *total_dev;
    return d < (uint64_t)capacity ? (int64_t)d : capacity;
}
// Keys per workgroup, decided on the device from the instance count: spread the keys over (at most) one workgroup per
// CU -- the scatter's fully divergent stores retire about one lane per cycle per CU, so idle CUs are lost time, while a
// 257th workgroup would wait for a whole CU (the LDS holds one) -- in steps of one key per thread.
constexpr int kTsCus = 256;
__device__ __host__ __forceinline__ int ts_items(int64_t n)
{
    const int64_t it = (n + (int64_t)kTsCus * kTsThreads - 1) / ((int64_t)kTsCus * kTsThreads);
    return it < 2 ? 2 : (it > kTsItems ? kTsItems : (int)it);
}

// hist[b][d] = number of keys of workgroup b that belong to tile d (rows of workgroups past the count are never read)
__global__ __launch_bounds__(kTsThreads) void tile_hist_kernel(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ total_dev,
                                                               int64_t capacity, int n_tiles, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_hist[kTsDigits];
    const int64_t n = ts_count(total_dev, capacity);
    const int items = ts_items(n);
    const int64_t base = (int64_t)blockIdx.x * items * kTsThreads;
    if (base >= n) return;
    for (int d = threadIdx.x; d < n_tiles; d += kTsThreads) s_hist[d] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTsItems; k++) {
        const int64_t i = base + k * kTsThreads + threadIdx.x;
        if (k < items && i < n) atomicAdd(&s_hist[keys[i] & (kTsDigits - 1)], 1u);
    }
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * kTsDigits;
    for (int d = threadIdx.x; d < n_tiles; d += kTsThreads) row[d] = s_hist[d];
}

// exclusive scan down every column of hist (over the workgroups that hold keys), column total -> totals[d].
// One workgroup = 64 columns x 16 segments of rows: sum the segment, exchange, rewrite the segment as prefixes.
__global__ __launch_bounds__(kTsThreads) void tile_colscan_kernel(uint32_t* __restrict__ hist, const uint64_t* __restrict__ total_dev,
                                                                  int64_t capacity, int n_tiles, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_seg[kTsWaves][kTsSegDigits];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int d = blockIdx.x * kTsSegDigits + lane;
    const int64_t n = ts_count(total_dev, capacity);
    const int64_t tile = (int64_t)ts_items(n) * kTsThreads;
    const int nb = (int)((n + tile - 1) / tile);
    const int per = (nb + kTsWaves - 1) / kTsWaves;
    const int r0 = min(seg * per, nb), r1 = min(r0 + per, nb);
    const bool live = d < n_tiles;
    uint32_t sum = 0;
    if (live) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            const uint32_t a = hist[(size_t)r * kTsDigits + d], b = hist[(size_t)(r + 1) * kTsDigits + d];
            const uint32_t c = hist[(size_t)(r + 2) * kTsDigits + d], e = hist[(size_t)(r + 3) * kTsDigits + d];
            sum += (a + b) + (c + e);
        }
        for (; r < r1; r++) sum += hist[(size_t)r * kTsDigits + d];
    }
    s_seg[seg][lane] = sum;
    __syncthreads();
    if (!live) return;
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (int s2 = 0; s2 < kTsWaves; s2++) { const uint32_t v = s_seg[s2][lane]; if (s2 < seg) run += v; tot += v; }
    if (seg == 0) totals[d] = tot;
    for (int r = r0; r < r1; r++) {
        uint32_t* cell = hist + (size_t)r * kTsDigits + d;
        const uint32_t v = *cell; *cell = run; run += v;
    }
}

// exclusive scan of one value per thread over a 1024-thread workgroup
__device__ __forceinline__ uint32_t block1024_excl_scan(uint32_t v, uint32_t* s_wave /*[16]*/, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kTsWaves; w++) { const uint32_t x = s_wave[w]; if (w < wave) base += x; tot += x; }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(kTsThreads) void tile_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                  const uint64_t* __restrict__ total_dev, int64_t capacity, int n_tiles,
                                                                  const uint32_t* __restrict__ hist_scanned, const uint32_t* __restrict__ totals,
                                                                  int chunk_force, uint32_t* __restrict__ ranges, uint32_t* __restrict__ chunk_base,
                                                                  uint64_t* __restrict__ info)
{
    __shared__ uint16_t s_cnt[kTsWaves][kTsDigits];      // per-wave digit counts, then per-wave offsets inside the workgroup
    __shared__ uint32_t s_base[kTsDigits];               // per tile: global slot minus position inside the workgroup
    __shared__ uint16_t s_lstart[kTsDigits];             // per tile: position of its first key inside the workgroup
    __shared__ uint32_t s_scan[kTsWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = ts_count(total_dev, capacity);
    const int items = ts_items(n);
    const int64_t base = (int64_t)blockIdx.x * items * kTsThreads;
    const bool active = base < n;
    if (!active && blockIdx.x != 0) return;
    // 32-bit positions inside the workgroup from here on (uniform 64-bit base + per-lane u32 offset: fewer registers)
    const int64_t left = n - base;
    const uint32_t here = active ? (uint32_t)(left < (int64_t)items * kTsThreads ? left : (int64_t)items * kTsThreads) : 0u;
    const uint32_t* kin = keys_in + (active ? base : 0);
    const uint32_t* vin = vals_in + (active ? base : 0);
    const uint32_t wlane = (uint32_t)(wave * items * 64 + lane);        // each wave owns a contiguous run of keys
    uint32_t key[kTsItems];
#pragma unroll
    for (int k = 0; k < kTsItems; k++) {                                // issued first: in flight during the scans below
        const uint32_t li = wlane + k * 64;
        key[k] = (k < items && li < here) ? (kin[li] & (kTsDigits - 1)) : 0u;
    }
    // tile bases = exclusive scan of the column totals; thread t owns tiles 4t .. 4t+3
    uint32_t tv[4], db[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int d = 4 * tid + i; tv[i] = d < n_tiles ? totals[d] : 0u; }
    uint32_t tot;
    const uint32_t ex = block1024_excl_scan((tv[0] + tv[1]) + (tv[2] + tv[3]), s_scan, &tot);
    db[0] = ex; db[1] = db[0] + tv[0]; db[2] = db[1] + tv[1]; db[3] = db[2] + tv[2];
    if (blockIdx.x == 0) {
        // what ranges_kernel and the chunk-table scan did: ranges[t] = [first, last), chunk_base = scan of ceil(len / chunk size)
        const uint32_t ch = chunk_size_for((uint64_t)n, chunk_force);
        uint32_t cc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int d = 4 * tid + i;
            cc[i] = d < n_tiles ? (tv[i] + ch - 1) / ch : 0u;
            if (d < n_tiles) { ranges[2 * d] = db[i]; ranges[2 * d + 1] = db[i] + tv[i]; }
        }
        uint32_t ctot;
        uint32_t cex = block1024_excl_scan((cc[0] + cc[1]) + (cc[2] + cc[3]), s_scan, &ctot);
#pragma unroll
        for (int i = 0; i < 4; i++) { const int d = 4 * tid + i; if (d < n_tiles) chunk_base[d] = cex; cex += cc[i]; }
        if (tid == 0) { chunk_base[n_tiles] = ctot; chunk_base[n_tiles + 1] = ch; }
        if (info) {
            // for the one-call trainer step (gsx_view.hip): 1 + the number of EMPTY tiles, straight into the host's pinned word.  An
            // empty tile's pixels stay at alpha 0, so the reference's global early exit cannot fire on this view, and the host,
            // which reads this with the instance count, leaves the exit path's three launches out
            uint32_t etot;
            uint32_t ne = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) ne += (4 * tid + i < n_tiles && tv[i] == 0u) ? 1u : 0u;
            block1024_excl_scan(ne, s_scan, &etot);
            if (tid == 0) info[0] = 1ull + (uint64_t)etot;
        }
        if (!active) return;
    }
    const uint32_t* prow = hist_scanned + (size_t)blockIdx.x * kTsDigits;
    uint32_t gb[4];                                                     // first global slot of this workgroup's keys of my 4 tiles
#pragma unroll
    for (int i = 0; i < 4; i++) { const int d = 4 * tid + i; gb[i] = db[i] + (d < n_tiles ? prow[d] : 0u); }
    {
        uint4* z = reinterpret_cast<uint4*>(&s_cnt[0][0]);
        constexpr int kVec = kTsWaves * kTsDigits * 2 / 16;
        for (int i = tid; i < kVec; i += kTsThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    uint16_t* cnt = s_cnt[wave];
    // from here key[k] also carries the key's position: tile id in bits 0..11, position (< 16384) above
#pragma unroll
    for (int k = 0; k < kTsItems; k++) {
        if (k < items) {                                                // uniform
            const bool valid = wlane + k * 64 < here;
            const uint32_t d = key[k];
            uint32_t rank, count;
            wave_match_rank<kTsBits>(d, valid, rank, count);
            const uint32_t old = cnt[d];                               // every lane reads before the leader writes
            if (valid && rank == 0) cnt[d] = (uint16_t)(old + count);
            key[k] = d | ((old + rank) << kTsBits);
        }
        __builtin_amdgcn_sched_barrier(0);                              // one round at a time: interleaved rounds spill (128 VGPRs at 1024 threads)
    }
    // the values are read only now (16 registers fewer across the ranking), and early enough that their latency hides
    // behind the prefix and scan steps below
    uint32_t val[kTsItems];
#pragma unroll
    for (int k = 0; k < kTsItems; k++) {
        const uint32_t li = wlane + k * 64;
        val[k] = (k < items && li < here) ? vin[li] : 0u;
    }
    __syncthreads();
    // per tile: exclusive prefix of the 16 waves' counts (fits 16 bits: <= 16384 keys per workgroup).  Thread t owns tiles
    // 4t .. 4t+3 = one uint2 of packed u16 pairs per wave row.
    uint32_t t01 = 0, t23 = 0;
    {
        uint2* c2 = reinterpret_cast<uint2*>(&s_cnt[0][0]);
#pragma unroll
        for (int w = 0; w < kTsWaves; w++) {
            const uint2 c = c2[w * (kTsDigits / 4) + tid];
            c2[w * (kTsDigits / 4) + tid] = make_uint2(t01, t23);
            t01 += c.x; t23 += c.y;
        }
    }
    // position of each tile's first key inside the workgroup's tile-sorted order, and what turns such a position into
    // the global slot: the keys go through LDS in that order so that the stores to HBM are runs of consecutive addresses
    // (scattered 4-byte stores cost one L2 transaction each: 2 per key, ~6 us per million keys on the whole chip)
    const uint32_t c0 = t01 & 0xffffu, c1 = t01 >> 16, c2n = t23 & 0xffffu, c3 = t23 >> 16;
    uint32_t ltot;
    const uint32_t l0 = block1024_excl_scan((c0 + c1) + (c2n + c3), s_scan, &ltot);
    const uint32_t l1 = l0 + c0, l2 = l1 + c1, l3 = l2 + c2n;
    reinterpret_cast<uint2*>(s_lstart)[tid] = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
    reinterpret_cast<uint4*>(s_base)[tid] = make_uint4(gb[0] - l0, gb[1] - l1, gb[2] - l2, gb[3] - l3);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTsItems; k++)
        if (k < items) { const uint32_t d = key[k] & (kTsDigits - 1); key[k] += ((uint32_t)s_lstart[d] + (uint32_t)cnt[d]) << kTsBits; }
    __syncthreads();                                                    // the counters are dead: their LDS stages the keys
    uint32_t* s_val = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);        // 16384 x 4 B
    uint16_t* s_key = reinterpret_cast<uint16_t*>(s_val + kTsThreads * kTsItems);      // 16384 x 2 B
#pragma unroll
    for (int k = 0; k < kTsItems; k++) {
        const uint32_t li = wlane + k * 64;
        if (k < items && li < here) { const uint32_t pos = key[k] >> kTsBits; s_val[pos] = val[k]; s_key[pos] = (uint16_t)(key[k] & (kTsDigits - 1)); }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTsItems; k++) {
        const uint32_t i = (uint32_t)(k * kTsThreads + tid);
        if (k < items && i < here) {
            const uint32_t d = s_key[i];
            const uint32_t dst = s_base[d] + i;
            if (keys_out) keys_out[dst] = d;              // (the view driver asks for no sorted keys: nothing reads them)
            vals_out[dst] = s_val[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same one-pass sort for images of up to 8192 tiles (round 3; 1920 x 1080 = 8160: BASELINE config 5 used to take the
// generic path -- two 8-bit passes of three launches, a launch of binary searches, a scan: the largest stage that was
// left of a config-5 step).  A 13-bit digit doubles the counters: 8 waves x 8192 u16 = 128 KB is what fits, so the
// workgroup is 512 threads (8 waves), each thread holds up to 32 keys (16384 keys per workgroup, as in the 12-bit form:
// the histogram table costs 2 B per key), and what the 12-bit form keeps BESIDE the counters -- per tile the position of
// its first key inside the workgroup and the global base -- moves: the first goes INTO the counters (added to every
// wave's prefix), the second into the counters' space once they are dead (next to the staged keys and values: 64 + 32 + 32
// KB = the 128 KB exactly).  Thread t owns tiles 16 t .. 16 t + 15.
// ------------------------------------------------------------------------------------------
constexpr int kWsBits = 13, kWsDigits = 1 << kWsBits;
constexpr int kWsThreads = 512, kWsWaves = kWsThreads / 64;
constexpr int kWsItems = 32;
constexpr int kWsOwn = kWsDigits / kWsThreads;               // 16 tiles per thread
__device__ __host__ __forceinline__ int ws_items(int64_t n)
{
    const int64_t it = (n + (int64_t)kTsCus * kWsThreads - 1) / ((int64_t)kTsCus * kWsThreads);
    return it < 2 ? 2 : (it > kWsItems ? kWsItems : (int)it);
}

__global__ __launch_bounds__(kWsThreads) void wtile_hist_kernel(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ total_dev,
                                                                int64_t capacity, int n_tiles, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_hist[kWsDigits];
    const int64_t n = ts_count(total_dev, capacity);
    const int items = ws_items(n);
    const int64_t base = (int64_t)blockIdx.x * items * kWsThreads;
    if (base >= n) return;
    for (int d = threadIdx.x; d < n_tiles; d += kWsThreads) s_hist[d] = 0;
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < kWsItems; k++) {
        const int64_t i = base + k * kWsThreads + threadIdx.x;
        if (k < items && i < n) atomicAdd(&s_hist[keys[i] & (kWsDigits - 1)], 1u);
    }
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * kWsDigits;
    for (int d = threadIdx.x; d < n_tiles; d += kWsThreads) row[d] = s_hist[d];
}

// column scan of the wide table: as tile_colscan_kernel, rows of kWsDigits words, workgroups of ws_items(n) * 512 keys
__global__ __launch_bounds__(kTsThreads) void wtile_colscan_kernel(uint32_t* __restrict__ hist, const uint64_t* __restrict__ total_dev,
                                                                   int64_t capacity, int n_tiles, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_seg[kTsWaves][kTsSegDigits];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int d = blockIdx.x * kTsSegDigits + lane;
    const int64_t n = ts_count(total_dev, capacity);
    const int64_t tile = (int64_t)ws_items(n) * kWsThreads;
    const int nb = (int)((n + tile - 1) / tile);
    const int per = (nb + kTsWaves - 1) / kTsWaves;
    const int r0 = min(seg * per, nb), r1 = min(r0 + per, nb);
    const bool live = d < n_tiles;
    uint32_t sum = 0;
    if (live) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            const uint32_t a = hist[(size_t)r * kWsDigits + d], b = hist[(size_t)(r + 1) * kWsDigits + d];
            const uint32_t c = hist[(size_t)(r + 2) * kWsDigits + d], e = hist[(size_t)(r + 3) * kWsDigits + d];
            sum += (a + b) + (c + e);
        }
        for (; r < r1; r++) sum += hist[(size_t)r * kWsDigits + d];
    }
    s_seg[seg][lane] = sum;
    __syncthreads();
    if (!live) return;
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (int s2 = 0; s2 < kTsWaves; s2++) { const uint32_t v = s_seg[s2][lane]; if (s2 < seg) run += v; tot += v; }
    if (seg == 0) totals[d] = tot;
    for (int r = r0; r < r1; r++) {
        uint32_t* cell = hist + (size_t)r * kWsDigits + d;
        const uint32_t v = *cell; *cell = run; run += v;
    }
}

// exclusive scan of one value per thread over a 512-thread workgroup
__device__ __forceinline__ uint32_t block512_excl_scan(uint32_t v, uint32_t* s_wave /*[8]*/, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWsWaves; w++) { const uint32_t x = s_wave[w]; if (w < wave) base += x; tot += x; }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(kWsThreads) void wtile_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                   const uint64_t* __restrict__ total_dev, int64_t capacity, int n_tiles,
                                                                   const uint32_t* __restrict__ hist_scanned, const uint32_t* __restrict__ totals,
                                                                   int chunk_force, uint32_t* __restrict__ ranges, uint32_t* __restrict__ chunk_base)
{
    // per-wave digit counts; then per wave and tile the position of the wave's first key of that tile inside the workgroup's
    // tile-sorted order; then (dead) the staged values, keys and the tiles' global bases
    __shared__ __align__(16) uint16_t s_cnt[kWsWaves][kWsDigits];           // 128 KB
    __shared__ uint32_t s_scan[kWsWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = ts_count(total_dev, capacity);
    const int items = ws_items(n);
    const int64_t base = (int64_t)blockIdx.x * items * kWsThreads;
    const bool active = base < n;
    if (!active && blockIdx.x != 0) return;
    const int64_t left = n - base;
    const uint32_t here = active ? (uint32_t)(left < (int64_t)items * kWsThreads ? left : (int64_t)items * kWsThreads) : 0u;
    const uint32_t* kin = keys_in + (active ? base : 0);
    const uint32_t* vin = vals_in + (active ? base : 0);
    const uint32_t wlane = (uint32_t)(wave * items * 64 + lane);        // each wave owns a contiguous run of keys
    uint32_t key[kWsItems];
#pragma unroll
    for (int k = 0; k < kWsItems; k++) {
        const uint32_t li = wlane + k * 64;
        key[k] = (k < items && li < here) ? (kin[li] & (kWsDigits - 1)) : 0u;
    }
    // tile bases = exclusive scan of the column totals; thread t owns tiles 16 t .. 16 t + 15
    uint32_t gb[kWsOwn];                                                // first global slot of this workgroup's keys of my tiles
    {
        uint32_t tv[kWsOwn], sum = 0;
#pragma unroll
        for (int i = 0; i < kWsOwn; i++) { const int d = kWsOwn * tid + i; tv[i] = d < n_tiles ? totals[d] : 0u; sum += tv[i]; }
        uint32_t tot;
        uint32_t run = block512_excl_scan(sum, s_scan, &tot);
        if (blockIdx.x == 0) {
            // ranges[t] = [first, last), chunk_base = scan of ceil(len / chunk size): what ranges_kernel and the chunk-table scan do
            const uint32_t ch = chunk_size_for((uint64_t)n, chunk_force);
            uint32_t csum = 0, r2 = run;
#pragma unroll
            for (int i = 0; i < kWsOwn; i++) {
                const int d = kWsOwn * tid + i;
                if (d < n_tiles) { ranges[2 * d] = r2; ranges[2 * d + 1] = r2 + tv[i]; csum += (tv[i] + ch - 1) / ch; }
                r2 += tv[i];
            }
            uint32_t ctot;
            uint32_t cex = block512_excl_scan(csum, s_scan, &ctot);
#pragma unroll
            for (int i = 0; i < kWsOwn; i++) {
                const int d = kWsOwn * tid + i;
                if (d < n_tiles) { chunk_base[d] = cex; cex += (tv[i] + ch - 1) / ch; }
            }
            if (tid == 0) { chunk_base[n_tiles] = ctot; chunk_base[n_tiles + 1] = ch; }
            if (!active) return;
        }
        const uint32_t* prow = hist_scanned + (size_t)blockIdx.x * kWsDigits;
#pragma unroll
        for (int i = 0; i < kWsOwn; i++) { const int d = kWsOwn * tid + i; gb[i] = run + (d < n_tiles ? prow[d] : 0u); run += tv[i]; }
    }
    {
        uint4* z = reinterpret_cast<uint4*>(&s_cnt[0][0]);
        constexpr int kVec = kWsWaves * kWsDigits * 2 / 16;
        for (int i = tid; i < kVec; i += kWsThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    uint16_t* cnt = s_cnt[wave];
    // from here key[k] also carries the key's position: tile id in bits 0..12, position (< 16384) above
#pragma unroll
    for (int k = 0; k < kWsItems; k++) {
        if (k < items) {                                                // uniform
            const bool valid = wlane + k * 64 < here;
            const uint32_t d = key[k];
            uint32_t rank, count;
            wave_match_rank<kWsBits>(d, valid, rank, count);
            const uint32_t old = cnt[d];                               // every lane reads before the leader writes
            if (valid && rank == 0) cnt[d] = (uint16_t)(old + count);
            key[k] = d | ((old + rank) << kWsBits);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t val[kWsItems];
#pragma unroll
    for (int k = 0; k < kWsItems; k++) {
        const uint32_t li = wlane + k * 64;
        val[k] = (k < items && li < here) ? vin[li] : 0u;
    }
    __syncthreads();
    // per tile: total over the 8 waves (pass A, read only), the tiles' first positions inside the workgroup (exclusive scan in
    // tile order), then each wave's entry becomes first position + the counts of the waves before it (pass B).  A thread's 16
    // tiles are two uint4 of packed u16 per wave row; sums stay below 2^15, so the packed adds never carry into a neighbour.
    uint4* c4 = reinterpret_cast<uint4*>(&s_cnt[0][0]);
    constexpr int kRow4 = kWsDigits * 2 / 16;                           // uint4 per wave row
    uint32_t tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};                        // 16 packed u16 totals
#pragma unroll
    for (int w = 0; w < kWsWaves; w++) {
        const uint4 a = c4[w * kRow4 + 2 * tid], b = c4[w * kRow4 + 2 * tid + 1];
        tsum[0] += a.x; tsum[1] += a.y; tsum[2] += a.z; tsum[3] += a.w; tsum[4] += b.x; tsum[5] += b.y; tsum[6] += b.z; tsum[7] += b.w;
    }
    uint32_t mysum = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) mysum += (tsum[i] & 0xffffu) + (tsum[i] >> 16);
    uint32_t ltot;
    uint32_t lrun = block512_excl_scan(mysum, s_scan, &ltot);
    uint32_t lstart[8];                                                 // 16 packed u16 first positions
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t lo = lrun; lrun += tsum[i] & 0xffffu;
        const uint32_t hi = lrun; lrun += tsum[i] >> 16;
        lstart[i] = lo | (hi << 16);
        gb[2 * i] -= lo; gb[2 * i + 1] -= hi;                           // global slot of local position 0 of the tile
    }
    {
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = lstart[i];
#pragma unroll
        for (int w = 0; w < kWsWaves; w++) {
            const uint4 a = c4[w * kRow4 + 2 * tid], b = c4[w * kRow4 + 2 * tid + 1];
            c4[w * kRow4 + 2 * tid] = make_uint4(r[0], r[1], r[2], r[3]);
            c4[w * kRow4 + 2 * tid + 1] = make_uint4(r[4], r[5], r[6], r[7]);
            r[0] += a.x; r[1] += a.y; r[2] += a.z; r[3] += a.w; r[4] += b.x; r[5] += b.y; r[6] += b.z; r[7] += b.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWsItems; k++)
        if (k < items) { const uint32_t d = key[k] & (kWsDigits - 1); key[k] += (uint32_t)cnt[d] << kWsBits; }
    __syncthreads();                                                    // the counters are dead: their LDS stages values, keys and bases
    uint32_t* s_val = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);        // 16384 x 4 B = 64 KB
    uint16_t* s_key = reinterpret_cast<uint16_t*>(s_val + kWsThreads * kWsItems);      // 16384 x 2 B = 32 KB
    uint32_t* s_base = reinterpret_cast<uint32_t*>(s_key + kWsThreads * kWsItems);     // 8192 x 4 B = 32 KB
#pragma unroll
    for (int i = 0; i < kWsOwn; i++) s_base[kWsOwn * tid + i] = gb[i];
#pragma unroll
    for (int k = 0; k < kWsItems; k++) {
        const uint32_t li = wlane + k * 64;
        if (k < items && li < here) { const uint32_t pos = key[k] >> kWsBits; s_val[pos] = val[k]; s_key[pos] = (uint16_t)(key[k] & (kWsDigits - 1)); }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWsItems; k++) {
        const uint32_t i = (uint32_t)(k * kWsThreads + tid);
        if (k < items && i < here) {
            const uint32_t d = s_key[i];
            const uint32_t dst = s_base[d] + i;
            if (keys_out) keys_out[dst] = d;              // (the view driver asks for no sorted keys: nothing reads them)
            vals_out[dst] = s_val[i];
        }
    }
}

// workgroups to launch for `capacity` instances: the device picks the keys per workgroup from the real count
static int64_t tile_sort_blocks(int64_t capacity)
{
    const int64_t nb = ceil_div(capacity > 0 ? capacity : 1, (int64_t)ts_items(capacity) * kTsThreads);
    return nb;
}
static size_t tile_sort_ws_bytes(int64_t capacity)
{
    int64_t nb = tile_sort_blocks(capacity);
    if (nb < kTsCus) nb = kTsCus;
    return align_up((size_t)nb * kTsDigits * 4) + align_up((size_t)kTsDigits * 4);
}
static int64_t wtile_sort_blocks(int64_t capacity)
{
    return ceil_div(capacity > 0 ? capacity : 1, (int64_t)ws_items(capacity) * kWsThreads);
}
static size_t wtile_sort_ws_bytes(int64_t capacity)
{
    int64_t nb = wtile_sort_blocks(capacity);
    if (nb < kTsCus) nb = kTsCus;
    return align_up((size_t)nb * kWsDigits * 4) + align_up((size_t)kWsDigits * 4);
}

}  // namespace gsx

using namespace gsx;

// scan workspace that covers the chunk table of the largest image (65535/16)^2 tiles
static size_t chunk_scan_ws_bytes() { return gsx_scan_workspace_bytes((int64_t)4096 * 4096 + 1); }

GSX_API size_t gsx_bin_workspace_bytes(int32_t N, int64_t capacity)
{
    size_t a = gsx_scan_workspace_bytes(N > 0 ? N : 1);
    size_t b = sort_ws_bytes_internal(capacity > 0 ? capacity : 1);
    size_t c = chunk_scan_ws_bytes();
    size_t t = tile_sort_ws_bytes(capacity), t13 = wtile_sort_ws_bytes(capacity);
    if (t13 > t) t = t13;
    size_t m = a > b ? a : b;
    if (t > m) m = t;
    return 256 + (m > c ? m : c);
}

GSX_API int64_t gsx_max_chunks(int64_t D, int32_t H, int32_t W, int32_t chunk_force)
{
    const int64_t n_tiles = (int64_t)((W + GSX_TILE - 1) / GSX_TILE) * ((H + GSX_TILE - 1) / GSX_TILE);
    if (D < 0) D = 0;
    // the largest chunk count any instance count <= D can produce: the device shortens the chunks of small views
    int64_t full;
    if (chunk_force == 64 || chunk_force == 128 || chunk_force == 256) full = D / chunk_force;
    else {
        const int64_t a = (D < (1ll << 19) ? D : (1ll << 19) - 1) / 64, b = (D < (1ll << 20) ? D : (1ll << 20) - 1) / 128, c = D / GSX_CHUNK;
        full = a > b ? a : b; if (c > full) full = c;
    }
    return full + (D < n_tiles ? D : n_tiles);
}

GSX_API int gsx_bin_count(const uint32_t* tiles_touched, const uint64_t* tt_block_sums, int32_t N, uint32_t* offsets, uint64_t* total_dev,
                          int64_t* D_pinned_host, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || !total_dev) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (N > 0 && (!tiles_touched || !offsets || !ws || ws_bytes < gsx_scan_workspace_bytes(N))) return GSX_E_INVALID;
    // D and the caller's word reach the host from the scan kernel itself when the 16 bytes are pinned memory the device can
    // address (hipHostMalloc / registered: what torch's pin_memory() gives) -- a hipMemcpyAsync of 16 bytes is a blit kernel,
    // i.e. one more dependent launch (~4.5 us) in the middle of every view; anything else gets the copy
    uint64_t* mirror = nullptr;
    if (D_pinned_host && N > 0) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, D_pinned_host, 0) == hipSuccess) mirror = static_cast<uint64_t*>(dp);
        else (void)hipGetLastError();
    }
    int s = scan_internal(tiles_touched, offsets, total_dev, N, ws, ws_bytes, st, tt_block_sums, kBlock, mirror);     // N == 0: total = 0
    if (s != GSX_OK) return s;
    if (D_pinned_host && !mirror) GSX_HIP(hipMemcpyAsync(D_pinned_host, total_dev, 16, hipMemcpyDeviceToHost, st));
    return GSX_OK;
}

GSX_API int gsx_bin_emit_sort(const float* rec, const uint32_t* tile_box, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                              int64_t capacity, const uint64_t* total_dev,
                              uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                              uint32_t* ranges, uint32_t* chunk_base, int32_t chunk_force, void* ws, size_t ws_bytes, void* stream)
{
    return gsx::bin_emit_sort_impl(rec, tile_box, offsets, N, H, W, capacity, total_dev, tiles_sorted, ranks_sorted, tiles_tmp, ranks_tmp,
                                   ranges, chunk_base, chunk_force, ws, ws_bytes, nullptr, stream, nullptr);
}

// info (nullable, device-visible pinned host memory): images of up to 4096 tiles leave 1 + their number of empty tiles there
int gsx::bin_emit_sort_impl(const float* rec, const uint32_t* tile_box, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                            int64_t capacity, const uint64_t* total_dev,
                            uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                            uint32_t* ranges, uint32_t* chunk_base, int32_t chunk_force, void* ws, size_t ws_bytes, uint64_t* info, void* stream,
                            const EmitScan* scan)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || capacity <= 0) return GSX_E_INVALID;
    if (chunk_force != 0 && chunk_force != 64 && chunk_force != 128 && chunk_force != 256) return GSX_E_INVALID;
    if (capacity >= 0xFFFFFFFFll) return GSX_E_RANGE;
    if (!ranges || !chunk_base || !total_dev || !ws || ws_bytes < gsx_bin_workspace_bytes(N, capacity)) return GSX_E_INVALID;
    if (!tiles_sorted || !ranks_sorted || !tiles_tmp || !ranks_tmp) return GSX_E_INVALID;
    if (N > 0 && (!rec || !offsets)) return GSX_E_INVALID;
    if (scan && (N <= 0 || !scan->sums || !scan->offsets_out || scan->offsets_out != offsets || scan->total != total_dev)) return GSX_E_INVALID;
    if (W > 256 * GSX_TILE || H > 256 * GSX_TILE) tile_box = nullptr;       // the packed boxes hold 8-bit tile coordinates
    hipStream_t st = (hipStream_t)stream;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE, n_tiles = tw * th;
    void* sub_ws = (char*)ws + 256; size_t sub_bytes = ws_bytes - 256;
    const int bits = tile_bits(n_tiles);
    // scan != nullptr (N > 0): offsets[] and *total_dev are this call's to write (emit_kernel<true>)
    auto launch_emit = [&](uint32_t* e_tiles, uint32_t* e_ranks) {
        const dim3 grid((unsigned)ceil_div(N, kBlock));
        if (scan) emit_kernel<true><<<grid, dim3(kBlock), 0, st>>>(rec, tile_box, nullptr, N, tw, (uint32_t)capacity, e_tiles, e_ranks,
                                                                   scan->sums, scan->offsets_out, scan->total, scan->mirror);
        else emit_kernel<false><<<grid, dim3(kBlock), 0, st>>>(rec, tile_box, offsets, N, tw, (uint32_t)capacity, e_tiles, e_ranks,
                                                               nullptr, nullptr, nullptr, nullptr);
    };
    if (n_tiles <= kTsDigits) {
        // one 12-bit pass; ranges and chunk table come out of the scatter's workgroup 0
        // grid for the largest workgroup count any instance count <= capacity can need (fewer instances -> fewer keys per
        // workgroup, up to one workgroup per CU; past 256 x 16384 instances the workgroups stay at 16384 keys)
        int64_t nbl = tile_sort_blocks(capacity);
        const int64_t small = ceil_div(capacity, 2 * kTsThreads);        // workgroups of the smallest tile
        if (nbl < kTsCus) nbl = small < kTsCus ? small : kTsCus;
        const unsigned nb = (unsigned)nbl;
        uint32_t* hist = (uint32_t*)sub_ws;
        uint32_t* totals = (uint32_t*)((char*)sub_ws + align_up((size_t)nb * kTsDigits * 4));
        if (N > 0) {
            launch_emit(tiles_tmp, ranks_tmp);
            tile_hist_kernel<<<dim3(nb), dim3(kTsThreads), 0, st>>>(tiles_tmp, total_dev, capacity, n_tiles, hist);
        }
        tile_colscan_kernel<<<dim3((unsigned)ceil_div(n_tiles, kTsSegDigits)), dim3(kTsThreads), 0, st>>>(hist, total_dev, capacity, n_tiles, totals);
        tile_scatter_kernel<<<dim3(nb), dim3(kTsThreads), 0, st>>>(tiles_tmp, ranks_tmp, (scan && scan->no_sorted_keys) ? nullptr : tiles_sorted, ranks_sorted, total_dev, capacity, n_tiles,
                                                                    hist, totals, chunk_force, ranges, chunk_base, info);
        GSX_LAUNCH_CHECK();
        return GSX_OK;
    }
    if (n_tiles <= kWsDigits) {
        // one 13-bit pass (a 1080p image: 8160 tiles): as above with 512-thread workgroups of up to 16384 keys
        int64_t nbl = wtile_sort_blocks(capacity);
        const int64_t small = ceil_div(capacity, 2 * kWsThreads);
        if (nbl < kTsCus) nbl = small < kTsCus ? small : kTsCus;
        const unsigned nb = (unsigned)nbl;
        uint32_t* hist = (uint32_t*)sub_ws;
        uint32_t* totals = (uint32_t*)((char*)sub_ws + align_up((size_t)nb * kWsDigits * 4));
        if (N > 0) {
            launch_emit(tiles_tmp, ranks_tmp);
            wtile_hist_kernel<<<dim3(nb), dim3(kWsThreads), 0, st>>>(tiles_tmp, total_dev, capacity, n_tiles, hist);
        }
        wtile_colscan_kernel<<<dim3((unsigned)ceil_div(n_tiles, kTsSegDigits)), dim3(kTsThreads), 0, st>>>(hist, total_dev, capacity, n_tiles, totals);
        wtile_scatter_kernel<<<dim3(nb), dim3(kWsThreads), 0, st>>>(tiles_tmp, ranks_tmp, (scan && scan->no_sorted_keys) ? nullptr : tiles_sorted, ranks_sorted, total_dev, capacity, n_tiles,
                                                                     hist, totals, chunk_force, ranges, chunk_base);
        GSX_LAUNCH_CHECK();
        return GSX_OK;
    }
    const int passes = (bits + 7) / 8;
    // emit into the buffer the sort does NOT write first, so that the last pass lands in *_sorted
    uint32_t* e_t = (passes & 1) ? tiles_tmp : tiles_sorted;
    uint32_t* e_r = (passes & 1) ? ranks_tmp : ranks_sorted;
    if (N > 0) {
        launch_emit(e_t, e_r);
        GSX_LAUNCH_CHECK();
        int s = sort_pairs_internal(e_t, e_r, tiles_sorted, ranks_sorted, tiles_tmp, ranks_tmp, capacity, total_dev, 0, bits, sub_ws, sub_bytes, st);
        if (s != GSX_OK) return s;
    }
    ranges_kernel<<<dim3((unsigned)ceil_div(n_tiles + 1, kBlock)), dim3(kBlock), 0, st>>>(tiles_sorted, total_dev, (uint32_t)capacity, n_tiles, chunk_force, ranges, chunk_base);
    GSX_LAUNCH_CHECK();
    return scan_internal(chunk_base, chunk_base, nullptr, (int64_t)n_tiles + 1, sub_ws, sub_bytes, st);
}

GSX_API int gsx_bin_instances(const float* rec, const uint32_t* tiles_touched, int32_t N, int32_t H, int32_t W,
                              int64_t capacity, uint32_t* offsets,
                              uint32_t* tiles_sorted, uint32_t* ranks_sorted, uint32_t* tiles_tmp, uint32_t* ranks_tmp,
                              uint32_t* ranges, uint32_t* chunk_base, int32_t chunk_force, int64_t* D_host, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || capacity < 0) return GSX_E_INVALID;
    if (!ranges || !chunk_base || !D_host || !ws || ws_bytes < gsx_bin_workspace_bytes(N, capacity)) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE, n_tiles = tw * th;
    *D_host = 0;
    auto empty = [&]() -> int {
        GSX_HIP(hipMemsetAsync(ranges, 0, (size_t)n_tiles * 8, st));
        GSX_HIP(hipMemsetAsync(chunk_base, 0, (size_t)(n_tiles + 1) * 4, st));
        GSX_HIP(hipMemsetD32Async((hipDeviceptr_t)(chunk_base + n_tiles + 1), (int)chunk_size_for(0, chunk_force), 1, st));
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
    return gsx_bin_emit_sort(rec, nullptr, offsets, N, H, W, capacity, total_dev, tiles_sorted, ranks_sorted, tiles_tmp, ranks_tmp,
                             ranges, chunk_base, chunk_force, ws, ws_bytes, stream);
}
