// gsx_sort.hip -- device exclusive scan and stable LSD radix sort of (u32 key, u32 value) pairs
// for gfx950 (wave64).  Used twice on the render path:
//   * depth order: keys = order-preserving depth bits (32 bits, 4 passes), values = Gaussian index
//     (replaces torch.argsort, /root/reference/model/gaussian_splatting.py:198)
//   * binning: keys = tile id (ceil(log2(tiles)) bits, usually 2 passes), values = depth rank;
//     instances are emitted in rank order, so a STABLE sort on the tile id alone yields the
//     (tile << 32 | rank) key order with 4-byte keys and 2 passes instead of 64-bit keys and 5.
// HBM-bound integer work: coalesced 4-byte streams, digit ranking with wave64 ballots, no MFMA.
#include <atomic>
#include "gsx_common.h"

namespace gsx {

thread_local int g_last_hip_error = 0;

// ------------------------------------------------------------------------------------------
// exclusive scan (u32 values, u64 block sums / total)
// ------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;   // 2048

__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(const uint32_t* __restrict__ in, int64_t n, uint64_t* __restrict__ block_sums)
{
    __shared__ uint64_t s_w[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        int64_t i = base + k * kBlock + threadIdx.x;
        if (i < n) s += in[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// Second (and last) level: workgroup b first adds up the sums of everything before its tile -- `ratio` entries of
// `sums` per tile, b * ratio values read by 256 threads, i.e. O(nb^2 / 2) reads of a table that sits in L2 -- instead of a
// third launch for the scan of the sums (a dependent launch costs ~4.6 us; at 1 M elements the table is 4 KB).  The
// last workgroup also writes the grand total.
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n,
                                                            const uint64_t* __restrict__ sums, int ratio, uint64_t* __restrict__ total,
                                                            uint64_t* __restrict__ mirror)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint64_t s_pre[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t acc = 0;
    const int64_t before = (int64_t)blockIdx.x * ratio;
    {
        int64_t j = threadIdx.x;                      // 4 loads in flight: the last workgroups walk ~16 strides of the table
        uint64_t a1 = 0, a2 = 0, a3 = 0;
        for (; j + 3 * kBlock < before; j += 4 * kBlock) { acc += sums[j]; a1 += sums[j + kBlock]; a2 += sums[j + 2 * kBlock]; a3 += sums[j + 3 * kBlock]; }
        for (; j < before; j += kBlock) acc += sums[j];
        acc += (a1 + a2) + a3;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if (lane == 0) s_pre[wave] = acc;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;   // blocked layout
    uint32_t v[kScanItems]; uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { int64_t i = base + k; v[k] = (i < n) ? in[i] : 0u; s += v[k]; }
    uint32_t tot;
    const uint32_t ex0 = block_excl_scan_u32(s, s_wave, &tot);          // barriers inside: s_pre is visible after it
    const uint64_t pre = (s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]);
    uint32_t ex = ex0 + (uint32_t)pre;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { int64_t i = base + k; if (i < n) out[i] = ex; ex += v[k]; }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        *total = pre + (uint64_t)tot;
        // mirror (nullable): two u64 of device-visible pinned HOST memory -- the total and the caller's word behind it go there
        // straight from this kernel (a 16-byte hipMemcpyAsync is a blit kernel of its own: one more dependent launch)
        if (mirror) { mirror[0] = pre + (uint64_t)tot; mirror[1] = total[1]; }
    }
}

// one workgroup of 1024 threads scans up to kSmallScan values in a single launch (radix histograms of
// <= 4M keys, the chunk table): a dependent launch costs ~4.6 us, three of them more than the work
constexpr int kSmallThreads = 1024;
constexpr int kSmallScan = 8192;         // one sweep
__global__ __launch_bounds__(kSmallThreads) void scan_small_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                   int64_t n, uint64_t* __restrict__ total, uint64_t* __restrict__ mirror)
{
    __shared__ uint32_t s_w[kSmallThreads / 64];
    __shared__ uint64_t s_carry;                              // 64-bit so that *total is exact (prefixes wrap to u32)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    constexpr int kPer = 8;                                   // values per thread per sweep (blocked: 32 B per lane)
    for (int64_t base = 0; base < n; base += (int64_t)kSmallThreads * kPer) {
        const int64_t i0 = base + (int64_t)threadIdx.x * kPer;
        uint32_t v[kPer]; uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) { v[k] = (i0 + k < n) ? in[i0 + k] : 0u; s += v[k]; }
        const uint32_t incl = wave_incl_scan_u32(s);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint64_t wbase = s_carry;
        for (int w = 0; w < wave; w++) wbase += s_w[w];
        uint32_t ex = (uint32_t)wbase + incl - s;
#pragma unroll
        for (int k = 0; k < kPer; k++) { if (i0 + k < n) out[i0 + k] = ex; ex += v[k]; }
        __syncthreads();
        if (threadIdx.x == kSmallThreads - 1) s_carry = wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) {
        *total = s_carry;
        if (mirror) { mirror[0] = s_carry; mirror[1] = total[1]; }
    }
}

// block_sums (nullable): sums of consecutive groups of `sums_tile` elements that the producer of `in` already has
// (build_records: one per 256 ranks); then the scan is ONE launch.
static int scan_launch(const uint32_t* in, uint32_t* out, uint64_t* total_dev, int64_t n, void* ws, size_t ws_bytes, hipStream_t st,
                       const uint64_t* block_sums = nullptr, int sums_tile = 0, uint64_t* mirror = nullptr)
{
    if (n <= kSmallScan && !block_sums) {
        scan_small_kernel<<<dim3(1), dim3(kSmallThreads), 0, st>>>(in, out, n, total_dev, mirror);
        GSX_LAUNCH_CHECK();
        return GSX_OK;
    }
    int64_t nb = ceil_div(n, kScanTile);
    if (block_sums) {
        if (sums_tile <= 0 || kScanTile % sums_tile != 0) return GSX_E_INVALID;
        scan_apply_kernel<<<dim3((unsigned)nb), dim3(kBlock), 0, st>>>(in, out, n, block_sums, kScanTile / sums_tile, total_dev, mirror);
        GSX_LAUNCH_CHECK();
        return GSX_OK;
    }
    if (ws_bytes < (size_t)nb * 8) return GSX_E_CAPACITY;
    uint64_t* sums = (uint64_t*)ws;
    scan_reduce_kernel<<<dim3((unsigned)nb), dim3(kBlock), 0, st>>>(in, n, sums);
    scan_apply_kernel<<<dim3((unsigned)nb), dim3(kBlock), 0, st>>>(in, out, n, sums, 1, total_dev, mirror);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

// ------------------------------------------------------------------------------------------
// radix sort, 8-bit digits
// ------------------------------------------------------------------------------------------
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortItems = 8;                       // keys per thread
constexpr int kSortTile = kBlock * kSortItems;      // 4096 keys per workgroup

// hist[d * nb + b] = number of keys of workgroup b with digit d
// n_dev (nullable): element count in device memory, clamped to n -- lets the binning stage enqueue the sort
// before the host knows the instance count (the grid is then sized by the buffers' capacity)
__device__ __forceinline__ int64_t device_count(const uint64_t* __restrict__ n_dev, int64_t n)
{
    if (!n_dev) return n;
    const uint64_t d = *n_dev;
    return d < (uint64_t)n ? (int64_t)d : n;
}

// check_bit < 32 (first pass of a sort only): row kRadix of the table gets the workgroup's number of keys whose bits
// [check_bit, 32) differ from keys[0]'s -- the caller's proof that a sort which skipped those bits was entitled to.
__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, const uint64_t* __restrict__ n_dev,
                                                            int shift, uint32_t mask, uint32_t* __restrict__ hist, uint32_t nb, int check_bit)
{
    __shared__ uint32_t s_hist[kRadix];
    __shared__ uint32_t s_mis[4];
    n = device_count(n_dev, n);
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    const uint32_t ref = (check_bit < 32 && n > 0) ? (keys[0] >> check_bit) : 0u;
    uint32_t mis = 0;
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        int64_t i = base + k * kBlock + threadIdx.x;
        if (i < n) {
            const uint32_t key = keys[i];
            atomicAdd(&s_hist[(key >> shift) & mask], 1u);
            if (check_bit < 32) mis += ((key >> check_bit) != ref) ? 1u : 0u;
        }
    }
    if (check_bit < 32) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mis += __shfl_down(mis, d, 64);
        if ((threadIdx.x & 63) == 0) s_mis[threadIdx.x >> 6] = mis;
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nb + blockIdx.x] = s_hist[threadIdx.x];
    if (check_bit < 32 && threadIdx.x == 0) hist[(size_t)kRadix * nb + blockIdx.x] = (s_mis[0] + s_mis[1]) + (s_mis[2] + s_mis[3]);
}

// one workgroup per digit: exclusive scan of that digit's per-workgroup counts in place, row total -> totals[d].
// (hist + rowscan + scatter = 3 launches per pass; the scatter kernel turns the 256 totals into digit bases itself)
// Workgroup kRadix, when launched, adds up the mismatch row instead (radix_hist_kernel) -> *mismatch.
__global__ __launch_bounds__(kBlock) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nb, uint32_t* __restrict__ totals,
                                                               uint64_t* __restrict__ mismatch)
{
    __shared__ uint32_t s_wave[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nb;
    uint32_t carry = 0;
    const bool plain_sum = blockIdx.x == kRadix;
    for (uint32_t base = 0; base < nb; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < nb) ? row[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, s_wave, &tot);
        if (i < nb && !plain_sum) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { if (plain_sum) *mismatch = (uint64_t)carry; else totals[blockIdx.x] = carry; }
}

// Stable scatter of one 2048-key tile.  Each wave owns a CONTIGUOUS quarter of the tile (8 rounds of 64
// consecutive keys), ranks its keys against its own running digit counters in LDS (same-wave LDS operations
// retire in order, so the rounds need no barrier), and only then do the four waves meet to turn the counters into
// positions.  The pairs then pass through LDS in digit order, so that the stores to HBM are runs of consecutive
// addresses: a scattered 4-byte store costs one L2 transaction, and 2 per key is what bounded this kernel.
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                               uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                               int64_t n, const uint64_t* __restrict__ n_dev, int shift, uint32_t mask,
                                                               const uint32_t* __restrict__ hist_scanned, uint32_t nb,
                                                               const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_cnt[4][kRadix];     // per-wave digit counts, then per-wave first positions inside the tile
    __shared__ uint32_t s_delta[kRadix];      // per digit: global slot minus position inside the tile
    __shared__ uint32_t s_key[kSortTile], s_val[kSortTile];
    n = device_count(n_dev, n);
    __shared__ uint32_t s_scan[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaveKeys = kSortTile / 4;  // 512
    const int64_t tile0 = (int64_t)blockIdx.x * kSortTile;
    const int64_t base = tile0 + (int64_t)wave * kWaveKeys;
    uint32_t tot;
    const uint32_t digit_base = block_excl_scan_u32(totals[tid], s_scan, &tot);   // two barriers inside
    s_cnt[0][tid] = 0; s_cnt[1][tid] = 0; s_cnt[2][tid] = 0; s_cnt[3][tid] = 0;

    uint32_t key[kSortItems], val[kSortItems], loc[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const int64_t i = base + k * 64 + lane;
        key[k] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
        val[k] = (i < n) ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
    }
    __syncthreads();                          // counters zeroed
    uint32_t* cnt = s_cnt[wave];
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const bool valid = base + k * 64 + lane < n;
        const uint32_t d = (key[k] >> shift) & mask;
        uint32_t rank, count;
        wave_match_rank<kRadixBits>(d, valid, rank, count);             // lanes of this wave holding the same digit (stable: lane order == key order)
        const uint32_t old = cnt[d];                                   // every lane reads before the leader writes
        if (valid && rank == 0) cnt[d] = old + count;
        loc[k] = old + rank;
    }
    __syncthreads();
    {
        const uint32_t c0 = s_cnt[0][tid], c1 = s_cnt[1][tid], c2 = s_cnt[2][tid], c3 = s_cnt[3][tid];
        uint32_t ltot;
        const uint32_t lstart = block_excl_scan_u32((c0 + c1) + (c2 + c3), s_scan, &ltot);      // digit tid's first position inside the tile
        s_cnt[0][tid] = lstart; s_cnt[1][tid] = lstart + c0; s_cnt[2][tid] = lstart + c0 + c1; s_cnt[3][tid] = lstart + c0 + c1 + c2;
        s_delta[tid] = digit_base + hist_scanned[(size_t)tid * nb + blockIdx.x] - lstart;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        if (base + k * 64 + lane < n) {
            const uint32_t pos = cnt[(key[k] >> shift) & mask] + loc[k];
            s_key[pos] = key[k]; s_val[pos] = val[k];
        }
    }
    __syncthreads();
    const int64_t left = n - tile0;
    const uint32_t here = (uint32_t)(left < (int64_t)kSortTile ? (left > 0 ? left : 0) : (int64_t)kSortTile);
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t i = (uint32_t)(k * kBlock + tid);
        if (i < here) {
            const uint32_t kk = s_key[i];
            const uint32_t dst = s_delta[(kk >> shift) & mask] + i;
            keys_out[dst] = kk;
            vals_out[dst] = s_val[i];
        }
    }
}

__global__ __launch_bounds__(kBlock) void iota_kernel(uint32_t* __restrict__ v, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) v[i] = i;
}

static size_t sort_ws_bytes(int64_t n)
{
    int64_t nb = ceil_div(n > 0 ? n : 1, kSortTile);
    int64_t hist_n = nb * (kRadix + 1);          // + the mismatch row of a checked sort
    return align_up((size_t)hist_n * 4) + align_up((size_t)kRadix * 4);
}

// Sorts n pairs on key bits [begin_bit, end_bit).  Pass 0 reads (k_in, v_in); passes alternate
// between (k_a, v_a) and (k_b, v_b) such that the LAST pass writes (k_a, v_a).  No pass reads and
// writes the same buffer provided k_in differs from the first destination:
//   odd pass count  -> first destination is a;  even -> first destination is b.
// n_dev != NULL: the real count lives in device memory (clamped to n, which then only sizes the grids).
int sort_pairs_internal(const uint32_t* k_in, const uint32_t* v_in, uint32_t* k_a, uint32_t* v_a,
                        uint32_t* k_b, uint32_t* v_b, int64_t n, const uint64_t* n_dev, int begin_bit, int end_bit,
                        void* ws, size_t ws_bytes, hipStream_t st, int check_bit, uint64_t* mismatch_dev)
{
    if (n <= 0) { if (mismatch_dev) GSX_HIP(hipMemsetAsync(mismatch_dev, 0, 8, st)); return GSX_OK; }
    if (n >= (int64_t)0xFFFFFFFFll) return GSX_E_RANGE;
    if (ws_bytes < sort_ws_bytes(n)) return GSX_E_CAPACITY;
    const int passes = (end_bit - begin_bit + kRadixBits - 1) / kRadixBits;
    const uint32_t nb = (uint32_t)ceil_div(n, kSortTile);
    const int64_t hist_n = (int64_t)nb * (kRadix + 1);
    uint32_t* hist = (uint32_t*)ws;
    uint32_t* totals = (uint32_t*)((char*)ws + align_up((size_t)hist_n * 4));
    const bool check = mismatch_dev != nullptr && check_bit < 32;
    if (mismatch_dev && !check) GSX_HIP(hipMemsetAsync(mismatch_dev, 0, 8, st));      // nothing above bit 32 can differ
    if (passes == 0) {   // nothing to sort on: copy through
        GSX_HIP(hipMemcpyAsync(k_a, k_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        if (v_in) GSX_HIP(hipMemcpyAsync(v_a, v_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        else iota_kernel<<<dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, st>>>(v_a, (uint32_t)n);     // "values_in == NULL means 0..n-1" holds here too
        if (check) {     // the check rides on a histogram launch
            radix_hist_kernel<<<dim3(nb), dim3(kBlock), 0, st>>>(k_in, n, n_dev, 0, 0u, hist, nb, check_bit);
            radix_rowscan_kernel<<<dim3(kRadix + 1), dim3(kBlock), 0, st>>>(hist, nb, totals, mismatch_dev);
            GSX_LAUNCH_CHECK();
        }
        return GSX_OK;
    }
    const uint32_t* ks = k_in; const uint32_t* vs = v_in;
    for (int p = 0; p < passes; p++) {
        const bool to_a = ((passes - 1 - p) % 2) == 0;
        uint32_t* kd = to_a ? k_a : k_b; uint32_t* vd = to_a ? v_a : v_b;
        const int shift = begin_bit + p * kRadixBits;
        const int bits = (end_bit - shift) < kRadixBits ? (end_bit - shift) : kRadixBits;
        const uint32_t mask = (1u << bits) - 1u;
        const bool chk = check && p == 0;
        radix_hist_kernel<<<dim3(nb), dim3(kBlock), 0, st>>>(ks, n, n_dev, shift, mask, hist, nb, chk ? check_bit : 32);
        radix_rowscan_kernel<<<dim3(kRadix + (chk ? 1 : 0)), dim3(kBlock), 0, st>>>(hist, nb, totals, mismatch_dev);
        radix_scatter_kernel<<<dim3(nb), dim3(kBlock), 0, st>>>(ks, vs, kd, vd, n, n_dev, shift, mask, hist, nb, totals);
        GSX_LAUNCH_CHECK();
        ks = kd; vs = vd;
    }
    return GSX_OK;
}

int scan_internal(const uint32_t* in, uint32_t* out, uint64_t* total_dev, int64_t n, void* ws, size_t ws_bytes, hipStream_t st,
                  const uint64_t* block_sums, int sums_tile, uint64_t* mirror)
{
    if (n <= 0) {
        if (total_dev) GSX_HIP(hipMemsetAsync(total_dev, 0, 8, st));
        if (total_dev && mirror) GSX_HIP(hipMemcpyAsync(mirror, total_dev, 16, hipMemcpyDeviceToHost, st));
        return GSX_OK;
    }
    return scan_launch(in, out, total_dev, n, ws, ws_bytes, st, block_sums, sums_tile, mirror);
}

size_t sort_ws_bytes_internal(int64_t n) { return sort_ws_bytes(n); }

}  // namespace gsx

using namespace gsx;

GSX_API size_t gsx_sort_workspace_bytes(int64_t n) { return sort_ws_bytes(n); }

GSX_API int gsx_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* values_in, uint32_t* keys_out, uint32_t* values_out,
                               uint32_t* keys_tmp, uint32_t* values_tmp, int64_t n, int begin_bit, int end_bit,
                               int check_bit, uint64_t* mismatch_dev, void* ws, size_t ws_bytes, void* stream)
{
    if (n < 0 || begin_bit < 0 || end_bit > 32 || begin_bit > end_bit || check_bit < 0 || check_bit > 32) return GSX_E_INVALID;
    if (n == 0) {       // nothing to sort, but the caller reads the mismatch count either way: no key differs
        if (mismatch_dev) GSX_HIP(hipMemsetAsync(mismatch_dev, 0, 8, (hipStream_t)stream));
        return GSX_OK;
    }
    if (!keys_in || !keys_out || !values_out || !keys_tmp || !values_tmp || !ws) return GSX_E_INVALID;
    if (keys_in == keys_out || keys_in == keys_tmp || keys_out == keys_tmp) return GSX_E_INVALID;
    // a pass that scatters from the buffer it scatters into corrupts the permutation silently
    if (values_in && (values_in == values_out || values_in == values_tmp)) return GSX_E_INVALID;
    if (values_out == values_tmp) return GSX_E_INVALID;
    return sort_pairs_internal(keys_in, values_in, keys_out, values_out, keys_tmp, values_tmp, n, nullptr, begin_bit, end_bit,
                               ws, ws_bytes, (hipStream_t)stream, check_bit, mismatch_dev);
}

GSX_API size_t gsx_scan_workspace_bytes(int64_t n) { return align_up((size_t)ceil_div(n > 0 ? n : 1, kScanTile) * 8); }

GSX_API int gsx_scan_u32(const uint32_t* in, uint32_t* out_exclusive, uint64_t* total_dev, int64_t n,
                         void* ws, size_t ws_bytes, void* stream)
{
    if (n < 0) return GSX_E_INVALID;
    if (n > 0 && (!in || !out_exclusive || !ws)) return GSX_E_INVALID;
    return scan_internal(in, out_exclusive, total_dev, n, ws, ws_bytes, (hipStream_t)stream, nullptr, 0, nullptr);
}

// ------------------------------------------------------------------------------------------
// The one entry point that allocates nothing and takes no workspace (no counterpart in the reference): the same sort,
// same kernels, with the ping-pong buffers and the histogram workspace in a file-scope device array.  For callers that
// cannot call hipMalloc or size a workspace: code under HIP graph capture, tight serving loops, harnesses that forbid it.
// Every GPU the library runs on holds its own copy of the array (kStaticSortBytes, ~8.6 MB) until the process ends.
// ------------------------------------------------------------------------------------------
namespace gsx {

constexpr int64_t kStaticSortMaxN = GSX_STATIC_SORT_MAX_N;
// keys_tmp | values_tmp | ws, each part 256-byte aligned (kStaticSortMaxN * 4 is)
constexpr size_t kStaticSortWs = ((size_t)((kStaticSortMaxN + kSortTile - 1) / kSortTile) * (kRadix + 1) * 4 + 255) / 256 * 256 + (size_t)kRadix * 4;
constexpr size_t kStaticSortBytes = 2 * (size_t)kStaticSortMaxN * 4 + kStaticSortWs;
static __device__ __align__(256) unsigned char scratch[kStaticSortBytes];

// In host code `scratch` is a stand-in, not a device address: each device's copy is looked up once, with that device
// current, and only that pointer is ever handed to a kernel or a memory call.
constexpr int kMaxDevices = 64;
static std::atomic<void*> g_scratch_ptr[kMaxDevices];

static int static_scratch(void** out)
{
    int dev = -1;
    GSX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return GSX_E_INVALID;
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {       // first call on this device (two threads racing here look up the same address)
        GSX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return GSX_E_HIP;
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return GSX_OK;
}

}  // namespace gsx

GSX_API int gsx_sort_pairs_u32_static(const uint32_t* keys_in, const uint32_t* values_in, uint32_t* keys_out, uint32_t* values_out,
                                      int64_t n, int begin_bit, int end_bit, void* stream)
{
    if (n < 0 || begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) return GSX_E_INVALID;
    if (n > kStaticSortMaxN) return GSX_E_CAPACITY;            // beyond the static scratch: use gsx_sort_pairs_u32 with a workspace
    if (n == 0) return GSX_OK;
    if (!keys_in || !keys_out || !values_out || keys_in == keys_out) return GSX_E_INVALID;
    if (values_in && values_in == values_out) return GSX_E_INVALID;      // with an odd pass count pass 0 would scatter in place
    void* base = nullptr;
    const int st = static_scratch(&base);
    if (st != GSX_OK) return st;
    unsigned char* b = static_cast<unsigned char*>(base);
    uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(b);
    uint32_t* values_tmp = reinterpret_cast<uint32_t*>(b + (size_t)kStaticSortMaxN * 4);
    void* ws = b + 2 * (size_t)kStaticSortMaxN * 4;
    // nothing in the scratch has to start at zero: every pass's histogram kernel writes each of the table entries the pass
    // reads, and the ping-pong buffers are written before they are read
    return sort_pairs_internal(keys_in, values_in, keys_out, values_out, keys_tmp, values_tmp, n, nullptr, begin_bit, end_bit,
                               ws, kStaticSortWs, (hipStream_t)stream, 32, nullptr);
}
