// gsx_composite.hip -- alpha compositing, forward and backward (gfx950, wave64).
// Replaces the inner loop of _render_chunk (/root/reference/model/gaussian_splatting.py:118-148)
// and the autograd graph recorded over it.
//
// Work decomposition (MI355X-first: >> 256 equal work items whatever the splats' screen
// distribution -- the orbit views of BASELINE config 3 put 1M splats on ~400 of 4096 tiles):
//   * a tile's depth-ordered list is cut into CHUNKS of <= GSX_CHUNK instances;
//   * one wave64 composites one chunk for the whole 16x16 tile, 4 pixels per lane (one per 8x8
//     quadrant), records staged through LDS 64 at a time and read back as broadcasts; a quadrant
//     the splat's pixel box misses is skipped with a scalar branch;
//   * compositing is associative: chunk c yields (C_loc, T_loc); a per-tile pass scans the chunks
//     (prefix transmittance forward, colour-behind backward) and writes the image.
// Transmittance is carried as (mantissa, exponent) so that long lists neither underflow nor lose
// the relative accuracy the backward needs.  The backward walks each chunk back to front:
//   T_k = T_{k+1} / (1 - alpha_k),  dL/dalpha_k = T_k (c_k.G - W_k),  W_{k-1} = alpha_k c_k.G + (1 - alpha_k) W_k
// starting from the chunk-end state (T_end, W_end = G . colour-behind) the forward left -- no
// subtraction of nearly equal sums anywhere.  Nine partial sums per (tile, splat) are reduced over
// the wave with DPP and stored as ONE row per instance: no atomics, bitwise reproducible.
//
// Bound: VALU (about 25 VALU + 1 v_exp_f32 per pixel-splat pair forward, ~45 + reduction
// backward); HBM traffic per instance is 52 B forward + 24 B of chunk state.  No MFMA: there is no
// contraction on this path.
#include "gsx_common.h"
#include <type_traits>

namespace gsx {

constexpr int kCBatch = 64;            // records staged per LDS batch (one per lane)
// backward work lists: classes 0..3 by expected cost (quadrant evaluations the forward counted, scaled by the
// share of quadrants that are not occluded), class 4 = fully occluded (zero rows only).  Layout behind the flag
// and dark words: 16 counters (two sets of 8: normal pass / pass after a global early exit), then for each set
// kWorkClasses lists of max_chunks chunk ids.
constexpr int kWorkClasses = 5;
// a list entry: chunk id | kItemFront (the first half of a chunk cut in two) | kItemBack (its second half); a chunk in two
// halves is two entries, so a list has room for 2 x max_chunks of them
constexpr uint32_t kItemFront = 0x80000000u, kItemBack = 0x40000000u, kItemChunk = 0x3FFFFFFFu;
__device__ __forceinline__ uint32_t* work_counters(uint32_t* work, bool pass1) { return work + (pass1 ? 8 : 0); }
__device__ __forceinline__ uint32_t* work_list(uint32_t* work, bool pass1, int cls, uint32_t max_chunks)
{
    return work + 16 + ((size_t)(pass1 ? kWorkClasses : 0) + (size_t)cls) * 2u * max_chunks;
}
constexpr int kPixPerTile = GSX_TILE * GSX_TILE;

__device__ __forceinline__ float pix_coord(int i, int steps, float step)
{
    // torch.linspace(-1, 1, steps) in fp32 (:100-102): symmetric fill, one fma per element
    if (steps == 1) return -1.0f;
    return (i < (steps >> 1)) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

// state[0] -> number of ranks to composite (N when the global early exit does not fire)
__device__ __forceinline__ uint32_t rank_limit_from_state(const uint32_t* state, uint32_t N)
{
    const uint32_t s = state[0];
    return (s != GSX_RANK_NONE && s + 1u < N) ? s + 1u : N;
}

// (1 - alpha), never exactly zero: an exactly opaque splat leaves 2^-30 of the light through
// (forward and backward agree on it, so T_{k+1}/(1 - alpha_k) is always defined)
__device__ __forceinline__ float one_minus(float al)
{
    // = (f == 0 ? 2^-30 : f) in ONE instruction (v_max_u32 on the bit patterns) instead of a compare and a select: +0 is the
    // only pattern below 2^-30's that 1 - al can produce -- the next positive value is 2^-24, half an ulp of 1; negative
    // values (al > 1) and NaN have the sign or exponent bits set and compare above -- and 1 - 1 is +0 in round-to-nearest
    const float f = 1.0f - al;
    return __uint_as_float(max(__float_as_uint(f), 0x30800000u));
}

__device__ __forceinline__ void renorm(float& m, int& e)
{
    e += __builtin_amdgcn_frexp_expf(m);
    m = __builtin_amdgcn_frexp_mantf(m);
}

// 4-bit mask of the tile's 8x8 quadrants (bit k: x half k&1, y half k>>1) in which the splat can reach
// |alpha| >= 2^-GSX_CUT_LOG2.  Computed once per record by the staging lane and kept in the LDS copy of
// the record, so the per-splat loop spends one v_readfirstlane + four scalar bit tests (the CU's single
// scalar ALU was the bottleneck when the box was compared per quadrant inside the loop).
// Two tests: the pixel box (exact integers), then -- for a proper (negative definite) conic -- the
// maximum of the exponent q' over the quadrant's rectangle of pixel centres, which is 0 if the centre
// lies inside and otherwise sits on one of the four edges (concave quadratic): the quadrant is dropped
// when o * 2^q'max < 2^-24 with a safety margin.  16% fewer pair evaluations on the orbit views.
struct QuadEdges { float x[4], y[4]; };   // pixel-centre coordinates of columns/rows 0, 7, 8, 15 of the tile (clamped)

__device__ __forceinline__ float edge_max(float kaa, float kab, float kbb, float fixed, float lo, float hi, float nh_inv_kaa)
{
    // max over t in [lo, hi] of kaa t^2 + kab fixed t + kbb fixed^2, kaa < 0; nh_inv_kaa ~ -0.5 / kaa.  The parabola is evaluated AT
    // the t found, so a t that is an ulp off its vertex costs a second-order nothing and never overstates the maximum: sixteen IEEE
    // divisions per record (ten instructions each) became two reciprocals
    float t = (kab * fixed) * nh_inv_kaa;
    t = fminf(fmaxf(t, lo), hi);
    return kaa * t * t + kab * fixed * t + kbb * fixed * fixed;
}

__device__ __forceinline__ uint32_t quad_mask(const float4& a, const float4& b, float box_x, float box_y, int qx, int qy,
                                              const QuadEdges& E, int cut_log2, int cut_min_exp, bool tiny)
{
    // an image less than a tile wide or high: every pair of the tile's list is evaluated, as the oracle's tiled mode does --
    // its gradient tensors can consist of sub-cut weights altogether (a one-pixel column in every splat's far tail), and
    // then ANY mask is a 100 % relative error (gsx_constants.h, GSX_CUT_LOG2_FOR)
    if (tiny) return 0xFu;
    const uint32_t bx = __float_as_uint(box_x), by = __float_as_uint(box_y);
    const int xlo = (int)(bx & 0xffff) - qx, xhi = (int)(bx >> 16) - qx;
    const int ylo = (int)(by & 0xffff) - qy, yhi = (int)(by >> 16) - qy;
    const uint32_t mx = (uint32_t)(xlo <= 7 && xhi >= 0) | ((uint32_t)(xlo <= 15 && xhi >= 8) << 1);
    const uint32_t my = (uint32_t)(ylo <= 7 && yhi >= 0) | ((uint32_t)(ylo <= 15 && yhi >= 8) << 1);
    uint32_t mask = ((my & 1u) ? mx : 0u) | ((my & 2u) ? (mx << 2) : 0u);     // bit k = mx[k&1] & my[k>>1]
    const float k00 = a.z, kb = a.w, k11 = b.x;
    const bool proper = (k00 < 0.0f) && (k11 < 0.0f) && (4.0f * k00 * k11 - kb * kb > 0.0f);
    if (proper) {
        const int e = max((int)((__float_as_uint(b.y) >> 23) & 0xff) - 127, cut_min_exp);           // |o| < 2^(e+1), floored like the box
        const float thr = -(float)(cut_log2 + 1 + e) - 0.0625f;               // margin for rounding
        const float r00 = -0.5f * __builtin_amdgcn_rcpf(k00), r11 = -0.5f * __builtin_amdgcn_rcpf(k11);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float ulo = E.x[(k & 1) * 2] - a.x, uhi = E.x[(k & 1) * 2 + 1] - a.x;
            const float vlo = E.y[(k >> 1) * 2] - a.y, vhi = E.y[(k >> 1) * 2 + 1] - a.y;
            float qmax = 0.0f;
            if (!(ulo <= 0.0f && uhi >= 0.0f && vlo <= 0.0f && vhi >= 0.0f)) {
                qmax = fmaxf(fmaxf(edge_max(k00, kb, k11, vlo, ulo, uhi, r00), edge_max(k00, kb, k11, vhi, ulo, uhi, r00)),
                             fmaxf(edge_max(k11, kb, k00, ulo, vlo, vhi, r11), edge_max(k11, kb, k00, uhi, vlo, vhi, r11)));
            }
            if (qmax < thr) mask &= ~(1u << k);
        }
    }
    return mask;
}

__device__ __forceinline__ QuadEdges quad_edges(int qx, int qy, int H, int W, float stepx, float stepy);

__device__ __forceinline__ QuadEdges quad_edges(int qx, int qy, int H, int W, float stepx, float stepy)
{
    QuadEdges E;
    const int off[4] = {0, 7, 8, 15};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        E.x[i] = pix_coord(min(qx + off[i], W - 1), W, stepx);
        E.y[i] = pix_coord(min(qy + off[i], H - 1), H, stepy);
    }
    return E;
}

// Workgroup -> chunk.  Consecutive chunks belong to the same tile and cost about the same (a dense tile's
// chunks all run 256 splats x 4 quadrants, a sparse tile's a tenth of that); the dispatcher fills a CU
// with consecutive workgroups, so the identity map left a few CUs with only heavy waves (measured: VALU
// 23% busy chip-wide, kernel time = the heaviest SIMD).  A stride walk (multiplication by a prime,
// bijective mod the chunk count) gives every CU a mix.
__device__ __forceinline__ uint32_t chunk_of_block(uint32_t block, uint32_t total)
{
    const uint32_t stride = (total % 7919u) ? 7919u : 7927u;
    return (uint32_t)(((uint64_t)block * stride) % total);
}

struct ChunkGeom { int tile, tx, ty; uint32_t begin, end; };

// chunk -> (tile, instance range).  chunk_base has n_tiles+2 entries: the exclusive scan of chunks per tile (n_tiles + 1)
// and, last, the chunk size the binning stage chose for this view (64, 128 or 256 instances: chunk_size_for).
__device__ __forceinline__ ChunkGeom chunk_locate(const uint32_t* __restrict__ chunk_base, int n_tiles, uint32_t chunk,
                                                  const uint32_t* __restrict__ ranges, int tw)
{
    const uint32_t ch = chunk_base[n_tiles + 1];
    int lo = 0, hi = n_tiles - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (chunk_base[mid + 1] > chunk) hi = mid; else lo = mid + 1; }
    ChunkGeom g;
    g.tile = lo; g.tx = lo % tw; g.ty = lo / tw;
    g.begin = ranges[2*lo] + (chunk - chunk_base[lo]) * ch;
    g.end = min(g.begin + ch, ranges[2*lo+1]);
    return g;
}

// the same with the tile known (a shallow view's chunk pass leaves it in the chunk's header word, which only a deep view
// publishes into): two round trips instead of the search's twelve at the start of every wave of the backward
__device__ __forceinline__ ChunkGeom chunk_locate_in(const uint32_t* __restrict__ chunk_base, int n_tiles, uint32_t chunk,
                                                     const uint32_t* __restrict__ ranges, int tw, int tile)
{
    const uint32_t ch = chunk_base[n_tiles + 1];
    ChunkGeom g;
    g.tile = tile; g.tx = tile % tw; g.ty = tile / tw;
    g.begin = ranges[2*tile] + (chunk - chunk_base[tile]) * ch;
    g.end = min(g.begin + ch, ranges[2*tile+1]);
    return g;
}

// pixel p of a tile in "quadrant-major" order: k = p >> 6 picks the 8x8 quadrant, lane = p & 63
__device__ __forceinline__ void pix_xy(int k, int lane, int tx, int ty, int& x, int& y)
{
    x = tx * GSX_TILE + (lane & 7) + ((k & 1) << 3);
    y = ty * GSX_TILE + (lane >> 3) + ((k >> 1) << 3);
}

__device__ __forceinline__ float* state_ptr(float* chunk_state, uint32_t chunk, int word, int pix)
{
    return chunk_state + ((size_t)chunk * GSX_STATE_WORDS + word) * kPixPerTile + pix;
}

// ------------------------------------------------------------------------------------------
// Forward occlusion skipping.  The chunk pass composites all chunks of a tile in parallel, blind to the transmittance in
// front of them; in a dense view a third to a half of the instances sit behind a wall of earlier splats (and in a
// config-5-like scene five sixths).  Every chunk therefore PUBLISHES, when it is done, a payload: per pixel, the DARKNESS
// q = floor(-4 log2 |T_local|) of what it lets through, in quarter-bits, rounded towards "brighter" (one byte per pixel:
// 64 words per chunk, lane l holds its own four pixels).  Integer sums of the payloads of a tile's chunks 0..k-1 bound
// -log2 |T| on entry to chunk k, per pixel.  The rule, per 8x8 QUADRANT w of the tile (64 pixels = one wave of the
// combine pass = one lane-quadruple slot of the chunk pass):
//   w is DARK from the first level L on at which every pixel p of w has  sum_{c < L} q_c(p)  >=  4 GSX_CUT_LOG2 + G(tile),
//   G(tile) = the growth bounds of ALL the tile's chunks together: by how much |T| can grow inside them, in quarter-bits
//   (0 unless an opacity lies outside [0, 2]: the header pass below; a chunk that bounds nothing -- NaN/inf transmittance,
//   15 bits and more -- makes its tile UNBOUNDED: nothing of it is ever dark).
// Behind L every weight T * alpha of the quadrant is below 2^-GSX_CUT_LOG2 (the entry bound times whatever the chunks from L
// on can add), the bound under which the footprint cut drops pairs.  The combine pass evaluates the rule from the
// published words -- integers, a function of the data alone -- and treats a dark quadrant of a chunk as the identity
// WHETHER OR NOT it was computed, so the image does not depend on timing.  A chunk may therefore leave out every quadrant
// it can PROVE dark when it starts, and skip itself when all four are.  Payloads of chunks in front of a dark point
// always exist: a quadrant that is not dark is always computed.  Workgroups are dispatched in LEVEL order (every tile's
// chunk 0, then every tile's chunk 1, ...; chunk_order_kernel builds the table), so that by the time the deep chunks of
// a tile start, its shallow ones have finished.  No waiting anywhere: a chunk that cannot tell yet computes.
// (Round 2's first rule tested each chunk against 4 GSX_CUT_LOG2 + its OWN growth and took that growth off the sums of a
// dark chunk: exact, but not monotone -- behind the dark point the sums stand still, every grower drains them, and a
// chunk's proof needed the headers of everything in front of it.  A dozen trainer steps push a few hundred of six million
// opacities below 0; that tripled the chunk pass of a config-5 view.)
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMetaNonMono = 0x80u;               // meta word bit 7 (the low byte's codes are 1..5)
constexpr uint32_t kHdrDone = 0x80000000u, kHdrUnbounded = 0x40000000u, kHdrGrowthMask = 0xFFFFFu;
constexpr uint32_t kDeepChunks = GSX_DEEP_CHUNKS;      // a view is DEEP from this many chunks on (see chunk_order_kernel)
constexpr int kDarkQ = 4 * GSX_CUT_LOG2;               // quarter-bits of darkness at which every weight is below the cut
constexpr int kPayBias = 65;                           // payload byte = q + 65 in [1, 255] (0 = not published): q in [-64, 190]
// tail of the chunk-state buffer behind the states, in words per chunk: meta 1, dark 1, work lists 2 x kWorkClasses (+ 16
// counters), header 1, order table 1, payload 64
__device__ __host__ __forceinline__ size_t tail_hdr_offset(size_t max_chunks) { return 2 * max_chunks + 16 + 2 * (size_t)kWorkClasses * 2 * max_chunks; }
// mid-chunk states of a shallow view: behind header, order table and payloads (gsx_constants.h)
__device__ __host__ __forceinline__ size_t tail_mid_offset(size_t max_chunks) { return tail_hdr_offset(max_chunks) + 66 * max_chunks; }
// quadrant masks of a shallow view's instances, one byte each, in the chunk's payload words (which only a deep view publishes
// into): the chunk pass computes every instance's mask (quad_mask: ~370 instructions per batch of 64) and leaves it here, so
// that the backward reads a byte instead of computing it again
static_assert(GSX_CHUNK <= 64 * 4, "a chunk's masks fit its 64 payload words");
__device__ __host__ __forceinline__ size_t tail_pay_offset(size_t max_chunks) { return tail_hdr_offset(max_chunks) + 2 * max_chunks; }
__device__ __forceinline__ float* mid_ptr(float* mid_state, uint32_t chunk, int word, int pix)
{
    return mid_state + ((size_t)chunk * GSX_MID_WORDS + word) * kPixPerTile + pix;
}

// one splat's contribution to its chunk's growth bound (bits): log2 |1 - o| for an opacity outside [0, 2] (+ slack for the
// rounding of the sum); NaN stays NaN
// a = (mu_x, mu_y, k00, kb), b = (k11, opacity, ...) of the record.  A splat whose exponent is not a proper (negative
// definite, finite) quadratic can reach exp2(q) = inf -- a Gaussian at the camera plane -- and bounds nothing: infinite.
// A function of the RECORD alone, so that a chunk that skips itself and one that computes agree on the header.
__device__ __forceinline__ float growth_term(const float4& a, const float4& b)
{
    const bool proper = (a.z < 0.0f) && (b.x < 0.0f) && (4.0f * a.z * b.x - a.w * a.w > 0.0f) && (fabsf(a.x) <= 3.0e38f) && (fabsf(a.y) <= 3.0e38f);
    if (!proper) return __builtin_inff();
    const float o = b.y;
    return (o >= 0.0f && o <= 2.0f) ? 0.0f : __builtin_amdgcn_logf(fabsf(o - 1.0f)) + 1e-3f;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// header word of a finished chunk from its growth sum (wave total; NaN or 15 bits and more: unbounded)
__device__ __forceinline__ uint32_t header_word(float growth)
{
    if (!(growth < 15.0f)) return kHdrDone | kHdrUnbounded;
    return kHdrDone | (growth == 0.0f ? 0u : (uint32_t)(4.0f * growth) + 2u);
}
// darkness of one pixel's local transmittance (m, e) after renormalisation (|m| in [0.5, 1) or 0), rounded down
__device__ __forceinline__ int darkness_q(float m, int e)
{
    const float am = fabsf(m);
    if (!(am < 1.0f)) return -64;              // not finite: only in a chunk whose header says unbounded
    if (am == 0.0f) return 190;
    const int q = (int)floorf(-4.0f * ((float)e + __builtin_amdgcn_logf(am)) - 0.05f);
    return max(-64, min(190, q));
}

// Entry darkness of chunk k of a tile.  Returns the quadrants PROVEN dark (bits 0..3) | a reason code << 8 (1 = level 0,
// 2 = an unbounded chunk in the tile, 3 = nothing proven and a payload was missing, 4 = nothing proven, 5 = some quadrants
// are dark; kept in the low byte of the chunk's meta word as a diagnostic: tools/skip_stats.py).  Whatever is not proven is
// computed, so a proof only has to be SOUND (never call dark what the rule of the combine pass does not), and everything
// below errs to that side: a word that is read too early, or stale, reads as "not there".
// The rule is monotone -- a quadrant is dark from the first level on at which its payload sums reach the tile's threshold
// -- so the proof is made once per (tile, quadrant): the first dark level is a fact about the data, whoever finds it; it
// is kept in tile_first[w] (kDarkUnknown until found; atomicMin), and a chunk reads four words and is done.
// Finding it means summing the payloads of the chunks in front from level 0 (`walk`); a quadrant whose next payload is
// not there yet is given up for this call, the others go on.  Who walks:
//   * a chunk of the first kWalkAll levels when it STARTS (cheap, and the shallow dark points are found at once);
//   * a chunk when it is DONE, for the entry of its successor (own payload included), at every fourth of the first levels
//     and behind them at levels of the form m * 2^e, m = 4..7: every next walk is at most a quarter deeper than the last,
//     so a dark point at level L is known by level 1.25 L while a tile's walks together read ~5 x its depth in payloads --
//     a quadrant at the rim of the scene never gets dark, and its tile must not walk its whole list again for every chunk.
//     Why at the end: the chunks of a tile that is not dark yet all compute, and once most other tiles skip, the machine
//     dispatches them faster than they finish -- a chunk that starts finds the last dozens of payloads in front of it
//     missing.  A chunk that is done started after its predecessors and took as long: their payloads are there.
// Visibility.  The words read here (levels, "published" bits, payloads) are written by other chunks of the same launch,
// on any XCD, with `sc1` stores or atomics, and read with `sc1` loads (which bypass this CU's L1; a per-lane address where
// the address is wave-uniform, so that the load does not become a scalar one through yet another cache).  The XCDs' L2s
// are not coherent with each other, so a reader may see an old value for a while.  Every such error goes the safe way --
// an unknown level, a bit that is not up yet, a zero payload word all read as "not there": the chunk computes -- so what
// is skipped depends on timing and placement, what the image is does not.  A payload is only loaded once its chunk's
// bit is up, and zeros under a raised bit (the payloads are not line-aligned) are taken for "not there" as well.
constexpr uint32_t kDarkUnknown = 0xFFFFFFFFu, kTileUnbounded = 0x80000000u, kHdrPublished = 0x40000000u, kWalkAll = 64;
__device__ __forceinline__ uint32_t sc1_read(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool walk_level(uint32_t k)         // the levels whose entry is examined by the chunk in front, when it is done
{
    if (k < kWalkAll) return (k & 3u) == 0u;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(k);      // k = m * 2^(e-2) + rest, m = 4..7
    return (k & ((1u << (e - 2u)) - 1u)) == 0u;
}
__device__ __forceinline__ uint32_t entry_walk(const uint32_t* hdr_t, const uint32_t* pay_t, uint32_t* tile_first, uint32_t k, int lane,
                                               uint32_t tile_growth, bool walk)
{
    if (k == 0u) return 1u << 8;
    if (!tile_first) return 4u << 8;                          // (an image with more tiles than the work lists' room holds words for)
    if (tile_growth & kTileUnbounded) return 2u << 8;
    const int thr = kDarkQ + (int)tile_growth;
    uint32_t trk = 0xFu, dark = 0u, missing = 0u;
    const uint32_t head = sc1_read(tile_first + (lane & 3));  // (a per-lane address: a uniform one becomes a scalar load, another cache)
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)head, w);
        if (L != kDarkUnknown) { trk &= ~(1u << w); if (L <= k) dark |= 1u << w; }
    }
    if (trk == 0u || !walk) return dark | ((dark ? 5u : 4u) << 8);
    uint32_t first[4] = { kDarkUnknown, kDarkUnknown, kDarkUnknown, kDarkUnknown };
    int s[4] = { 0, 0, 0, 0 };
#pragma unroll 1
    for (uint32_t j0 = 0; j0 < k && trk != 0u; j0 += 8) {
        // which of the next eight chunks have published (lane u asks for chunk j0 + u)
        uint32_t fl = 0u;
        if (lane < 8 && j0 + (uint32_t)lane < k) fl = sc1_read(hdr_t + j0 + lane);
        const uint32_t there8 = (uint32_t)__builtin_amdgcn_ballot_w64((fl & kHdrPublished) != 0u) & 0xffu;
        uint32_t pw[8];
#pragma unroll
        for (int u = 0; u < 8; u++)            // eight payloads in flight
            pw[u] = ((there8 >> u) & 1u) ? __hip_atomic_load(pay_t + (size_t)(j0 + u) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (j0 + u < k) {
                // a published word is never 0: zeros under a raised bit are a stale line (the payloads are not line-aligned: the
                // neighbour's read brought this one's first or last words in too early)
                const bool there = ((there8 >> u) & 1u) != 0u && __builtin_amdgcn_ballot_w64(pw[u] == 0u) == 0ull;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t bit = 1u << w;
                    if (trk & bit) {
                        if (__builtin_amdgcn_ballot_w64(s[w] < thr) == 0ull) {                // dark from this level on
                            first[w] = j0 + (uint32_t)u; trk &= ~bit; dark |= bit;
                        } else if (there) {
                            s[w] += (int)((pw[u] >> (8 * w)) & 0xffu) - kPayBias;
                        } else {                                                              // not finished yet: give this one up
                            trk &= ~bit; missing |= bit;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < 4; w++)
        if ((trk & (1u << w)) && __builtin_amdgcn_ballot_w64(s[w] < thr) == 0ull) { first[w] = k; dark |= 1u << w; }
    if (lane == 0) {
#pragma unroll
        for (int w = 0; w < 4; w++) if (first[w] != kDarkUnknown) atomicMin(tile_first + w, first[w]);
    }
    return dark | ((dark ? 5u : missing ? 3u : 4u) << 8);
}

// ------------------------------------------------------------------------------------------
// Header pass of a deep view (two small launches in front of the chunk pass): the growth of every tile, in the header
// word of the tile's FIRST chunk (hdr[chunk_base[tile]]: sum of its chunks' bounds in quarter-bits | kTileUnbounded).
//   grower_bitmap_kernel: one bit per RANK -- can this splat make |T| grow (opacity outside [0, 2], or an improper
//     exponent)? -- and their number.  Almost always zero: then the second launch returns at once (the words are zero).
//   tile_growth_kernel: one wave per chunk tests its ranks against the bitmap (L2-resident: N / 8 bytes) and reads the
//     records of the growers only; a chunk's bound (the same terms in the same order as the chunk pass's own sum for the
//     backward's cut) goes into its tile's word.  Integer additions: the total does not depend on their order.  Without a
//     bitmap (no room: see the host side) every record is read.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void grower_bitmap_kernel(const float4* __restrict__ rec, uint32_t N, const uint32_t* __restrict__ chunk_base,
                                                               int n_tiles, uint32_t max_chunks, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ count)
{
    if (min(chunk_base[n_tiles], max_chunks) < kDeepChunks) return;
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;      // N is rounded up to whole waves: every lane reaches the ballot
    bool grower = false;
    if (r < N) { const float4* R = rec + (size_t)kRecVec * r; grower = growth_term(R[0], R[1]) != 0.0f; }      // NaN counts
    const unsigned long long m = __builtin_amdgcn_ballot_w64(grower);
    if ((threadIdx.x & 63) == 0 && r < N) {                    // (a wave wholly past the end has no words: the count sits there)
        bitmap[(r >> 5)] = (uint32_t)m; bitmap[(r >> 5) + 1] = (uint32_t)(m >> 32);
        if (m) atomicAdd(count, (uint32_t)__popcll(m));
    }
}

constexpr int kHeaderBlocks = 4096;
__global__ __launch_bounds__(kBlock) void tile_growth_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
                                                             const uint32_t* __restrict__ chunk_base, int n_tiles, int tw, uint32_t max_chunks,
                                                             const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ count, uint32_t* __restrict__ hdr)
{
    const uint32_t n_chunks = min(chunk_base[n_tiles], max_chunks);
    if (n_chunks < kDeepChunks) return;
    if (bitmap && *count == 0u) return;                       // no grower in the view: every tile's word stays 0
    const int lane = threadIdx.x & 63;
    for (uint32_t chunk = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); chunk < n_chunks; chunk += gridDim.x * (kBlock / 64)) {
        const ChunkGeom g = chunk_locate(chunk_base, n_tiles, chunk, ranges, tw);
        float pre = 0.0f;
        bool any = false;
        for (uint32_t i = g.begin + lane; i < g.end; i += kCBatch) {
            const uint32_t r = ranks[i];
            if (!bitmap || ((bitmap[r >> 5] >> (r & 31u)) & 1u)) {
                const float4* R = rec + (size_t)kRecVec * r;
                pre += growth_term(R[0], R[1]);
                any = true;
            }
        }
        if (!__builtin_amdgcn_ballot_w64(any)) continue;
        const uint32_t hw = header_word(wave_sum(pre));
        if (lane == 0) {
            uint32_t* word = hdr + chunk_base[g.tile];
            if (hw & kHdrUnbounded) atomicOr(word, kTileUnbounded);
            else if (hw & kHdrGrowthMask) atomicAdd(word, hw & kHdrGrowthMask);
        }
    }
}

// Level-order work table: order_tab[i] = chunk id, all tiles' chunk 0 first (tile order), then all chunk 1s, ...  Workgroup l
// writes the levels l, l + kOrderLevels, ...: a level's offset is sum_t min(n_t, l), a tile's slot its rank among the tiles
// that reach the level.  Also clears the published headers and payloads.
constexpr int kOrderThreads = 1024, kOrderLevels = 256;
// Skipping pays where a tile's list is many rounds of the machine deep: then the chunks that decide a tile's dark point have
// long finished when its deep chunks start.  In a shallow view (config 3: <= 10 k chunks on ~5 k wave slots, dark points
// at level 15-20, i.e. exactly one round deep) the deep chunks start while the shallow ones still run, nothing can be
// proven in time, and reading the predecessors' payloads only costs (measured: +25 % on the chunk pass, no chunk skipped).
// A view is DEEP from this many chunks on (a function of the data, like everything the rule depends on); below it nothing is
// published, nothing skipped, and the combine pass ignores the rule.
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* s_w /*[16]*/, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kOrderThreads / 64; w++) { const uint32_t x = s_w[w]; if (w < wave) base += x; tot += x; }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(kOrderThreads) void chunk_order_kernel(const uint32_t* __restrict__ chunk_base, int n_tiles, uint32_t max_chunks,
                                                                    uint32_t* __restrict__ hdr, uint32_t* __restrict__ pay,
                                                                    uint32_t* __restrict__ order_tab, uint32_t* __restrict__ tile_dark)
{
    __shared__ uint32_t s_w[kOrderThreads / 64];
    const uint32_t n_chunks = min(chunk_base[n_tiles], max_chunks);
    if (n_chunks < kDeepChunks) return;                       // a shallow view: no table, and every word is written before it is read
    // "not published": payloads of this view's chunks (not of the buffers' capacity); header words (the tiles' growth) and
    // first dark levels start from nothing
    for (size_t i = (size_t)blockIdx.x * kOrderThreads + threadIdx.x; i < (size_t)n_chunks; i += (size_t)gridDim.x * kOrderThreads) hdr[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * kOrderThreads + threadIdx.x; i < (size_t)n_chunks * 64; i += (size_t)gridDim.x * kOrderThreads) pay[i] = 0u;
    if (tile_dark)
        for (size_t i = (size_t)blockIdx.x * kOrderThreads + threadIdx.x; i < (size_t)n_tiles * 4; i += (size_t)gridDim.x * kOrderThreads) tile_dark[i] = kDarkUnknown;
    // workgroup b writes the levels b, b + gridDim.x, ...: level order all the way down.  (First version: the last
    // workgroup wrote everything from level kOrderLevels - 1 on tile by tile -- the deep end of every tile was dispatched
    // in one burst, a thousand chunks of one tile in flight together, none of which could know anything about the other.)
    for (uint32_t l = blockIdx.x; ; l += gridDim.x) {
        uint32_t below = 0;                                   // chunks of the levels before l
        for (int t0 = 0; t0 < n_tiles; t0 += kOrderThreads) {
            const int t = t0 + (int)threadIdx.x;
            if (t < n_tiles) below += min(chunk_base[t + 1] - chunk_base[t], l);
        }
        uint32_t level_off;
        block_excl_scan_1024(below, s_w, &level_off);
        if (level_off >= n_chunks) return;                    // no tile is this deep (uniform)
        uint32_t carry = level_off;
        for (int t0 = 0; t0 < n_tiles; t0 += kOrderThreads) {
            const int t = t0 + (int)threadIdx.x;
            uint32_t cb = 0, cnt = 0;
            if (t < n_tiles) { cb = chunk_base[t]; cnt = chunk_base[t + 1] - cb > l ? 1u : 0u; }
            uint32_t tot;
            const uint32_t pos = carry + block_excl_scan_1024(cnt, s_w, &tot);
            if (cnt && pos < max_chunks) order_tab[pos] = cb + l;
            carry += tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// F1: one wave per chunk, local compositing from (C = 0, T = 1)
// ------------------------------------------------------------------------------------------
// DEEP: the instantiation for deep views (level order, forward occlusion skipping); the other one never looks at a header.
// Which of the two a view needs is known on the device only: both are launched when the buffers could hold a deep view,
// and the one that does not apply returns at once.  (One kernel with the deep prologue behind a branch, round 2's first
// version, tied the shallow path's register allocation to the prologue's shape: 92 VGPRs became 117-120 when the walk
// gained its shortcuts.)
template <bool PASS1, bool DEPTH, bool DEEP>
__global__ __launch_bounds__(64) void chunk_fwd_kernel(
    const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
    const uint32_t* __restrict__ chunk_base, int n_tiles, uint32_t N, int H, int W, int tw, float stepx, float stepy,
    uint32_t* __restrict__ state, float* __restrict__ chunk_state, uint32_t* __restrict__ chunk_meta, uint32_t* __restrict__ work,
    uint32_t* __restrict__ hdr, uint32_t* __restrict__ pay, const uint32_t* __restrict__ order_tab, uint32_t* __restrict__ tile_dark)
{
    __shared__ float4 s_a[kCBatch], s_b[kCBatch], s_c[kCBatch];
    // (state[] and the work counters are cleared by block 0 of chunk_fwd_wg_kernel, which gsx_composite_fwd always launches in front)
    const uint32_t n_chunks = chunk_base[n_tiles];
    if (!PASS1 && DEEP != (n_chunks >= kDeepChunks)) return;          // the other instantiation's view
    uint32_t limit = N;
    if (PASS1) { limit = rank_limit_from_state(state, N); if (limit >= N) return; }
    const int lane = threadIdx.x;
    // pass 0: one chunk per workgroup (gridDim.x >= n_chunks).  Pass 1 only has work after a global early exit, so it is
    // launched with a small fixed grid -- an unsaturated view pays for a few hundred early returns, not for max_chunks of
    // them -- and walks the chunks with a grid stride when it does run.
    for (uint32_t blk = blockIdx.x; blk < n_chunks; blk += gridDim.x) {
    // pass 0 of a deep view walks the chunks in level order (the table of chunk_order_kernel); otherwise the stride walk
    constexpr bool deep = DEEP;
    const uint32_t chunk = deep ? order_tab[blk] : chunk_of_block(blk, n_chunks);
    const ChunkGeom g = chunk_locate(chunk_base, n_tiles, chunk, ranges, tw);
    // quadrants proven dark on entry: left out (the combine pass ignores them anyway)
    uint32_t dark_quads = 0, why = 0;
    if (deep) {
        const uint32_t cb = chunk_base[g.tile];
        const uint32_t pre = entry_walk(hdr + cb, pay + (size_t)cb * 64, tile_dark ? tile_dark + 4 * (size_t)g.tile : nullptr, chunk - cb, lane,
                                        hdr[cb] & ~kHdrPublished, chunk - cb < kWalkAll);
        dark_quads = pre & 0xFu; why = pre >> 8;
        if (dark_quads == 0xFu) {                             // every quadrant dark on entry: nothing to do
            if (lane == 0) chunk_meta[chunk] = 0u;
            return;
        }
    }
    // (wave-uniform by construction -- ballots -- but the compiler cannot see that: without the scalar copy the per-splat
    // quadrant tests become exec-mask branches instead of scalar bit tests, +25 % on this kernel)
    dark_quads = __builtin_amdgcn_readfirstlane(dark_quads);
    why = __builtin_amdgcn_readfirstlane(why);
    const int x0 = g.tx * GSX_TILE + (lane & 7), y0 = g.ty * GSX_TILE + (lane >> 3);
    float gx[2], gy[2];
    gx[0] = pix_coord(min(x0, W - 1), W, stepx); gx[1] = pix_coord(min(x0 + 8, W - 1), W, stepx);
    gy[0] = pix_coord(min(y0, H - 1), H, stepy); gy[1] = pix_coord(min(y0 + 8, H - 1), H, stepy);
    const int qx = g.tx * GSX_TILE, qy = g.ty * GSX_TILE;
    const QuadEdges QE = quad_edges(qx, qy, H, W, stepx, stepy);
    float C0[4], C1[4], C2[4], Dm[4], m[4]; int e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { C0[k] = C1[k] = C2[k] = Dm[k] = 0.f; m[k] = 1.f; e[k] = 0; }
    bool stop = false;
    uint32_t active = 0;
    // Bound on how much |T| can GROW inside the chunk, in bits: alpha = o * G with G in (0, 1], so |1 - alpha| <= 1 for
    // o in [0, 2], <= 1 - o for o < 0 and <= o - 1 for o > 2; the sum of log2 of those factors over the chunk's splats
    // bounds log2(|T_k| / |T_start|) for every k.  Zero for ordinary opacities.
    float growth = 0.0f;
    // can a splat of this chunk take a pixel's accumulated alpha DOWN (alpha outside [0, 1]: an opacity outside [0, 1], or an
    // improper exponent)?  Bit 7 of the meta word; the combine pass needs it for the early-exit gate (kMetaNonMono)
    bool nonmono = false;
    uint32_t cost = 0;                        // quadrant evaluations of this chunk (the backward's cost estimate)
    for (uint32_t bs = g.begin; bs < g.end && !stop; bs += kCBatch) {
        const uint32_t i = bs + lane;
        uint32_t my_mask = 0u;
        bool past_exit = false;
        if (i < g.end) {
            const uint32_t r = ranks[i];
            const float4* R = rec + (size_t)kRecVec * r;
            const float4 a = R[0], b = R[1];
            const float gt = growth_term(a, b);
            growth += gt;
            nonmono = nonmono || (gt != 0.0f) || !(b.y <= 1.0f);      // (gt: o < 0, o > 2, improper, NaN; o in (1, 2] grows nothing but overshoots)
            const float4 c = R[2];
            my_mask = quad_mask(a, b, c.z, c.w, qx, qy, QE, GSX_CUT_LOG2_FOR(H, W), GSX_CUT_MIN_EXP_FOR(H, W), H < GSX_TILE || W < GSX_TILE) & ~dark_quads;
            if (PASS1) { past_exit = r >= limit; if (past_exit) my_mask = 0u; }      // ranks ascend: everything from the exit rank on is left out
            s_a[lane] = a; s_b[lane] = b; s_c[lane] = c;
        }
        // the batch's quadrant masks as four wave-uniform bit sets in scalar registers (see chunk_bwd_kernel): no mask read
        // from LDS per instance, and instances that reach no live quadrant (inside the tile's box but outside the ellipse,
        // or behind the wall) are never visited
        const unsigned long long q0 = __builtin_amdgcn_ballot_w64((my_mask & 1u) != 0u), q1 = __builtin_amdgcn_ballot_w64((my_mask & 2u) != 0u);
        const unsigned long long q2 = __builtin_amdgcn_ballot_w64((my_mask & 4u) != 0u), q3 = __builtin_amdgcn_ballot_w64((my_mask & 8u) != 0u);
        if (PASS1) { if (__builtin_amdgcn_ballot_w64(past_exit) != 0ull) stop = true; }
        __syncthreads();
        for (unsigned long long todo = (q0 | q1) | (q2 | q3); todo != 0ull; ) {
            const uint32_t j = (uint32_t)__builtin_ctzll(todo);           // front to back
            todo &= todo - 1ull;
            const float4 a = s_a[j], b = s_b[j], c = s_c[j];
            const unsigned long long bit = 1ull << j;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((k == 0 ? q0 : k == 1 ? q1 : k == 2 ? q2 : q3) & bit) {      // one scalar test per quadrant
                    const float dx = gx[k & 1] - a.x, dy = gy[k >> 1] - a.y;
                    const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
                    const float al = b.y * __builtin_amdgcn_exp2f(q);
                    const float w = ldexpf(m[k], e[k]) * al;
                    C0[k] += w * b.z; C1[k] += w * b.w; C2[k] += w * c.x;
                    if (DEPTH) Dm[k] += w * c.y;
                    m[k] *= one_minus(al);
                }
            }
            if ((++active & 3u) == 0u) {                       // every 4 composited splats: |m| stays within 2^+-120
#pragma unroll
                for (int k = 0; k < 4; k++) renorm(m[k], e[k]);
            }
        }
        cost += (uint32_t)(__popcll(q0) + __popcll(q1)) + (uint32_t)(__popcll(q2) + __popcll(q3));
        __syncthreads();
    }
    const bool publish = n_chunks >= kDeepChunks;            // (pass 1 of a deep view republishes what it recomputed)
    uint32_t payload = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        renorm(m[k], e[k]);
        // (a pixel outside the image shows nothing: it must not keep its quadrant from getting dark)
        if (publish) payload |= (uint32_t)(((x0 + ((k & 1) << 3) < W && y0 + ((k >> 1) << 3) < H) ? darkness_q(m[k], e[k]) : 190) + kPayBias) << (8 * k);
        const int pix = k * 64 + lane;
        *state_ptr(chunk_state, chunk, 0, pix) = C0[k];
        *state_ptr(chunk_state, chunk, 1, pix) = C1[k];
        *state_ptr(chunk_state, chunk, 2, pix) = C2[k];
        *state_ptr(chunk_state, chunk, 3, pix) = m[k];
        *state_ptr(chunk_state, chunk, 4, pix) = __int_as_float(e[k]);
        *state_ptr(chunk_state, chunk, 5, pix) = Dm[k];
    }
    growth = wave_sum(growth);
    const bool any_nonmono = __builtin_amdgcn_ballot_w64(nonmono) != 0ull;
    // whole bits, rounded up, + 1 of slack for the rounding of the sum; GSX_META_GROWTH_MAX = "unbounded" (no cut)
    const uint32_t gbits = (growth == 0.0f) ? 0u : (growth < (float)(GSX_META_GROWTH_MAX - 2)) ? (uint32_t)growth + 2u : (uint32_t)GSX_META_GROWTH_MAX;
    // publish the payload (every lane its own word; zero = not there yet)
    if (publish) __hip_atomic_store(pay + (size_t)chunk * 64 + lane, payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
        chunk_meta[chunk] = (gbits << GSX_META_GROWTH_SHIFT) | (cost << GSX_META_COST_SHIFT) | why | (any_nonmono ? kMetaNonMono : 0u);
    }
    if (DEEP) {
        // every lane's word is on its way before the chunk's bit goes up; then, done: look for the tile's dark points on
        // behalf of the chunks behind (entry_walk)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_or(hdr + chunk, kHdrPublished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t cb = chunk_base[g.tile], k1 = chunk - cb + 1u;
        if (k1 < chunk_base[g.tile + 1] - cb && walk_level(k1))
            entry_walk(hdr + cb, pay + (size_t)cb * 64, tile_dark ? tile_dark + 4 * (size_t)g.tile : nullptr, k1, lane, hdr[cb] & ~kHdrPublished, true);
    }
    if (!PASS1) break;                        // compile-time: pass 0 has no back edge
    __syncthreads();
    }
}

constexpr int kWorkNonEmpty = 5;          // work[5]: number of non-empty tiles (written by the last list block of the chunk pass)
// The exit search's signal words (saturation_rank_kernel): one 64-bit word {round, rank} per copy, the copies kSigStride words apart
// in pass 1's work lists (which nothing reads before the search is over); a view too small for that has the one copy in work[14..15].
// The combine pass in front of the search zeroes them.  work[13]: the search's barrier arrivals.
constexpr int kWorkArrivals = 13;
constexpr uint32_t kSigCopies = 256u, kSigStride = 32u;
__device__ __forceinline__ uint32_t sig_copies(uint32_t max_chunks)
{
    return max_chunks >= 1024u ? kSigCopies : min(kSigCopies, 2u * (uint32_t)kWorkClasses * max_chunks / kSigStride);
}
__device__ __forceinline__ unsigned long long* sig_word(uint32_t* work, uint32_t max_chunks, uint32_t c)
{
    return sig_copies(max_chunks) ? reinterpret_cast<unsigned long long*>(work + 16 + (size_t)2 * kWorkClasses * max_chunks + (size_t)c * kSigStride)
                                  : reinterpret_cast<unsigned long long*>(work + 14);
}
constexpr int kWorkBig = 6;               // work[6]: how many of them have more than kParSmall chunks (they come first in the list)
constexpr int kListTiles = 16;            // tiles per list block
constexpr int kParSmall = 12;             // = ParShape::U: a tile of up to this many chunks is one GROUP's work in the combine pass

// The tiles that hold something, as a list (order_tab's room: a shallow view has no level-order table), so that the combine pass
// runs on those alone: the bench scene leaves 3600-3900 of its 4096 tiles empty, and a grid with one workgroup per
// tile spent most of its time passing empty workgroups through the dispatcher in front of the 250-400 that have work.
// The empty tiles' pixels are written here (zeros; alpha 0 is "not saturated": an empty tile vetoes the global exit).
// Runs as the LAST blocks of the chunk pass's grid (a launch of its own until late in round 3: 6.4 us of launch floor), beside the
// pass's tail.  No atomics: a block counts the non-empty tiles in front of its own from the chunk table (n_tiles words out of the
// L2), so the list is in tile order and the same from run to run; the block that holds the last tile writes the count.
__device__ __forceinline__ void tile_list_block(int lb, uint32_t* s_scratch /* >= 13 words */,
    const uint32_t* __restrict__ chunk_base, int H, int W, int tw, int n_tiles,
    uint32_t* __restrict__ work, uint32_t* __restrict__ list, float* __restrict__ rgb, float* __restrict__ alpha, float* __restrict__ depth_map)
{
    // Two parts: first the tiles of more than kParSmall chunks (the combine pass gives each a workgroup), then the small ones (four
    // to a workgroup, one per group of 256 threads).  A block needs the number of big tiles in front of its own, of small ones in
    // front of its own, and of big ones altogether.
    const int t0 = lb * kListTiles;
    uint32_t big_before = 0, small_before = 0, big_all = 0;
    for (int t = (int)threadIdx.x; t < n_tiles; t += kBlock) {
        const uint32_t nc = chunk_base[t + 1] - chunk_base[t];
        const uint32_t big = nc > (uint32_t)kParSmall ? 1u : 0u;
        big_all += big;
        if (t < t0) { big_before += big; small_before += (nc != 0u && !big) ? 1u : 0u; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        big_before += (uint32_t)__shfl_down((int)big_before, d, 64);
        small_before += (uint32_t)__shfl_down((int)small_before, d, 64);
        big_all += (uint32_t)__shfl_down((int)big_all, d, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        s_scratch[threadIdx.x >> 6] = big_before; s_scratch[4 + (threadIdx.x >> 6)] = small_before; s_scratch[8 + (threadIdx.x >> 6)] = big_all;
    }
    __syncthreads();
    const uint32_t n_big = (s_scratch[8] + s_scratch[9]) + (s_scratch[10] + s_scratch[11]);
    const uint32_t base_big = (s_scratch[0] + s_scratch[1]) + (s_scratch[2] + s_scratch[3]);
    const uint32_t base_small = n_big + ((s_scratch[4] + s_scratch[5]) + (s_scratch[6] + s_scratch[7]));
    if (threadIdx.x < 64u) {
        const int t = t0 + (int)threadIdx.x;
        const bool valid = threadIdx.x < (uint32_t)kListTiles && t < n_tiles;
        const uint32_t nc = valid ? chunk_base[t + 1] - chunk_base[t] : 0u;
        const bool big = nc > (uint32_t)kParSmall, small = nc != 0u && !big;
        const unsigned long long mb = __builtin_amdgcn_ballot_w64(big), ms = __builtin_amdgcn_ballot_w64(small);
        const unsigned long long me = __builtin_amdgcn_ballot_w64(valid && nc == 0u), lt = (1ull << threadIdx.x) - 1ull;
        if (big) list[base_big + (uint32_t)__popcll(mb & lt)] = (uint32_t)t;
        if (small) list[base_small + (uint32_t)__popcll(ms & lt)] = (uint32_t)t;
        if (threadIdx.x == 0u) {
            s_scratch[12] = (uint32_t)me;
            if (t0 + kListTiles >= n_tiles) { work[kWorkNonEmpty] = base_small + (uint32_t)__popcll(ms); work[kWorkBig] = n_big; }
        }
    }
    __syncthreads();
    uint32_t em = s_scratch[12];      // (the veto of the empty tiles: tile_combine_par_kernel, from the count)
    const size_t P = (size_t)H * W;
    while (em) {
        const int tile = t0 + __builtin_ctz(em);
        em &= em - 1u;
        int x, y; pix_xy((int)(threadIdx.x >> 6), (int)(threadIdx.x & 63u), tile % tw, tile / tw, x, y);
        if (x < W && y < H) {
            const size_t p = (size_t)y * W + x;
            rgb[p] = 0.f; rgb[P + p] = 0.f; rgb[2*P + p] = 0.f;
            alpha[p] = 0.f;
            if (depth_map) depth_map[p] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// F1w (round 3): the chunk pass of a SHALLOW view -- one WORKGROUP of four waves per chunk.  Wave w composites the chunk's
// w-th quarter (at most 64 instances: one staged batch, no batch loop) from (C = 0, T = 1); the four partial results are
// merged IN ORDER through LDS (compositing is associative), wave w merging the tile's quadrant w, and one state per chunk
// is written as before -- so the chunk stays the unit of state and of the combine pass, while the unit of LATENCY is a
// quarter of it.  Why: with one wave per chunk the kernel took about as long as two of its longest work items (a full
// 256-instance chunk is ~17 k wave-instructions: ~80 us on a SIMD it shares with four others), the machine spent the
// second half of the launch draining (3.3 of 5 wave slots resident on average, SQ counters of round 2), and the VALU ran
// at the two-waves-per-SIMD rate (profiles/r3_valu_issue_rate.txt).  Four waves of a workgroup sit on the four SIMDs of
// one CU: every SIMD gets an equal share of every chunk.
// The merge also leaves the local result of the chunk's SECOND HALF (quarters 2 and 3) in mid_state: the backward starts
// the first half from it on a wave of its own (chunk_bwd_kernel).
// Not bit-identical to the one-wave kernel (the products are associated differently); deterministic like it.
// ------------------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(kBlock) void chunk_fwd_wg_kernel(
    const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
    const uint32_t* __restrict__ chunk_base, int n_tiles, int H, int W, int tw, float stepx, float stepy,
    uint32_t* __restrict__ state, float* __restrict__ chunk_state, uint32_t* __restrict__ chunk_meta, uint32_t* __restrict__ work,
    float* __restrict__ mid_state, uint32_t n_list_blocks, uint32_t* __restrict__ list, float* __restrict__ rgb, float* __restrict__ alpha,
    float* __restrict__ depth_map, uint8_t* __restrict__ masks, uint32_t* __restrict__ chunk_tile)
{
    // C0 C1 C2 m e (Dm).  Without a depth map the parts are 20 KB instead of 24: seven workgroups per CU instead of six, i.e.
    // seven waves per SIMD (65 VGPRs), and the chunk pass of the bench scene 195 -> 189 us with the combine pass
    constexpr int kWords = DEPTH ? 6 : 5;
    // [wave][word][pixel of the tile]; the head of a wave's part doubles as its record staging (3 x 64 float4 = 3 KB of 6 KB),
    // which is dead when the wave parks its result there
    __shared__ float s_part[4][kWords][kPixPerTile];
    __shared__ uint32_t s_cost[4], s_nonmono[4];
    __shared__ float s_growth[4];
    if (blockIdx.x == 0) {                    // the combine pass (next launch) counts in these; the chunk pass itself reads neither
        if (threadIdx.x < 4) state[threadIdx.x] = 0u;
        if (threadIdx.x < 16 && threadIdx.x != (uint32_t)kWorkNonEmpty && threadIdx.x != (uint32_t)kWorkBig) work[threadIdx.x] = 0u;   // (work[5], [6]: the last list block's)
    }
    const uint32_t n_chunks = chunk_base[n_tiles];
    if (n_chunks >= kDeepChunks) return;                                // a deep view is the other kernel's (uniform over the workgroup)
    const uint32_t n_chunk_blocks = gridDim.x - n_list_blocks;
    if (blockIdx.x >= n_chunk_blocks) {                                 // the grid's last blocks: list of non-empty tiles, empty tiles' pixels
        tile_list_block((int)(blockIdx.x - n_chunk_blocks), reinterpret_cast<uint32_t*>(&s_part[0][0][0]), chunk_base, H, W, tw, n_tiles,
                        work, list, rgb, alpha, depth_map);
        return;
    }
    if (blockIdx.x >= n_chunks) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t chunk = chunk_of_block(blockIdx.x, n_chunks);
    const ChunkGeom g = chunk_locate(chunk_base, n_tiles, chunk, ranges, tw);
    const uint32_t sub = chunk_base[n_tiles + 1] >> 2;                  // 64, 32 or 16 instances per wave
    const uint32_t my_begin = min(g.begin + (uint32_t)wave * sub, g.end), my_end = min(my_begin + sub, g.end);
    float4* s_a = reinterpret_cast<float4*>(&s_part[wave][0][0]);
    float4* s_b = s_a + kCBatch;
    float4* s_c = s_b + kCBatch;
    const int x0 = g.tx * GSX_TILE + (lane & 7), y0 = g.ty * GSX_TILE + (lane >> 3);
    float gx[2], gy[2];
    gx[0] = pix_coord(min(x0, W - 1), W, stepx); gx[1] = pix_coord(min(x0 + 8, W - 1), W, stepx);
    gy[0] = pix_coord(min(y0, H - 1), H, stepy); gy[1] = pix_coord(min(y0 + 8, H - 1), H, stepy);
    const int qx = g.tx * GSX_TILE, qy = g.ty * GSX_TILE;
    float C0[4], C1[4], C2[4], Dm[4], m[4]; int e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { C0[k] = C1[k] = C2[k] = Dm[k] = 0.f; m[k] = 1.f; e[k] = 0; }
    float growth = 0.0f;
    bool nonmono = false;
    uint32_t cost = 0;
    if (my_begin < my_end) {                  // (wave-uniform)
        const QuadEdges QE = quad_edges(qx, qy, H, W, stepx, stepy);
        const uint32_t i = my_begin + lane;
        uint32_t my_mask = 0u;
        if (i < my_end) {
            const uint32_t r = ranks[i];
            const float4* R = rec + (size_t)kRecVec * r;
            const float4 a = R[0], b = R[1];
            const float gt = growth_term(a, b);
            growth = gt;
            nonmono = (gt != 0.0f) || !(b.y <= 1.0f);
            const float4 c = R[2];
            my_mask = quad_mask(a, b, c.z, c.w, qx, qy, QE, GSX_CUT_LOG2_FOR(H, W), GSX_CUT_MIN_EXP_FOR(H, W), H < GSX_TILE || W < GSX_TILE);
            s_a[lane] = a; s_b[lane] = b; s_c[lane] = c;
            if (masks) masks[(size_t)chunk * (64 * 4) + (i - g.begin)] = (uint8_t)my_mask;          // for the backward
        }
        const unsigned long long q0 = __builtin_amdgcn_ballot_w64((my_mask & 1u) != 0u), q1 = __builtin_amdgcn_ballot_w64((my_mask & 2u) != 0u);
        const unsigned long long q2 = __builtin_amdgcn_ballot_w64((my_mask & 4u) != 0u), q3 = __builtin_amdgcn_ballot_w64((my_mask & 8u) != 0u);
        // the staging area is this wave's own: its LDS instructions execute in order, no workgroup barrier needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t active = 0;
        for (unsigned long long todo = (q0 | q1) | (q2 | q3); todo != 0ull; ) {
            const uint32_t j = (uint32_t)__builtin_ctzll(todo);           // front to back
            todo &= todo - 1ull;
            const float4 a = s_a[j], b = s_b[j], c = s_c[j];
            const unsigned long long bit = 1ull << j;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((k == 0 ? q0 : k == 1 ? q1 : k == 2 ? q2 : q3) & bit) {      // one scalar test per quadrant
                    const float dx = gx[k & 1] - a.x, dy = gy[k >> 1] - a.y;
                    const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
                    const float al = b.y * __builtin_amdgcn_exp2f(q);
                    const float w = ldexpf(m[k], e[k]) * al;
                    C0[k] += w * b.z; C1[k] += w * b.w; C2[k] += w * c.x;
                    if (DEPTH) Dm[k] += w * c.y;
                    m[k] *= one_minus(al);
                }
            }
            if ((++active & 3u) == 0u) {                       // every 4 composited splats: |m| stays within 2^+-120
#pragma unroll
                for (int k = 0; k < 4; k++) renorm(m[k], e[k]);
            }
        }
        cost = (uint32_t)(__popcll(q0) + __popcll(q1)) + (uint32_t)(__popcll(q2) + __popcll(q3));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();       // every lane has read its last record before the result goes over the staging area
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        renorm(m[k], e[k]);
        const int pix = k * 64 + lane;
        s_part[wave][0][pix] = C0[k]; s_part[wave][1][pix] = C1[k]; s_part[wave][2][pix] = C2[k];
        s_part[wave][3][pix] = m[k]; s_part[wave][4][pix] = __int_as_float(e[k]);
        if (DEPTH) s_part[wave][5][pix] = Dm[k];
    }
    growth = wave_sum(growth);
    const bool any_nonmono = __builtin_amdgcn_ballot_w64(nonmono) != 0ull;
    if (lane == 0) { s_cost[wave] = cost; s_growth[wave] = growth; s_nonmono[wave] = any_nonmono ? 1u : 0u; }
    __syncthreads();
    // merge: wave w takes quadrant w (pixel = threadIdx.x), the four quarters in list order
    {
        const int pix = threadIdx.x;
        float A0 = 0.f, A1 = 0.f, A2 = 0.f, AD = 0.f, am = 1.f; int ae = 0;        // the whole chunk
        float B0 = 0.f, B1 = 0.f, B2 = 0.f, bm = 1.f; int be = 0;                  // its second half (quarters 2, 3)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float c0 = s_part[j][0][pix], c1 = s_part[j][1][pix], c2 = s_part[j][2][pix];
            const float mj = s_part[j][3][pix]; const int ej = __float_as_int(s_part[j][4][pix]);
            const float Tv = ldexpf(am, ae);
            A0 += Tv * c0; A1 += Tv * c1; A2 += Tv * c2;
            if (DEPTH) AD += Tv * s_part[j][5][pix];
            am *= mj; ae += ej; renorm(am, ae);
            if (j >= 2) {
                const float Tb = ldexpf(bm, be);
                B0 += Tb * c0; B1 += Tb * c1; B2 += Tb * c2;
                bm *= mj; be += ej; renorm(bm, be);
            }
        }
        *state_ptr(chunk_state, chunk, 0, pix) = A0; *state_ptr(chunk_state, chunk, 1, pix) = A1; *state_ptr(chunk_state, chunk, 2, pix) = A2;
        *state_ptr(chunk_state, chunk, 3, pix) = am; *state_ptr(chunk_state, chunk, 4, pix) = __int_as_float(ae);
        *state_ptr(chunk_state, chunk, 5, pix) = AD;
        if (mid_state) {
            *mid_ptr(mid_state, chunk, 0, pix) = B0; *mid_ptr(mid_state, chunk, 1, pix) = B1; *mid_ptr(mid_state, chunk, 2, pix) = B2;
            *mid_ptr(mid_state, chunk, 3, pix) = bm; *mid_ptr(mid_state, chunk, 4, pix) = __int_as_float(be);
        }
    }
    if (threadIdx.x == 0) {
        const float gsum = (s_growth[0] + s_growth[1]) + (s_growth[2] + s_growth[3]);
        const uint32_t csum = (s_cost[0] + s_cost[1]) + (s_cost[2] + s_cost[3]);
        const uint32_t nm = (s_nonmono[0] | s_nonmono[1]) | (s_nonmono[2] | s_nonmono[3]);
        // whole bits, rounded up, + 1 of slack for the rounding of the sum; GSX_META_GROWTH_MAX = "unbounded" (no cut)
        const uint32_t gbits = (gsum == 0.0f) ? 0u : (gsum < (float)(GSX_META_GROWTH_MAX - 2)) ? (uint32_t)gsum + 2u : (uint32_t)GSX_META_GROWTH_MAX;
        chunk_meta[chunk] = (gbits << GSX_META_GROWTH_SHIFT) | (csum << GSX_META_COST_SHIFT) | (nm ? kMetaNonMono : 0u);
        if (chunk_tile) chunk_tile[chunk] = (uint32_t)g.tile;             // for the backward (chunk_locate_in)
    }
}

// ------------------------------------------------------------------------------------------
// F2: one workgroup per tile, thread = pixel.  Forward scan over the tile's chunks (prefix
// transmittance, image), then -- for training -- a backward scan that replaces each chunk's
// local colour by the colour BEHIND it (per unit transmittance at the chunk's end).
// state[1] is raised if some pixel is clearly not saturated (the global early exit cannot fire).
// ------------------------------------------------------------------------------------------
// DEEP: as for the chunk pass.  Instantiated with it only since round 3: the deep views' pass 0 and every view's pass 1
// (after a global early exit); a shallow view's pass 0 is tile_combine_par_kernel below, on the list the chunk pass's last blocks leave.
template <bool PASS1, bool DEPTH, bool DEEP>
__global__ __launch_bounds__(kBlock) void tile_combine_kernel(
    const uint32_t* __restrict__ chunk_base, uint32_t N, int H, int W, int tw, int need_bwd,
    uint32_t* __restrict__ state, float* __restrict__ chunk_state, const uint32_t* __restrict__ chunk_meta,
    uint32_t* __restrict__ chunk_dark, uint32_t* __restrict__ work, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ pay,
    uint32_t max_chunks, int n_tiles,
    float* __restrict__ rgb, float* __restrict__ alpha, float* __restrict__ depth_map, const uint32_t* __restrict__ ranges, int split_halves)
{
    if (PASS1) { if (rank_limit_from_state(state, N) >= N) return; }
    if (!PASS1 && DEEP != (chunk_base[n_tiles] >= kDeepChunks)) return;              // the other instantiation's view
    const int pix = threadIdx.x;
    // pass 0: one tile per workgroup; pass 1 (a no-op unless the global early exit fired): small grid, grid stride
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t c0 = chunk_base[tile], c1 = chunk_base[tile + 1];
    int x, y; pix_xy(pix >> 6, pix & 63, tile % tw, tile / tw, x, y);
    const bool inside = x < W && y < H;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dm = 0.f, m = 1.f; int e = 0;
    // chunk states are independent loads: fetch kCombineBatch chunks ahead of the dependent scan so that the
    // longest tile (tens of chunks) is not one memory latency per chunk
    constexpr int kCombineBatch = 16;
    __shared__ uint8_t s_dark[kCombineBatch][4];
    // the last batch's local results stay in registers for the backward scan (most tiles have one batch)
    float l0[kCombineBatch], l1[kCombineBatch], l2[kCombineBatch], l3[kCombineBatch], l4[kCombineBatch];
    // "Forward occlusion skipping": this pixel's darkness sum over the chunks in front, and whether an unbounded chunk has
    // been passed (nothing is dark behind one).  One wave = one 8x8 quadrant, so the quadrant's verdict is a ballot.
    // Only in a DEEP view (kDeepChunks: a function of the view's chunk count): shallow views neither publish nor skip.
    int dsum = 0;
    const uint32_t tile_growth = (DEEP && c1 > c0 && chunk_base[n_tiles] >= kDeepChunks) ? (hdr[c0] & ~kHdrPublished) : kTileUnbounded;
    const bool blocked = (tile_growth & kTileUnbounded) != 0u;
    const int dark_thr = kDarkQ + (int)(tile_growth & ~kTileUnbounded);
    bool for_good = false;                    // this wave's quadrant has passed its dark point (never without DEEP: `blocked`)
    bool tile_nonmono = false;                // some splat of the tile's list can take A down (kMetaNonMono of its chunks)
    uint32_t skipped = 0;                     // bit u: this quadrant of chunk u of the (last) batch is dark -- the identity, computed or not
    for (uint32_t cb = c0; cb < c1; cb += kCombineBatch) {
        float l5[kCombineBatch];
        skipped = 0;
        // lane u of every wave fetches chunk cb + u's flag word together with the states; the scan below reads it as a lane
        // broadcast (a scalar load inside the dependent scan is one more memory latency per chunk)
        // (a quadrant that is past its dark point needs none of this: in a deep tile that is most of the list -- the states
        // of a config-5 view are 5 KB a chunk, and reading them all cost as much as the chunk pass)
        uint32_t meta_l = 0u;
        if (!for_good) {
            if ((threadIdx.x & 63) < kCombineBatch) meta_l = chunk_meta[min(cb + (threadIdx.x & 63u), c1 - 1)];
            tile_nonmono = tile_nonmono || __builtin_amdgcn_ballot_w64((meta_l & kMetaNonMono) != 0u) != 0ull;
#pragma unroll
            for (int u = 0; u < kCombineBatch; u++) {
                const uint32_t c = min(cb + u, c1 - 1);
                l0[u] = *state_ptr(chunk_state, c, 0, pix); l1[u] = *state_ptr(chunk_state, c, 1, pix);
                l2[u] = *state_ptr(chunk_state, c, 2, pix); l3[u] = *state_ptr(chunk_state, c, 3, pix);
                l4[u] = *state_ptr(chunk_state, c, 4, pix);
                l5[u] = DEPTH ? *state_ptr(chunk_state, c, 5, pix) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < kCombineBatch; u++) {
            const uint32_t c = cb + u;
            if (c < c1) {
                if (!blocked && !for_good) for_good = __builtin_amdgcn_ballot_w64(dsum < dark_thr) == 0ull;
                if (for_good) {                               // the identity here, whether or not the chunk pass computed it
                    skipped |= 1u << u;
                    if (need_bwd && (threadIdx.x & 63) == 0) {
                        reinterpret_cast<uint8_t*>(chunk_dark + c)[threadIdx.x >> 6] = 0x81;      // dark (bit 0), by this rule (bit 7)
                        s_dark[u][threadIdx.x >> 6] = 1;
                    }
                    continue;
                }
                // not dark: the chunk pass computed this quadrant and published what it lets through
                if (!blocked) dsum += (int)((pay[(size_t)c * 64 + (threadIdx.x & 63)] >> (8 * (threadIdx.x >> 6))) & 0xffu) - kPayBias;
                if (need_bwd) {
                    // occlusion cut: byte w of chunk_dark[c] = 1 when every pixel of quadrant w (this wave's 64 pixels)
                    // enters chunk c with |T| < 2^-(GSX_CUT_LOG2 + growth bits of c) (|m| < 1 after renorm, so a test
                    // on e suffices): then |T| stays below 2^-24 throughout the chunk.  Pixels outside the image carry
                    // no gradient and count as dark
                    const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)meta_l, u) >> GSX_META_GROWTH_SHIFT;          // bits |T| may grow inside chunk c
                    const bool darkp = gb < GSX_META_GROWTH_MAX && !(H < GSX_TILE || W < GSX_TILE) && (!inside || m == 0.0f || e <= -GSX_CUT_LOG2_FOR(H, W) - (int)gb);
                    const bool all_dark = __builtin_amdgcn_ballot_w64(!darkp) == 0ull;
                    if ((threadIdx.x & 63) == 0) {
                        reinterpret_cast<uint8_t*>(chunk_dark + c)[threadIdx.x >> 6] = all_dark ? 1 : 0;
                        s_dark[u][threadIdx.x >> 6] = all_dark ? 1 : 0;
                    }
                }
                const float Tv = ldexpf(m, e);
                C0 += Tv * l0[u]; C1 += Tv * l1[u]; C2 += Tv * l2[u];
                if (DEPTH) Dm += Tv * l5[u];
                m *= l3[u];
                e += __float_as_int(l4[u]);
                renorm(m, e);
                if (need_bwd) { *state_ptr(chunk_state, c, 6, pix) = m; *state_ptr(chunk_state, c, 7, pix) = __int_as_float(e); }
            }
        }
        if (need_bwd) {
            // Work lists of the backward, by expected cost, so that the dispatcher hands out the long work items
            // first and the trivial ones (occluded chunks: zero rows only) fill the tail.  The order inside a
            // list depends on timing; the results do not (every chunk writes its own rows).
            __syncthreads();
            if (threadIdx.x < 64) {                       // wave 0: one atomic per class per batch, not per chunk
                const uint32_t c = cb + threadIdx.x;
                const bool valid = threadIdx.x < kCombineBatch && c < c1;
                int cls = -1;
                bool halves = false;          // this chunk goes into the list as two entries (first half, second half)
                if (valid) {
                    const uint32_t meta = chunk_meta[c];
                    const uint32_t n_dark = (uint32_t)(s_dark[threadIdx.x][0] + s_dark[threadIdx.x][1] + s_dark[threadIdx.x][2] + s_dark[threadIdx.x][3]);
                    uint32_t est = ((meta >> GSX_META_COST_SHIFT) & GSX_META_COST_MASK) * (4u - n_dark);      // <= 4 * 4 * chunk size
                    const uint32_t ch = chunk_base[n_tiles + 1];
                    // A shallow view's chunks of 128 or 256 instances with something in both halves: the chunk pass left the
                    // second half's local result (mid_state), so the first half can start on a wave of its own.  Not the
                    // occluded ones (nothing to compute), not after a global early exit (pass 1 rewrites the states only).
                    if (!PASS1 && split_halves && ch >= 128u && n_dark != 4u) {
                        const uint32_t len = (c + 1u < c1) ? ch : (ranges[2 * tile + 1] - ranges[2 * tile]) - (c - c0) * ch;
                        halves = len > (ch >> 1);
                    }
                    if (halves) est >>= 1;
                    cls = n_dark == 4u ? 4 : est >= 10u * ch ? 0 : est >= 6u * ch ? 1 : est >= 3u * ch ? 2 : 3;
                    if (halves) cls = est >= 5u * ch ? 0 : est >= 3u * ch ? 1 : est >= 3u * (ch >> 1) ? 2 : 3;
                }
                const unsigned long long lt = (1ull << threadIdx.x) - 1ull;
                const unsigned long long hv = __builtin_amdgcn_ballot_w64(halves);
                uint32_t* cnt = work_counters(work, PASS1);
                // lane k reserves class k's slots: the five atomics are in flight together (one L2 round trip per batch;
                // issued one after the other they were most of a long tile's time in this kernel)
                uint32_t mine = 0, below = 0;
#pragma unroll
                for (int k = 0; k < kWorkClasses; k++) {
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(cls == k);
                    if ((int)threadIdx.x == k) mine = (uint32_t)(__popcll(mk) + __popcll(mk & hv));
                    if (cls == k) below = (uint32_t)(__popcll(mk & lt) + __popcll(mk & hv & lt));
                }
                uint32_t base = 0;
                if (threadIdx.x < kWorkClasses && mine) base = atomicAdd(cnt + threadIdx.x, mine);
                base = (uint32_t)__shfl((int)base, cls < 0 ? 0 : cls, 64);
                if (cls >= 0) {
                    uint32_t* list = work_list(work, PASS1, cls, max_chunks);
                    if (halves) { list[base + below] = c | kItemBack; list[base + below + 1u] = c | kItemFront; }
                    else list[base + below] = c;
                }
            }
            __syncthreads();
        }
    }
    const float A = 1.0f - ldexpf(m, e);
    if (inside) {
        const size_t P = (size_t)H * W, p = (size_t)y * W + x;
        rgb[p] = C0; rgb[P + p] = C1; rgb[2*P + p] = C2;
        alpha[p] = A;
        if (DEPTH) depth_map[p] = Dm;
    }
    if (!PASS1) {
        // a pixel at least 1e-5 below the threshold is unsaturated in any arithmetic.  A NaN does not veto: it comes from a
        // splat with exp2(q) = inf (a Gaussian at the camera plane: its 2x2 covariance is not positive definite) somewhere in
        // the list, possibly BEHIND the rank at which the reference had already left its loop -- the exact search (F3) then
        // decides with the reference's own arithmetic, and finds no exit if such a pixel never saturated before the NaN
        // Only a pixel whose list is MONOTONE vetoes: with every alpha in [0, 1] its A never was above its final value, so
        // "not saturated at the end" means "never saturated" and the reference's test (every pixel > 0.99 at once) never
        // held.  A pixel of a tile with a non-monotone splat may have been above 0.99 in the middle of its list and come
        // back down (opacities in (1, 3]: tools/fuzz_parity.py seed 3480, exit after 2 of 7 with a final alpha of 0.84):
        // it does not veto, and the exact search decides.  An empty tile is monotone: views that leave tiles empty -- every
        // training view of the bench scene -- never reach the search, whatever their opacities do.
        const bool unsat = inside && (A <= GSX_SAT_ALPHA - 1e-5f) && !tile_nonmono;
        // thousands of tiles raise the same flag: look before the atomic (4096 same-address atomics serialise in L2)
        // (bit 1: some tile holds a splat that can take A down -- the exit search then needs its rounds)
        const uint32_t raise = (__syncthreads_or(unsat) ? 1u : 0u) | (__syncthreads_or(tile_nonmono) ? 2u : 0u);      // (a wave past its dark point saw fewer flags)
        if (raise && threadIdx.x == 0 &&
            (__hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & raise) != raise) atomicOr(state + 1, raise);
    }
    if (need_bwd && c1 > c0) {
        float B0 = 0.f, B1 = 0.f, B2 = 0.f;
        const uint32_t cbl = c0 + ((c1 - c0 - 1u) / kCombineBatch) * kCombineBatch;    // first chunk of the last forward batch
#pragma unroll
        for (int u = kCombineBatch - 1; u >= 0; u--) {
            const uint32_t c = cbl + u;
            if (c < c1 && !(skipped & (1u << u))) {
                const float Tl = ldexpf(l3[u], __float_as_int(l4[u]));
                *state_ptr(chunk_state, c, 0, pix) = B0; *state_ptr(chunk_state, c, 1, pix) = B1; *state_ptr(chunk_state, c, 2, pix) = B2;
                B0 = l0[u] + Tl * B0; B1 = l1[u] + Tl * B1; B2 = l2[u] + Tl * B2;
            }
        }
        for (uint32_t ce = cbl; ce > c0; ce -= kCombineBatch) {                          // earlier batches: full, reloaded
            // dark by the rule is dark for good: if the batch's first chunk is, all of it is the identity for this quadrant
            // (lane u fetches this quadrant's dark byte of chunk ce - 1 - u: one load per batch, not one in front of every chunk)
            uint32_t dkb = 0u;
            if ((threadIdx.x & 63u) < (uint32_t)kCombineBatch)
                dkb = reinterpret_cast<const uint8_t*>(chunk_dark + (ce - 1u - (threadIdx.x & 63u)))[threadIdx.x >> 6];
            const uint32_t by_rule = (uint32_t)__builtin_amdgcn_ballot_w64(dkb == 0x81u);
            if (by_rule & (1u << (kCombineBatch - 1))) continue;
#pragma unroll
            for (int u = 0; u < kCombineBatch; u++) {
                const uint32_t c = ce - 1 - u;               // (earlier batches are full: ce - c0 >= kCombineBatch)
                l0[u] = *state_ptr(chunk_state, c, 0, pix); l1[u] = *state_ptr(chunk_state, c, 1, pix);
                l2[u] = *state_ptr(chunk_state, c, 2, pix); l3[u] = *state_ptr(chunk_state, c, 3, pix);
                l4[u] = *state_ptr(chunk_state, c, 4, pix);
            }
#pragma unroll
            for (int u = 0; u < kCombineBatch; u++) {
                if (!(by_rule & (1u << u))) {                 // (dark by the rule: the identity)
                    const uint32_t c = ce - 1 - u;
                    const float Tl = ldexpf(l3[u], __float_as_int(l4[u]));
                    *state_ptr(chunk_state, c, 0, pix) = B0; *state_ptr(chunk_state, c, 1, pix) = B1; *state_ptr(chunk_state, c, 2, pix) = B2;
                    B0 = l0[u] + Tl * B0; B1 = l1[u] + Tl * B1; B2 = l2[u] + Tl * B2;
                }
            }
        }
    }
    if (!PASS1) break;                        // compile-time: pass 0 has no back edge
    __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// F2 for SHALLOW views (pass 0 of a view of fewer than kDeepChunks chunks; round 3): the same results from a workgroup
// of 1024 threads that takes a tile's chunks 64 at a time, 16 per GROUP of 256 threads (thread = pixel as above, the four
// groups side by side along the list).  Compositing is associative: every group folds its 16 local results into one
// (colour, transmittance), the four meet in LDS, and each group then knows what stands in front of it and what comes
// behind it.  Why: the bench scene puts all of a view's instances on 250-400 of its 4096 tiles, 30-60 chunks each
// (tools/tile_depth_stats.py), so the kernel above was ~300 workgroups walking three or four dependent batches each --
// 40 us of memory round trips for the forward scan alone (ablation: tools/ab_kstats.sh) with the chip empty.  Here such a tile
// is ONE batch: one round trip for the states, one barrier, one reservation in the work lists.
// Tiles of more than 48 chunks loop (forward, then backward with the states reloaded, as above).
// A tile of at most U = 12 chunks is one group's work, and a workgroup takes FOUR of them side by side (the list's second part:
// tile_list_block).  After a thousand trainer steps the bench scene has splats on all 4096 tiles, 3700 of them with one or two
// chunks: one tile per workgroup and iteration was 8 dependent iterations per workgroup with three groups idle (97 us); four
// at a time it is 47 us, and the bench scene's own 300 deep tiles are untouched (tools/soak_ktrace.sh).
// ------------------------------------------------------------------------------------------
// four groups of 256 threads, 12 chunks per group and batch: what 128 registers hold without scratch (with the depth map's
// sixth word 9 registers go to scratch; the two instantiations must cut the list alike, or the image would depend in its
// last bit on whether a depth map was asked for)
constexpr int kParGroups = 4;
template <bool DEPTH> struct ParShape { static constexpr int U = 12, Batch = kParGroups * U; };
constexpr int kParGrid = 512;             // workgroups of the combine pass at most (one of 1024 threads fits a CU): beyond that, a stride
template <bool DEPTH>
__global__ __launch_bounds__(kParGroups * kBlock) void tile_combine_par_kernel(
    const uint32_t* __restrict__ chunk_base, int H, int W, int tw, int need_bwd,
    uint32_t* __restrict__ state, float* __restrict__ chunk_state, const uint32_t* __restrict__ chunk_meta,
    uint32_t* __restrict__ chunk_dark, uint32_t* __restrict__ work, uint32_t max_chunks, int n_tiles,
    float* __restrict__ rgb, float* __restrict__ alpha, float* __restrict__ depth_map, const uint32_t* __restrict__ ranges,
    const uint32_t* __restrict__ list)
{
    constexpr int G = kParGroups, U = ParShape<DEPTH>::U, kParBatch = ParShape<DEPTH>::Batch, NA = DEPTH ? 6 : 5;
    static_assert(kParBatch <= 64 && G * kBlock <= 1024, "one wave builds a batch's work-list entries, lane = chunk");
    static_assert(U == kParSmall, "the list's second part holds the tiles that fit one group");
    // per group and pixel: what the group's chunks add (colour 3, depth with DEPTH) and let through (mantissa, exponent)
    __shared__ float s_agg[G][NA][kPixPerTile];
    __shared__ uint8_t s_dark[kParBatch][4];
    // (the exit search's signal words, for shallow and deep views alike: this launch is in front of every search)
    if (blockIdx.x == 0 && threadIdx.x < max(sig_copies(max_chunks), 1u)) *sig_word(work, max_chunks, threadIdx.x) = 0ull;
    if (chunk_base[n_tiles] >= kDeepChunks) return;                                  // the deep instantiation's view
    const uint32_t n_held = work[kWorkNonEmpty];
    // alpha = 0 on an empty tile is "not saturated": it vetoes the global exit (one workgroup says so for all of them; a
    // look-and-raise per empty tile was 3700 reads of one L2 line)
    // (and says "no exit rank" itself: the search behind this launch would, but the one-call trainer step does not even enqueue it
    // when the host knows of an empty tile -- gsx_view.hip)
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_held < (uint32_t)n_tiles) { atomicOr(state + 1, 1u); state[0] = GSX_RANK_NONE; }
    const uint32_t n_big = min(work[kWorkBig], n_held);
    // One tile.  SOLO: the tile is this GROUP's alone (up to U chunks: one batch, nothing to exchange), four tiles to the workgroup;
    // `slot0` is the list position of group 0's tile, `valid` false for a group past the list's end (it still meets every barrier).
    // The same operations in the same order as a tile that has the four groups to itself and chunks for the first only: the same bits.
    auto one_tile = [&](auto solo_c, const int tile, const bool valid, const uint32_t slot0) {
    constexpr bool SOLO = decltype(solo_c)::value;
    const uint32_t c0 = valid ? chunk_base[tile] : 0u, c1 = valid ? chunk_base[tile + 1] : 0u;
    // (group and quadrant are wave-uniform: in scalar registers, so that a chunk's address is a scalar base + the pixel's offset)
    const int pix = threadIdx.x & (kPixPerTile - 1);
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int grp_at = SOLO ? 0 : grp;                                // where in a batch the group's chunks stand
    const int quad = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 3u));
    const uint32_t lane = threadIdx.x & 63u;
    int x, y; pix_xy(quad, pix & 63, tile % tw, tile / tw, x, y);
    const bool inside = x < W && y < H;
    const bool tiny = H < GSX_TILE || W < GSX_TILE;
    const int cut = GSX_CUT_LOG2_FOR(H, W);
    // the tile's state in front of the batch at hand: the same bits in every group (same values, same order)
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dm = 0.f, m = 1.f; int e = 0;
    bool nonmono = false;
    float l0[U], l1[U], l2[U], l3[U], l4[U];
    uint32_t sb_last = c0;
    // work-list reservation of the last batch (wave 0), emitted after the backward scan
    int wl_cls = -1; bool wl_halves = false; uint32_t wl_base = 0u, wl_below = 0u, wl_chunk = 0u;
    const uint32_t ch = chunk_base[n_tiles + 1];

    // the states of this group's chunks [g0, g0 + U) (clamped: a group past the list's end loads nothing)
    auto load_states = [&](uint32_t g0) {
        if (g0 < c1) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (SOLO && g0 + u >= c1) break;                  // (uniform over the wave; a small tile is mostly one or two chunks)
                const uint32_t c = min(g0 + u, c1 - 1u);
                l0[u] = *state_ptr(chunk_state, c, 0, pix); l1[u] = *state_ptr(chunk_state, c, 1, pix);
                l2[u] = *state_ptr(chunk_state, c, 2, pix); l3[u] = *state_ptr(chunk_state, c, 3, pix);
                l4[u] = *state_ptr(chunk_state, c, 4, pix);
            }
        }
    };
    // ... folded into one (identity for a group past the end)
    auto fold = [&](uint32_t g0, const float* l5, float& a0, float& a1, float& a2, float& ad, float& am, int& ae) {
        a0 = 0.f; a1 = 0.f; a2 = 0.f; ad = 0.f; am = 1.f; ae = 0;
        if (g0 < c1) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (g0 + u < c1) {
                    const float Tv = ldexpf(am, ae);
                    a0 += Tv * l0[u]; a1 += Tv * l1[u]; a2 += Tv * l2[u];
                    if (DEPTH && l5) ad += Tv * l5[u];
                    am *= l3[u];
                    ae += __float_as_int(l4[u]);
                    renorm(am, ae);
                }
            }
        }
    };
    // ... and published
    auto publish = [&](uint32_t g0, const float* l5) {
        float a0, a1, a2, ad, am; int ae;
        fold(g0, l5, a0, a1, a2, ad, am, ae);
        s_agg[grp][0][pix] = a0; s_agg[grp][1][pix] = a1; s_agg[grp][2][pix] = a2;
        if (DEPTH) s_agg[grp][3][pix] = ad;
        s_agg[grp][NA - 2][pix] = am; s_agg[grp][NA - 1][pix] = __int_as_float(ae);
    };
    // colour behind this group's last chunk (b) and in front of the whole batch (F), per unit of transmittance there,
    // from the colour F behind the batch
    auto behind = [&](float& F0, float& F1, float& F2, float& b0, float& b1, float& b2) {
#pragma unroll
        for (int g = G - 1; g >= 0; g--) {
            if (g == grp) { b0 = F0; b1 = F1; b2 = F2; }
            const float Tg = ldexpf(s_agg[g][NA - 2][pix], __float_as_int(s_agg[g][NA - 1][pix]));
            F0 = s_agg[g][0][pix] + Tg * F0; F1 = s_agg[g][1][pix] + Tg * F1; F2 = s_agg[g][2][pix] + Tg * F2;
        }
    };
    // each chunk's local colour replaced by the colour behind it, back to front
    auto scan_back = [&](uint32_t g0, float b0, float b1, float b2) {
        if (g0 < c1) {
#pragma unroll
            for (int u = U - 1; u >= 0; u--) {
                const uint32_t c = g0 + u;
                if (c < c1) {
                    const float Tl = ldexpf(l3[u], __float_as_int(l4[u]));
                    *state_ptr(chunk_state, c, 0, pix) = b0; *state_ptr(chunk_state, c, 1, pix) = b1; *state_ptr(chunk_state, c, 2, pix) = b2;
                    b0 = l0[u] + Tl * b0; b1 = l1[u] + Tl * b1; b2 = l2[u] + Tl * b2;
                }
            }
        }
    };
    auto list_emit = [&](uint32_t c) {
        const uint32_t base = (uint32_t)__shfl((int)wl_base, wl_cls < 0 ? 0 : wl_cls, 64);      // (lane k holds class k's base)
        if (wl_cls >= 0) {
            uint32_t* list = work_list(work, false, wl_cls, max_chunks);
            if (wl_halves) { list[base + wl_below] = c | kItemBack; list[base + wl_below + 1u] = c | kItemFront; }
            else list[base + wl_below] = c;
        }
    };

    for (uint32_t sb = c0; SOLO ? sb == c0 : sb < c1; sb += kParBatch) {
        sb_last = sb;
        const uint32_t g0 = sb + (uint32_t)grp_at * U;
        uint32_t meta_l = 0u;                 // lane u: chunk g0 + u's flag word
        float l5[DEPTH ? U : 1];
        if (g0 < c1) {
            if (lane < (uint32_t)U) meta_l = chunk_meta[min(g0 + lane, c1 - 1u)];
            if (DEPTH) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (SOLO && g0 + u >= c1) break;
                    l5[u] = *state_ptr(chunk_state, min(g0 + u, c1 - 1u), 5, pix);
                }
            }
        }
        load_states(g0);
        nonmono = nonmono || __builtin_amdgcn_ballot_w64((meta_l & kMetaNonMono) != 0u) != 0ull;
        // what stands in front of this group (pm, pe); the state behind the whole batch
        float pm = 1.f; int pe = 0;
        if constexpr (SOLO) fold(g0, l5, C0, C1, C2, Dm, m, e);
        else {
        publish(g0, l5);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g == grp) { pm = m; pe = e; }
            const float Tv = ldexpf(m, e);
            C0 += Tv * s_agg[g][0][pix]; C1 += Tv * s_agg[g][1][pix]; C2 += Tv * s_agg[g][2][pix];
            if (DEPTH) Dm += Tv * s_agg[g][3][pix];
            m *= s_agg[g][NA - 2][pix];
            e += __float_as_int(s_agg[g][NA - 1][pix]);
            renorm(m, e);
        }
        }
        if (need_bwd && g0 < c1) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = g0 + u;
                if (SOLO && c >= c1) break;
                if (c < c1) {
                    // occlusion cut (see the kernel above): quadrant `quad` of chunk c is dark when every pixel enters with
                    // |T| < 2^-(cut + growth bits of c)
                    const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)meta_l, u) >> GSX_META_GROWTH_SHIFT;
                    const bool darkp = gb < GSX_META_GROWTH_MAX && !tiny && (!inside || pm == 0.0f || pe <= -cut - (int)gb);
                    const bool all_dark = __builtin_amdgcn_ballot_w64(!darkp) == 0ull;
                    if (lane == 0u) {
                        reinterpret_cast<uint8_t*>(chunk_dark + c)[quad] = all_dark ? 1 : 0;
                        s_dark[grp * U + u][quad] = all_dark ? 1 : 0;
                    }
                    pm *= l3[u];
                    pe += __float_as_int(l4[u]);
                    renorm(pm, pe);
                    *state_ptr(chunk_state, c, 6, pix) = pm; *state_ptr(chunk_state, c, 7, pix) = __int_as_float(pe);
                }
            }
        }
        __syncthreads();                      // s_dark is complete; s_agg may be rewritten
        if (need_bwd && threadIdx.x < 64u) {
            // Work lists of the backward by expected cost (see the kernel above), lane = chunk: one reservation per class per batch
            // (SOLO: lanes [g * U, g * U + U) are the chunks of group g's tile)
            uint32_t c = sb + lane, wc0 = c0, wc1 = c1; int wtile = tile;
            if constexpr (SOLO) {
                const uint32_t g = lane / (uint32_t)U, slot = slot0 + g;
                wc0 = wc1 = 0u;
                if (g < (uint32_t)G && slot < n_held) { wtile = (int)list[slot]; wc0 = chunk_base[wtile]; wc1 = chunk_base[wtile + 1]; }
                c = wc0 + (lane - g * (uint32_t)U);
            }
            int cls = -1; bool halves = false;
            if (c < wc1 && lane < (uint32_t)kParBatch) {
                const uint32_t meta = chunk_meta[c];
                const uint32_t n_dark = (uint32_t)(s_dark[lane][0] + s_dark[lane][1] + s_dark[lane][2] + s_dark[lane][3]);
                uint32_t est = ((meta >> GSX_META_COST_SHIFT) & GSX_META_COST_MASK) * (4u - n_dark);
                if (ch >= 128u && n_dark != 4u) {
                    const uint32_t len = (c + 1u < wc1) ? ch : (ranges[2 * wtile + 1] - ranges[2 * wtile]) - (c - wc0) * ch;
                    halves = len > (ch >> 1);
                }
                if (halves) { est >>= 1; cls = est >= 5u * ch ? 0 : est >= 3u * ch ? 1 : est >= 3u * (ch >> 1) ? 2 : 3; }
                else cls = n_dark == 4u ? 4 : est >= 10u * ch ? 0 : est >= 6u * ch ? 1 : est >= 3u * ch ? 2 : 3;
            }
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long hv = __builtin_amdgcn_ballot_w64(halves);
            uint32_t mine = 0u, below = 0u;
#pragma unroll
            for (int k = 0; k < kWorkClasses; k++) {
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(cls == k);
                if ((int)lane == k) mine = (uint32_t)(__popcll(mk) + __popcll(mk & hv));
                if (cls == k) below = (uint32_t)(__popcll(mk & lt) + __popcll(mk & hv & lt));
            }
            uint32_t base = 0u;
            if (lane < (uint32_t)kWorkClasses && mine) base = atomicAdd(work_counters(work, false) + lane, mine);
            wl_cls = cls; wl_halves = halves; wl_base = base; wl_below = below; wl_chunk = c;
            if (!SOLO && sb + kParBatch < c1) list_emit(c); // (the last batch's round trip runs under the backward scan)
        }
    }
    const float A = 1.0f - ldexpf(m, e);
    if ((SOLO ? valid : grp == 0) && inside) {
        const size_t P = (size_t)H * W, p = (size_t)y * W + x;
        rgb[p] = C0; rgb[P + p] = C1; rgb[2*P + p] = C2;
        alpha[p] = A;
        if (DEPTH) depth_map[p] = Dm;
    }
    {
        // the veto of the global early exit and the non-monotone flag: as in the kernel above
        // (SOLO: every wave of a group has read its tile's flag words, so `nonmono` is the tile's already)
        const bool any_nonmono = __syncthreads_or(nonmono) != 0;
        const bool tile_nonmono = SOLO ? nonmono : any_nonmono;
        const bool unsat = (SOLO ? valid : grp == 0) && inside && (A <= GSX_SAT_ALPHA - 1e-5f) && !tile_nonmono;
        const uint32_t raise = (__syncthreads_or(unsat) ? 1u : 0u) | (any_nonmono ? 2u : 0u);
        if (raise && threadIdx.x == 0 &&
            (__hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & raise) != raise) atomicOr(state + 1, raise);
    }
    if constexpr (SOLO) {
        if (need_bwd) {
            scan_back(c0, 0.f, 0.f, 0.f);                        // (nothing behind the tile's only batch; past-the-end groups: c0 = c1)
            if (threadIdx.x < 64u) list_emit(wl_chunk);
        }
    } else if (need_bwd && c1 > c0) {
        // the last batch: states still in registers, the groups' results still in LDS
        float F0 = 0.f, F1 = 0.f, F2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        behind(F0, F1, F2, b0, b1, b2);
        scan_back(sb_last + (uint32_t)grp * U, b0, b1, b2);
        if (threadIdx.x < 64u) list_emit(sb_last + lane);
        for (uint32_t sb = sb_last; sb > c0; ) {                 // earlier batches: full, reloaded
            sb -= kParBatch;
            const uint32_t g0 = sb + (uint32_t)grp * U;
            load_states(g0);
            __syncthreads();                                     // everybody has read the previous batch's results
            publish(g0, nullptr);
            __syncthreads();
            behind(F0, F1, F2, b0, b1, b2);
            scan_back(g0, b0, b1, b2);
        }
    }
    __syncthreads();                          // the next tile rewrites s_agg and s_dark
    };
    // the big tiles, one at a time; then the small ones, four at a time
    for (uint32_t item = blockIdx.x; item < n_big; item += gridDim.x)
        one_tile(std::false_type{}, (int)list[item], true, 0u);
    const int grp_k = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const uint32_t first_q = (blockIdx.x + gridDim.x - n_big % gridDim.x) % gridDim.x;      // (the first workgroup without a big tile takes the first four)
    for (uint32_t slot0 = n_big + first_q * G; slot0 < n_held; slot0 += gridDim.x * G) {
        const bool valid = slot0 + (uint32_t)grp_k < n_held;
        one_tile(std::true_type{}, valid ? (int)list[slot0 + (uint32_t)grp_k] : 0, valid, slot0);
    }
}

// ------------------------------------------------------------------------------------------
// F3: exact global early-exit index (gaussian_splatting.py:147): the first rank k* after which EVERY pixel's accumulated
// alpha is > 0.99 at once.  Runs only when F2 found every pixel saturated at the end of its list.  Workgroups walk the tile
// lists with the REFERENCE's arithmetic (A += (1-A)*alpha, :144).
// With every alpha in [0, 1] a pixel's A only grows, and k* is the maximum over the pixels of the first rank after which
// A > 0.99.  An opacity outside [0, 1] lets A overshoot and come back under the threshold (training pushes raw opacities
// past 1; with opacities in (1, 3] the first-crossing rule was off on most scenes: tests/golden exitop*), so the search is
// the fixpoint of
//     k <- max over pixels p of f_p(k),   f_p(k) = the first rank k' >= k at which p is saturated
// ("at k'": after the last instance of p's tile with rank <= k').  k never decreases, never passes k* (at k* every pixel is
// saturated, so f_p(k) <= k* for k <= k*), and a fixpoint is a rank at which all pixels are saturated: it is k*.  One pass
// over the lists per round, a grid barrier between rounds (the grid is at most kSatBlocks workgroups, all resident; every
// workgroup arrives at every barrier, and the rounds end because k grows until it stands or some pixel reports "never");
// a view without a non-monotone splat (state[1] bit 1 clear: the combine pass raises it from the chunks' flags) takes the
// one pass it always took, on the whole grid, without a barrier.  state[2] = the running maximum (monotone: no reset between
// rounds), zeroed with state[0..1] by pass 0; state[3] gets a copy of the barrier's arrivals at the end.
// What the rounds cost on a trained scene (round 3; profiles/r3_soak_3000_steps.txt): NOT the walks.  Loads and atomics of one
// address from all over the chip are served one after the other, some 60 ns each, so every word that 512 workgroups look at
// costs each of them 30 us a look -- the running maximum, the barrier's counter, a "search is over" flag; with a look per
// batch of 256 splats a view's search took 5 ms where its walks are a tenth of that.  Hence: (i) what the workgroups wait
// for and look at while they walk is a SIGNAL WORD {round, rank}, kept in up to 256 copies on lines of their own (eight
// waves to a copy); the last workgroup to arrive at a barrier reads the maximum and writes the round's word to every copy,
// the first workgroup to find a pixel that never saturates writes all-ones; (ii) a workgroup keeps, per tile, where its walk
// stood (checkpoints, below), so the rounds together are one pass over the lists; (iii) the tiles are dealt from the combine
// pass's list, deep ones first, one per workgroup, and a workgroup takes its sparse tiles first; (iv) a wave walks one 8x8
// quadrant and only the splats whose mask byte (the chunk pass's) reaches it; (v) a splat with a rank below the round's k only
// moves A, in a loop of half the instructions (a wave nearly alone on its SIMD is bound by instruction issue).
// ------------------------------------------------------------------------------------------
constexpr int kSatBlocks = 512;           // workgroups that take part in the rounds of a non-monotone view: a quarter of what the chip holds
                                          // resident (8 of these per CU), so that up to four streams' searches can spin side by side
// (eight waves per SIMD, i.e. at most 64 VGPRs: with the LDS bound below, eight of these workgroups fit a CU -- what "four searches side
// by side" needs: 4 x 512 workgroups resident at once, or four searches could each hold a part of the chip and wait for the rest)
__global__ __launch_bounds__(kBlock, 8) void saturation_rank_kernel(
    const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
    int H, int W, int tw, int n_tiles, float stepx, float stepy, uint32_t* __restrict__ state,
    const uint32_t* __restrict__ chunk_base, const uint8_t* __restrict__ masks, uint32_t* __restrict__ work,
    const uint32_t* __restrict__ list, uint32_t max_chunks)
{
    // The barrier's arrivals count in the work counters' line (a word nothing else uses between the passes), NOT next to state[2];
    // nobody waits on it: the waiting is on the signal words.
    uint32_t* const bar_arrivals = work + kWorkArrivals;
    const uint32_t n_sig = max(sig_copies(max_chunks), 1u);
    // (a copy per wave: every wave looks for itself while it walks, below)
    unsigned long long* const my_sig = sig_word(work, max_chunks, (blockIdx.x * 4u + (threadIdx.x >> 6)) % n_sig);
    constexpr unsigned long long kSigNever = ~0ull;
    __shared__ float4 s_a[kBlock], s_b[kBlock];
    __shared__ uint32_t s_rank[kBlock];
    // which of the batch's 256 splats reach which 8x8 quadrant of the tile: [staging wave][quadrant] -> 64 bits.  Wave w walks
    // quadrant w and only the splats that reach it -- the masks are the chunk pass's (a shallow view leaves one byte per instance in
    // its chunks' payload words); what they drop is what the footprint cut drops from the tile lists themselves: pairs below the cut.
    // A trained bench scene's walk takes 1.5 of a splat's 4 quadrants on average.  Without masks (deep views) every splat is taken.
    __shared__ unsigned long long s_sets[4][4];
    __shared__ unsigned long long s_below[4];         // [staging wave] -> which of its 64 splats have a rank below the round's k (a prefix: the lists are in rank order)
    __shared__ uint32_t s_red[4];
    __shared__ uint32_t s_k;
    // Checkpoints of a non-monotone view's rounds: where the walk of the workgroup's j-th tile stood at the START of the last batch it
    // took (position in the list, A of every pixel there).  k only grows, and every instance in front of that batch has a rank below
    // the round's new k (the batch was walked because a pixel found its rank in it, or met the first rank above the old k in it), so
    // the next round resumes there with the very A the walk from the list's head would arrive with -- the rounds together are ONE
    // pass over the lists instead of one each.  Tiles beyond the kSatCk-th of a workgroup are walked from their head as before.
    // Eight slots (every tile of an image of up to 4096 tiles): with them the kernel's LDS is 17.8 KB, i.e. eight workgroups per CU
    // -- the four searches of kSatBlocks workgroups that may spin side by side (above) still all fit the chip.
    constexpr int kSatCk = 8;
    static_assert((size_t)kSatCk * kBlock * 4 + 2 * kBlock * 16 + kBlock * 4 + 256 <= 160 * 1024 / 8, "eight searching workgroups per CU");
    __shared__ float s_ck_a[kSatCk][kBlock];
    __shared__ uint32_t s_ck_pos[kSatCk];
    const uint32_t flags = state[1];          // (uniform over the grid: written by pass 0's combine launch)
    if (flags & 1u) {                        // some pixel never saturates: no early exit
        if (blockIdx.x == 0 && threadIdx.x == 0) state[0] = GSX_RANK_NONE;
        return;
    }
    // A view without a non-monotone splat needs ONE pass and no barrier: every workgroup adds its maximum to state[0]
    // (zeroed by pass 0) and is done.  Otherwise the rounds below, on the first kSatBlocks workgroups only.
    const bool monotone = (flags & 2u) == 0u;
    const bool use_masks = masks != nullptr && chunk_base[n_tiles] < kDeepChunks;
    const bool by_list = list != nullptr && chunk_base[n_tiles] < kDeepChunks && work[kWorkNonEmpty] == (uint32_t)n_tiles;
    const uint32_t ch_mask = chunk_base[n_tiles + 1] - 1u;              // chunk size (64, 128 or 256) - 1
    const int ch_shift = 31 - __builtin_clz(chunk_base[n_tiles + 1]);
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const uint32_t n_blocks = monotone ? gridDim.x : min(gridDim.x, (uint32_t)kSatBlocks);
    if (blockIdx.x >= n_blocks) return;
    uint32_t k_cur = 0u, result;
    for (uint32_t round = 1u; ; round++) {
        uint32_t mine = 0u;                  // max of f_p(k_cur) over this thread's pixels (GSX_RANK_NONE = never: the largest value)
        // A pixel that is never saturated from k_cur on settles the search ("no exit"), whatever the other tiles say.  In the rounds of
        // a non-monotone view the workgroup that finds one says so at once (signal words, below) and every wave looks at its copy four
        // times a batch: a trained scene -- a few screen-sized splats put something on every tile, opacities have drifted out of
        // [0, 1] everywhere, nothing vetoes -- otherwise walks its deepest tile list to the end in ONE workgroup (17 k splats, 1.4 ms
        // on the bench scene after 1200 steps), while a sparse tile at the image's edge answers in microseconds.
        bool quit = false;
        // (the workgroup's tiles LAST FIRST: in list order its deep tile comes first, and a view with a pixel that never saturates --
        // nearly always on a sparse tile -- would have every workgroup a third of a millisecond into a deep list before anybody
        // found it)
        const int n_mine = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)n_blocks + 1 : 0;
        unsigned long long pending = 0ull;   // the wave's signal word as loaded one look ago
        bool leave = false;
        for (int slot = n_mine - 1; slot >= 0 && !quit; slot--) {
            const int ti = (int)blockIdx.x + slot * (int)n_blocks;
            const bool ck = !monotone && slot < kSatCk;
            // (a stride walk over the tiles: with the identity map the first tiles of all workgroups are the image's top rows -- on
            // the bench orbit the dense band itself, so that the first sparse tile, the one with the pixel that never saturates,
            // only came up in the second round of tiles, 60 us into the search)
            // A shallow view that got here has no empty tile, so the combine pass's list holds every tile, the deep ones (more than
            // twelve chunks) first: dealt out in that order every workgroup gets at most one of up to kSatBlocks deep tiles -- the
            // stride walk gave some workgroup three of a trained bench scene's 180 lists of 8-19 k splats, and the search is as long
            // as the longest workgroup's walks.
            // (the rest of the list in a stride walk: in list order all workgroups would be on the same eight rows of tiles at a time,
            // and the sparse tile whose pixel never saturates -- which ends the search for everybody -- may be anywhere)
            const uint32_t n_big = by_list ? min(work[kWorkBig], (uint32_t)n_tiles) : 0u;
            const int tile = !by_list ? (int)chunk_of_block((uint32_t)ti, (uint32_t)n_tiles)
                           : (uint32_t)ti < n_big ? (int)list[ti]
                           : (int)list[n_big + chunk_of_block((uint32_t)ti - n_big, (uint32_t)n_tiles - n_big)];
            const uint32_t begin = ranges[2*tile], end = ranges[2*tile+1];
            int x, y; pix_xy(wave, lane, tile % tw, tile / tw, x, y);      // (wave = quadrant)
            const bool inside = x < W && y < H;
            const uint32_t chunk0 = chunk_base[tile];
            const float gx = pix_coord(min(x, W - 1), W, stepx), gy = pix_coord(min(y, H - 1), H, stepy);
            float A = 0.f;
            uint32_t f = inside ? GSX_RANK_NONE : k_cur;      // pixels outside the image do not vote
            bool looked = false;                              // has the state AT k_cur been looked at
            uint32_t start0 = begin;
            if (ck && round > 1u && begin < end) { start0 = s_ck_pos[slot]; A = s_ck_a[slot][threadIdx.x]; }       // (own words / read behind barriers)
            // the next batch's records travel while this one is walked (rank, then record: two dependent round trips that every batch
            // of 256 splats used to wait for)
            uint32_t nr = 0u, nm = 0u; float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
            auto fetch = [&](uint32_t at) {
                const uint32_t i = at + threadIdx.x;
                nm = 0u;
                if (i < end) {
                    nr = ranks[i]; const float4* R = rec + (size_t)kRecVec * nr; na = R[0]; nb = R[1];
                    const uint32_t rel = i - begin;
                    nm = use_masks ? (uint32_t)masks[(size_t)(chunk0 + (rel >> ch_shift)) * (64 * 4) + (rel & ch_mask)] : 0xFu;
                }
            };
            fetch(start0);
            for (uint32_t start = start0; start < end; start += kBlock) {
                if (ck) { s_ck_a[slot][threadIdx.x] = A; if (threadIdx.x == 0) s_ck_pos[slot] = start; }
                if (start + threadIdx.x < end) { s_rank[threadIdx.x] = nr; s_a[threadIdx.x] = na; s_b[threadIdx.x] = nb; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned long long bset = __builtin_amdgcn_ballot_w64(((nm >> k) & 1u) != 0u);
                    if (lane == 0) s_sets[wave][k] = bset;
                }
                {
                    const unsigned long long bel = __builtin_amdgcn_ballot_w64(start + threadIdx.x < end && nr < k_cur);
                    if (lane == 0) s_below[wave] = bel;
                }
                __syncthreads();
                fetch(start + kBlock);
                const uint32_t cnt = min((uint32_t)kBlock, end - start);
                // Has somebody found a pixel that never gets there?  Every wave looks at its copy of the signal word four times a batch
                // (all lanes load the one address: one request; the answer is taken at the NEXT look, so nobody waits for it) and
                // leaves the batch where it stands: a batch of 256 splats is up to 12 us of walking, and the search of a view that
                // has no exit is as long as the slowest workgroup takes to notice.
                for (int seg = 0; seg < 4 && !leave; seg++) {
                    if (!monotone) {
                        if (pending == kSigNever) { leave = true; break; }
                        pending = __hip_atomic_load(my_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // A splat with a rank below k only moves A: nothing is looked at or found in front of k.  From the second round on
                    // that is most of what a deep tile walks (from where the last round stood up to the new k), at half the
                    // instructions of the full step -- and a wave that has its SIMD nearly to itself issues one instruction per
                    // ~5 cycles whatever they depend on, so the instruction count is the walk's time.
                    const unsigned long long set = s_sets[seg][wave], bel = s_below[seg];
                    for (unsigned long long todo = set & bel; todo != 0ull; ) {
                        const uint32_t j = (uint32_t)seg * 64u + (uint32_t)__builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        const float4 a = s_a[j], b = s_b[j];
                        const float dx = gx - a.x, dy = gy - a.y;
                        const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
                        const float am = b.y * __builtin_amdgcn_exp2f(q);
                        A = A + (1.0f - A) * am;
                    }
                    for (unsigned long long todo = set & ~bel; todo != 0ull; ) {
                        const uint32_t j = (uint32_t)seg * 64u + (uint32_t)__builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        const uint32_t r = s_rank[j];
                        const float4 a = s_a[j], b = s_b[j];
                        const float dx = gx - a.x, dy = gy - a.y;
                        const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
                        const float am = b.y * __builtin_amdgcn_exp2f(q);
                        if (r > k_cur && !looked) { looked = true; if (f == GSX_RANK_NONE && A > GSX_SAT_ALPHA) f = k_cur; }
                        A = A + (1.0f - A) * am;
                        if (f == GSX_RANK_NONE && r >= k_cur && A > GSX_SAT_ALPHA) f = r;
                    }
                }
                // (the batch went past rank k_cur without this quadrant meeting a splat behind it: A is the state AT k_cur)
                if (!leave && !looked && s_rank[cnt - 1u] > k_cur) { looked = true; if (f == GSX_RANK_NONE && A > GSX_SAT_ALPHA) f = k_cur; }
                const bool all_found = __syncthreads_and(f != GSX_RANK_NONE);      // (also the barrier before the next batch is staged)
                if (!monotone && __syncthreads_or(leave)) { quit = true; break; }
                if (all_found) break;
            }
            if (quit) break;
            if (f == GSX_RANK_NONE && !looked && A > GSX_SAT_ALPHA) f = k_cur;     // the list ended at or before rank k_cur
            mine = max(mine, f);
            if (!monotone && __syncthreads_or(f == GSX_RANK_NONE)) {               // a pixel of this tile never gets there: tell everybody now
                // A trained scene has such pixels on hundreds of tiles, and every workgroup that meets one in its first tile gets here
                // within the same few microseconds: the first of the workgroups that share a signal word goes on to the running maximum,
                // the first of those to all the copies -- 8, then 64 atomics on a line at worst, then one per copy, not 512 on every line
                // (30 us of queueing for every look the workgroups still walking took).  (Maxima, like the rounds' words below: the
                // largest value stays whichever lands last.)
                if (threadIdx.x == 0 &&
                    __hip_atomic_fetch_max(my_sig, kSigNever, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kSigNever &&
                    atomicMax(state + 2, GSX_RANK_NONE) != GSX_RANK_NONE) {
                    for (uint32_t c = 0; c < n_sig; c++)
                        __hip_atomic_fetch_max(sig_word(work, max_chunks, c), kSigNever, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                quit = true;
            }
        }
        uint32_t v = mine;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (monotone) {                      // (uniform) the first crossings' maximum is the exit rank
            if (threadIdx.x == 0) atomicMax(state, max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
            return;
        }
        if (threadIdx.x == 0) {
            // A workgroup that has seen (or raised) "never" neither adds to the maximum nor arrives: the search is over, and whoever
            // waits at the barrier leaves it on the same word.  (512 arrivals + 512 maxima on one cache line are ~65 us of
            // serialised atomics -- what was left of the search on a trained bench scene once the walks ended early.)
            uint32_t k_seen = GSX_RANK_NONE;
            if (!quit) {
                atomicMax(state + 2, max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
                // grid barrier: arrive (release: the maximum above is out; acquire: the last to arrive sees everybody's).  The last one
                // reads the round's maximum and hands it to everybody through the signal words, {round, maximum} in every copy; the
                // others wait on THEIR copy (eight workgroups to a line: a look is a microsecond, not the tens of microseconds that
                // five hundred waiting workgroups looking at one word cost each other -- and the workgroups still walking with them).
                const uint32_t target = round * n_blocks;
                if (__hip_atomic_fetch_add(bar_arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == target) {
                    const unsigned long long word = ((unsigned long long)round << 32) |
                                                    __hip_atomic_load(state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (uint32_t c = 0; c < n_sig; c++)
                        __hip_atomic_fetch_max(sig_word(work, max_chunks, c), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                for (;;) {
                    const unsigned long long w = __hip_atomic_load(my_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (w == kSigNever) break;
                    if ((uint32_t)(w >> 32) == round) { k_seen = (uint32_t)w; break; }
                    __builtin_amdgcn_s_sleep(32);            // (~1 us between looks)
                }
            }
            s_k = k_seen;
        }
        __syncthreads();
        const uint32_t k_new = s_k;
        __syncthreads();                     // s_k and s_red are rewritten next round
        result = k_new;
        if (k_new == GSX_RANK_NONE || k_new == k_cur) break;      // never / a fixpoint (every workgroup reads the same value: uniform)
        k_cur = k_new;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = result; state[3] = __hip_atomic_load(bar_arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// ------------------------------------------------------------------------------------------
// B1: backward, one wave per chunk, back to front
// ------------------------------------------------------------------------------------------
// The nine per-lane sums of one splat reach their row through LDS (round 3, second half): every lane parks its nine values in a
// staging slot -- nine `ds_write_addtid_b32` (address = M0 + offset + 4 * lane: no address register, 2 cycles each), row v of the slot
// = value v of all 64 lanes -- and after kRedK splats lane t < 2 * 9 * kRedK adds up HALF a row (8 ds_read_b128, 31 adds), one DPP add
// joins the halves, and lane (slot j, value v) stores its total straight into the instance's row.  ~15 VALU instructions per splat at
// the plain-add rate instead of the 25 cross-lane ones of `wave_reduce9` (permlane swaps and DPP adds issue at a third to a half of that
// rate: profiles/r3_valu_issue_rate.txt), and no output staging.  Rows are 64 + 4 floats apart so that the 16 lanes of a ds_read_b128
// group (two halves of eight consecutive rows) hit 16 different bank quads.  Fixed order of summation: bitwise reproducible.
#ifndef GSX_RED_K
#define GSX_RED_K 2
#endif
constexpr int kRedK = GSX_RED_K;               // splats per flush (2: 36 of the 64 lanes sum, 7.3 KB of LDS per wave; 3: 54 lanes, 9.7 KB -- fewer
                                               // instructions and one resident wave per SIMD less: 0.744 against 0.735 ms per bench step)
constexpr int kRedRow = 68;                    // floats from one row of a slot to the next
constexpr int kRedSlot = 9 * kRedRow;          // floats per slot

// row order of a slot = order of the instance's row in `partials`: S(dq dx), S(dq dy), S(dq dx dx), S(dq dx dy), S(dq dy dy), S(dq), S(w G0..2)
__device__ __forceinline__ void stage_sums(uint32_t lds_byte_base, float r0, float r1, float r2, float r3, float r4, float r5,
                                           float r6, float r7, float r8)
{
    asm volatile(
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"                                  // SALU write of M0 -> LDS add-TID instruction: one wait state
        "ds_write_addtid_b32 %0 offset:0\n\t"
        "ds_write_addtid_b32 %1 offset:272\n\t"
        "ds_write_addtid_b32 %2 offset:544\n\t"
        "ds_write_addtid_b32 %3 offset:816\n\t"
        "ds_write_addtid_b32 %4 offset:1088\n\t"
        "ds_write_addtid_b32 %5 offset:1360\n\t"
        "ds_write_addtid_b32 %6 offset:1632\n\t"
        "ds_write_addtid_b32 %7 offset:1904\n\t"
        "ds_write_addtid_b32 %8 offset:2176"
        :: "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(r8), "s"(lds_byte_base)
        : "memory");
}
static_assert(kRedRow * 4 == 272, "the offsets in stage_sums are rows of kRedRow floats");

// One slice of the image loss's partial sums (gsx_loss.hip: loss_partial_kernel, kLossBlocks slices, 256 threads each) by ONE wave,
// which plays the slice's four waves one after the other: the same elements per thread, the same order of additions, the same bits.
template <int KIND>
__device__ __forceinline__ void loss_slice_wave(const LossInBwd& lw, int slice)
{
    const long long n4 = lw.n >> 2;
    const long long per = (n4 + kLossBlocks - 1) / kLossBlocks;
    const long long lo = (long long)slice * per, hi = min(lo + per, n4);
    const float4* p4 = reinterpret_cast<const float4*>(lw.pred);
    const float4* t4 = reinterpret_cast<const float4*>(lw.target);
    double sw[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int t = w * 64 + (int)threadIdx.x;            // the slice's thread this lane plays now
        float s = 0.0f;
        for (long long q = lo + t; q < hi; q += kBlock) {
            float4 p, tg;
            if (lw.vec) { p = p4[q]; tg = t4[q]; }
            else {
                p = make_float4(lw.pred[4*q], lw.pred[4*q+1], lw.pred[4*q+2], lw.pred[4*q+3]);
                tg = make_float4(lw.target[4*q], lw.target[4*q+1], lw.target[4*q+2], lw.target[4*q+3]);
            }
            s += (loss_term<KIND>(p.x, tg.x) + loss_term<KIND>(p.y, tg.y)) + (loss_term<KIND>(p.z, tg.z) + loss_term<KIND>(p.w, tg.w));
        }
        if (slice == 0 && t < (int)(lw.n & 3)) s += loss_term<KIND>(lw.pred[n4 * 4 + t], lw.target[n4 * 4 + t]);
        double sd = (double)s;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sd += __shfl_down(sd, d, 64);
        sw[w] = sd;
    }
    if (threadIdx.x == 0) lw.partials[slice] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

__global__ __launch_bounds__(64) void chunk_bwd_kernel(
    const float4* __restrict__ rec, const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ ranges,
    const uint32_t* __restrict__ chunk_base, const uint32_t* __restrict__ offsets, int n_tiles,
    uint32_t N, int H, int W, int tw, float stepx, float stepy, const uint32_t* __restrict__ state,
    const float* __restrict__ chunk_state, const uint32_t* __restrict__ chunk_meta, const uint32_t* __restrict__ chunk_dark,
    const uint32_t* __restrict__ work, const float* __restrict__ d_rgb, float* __restrict__ partials, int prezeroed,
    uint32_t max_chunks, const float* __restrict__ mid_state, const uint8_t* __restrict__ masks, const uint32_t* __restrict__ chunk_tile,
    LossInBwd lw)
{
    // the one-call trainer step's criterion (L1 or clamped MSE against lw.target, unit upstream gradient) rides in this launch: the
    // grid's last kLossBlocks workgroups leave the loss's partial sums, and every work item forms dL/d(rgb) from the image itself
    if (lw.target && blockIdx.x >= gridDim.x - (uint32_t)kLossBlocks) {
        if (lw.kind == GSX_LOSS_L1) loss_slice_wave<GSX_LOSS_L1>(lw, (int)(blockIdx.x - (gridDim.x - (uint32_t)kLossBlocks)));
        else loss_slice_wave<GSX_LOSS_CLAMPED_MSE>(lw, (int)(blockIdx.x - (gridDim.x - (uint32_t)kLossBlocks)));
        return;
    }
    __shared__ float4 s_a[kCBatch], s_b[kCBatch];
    __shared__ float s_cx[kCBatch];
    __shared__ uint32_t s_slot[kCBatch];
    // staging slots of the row sums (above)
    __shared__ __align__(16) float s_stage[kRedK * kRedSlot];
    // work lists left by the combine pass, costliest class first, occluded chunks last;
    // the second set is the one rebuilt after a global early exit
    const uint32_t limit = rank_limit_from_state(state, N);
    const bool after_exit = limit < N;
    const uint32_t* cnt = work_counters(const_cast<uint32_t*>(work), after_exit);
    const int lane = threadIdx.x;
    // one work item (a chunk, or half a chunk of a shallow view)
    auto bwd_item = [&](uint32_t b) {
    int cls = 0;
    while (cls < kWorkClasses && b >= cnt[cls]) { b -= cnt[cls]; cls++; }       // costliest class first
    if (cls == kWorkClasses) return;
    const uint32_t item = work_list(const_cast<uint32_t*>(work), after_exit, cls, max_chunks)[b];
    const uint32_t chunk = item & kItemChunk;
    // a shallow view's chunk pass left the chunk's tile (header word) and every instance's quadrant mask (payload words)
    const bool shallow = chunk_base[n_tiles] < kDeepChunks;
    ChunkGeom g = shallow ? chunk_locate_in(chunk_base, n_tiles, chunk, ranges, tw, (int)chunk_tile[chunk])
                                                  : chunk_locate(chunk_base, n_tiles, chunk, ranges, tw);
    const uint8_t* chunk_masks = shallow ? masks + (size_t)chunk * (64 * 4) - g.begin : nullptr;
    // a chunk in two halves (shallow views: the combine pass made two entries of it): this wave walks one of them
    const bool front = (item & kItemFront) != 0u;
    if (item & (kItemFront | kItemBack)) {
        const uint32_t mid = g.begin + (chunk_base[n_tiles + 1] >> 1);
        if (front) g.end = mid; else g.begin = mid;
    }
    // instances of this chunk that were composited: ranks ascend, keep those below the exit rank
    uint32_t eff_end = g.end;
    if (limit < N) {
        uint32_t lo = g.begin, hi = g.end;
        while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (ranks[mid] < limit) lo = mid + 1; else hi = mid; }
        eff_end = lo;
    }
    if (eff_end == g.begin) return;
    // Occlusion cut (the transmittance twin of the footprint cut): if every pixel of a quadrant enters this chunk
    // with |T| < 2^-(GSX_CUT_LOG2 + g), g = the chunk's bound on the growth of |T| in bits (0 unless an opacity lies outside [0, 2]),
    // every weight T*alpha there is below 2^-24 and so is every dL/dalpha per unit of upstream gradient: the
    // quadrant is masked out for the whole chunk, and a chunk with four dark quadrants writes its rows as zeros
    // without evaluating a pair.  Half of a dense view's chunks go this way.
    // chunk_dark[chunk]: one byte per 8x8 quadrant, written by the combine pass (growth bound included)
    const uint32_t dw = chunk_dark[chunk];
    const uint32_t dark_mask = (dw & 1u) | ((dw >> 7) & 2u) | ((dw >> 14) & 4u) | ((dw >> 21) & 8u);
    {
        if (dark_mask == 0xFu) {
            if (prezeroed) return;            // the host zero-filled every row up front (deep views: gsx_composite_bwd)
            for (uint32_t i = g.begin + lane; i < eff_end; i += 64) {
                const uint32_t r = ranks[i];
                const float2 bb = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(rec) + (size_t)GSX_REC_FLOATS * r + 10);
                const uint32_t bx = __float_as_uint(bb.x), by = __float_as_uint(bb.y);
                const int tx0 = (int)(bx & 0xffff) >> GSX_TILE_SHIFT, tx1 = ((int)(bx >> 16) >> GSX_TILE_SHIFT) + 1;
                const int ty0 = (int)(by & 0xffff) >> GSX_TILE_SHIFT;
                float* row = partials + (size_t)(offsets[r] + (uint32_t)((g.ty - ty0) * (tx1 - tx0) + (g.tx - tx0))) * 9u;
#pragma unroll
                for (int c = 0; c < 9; c++) row[c] = 0.0f;
            }
            return;
        }
    }
    const int x0 = g.tx * GSX_TILE + (lane & 7), y0 = g.ty * GSX_TILE + (lane >> 3);
    float gx[2], gy[2];
    gx[0] = pix_coord(min(x0, W - 1), W, stepx); gx[1] = pix_coord(min(x0 + 8, W - 1), W, stepx);
    gy[0] = pix_coord(min(y0, H - 1), H, stepy); gy[1] = pix_coord(min(y0 + 8, H - 1), H, stepy);
    const int qx = g.tx * GSX_TILE, qy = g.ty * GSX_TILE;
    const QuadEdges QE = quad_edges(qx, qy, H, W, stepx, stepy);
    const size_t P = (size_t)H * W;
    // the summing lanes of a flush: lane t adds up half `t & 1` of row `t >> 1` (= slot tj, value tv)
    const int t_row = min(lane >> 1, 9 * kRedK - 1);
    const int tj = t_row / 9, tv = t_row - tj * 9;
    const float* t_src = s_stage + t_row * kRedRow + (lane & 1) * 32;
    const bool t_store = (lane & 1) == 0 && lane < 2 * 9 * kRedK;
    const float t_fac = (tv >= 2 && tv <= 4) ? GSX_LN2 : 1.0f;
    const uint32_t stage_base = (uint32_t)(uintptr_t)s_stage;          // byte offset of the staging area in the LDS
    float G0[4], G1[4], G2[4], Wb[4], m[4]; int e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x0 + ((k & 1) << 3), y = y0 + ((k >> 1) << 3);
        G0[k] = G1[k] = G2[k] = 0.f;
        if (x < W && y < H) {
            const size_t p = (size_t)y * W + x;
            if (lw.target) {          // what loss_partial_grad_kernel would have left in d_rgb (unit upstream gradient)
                if (lw.kind == GSX_LOSS_L1) {
                    G0[k] = loss_grad<GSX_LOSS_L1>(lw.pred[p], lw.target[p], lw.g_over_n); G1[k] = loss_grad<GSX_LOSS_L1>(lw.pred[P + p], lw.target[P + p], lw.g_over_n);
                    G2[k] = loss_grad<GSX_LOSS_L1>(lw.pred[2*P + p], lw.target[2*P + p], lw.g_over_n);
                } else {
                    G0[k] = loss_grad<GSX_LOSS_CLAMPED_MSE>(lw.pred[p], lw.target[p], lw.g_over_n); G1[k] = loss_grad<GSX_LOSS_CLAMPED_MSE>(lw.pred[P + p], lw.target[P + p], lw.g_over_n);
                    G2[k] = loss_grad<GSX_LOSS_CLAMPED_MSE>(lw.pred[2*P + p], lw.target[2*P + p], lw.g_over_n);
                }
            } else { G0[k] = d_rgb[p]; G1[k] = d_rgb[P + p]; G2[k] = d_rgb[2*P + p]; }
        }
        const int pix = k * 64 + lane;
        const float* cs = chunk_state + (size_t)chunk * GSX_STATE_WORDS * kPixPerTile + pix;
        float b0 = cs[0 * kPixPerTile], b1 = cs[1 * kPixPerTile], b2 = cs[2 * kPixPerTile];      // colour behind the chunk
        m[k] = cs[6 * kPixPerTile]; e[k] = __float_as_int(cs[7 * kPixPerTile]);                  // transmittance at its end
        if (front) {
            // the state at the chunk's MIDDLE from the second half's local result (C_b, T_b) the chunk pass left: the colour
            // behind the middle is C_b + T_b * (colour behind the chunk), the transmittance there T_end / T_b -- the quotient
            // the second half's wave reaches by dividing its (1 - alpha)s out one by one
            const float* ms = mid_state + (size_t)chunk * GSX_MID_WORDS * kPixPerTile + pix;
            const float tb = ldexpf(ms[3 * kPixPerTile], __float_as_int(ms[4 * kPixPerTile]));
            b0 = ms[0 * kPixPerTile] + tb * b0; b1 = ms[1 * kPixPerTile] + tb * b1; b2 = ms[2 * kPixPerTile] + tb * b2;
            m[k] *= __builtin_amdgcn_rcpf(ms[3 * kPixPerTile]); e[k] -= __float_as_int(ms[4 * kPixPerTile]);
            renorm(m[k], e[k]);
        }
        Wb[k] = G0[k] * b0 + G1[k] * b1 + G2[k] * b2;
    }
    // one flush: the staged splats j0, j1, j2 (the first `n` of them) get their rows
    auto flush = [&](uint32_t n, uint32_t j0, uint32_t j1, uint32_t j2) {
        const float4* src = reinterpret_cast<const float4*>(t_src);
        const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3], r4 = src[4], r5 = src[5], r6 = src[6], r7 = src[7];
        const uint32_t js = tj == 0 ? j0 : tj == 1 ? j1 : j2;
        const uint32_t slot = s_slot[js];
        const float opa = s_b[js].y;
        float acc = (((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w)))
                  + (((r2.x + r2.y) + (r2.z + r2.w)) + ((r3.x + r3.y) + (r3.z + r3.w)));
        acc += (((r4.x + r4.y) + (r4.z + r4.w)) + ((r5.x + r5.y) + (r5.z + r5.w)))
             + (((r6.x + r6.y) + (r6.z + r6.w)) + ((r7.x + r7.y) + (r7.z + r7.w)));
        // the other half of the row sits in the neighbouring lane
        acc += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
        const float out = acc * ((tv == 5) ? __builtin_amdgcn_rcpf(opa) : t_fac);          // d opacity = S(dq) / o
        if (t_store && (uint32_t)tj < n) partials[(size_t)slot * 9u + (uint32_t)tv] = out;
    };
    const uint32_t nb = (eff_end - g.begin + kCBatch - 1) / kCBatch;
    uint32_t active = 0;
    for (uint32_t bi = nb; bi-- > 0; ) {
        const uint32_t bs = g.begin + bi * kCBatch;
        const uint32_t cnt = min((uint32_t)kCBatch, eff_end - bs);
        uint32_t my_mask = 0u;
        if ((uint32_t)lane < cnt) {
            const uint32_t r = ranks[bs + lane];
            const float4* R = rec + (size_t)kRecVec * r;
            float4 c = R[2];
            const uint32_t bx = __float_as_uint(c.z), by = __float_as_uint(c.w);
            const int tx0 = (int)(bx & 0xffff) >> GSX_TILE_SHIFT, tx1 = ((int)(bx >> 16) >> GSX_TILE_SHIFT) + 1;
            const int ty0 = (int)(by & 0xffff) >> GSX_TILE_SHIFT;
            s_slot[lane] = offsets[r] + (uint32_t)((g.ty - ty0) * (tx1 - tx0) + (g.tx - tx0));
            const float4 a = R[0], b = R[1];
            my_mask = (chunk_masks ? (uint32_t)chunk_masks[bs + lane]
                                   : quad_mask(a, b, c.z, c.w, qx, qy, QE, GSX_CUT_LOG2_FOR(H, W), GSX_CUT_MIN_EXP_FOR(H, W), H < GSX_TILE || W < GSX_TILE)) & ~dark_mask;
            s_a[lane] = a; s_b[lane] = b; s_cx[lane] = c.x;
        }
        // The batch's quadrant masks as four wave-uniform bit sets (bit j of qk: instance j reaches quadrant k, dark quadrants
        // taken out): they live in scalar registers, so the loop below neither reads a mask from LDS nor waits for it before
        // it may fetch the record, and instances that reach no live quadrant (inside the tile's box but outside the ellipse,
        // or occluded) are never visited -- the epilogue writes their zero rows.
        const unsigned long long q0 = __builtin_amdgcn_ballot_w64((my_mask & 1u) != 0u), q1 = __builtin_amdgcn_ballot_w64((my_mask & 2u) != 0u);
        const unsigned long long q2 = __builtin_amdgcn_ballot_w64((my_mask & 4u) != 0u), q3 = __builtin_amdgcn_ballot_w64((my_mask & 8u) != 0u);
        const unsigned long long live = (q0 | q1) | (q2 | q3);
        __syncthreads();
        uint32_t nst = 0, j0 = 0, j1 = 0, j2 = 0;                       // staged splats of the current flush (wave-uniform)
        for (unsigned long long todo = live; todo != 0ull; ) {
            const uint32_t j = 63u - (uint32_t)__builtin_clzll(todo);       // back to front
            todo &= ~(1ull << j);
            const float4 a = s_a[j], b = s_b[j]; const float cx = s_cx[j];
            const unsigned long long bit = 1ull << j;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f, s8 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((k == 0 ? q0 : k == 1 ? q1 : k == 2 ? q2 : q3) & bit) {      // one scalar test per quadrant
                    const float dx = gx[k & 1] - a.x, dy = gy[k >> 1] - a.y;
                    const float q = a.z*dx*dx + a.w*dx*dy + b.x*dy*dy;
                    const float al = b.y * __builtin_amdgcn_exp2f(q);
                    const float f = one_minus(al);
                    m[k] *= __builtin_amdgcn_rcpf(f);                 // T_k = T_{k+1} / (1 - alpha_k)
                    const float Tv = ldexpf(m[k], e[k]);
                    const float cG = b.z * G0[k] + b.w * G1[k] + cx * G2[k];
                    const float diff = cG - Wb[k];                    // dL/dalpha_k = T_k (c_k.G - W_k)
                    const float wv = Tv * al;
                    const float dq = wv * diff;                       // dL/dq' / ln2 = alpha_k dL/dalpha_k  (alpha = o * exp2(q'))
                    // (not fma(al, diff, Wb): one instruction less, but where an opaque splat stands in front of something
                    // bright the sum Wb + (cG - Wb) cancels -- fixture exit8's worst element went from 6e-4 to 3.4e-3 of the oracle)
                    Wb[k] = al * cG + f * Wb[k];
                    s0 += dq; s1 += dq * dx; s2 += dq * dy;
                    s3 += dq * dx * dx; s4 += dq * dx * dy; s5 += dq * dy * dy;
                    s6 += wv * G0[k]; s7 += wv * G1[k]; s8 += wv * G2[k];
                }
            }
            if ((++active & 3u) == 0u) {                       // every 4 composited splats: |m| stays within 2^+-120
#pragma unroll
                for (int k = 0; k < 4; k++) renorm(m[k], e[k]);
            }
            stage_sums((uint32_t)__builtin_amdgcn_readfirstlane((int)(stage_base + nst * (uint32_t)(kRedSlot * 4))), s1, s2, s3, s4, s5, s0, s6, s7, s8);
            if (nst == 0) j0 = j; else if (nst == 1) j1 = j; else j2 = j;
            if (++nst == (uint32_t)kRedK) { flush(kRedK, j0, j1, j2); nst = 0; }
        }
        if (nst) flush(nst, j0, j1, j2);
        // instances the loop never visited (nothing of them reaches a live quadrant): zero rows
        const unsigned long long unvisited = (cnt == 64u ? ~0ull : ((1ull << cnt) - 1ull)) & ~live;
        if (unvisited != 0ull) {
            for (uint32_t idx = lane; idx < cnt * 9u; idx += 64) {
                const uint32_t j = idx / 9u, c = idx - j * 9u;
                if ((unvisited >> j) & 1ull) partials[(size_t)s_slot[j] * 9u + c] = 0.0f;
            }
        }
        __syncthreads();
    }
    };
    bwd_item(blockIdx.x);
}

}  // namespace gsx

using namespace gsx;

static inline float lin_step(int steps) { return steps > 1 ? (1.0f - (-1.0f)) / (float)(steps - 1) : 0.0f; }

GSX_API int gsx_composite_fwd(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                              const uint32_t* chunk_base, int32_t N, int32_t H, int32_t W,
                              int64_t max_chunks, float* chunk_state, int32_t need_bwd, uint32_t* state,
                              float* rgb, float* alpha, float* depth_map, void* stream)
{
    return gsx::composite_fwd_impl(rec, ranks_sorted, ranges, chunk_base, N, H, W, max_chunks, chunk_state, need_bwd, state, rgb, alpha, depth_map,
                                   3, stream);
}

int gsx::composite_fwd_impl(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                            const uint32_t* chunk_base, int32_t N, int32_t H, int32_t W,
                            int64_t max_chunks, float* chunk_state, int32_t need_bwd, uint32_t* state,
                            float* rgb, float* alpha, float* depth_map, int which, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || max_chunks < 0) return GSX_E_INVALID;
    if (!ranges || !chunk_base || !state || !rgb || !alpha) return GSX_E_INVALID;
    // an empty model (N == 0) has no records: every tile's list is empty, the chunk table says 0 chunks, nothing reads `rec`
    if (max_chunks > 0 && ((N > 0 && !rec) || !ranks_sorted || !chunk_state)) return GSX_E_INVALID;
    if (max_chunks > (int64_t)kItemChunk) return GSX_E_RANGE;            // (the work lists' entries keep two bits for the halves)
    if ((uintptr_t)rec & 15u) return GSX_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE, n_tiles = tw * th;
    const float sx = lin_step(W), sy = lin_step(H);
    const float4* r4 = (const float4*)rec;
    uint32_t* meta = reinterpret_cast<uint32_t*>(chunk_state + (size_t)max_chunks * GSX_STATE_WORDS * kPixPerTile);
    uint32_t* work = meta + 2 * (size_t)max_chunks;          // 16 counters, 2 x kWorkClasses lists of 2 x max_chunks entries
    float* mid_state = reinterpret_cast<float*>(meta + tail_mid_offset((size_t)max_chunks));      // shallow views: second-half results per chunk
    const bool could_be_deep = (uint64_t)max_chunks >= kDeepChunks;
    // first dark level per (tile, quadrant) of a deep view (entry_walk): in the work lists' room, which the combine pass only
    // fills after the chunk pass (none if the image has more tiles than that holds: then nothing is skipped)
    uint32_t* tile_dark = (could_be_deep && 4 * (size_t)n_tiles <= 4 * (size_t)kWorkClasses * (size_t)max_chunks) ? work + 16 : nullptr;
    // the shallow instantiation only has work below kDeepChunks chunks: its grid need not be larger
    const dim3 gc((unsigned)max_chunks), gt((unsigned)n_tiles);
    // the four-wave chunk pass: one block per chunk, then the blocks that list the non-empty tiles and clear the empty ones' pixels
    const uint32_t n_list = (uint32_t)((n_tiles + kListTiles - 1) / kListTiles);
    const dim3 gc0l((unsigned)(could_be_deep ? kDeepChunks - 1 : max_chunks) + n_list);
    uint32_t* hdr = meta + tail_hdr_offset((size_t)max_chunks);                             // published headers, level-order table, payloads
    uint32_t* order_tab = hdr + max_chunks;
    uint32_t* pay = order_tab + max_chunks;
    // shallow views' combine pass: one workgroup per non-empty tile up to kParGrid of them, then a stride
    const dim3 gp((unsigned)std::min<int64_t>(std::min<int64_t>(n_tiles, max_chunks > 0 ? max_chunks : 1), kParGrid));
    if (which & 1) {
    // (block 0 of the four-wave chunk pass clears state[] and the work counters; with no chunks at all its grid is the list blocks alone)
    if (could_be_deep) {                               // (max_chunks bounds the view's chunk count: below the threshold no view can be deep)
        // the level-order table (its kernel also clears the payloads and the tiles' words), then the header pass: four
        // launches that return at once unless the view at hand is deep.  The grower bitmap (2 words per 64 ranks + the
        // count) borrows the head of the chunk states, which nothing writes before the chunk pass; a model too large
        // even for that gets its headers from the records.
        chunk_order_kernel<<<dim3(kOrderLevels), dim3(kOrderThreads), 0, st>>>(chunk_base, n_tiles, (uint32_t)max_chunks, hdr, pay, order_tab, tile_dark);
        const size_t bm_words = ((size_t)N + 63) / 64 * 2;
        uint32_t* bitmap = (N > 0 && bm_words + 1 <= (size_t)max_chunks * GSX_STATE_WORDS * kPixPerTile) ? reinterpret_cast<uint32_t*>(chunk_state) : nullptr;
        if (bitmap) {
            GSX_HIP(hipMemsetAsync(bitmap + bm_words, 0, sizeof(uint32_t), st));
            grower_bitmap_kernel<<<dim3((unsigned)(((size_t)N + kBlock - 1) / kBlock)), dim3(kBlock), 0, st>>>(r4, (uint32_t)N, chunk_base, n_tiles, (uint32_t)max_chunks, bitmap, bitmap + bm_words);
        }
        tile_growth_kernel<<<dim3(kHeaderBlocks), dim3(kBlock), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, tw, (uint32_t)max_chunks,
                                                                      bitmap, bitmap ? bitmap + bm_words : nullptr, hdr);
    }
    // pass 0: everything; F3: exact exit rank if every pixel saturated; pass 1: re-composite up to it
    if (depth_map) {
        chunk_fwd_wg_kernel<true><<<gc0l, dim3(kBlock), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, H, W, tw, sx, sy, state, chunk_state, meta, work, need_bwd ? mid_state : nullptr, n_list, order_tab, rgb, alpha, depth_map, reinterpret_cast<uint8_t*>(pay), need_bwd ? hdr : nullptr);
        if (could_be_deep) chunk_fwd_kernel<false, true, true><<<gc, dim3(64), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, (uint32_t)N, H, W, tw, sx, sy, state, chunk_state, meta, work, hdr, pay, order_tab, tile_dark);
        tile_combine_par_kernel<true><<<gp, dim3(kParGroups * kBlock), 0, st>>>(chunk_base, H, W, tw, need_bwd, state, chunk_state, meta, meta + max_chunks, work, (uint32_t)max_chunks, n_tiles, rgb, alpha, depth_map, ranges, order_tab);
        if (could_be_deep) tile_combine_kernel<false, true, true><<<gt, dim3(kBlock), 0, st>>>(chunk_base, (uint32_t)N, H, W, tw, need_bwd, state, chunk_state, meta, meta + max_chunks, work, hdr, pay, (uint32_t)max_chunks, n_tiles, rgb, alpha, depth_map, ranges, 0);
    } else {
        chunk_fwd_wg_kernel<false><<<gc0l, dim3(kBlock), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, H, W, tw, sx, sy, state, chunk_state, meta, work, need_bwd ? mid_state : nullptr, n_list, order_tab, rgb, alpha, nullptr, reinterpret_cast<uint8_t*>(pay), need_bwd ? hdr : nullptr);
        if (could_be_deep) chunk_fwd_kernel<false, false, true><<<gc, dim3(64), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, (uint32_t)N, H, W, tw, sx, sy, state, chunk_state, meta, work, hdr, pay, order_tab, tile_dark);
        tile_combine_par_kernel<false><<<gp, dim3(kParGroups * kBlock), 0, st>>>(chunk_base, H, W, tw, need_bwd, state, chunk_state, meta, meta + max_chunks, work, (uint32_t)max_chunks, n_tiles, rgb, alpha, depth_map, ranges, order_tab);
        if (could_be_deep) tile_combine_kernel<false, false, true><<<gt, dim3(kBlock), 0, st>>>(chunk_base, (uint32_t)N, H, W, tw, need_bwd, state, chunk_state, meta, meta + max_chunks, work, hdr, pay, (uint32_t)max_chunks, n_tiles, rgb, alpha, depth_map, ranges, 0);
    }
    }
    if (!(which & 2)) { GSX_LAUNCH_CHECK(); return GSX_OK; }
    // the three launches below only have work on a saturated view (the reference's global exit): small grids, grid stride
    const dim3 gc1((unsigned)(max_chunks < 2048 ? max_chunks : 2048)), gt1((unsigned)(n_tiles < 1024 ? n_tiles : 1024));
    saturation_rank_kernel<<<gt1, dim3(kBlock), 0, st>>>(r4, ranks_sorted, ranges, H, W, tw, n_tiles, sx, sy, state, chunk_base,
                                                         max_chunks > 0 ? reinterpret_cast<const uint8_t*>(pay) : nullptr, work,
                                                         max_chunks > 0 ? order_tab : nullptr, (uint32_t)max_chunks);
    if (max_chunks > 0) {
        if (depth_map) {
            chunk_fwd_kernel<true, true, false><<<gc1, dim3(64), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, (uint32_t)N, H, W, tw, sx, sy, state, chunk_state, meta, work, hdr, pay, order_tab, nullptr);
            tile_combine_kernel<true, true, true><<<gt1, dim3(kBlock), 0, st>>>(chunk_base, (uint32_t)N, H, W, tw, need_bwd, state, chunk_state, meta, meta + max_chunks, work, hdr, pay, (uint32_t)max_chunks, n_tiles, rgb, alpha, depth_map, ranges, 1);
        } else {
            chunk_fwd_kernel<true, false, false><<<gc1, dim3(64), 0, st>>>(r4, ranks_sorted, ranges, chunk_base, n_tiles, (uint32_t)N, H, W, tw, sx, sy, state, chunk_state, meta, work, hdr, pay, order_tab, nullptr);
            tile_combine_kernel<true, false, true><<<gt1, dim3(kBlock), 0, st>>>(chunk_base, (uint32_t)N, H, W, tw, need_bwd, state, chunk_state, meta, meta + max_chunks, work, hdr, pay, (uint32_t)max_chunks, n_tiles, rgb, alpha, depth_map, ranges, 1);
        }
    }
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int gsx_composite_bwd(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                              const uint32_t* chunk_base, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                              int64_t max_chunks, const float* chunk_state, const uint32_t* state,
                              const float* d_rgb, float* partials, int64_t n_rows, void* stream)
{
    return gsx::composite_bwd_impl(rec, ranks_sorted, ranges, chunk_base, offsets, N, H, W, max_chunks, chunk_state, state, d_rgb, partials, n_rows,
                                   nullptr, stream);
}

// lw (nullable; N > 0 and max_chunks > 0): the criterion inside the launch (LossInBwd, gsx_common.h); d_rgb is then not read
int gsx::composite_bwd_impl(const float* rec, const uint32_t* ranks_sorted, const uint32_t* ranges,
                            const uint32_t* chunk_base, const uint32_t* offsets, int32_t N, int32_t H, int32_t W,
                            int64_t max_chunks, const float* chunk_state, const uint32_t* state,
                            const float* d_rgb, float* partials, int64_t n_rows, const LossInBwd* lw, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || max_chunks < 0) return GSX_E_INVALID;
    if (lw && (N == 0 || max_chunks == 0 || !lw->pred || !lw->target || !lw->partials || lw->n <= 0)) return GSX_E_INVALID;
    if (N == 0 || max_chunks == 0) return GSX_OK;
    if (!rec || !ranks_sorted || !ranges || !chunk_base || !offsets || !chunk_state || !state || (!d_rgb && !lw) || !partials) return GSX_E_INVALID;
    if (max_chunks > (int64_t)kItemChunk) return GSX_E_RANGE;
    if ((uintptr_t)rec & 15u) return GSX_E_INVALID;
    if (n_rows < 0) return GSX_E_INVALID;
    const int tw = (W + GSX_TILE - 1) / GSX_TILE, th = (H + GSX_TILE - 1) / GSX_TILE;
    // A view of tens of millions of instances is mostly occluded chunks, whose rows are zeros: filling all rows in one
    // streaming pass and letting those chunks return at once beats having each of them find its rows' places (two gathers
    // per instance) and write 36 scattered bytes there (config 5, 1.9 G instances: backward 147 -> see DESIGN).  A small
    // view has few such chunks and the fill would only cost.
    const int prezeroed = n_rows >= GSX_PREZERO_ROWS ? 1 : 0;
    if (prezeroed) GSX_HIP(hipMemsetAsync(partials, 0, (size_t)n_rows * 9 * sizeof(float), (hipStream_t)stream));
    const uint32_t* meta = reinterpret_cast<const uint32_t*>(chunk_state + (size_t)max_chunks * GSX_STATE_WORDS * kPixPerTile);
    // one workgroup per work item: a chunk, or -- in a shallow view -- half a chunk (at most kDeepChunks - 1 chunks are cut)
    const int64_t grid = max_chunks + (max_chunks < (int64_t)kDeepChunks ? max_chunks : (int64_t)kDeepChunks);
    LossInBwd none = {};
    chunk_bwd_kernel<<<dim3((unsigned)(grid + (lw ? kLossBlocks : 0))), dim3(64), 0, (hipStream_t)stream>>>(
        (const float4*)rec, ranks_sorted, ranges, chunk_base, offsets, tw * th, (uint32_t)N, H, W, tw,
        lin_step(W), lin_step(H), state, chunk_state, meta, meta + max_chunks, meta + 2 * (size_t)max_chunks,
        d_rgb, partials, prezeroed, (uint32_t)max_chunks, reinterpret_cast<const float*>(meta + tail_mid_offset((size_t)max_chunks)),
        reinterpret_cast<const uint8_t*>(meta + tail_pay_offset((size_t)max_chunks)), meta + tail_hdr_offset((size_t)max_chunks),
        lw ? *lw : none);
    GSX_LAUNCH_CHECK();
    return GSX_OK;
}

GSX_API int64_t gsx_chunk_state_floats(int64_t max_chunks)
{
    if (max_chunks < 0 || max_chunks > (int64_t)kItemChunk) return GSX_E_RANGE;
    return (int64_t)GSX_CHUNK_STATE_FLOATS(max_chunks > 0 ? max_chunks : 1);
}

GSX_API const char* gsx_strerror(int status)
{
    switch (status) {
        case GSX_OK: return "ok";
        case GSX_E_INVALID: return "invalid argument";
        case GSX_E_CAPACITY: return "instance buffers or workspace too small";
        case GSX_E_HIP: return "HIP runtime error";
        case GSX_E_RANGE: return "problem exceeds 32-bit instance indexing";
        case GSX_E_COMM: return "RCCL is not available or one of its calls failed";
        case GSX_E_AGAIN: return "optimistic assumption did not hold: call again as documented";
        default: return "unknown status";
    }
}
GSX_API int gsx_last_hip_error(void) { return gsx::g_last_hip_error; }
GSX_API int gsx_abi_version(void) { return GSX_ABI_VERSION; }
