// Ranking: segmented (one segment per query) stable LSD radix sort of
// (desc_key(score), id), plus rank-of-labelled-ids counting.
//
// Replaces `ranks = np.argsort(-scores, axis=0)` (mdir/components/optim/score/
// cirscore.py:70) and the `np.in1d` position lookups of compute_map
// (mdir/external/cirtorch/utils/evaluate.py:80-81).
//
// All of it is HBM-bound integer work: 4 passes of 8 bits; per pass a per-tile
// histogram, a per-query scan, and a stable scatter.  Stability (equal keys keep
// ascending id) is what fixes the tie order documented in include/mdx.h.
#include <atomic>

#include "mdx_common.h"

#ifndef MDX_SORT_NT
#define MDX_SORT_NT 1     // the final ranking is stored non-temporally (0.859 -> 0.846 ms at 70 x 1 M; -DMDX_SORT_NT=0: plain)
#endif

namespace mdx {

#ifndef MDX_SORT_ITEMS
#define MDX_SORT_ITEMS 8
#endif
constexpr int SORT_ITEMS = MDX_SORT_ITEMS;     // elements per lane (8 x 512 threads = 4096-element tiles measured best)
#ifndef MDX_SORT_WAVES
#define MDX_SORT_WAVES 8
#endif
constexpr int SORT_WAVES = MDX_SORT_WAVES;     // waves per workgroup
constexpr int SORT_THREADS = 64 * SORT_WAVES;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;   // elements per workgroup
constexpr int SUB_TILE = 64 * SORT_ITEMS;      // elements per wave
constexpr int RADIX = 256;

// ascending bitonic sort of buf[0..P) (P a power of two) by all threads of the workgroup
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *buf, int P, int tid, int nthreads)
{
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += nthreads) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const uint64_t a = buf[i], b = buf[j];
                if ((a > b) == ((i & size) == 0)) {
                    buf[i] = b;
                    buf[j] = a;
                }
            }
            __syncthreads();
        }
}

// What a pass reads and writes.  A pass only needs the key bits it has not consumed yet, and ids of a database of
// n <= 2^24 rows fit 24 bits, so between passes an element is NOT (key word, id word) but
//      w = (key bits 24..31 : id)   one word, the same in every packed format, and
//      h = the middle key bits still to come: bits 8..23 (FMT_A, a u16) after pass 0, bits 16..23 (FMT_B, a u8)
//          after pass 1, nothing (FMT_C) after pass 2.
// Per element the four passes then move 4+6, 6+5, 5+4, 4+8 bytes instead of 4+8, 8+8, 8+8, 8+8 and their
// histograms read 4, 2, 1, 4 bytes instead of 4 x 4: 53 bytes instead of 76.  FMT_KV (key word, id word) is the
// layout for n > 2^24.  Rows of the intermediate arrays start at multiples of `stride` (a multiple of 256 elements),
// so that their wide loads are aligned; the scores keep their own row length n.
enum { FMT_SCORES = 0, FMT_KV = 1, FMT_A = 2, FMT_B = 3, FMT_C = 4, FMT_RANKS = 5 };

// Scores that arrive as column blocks (the peer blocks of the multi-GPU exchange: block g is a [nq, width_g] row-major
// matrix, row q of the problem = its rows q side by side): pass 0 reads them where they lie instead of a re-blocked copy.
constexpr int MAX_SEG = 32;
// The table travels in the kernel arguments (536 bytes).  Launches that do not read column blocks -- passes 1-3 of every
// ranking, all passes of a dense one -- take the empty NoSeg instead: twelve 600-byte argument blocks per ranking cost ~20 us.
struct NoSeg { static constexpr bool HAS = false; };
struct SegTable {
    static constexpr bool HAS = true;
    const float *p[MAX_SEG];
    int64_t start[MAX_SEG + 1];     // first column of block g; start[nseg] = n
    int nseg;                       // 0: plain [nq, n] scores
    uint32_t inv_w;                 // floor(2^32 / ceil(n / nseg)): where equally wide blocks would put a column (seg_find's guess)
};

// The table lives in the kernel arguments: every look into it is a scalar load that the tile's data loads wait for, and a
// workgroup that walked it from block 0 had four or five such round trips in a row at the start of a 3-10 us life (9-query
// ranking at G = 8, tools/g8_budget.py: 163-194 us over 8-24 peer blocks against 140 dense; per kernel hist<0> 14 against 8 us,
// scatter<0> 33 against 23).  The peer blocks of an exchange are equally wide to within a row, so the block is GUESSED from the
// column (where equal widths would put it) and its pointer and bounds are fetched together: one round trip when the guess
// holds; any widths stay correct, a wrong guess walks from there.
struct SegHit { const float *p; int64_t s0, s1; int g; };

__device__ __forceinline__ SegHit seg_find(const SegTable &t, int64_t i)
{
    int g = (int)__umulhi((uint32_t)i, t.inv_w);
    if (g > t.nseg - 1) g = t.nseg - 1;
    SegHit h = {t.p[g], t.start[g], t.start[g + 1], g};
    while (i >= h.s1 && g + 1 < t.nseg) { ++g; h.p = t.p[g]; h.s0 = h.s1; h.s1 = t.start[g + 1]; }
    while (i < h.s0 && g > 0) { --g; h.p = t.p[g]; h.s1 = h.s0; h.s0 = t.start[g]; }
    h.g = g;
    return h;
}

// A tile that ends in the NEXT block (every block of an exchange is many tiles wide, so a tile meets at most one block
// boundary): row q of both blocks as pointers that take the element's index in the whole row, and the boundary.  Round 6: such
// a tile went element by element through seg_elem before -- a table look-up per element and lane, five to eight times the life
// of a plain tile, and with 16 peer blocks every sixteenth workgroup was one (tools/seg_probe.py).
struct SegPair { const uint32_t *a, *b; int64_t cut; bool ok; };

__device__ __forceinline__ SegPair seg_pair(const SegTable &t, const SegHit &h, int64_t q, int64_t tile_end)
{
    SegPair r = {nullptr, nullptr, h.s1, false};
    if (h.g + 1 >= t.nseg) return r;
    const int64_t n0 = t.start[h.g + 1], n1 = t.start[h.g + 2];
    if (tile_end > n1) return r;            // narrower blocks than a tile: the general path
    r.a = (const uint32_t *)(h.p + q * (h.s1 - h.s0) - h.s0);
    r.b = (const uint32_t *)(t.p[h.g + 1] + q * (n1 - n0) - n0);
    r.ok = true;
    return r;
}

// element i of row q
__device__ __forceinline__ const float *seg_elem(const SegTable &t, int64_t q, int64_t i)
{
    const SegHit h = seg_find(t, i);
    return h.p + q * (h.s1 - h.s0) + (i - h.s0);
}

// per-tile digit histogram -> block_hist[q][b][digit].  A histogram does not care which lane counts which
// element, so a lane takes consecutive elements with one wide load: 4 scores / key words (16 bytes,
// dword-aligned for the scores: rows of an odd length start anywhere), or 8 of the u16 / u8 middle parts.
// It matters for pass 0, whose input comes from HBM (112 -> ~60 us).
//
// LDS sub-histograms are selected by LANE (not by wave): when a digit is concentrated in a few
// values (the top byte of cosine scores: sign + 7 exponent bits) at most 64/HIST_COPIES lanes of
// one ds_add hit the same address.  Row stride RADIX+1 keeps the copies in different banks.
constexpr int HIST_COPIES = 8;

template <int SRC, typename SEGT = NoSeg>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const float *__restrict__ scores,
                                                                 const void *__restrict__ src, int64_t n,
                                                                 int64_t stride, int nblk, int shift,
                                                                 uint32_t *__restrict__ block_hist, SEGT seg)
{
    static_assert(SORT_ITEMS == 8, "a lane takes 8 elements of a tile");
    constexpr bool FIRST = SRC == FMT_SCORES;
    __shared__ uint32_t h[HIST_COPIES][RADIX + 1];
    const int tid = threadIdx.x, copy = tid & (HIST_COPIES - 1);
    // pass 0 reads the scores newest rows first (the similarity kernel has just written them: the tail of its
    // output is still in the Infinity Cache) and leaves the head cached for the scatter, which then runs forward
    const int64_t q = FIRST ? (int64_t)gridDim.y - 1 - blockIdx.y : (int64_t)blockIdx.y;
    const int64_t b0 = FIRST ? (int64_t)gridDim.x - 1 - blockIdx.x : (int64_t)blockIdx.x;
    for (int e = tid; e < HIST_COPIES * (RADIX + 1); e += SORT_THREADS) (&h[0][0])[e] = 0;
    uint32_t d[SORT_ITEMS];
    int64_t first[2];           // element index of d[0] and of d[4]
    if (SRC == FMT_A || SRC == FMT_B) {
        const int64_t i = b0 * SORT_TILE + tid * 8;
        first[0] = i;
        first[1] = i + 4;
        if (SRC == FMT_A) {
            typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
            u16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
            if (i < n) w = *(const u16x8 *)((const uint16_t *)src + q * stride + i);      // rows are padded to the stride
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = w[j] & 255u;
        } else {
            typedef uint8_t u8x8 __attribute__((ext_vector_type(8)));
            u8x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
            if (i < n) w = *(const u8x8 *)((const uint8_t *)src + q * stride + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = w[j];
        }
    } else {
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t *p = FIRST ? (const uint32_t *)scores + q * n : (const uint32_t *)src + q * stride;
        u32x4u w[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            first[v] = b0 * SORT_TILE + (v * SORT_THREADS + tid) * 4;
            w[v] = u32x4u{0u, 0u, 0u, 0u};
        }
        bool straddles = false;                 // (uniform) the tile lies in two column blocks
        const int64_t tile_end = (b0 + 1) * SORT_TILE < n ? (b0 + 1) * SORT_TILE : n;
        SegHit h = {nullptr, 0, 0, 0};
        if constexpr (FIRST && SEGT::HAS) {
            h = seg_find(seg, b0 * SORT_TILE);
            straddles = tile_end > h.s1;
            p = (const uint32_t *)(h.p + q * (h.s1 - h.s0) - h.s0);
        }
        if (straddles) {
            if constexpr (SEGT::HAS) {
                const SegPair pr = seg_pair(seg, h, q, tile_end);
                if (pr.ok) {
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const int64_t i0 = first[v];
                        if (i0 + 4 <= pr.cut) w[v] = *(const u32x4u *)(pr.a + i0);
                        else if (i0 >= pr.cut && i0 + 4 <= tile_end) w[v] = *(const u32x4u *)(pr.b + i0);
                        else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (i0 + j < tile_end) w[v][j] = (i0 + j < pr.cut ? pr.a : pr.b)[i0 + j];
                        }
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (first[v] + j < n) w[v][j] = *(const uint32_t *)seg_elem(seg, q, first[v] + j);
                }
            }
        } else if ((b0 + 1) * SORT_TILE <= n) {        // whole tile (uniform branch): both loads in flight together
#pragma unroll
            for (int v = 0; v < 2; ++v) w[v] = *(const u32x4u *)(p + first[v]);
        } else {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (first[v] + j < n) w[v][j] = p[first[v] + j];
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[v * 4 + j] = ((FIRST ? desc_key(__uint_as_float(w[v][j])) : w[v][j]) >> shift) & 255u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r)
        if (first[r / 4] + (r % 4) < n) atomicAdd(&h[copy][d[r]], 1u);
    __syncthreads();
    for (int e = tid; e < RADIX; e += SORT_THREADS) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < HIST_COPIES; ++w) tot += h[w][e];
        block_hist[(q * nblk + b0) * RADIX + e] = tot;
    }
}

// per query and per slice of 32 digits: exclusive prefix over tiles for each digit (in place)
// + the per-query digit totals (the scatter kernel turns them into digit bases in the same
// single-wave scan it already runs).  1024 threads = 32 groups x 32 digits; group g scans 1/32 of
// the tiles (sum pass, then prefix pass -- block_hist is L2-resident), so the serial chain is
// nblk/32 long and 8 workgroups per query run in parallel: this kernel is pure latency, and with
// one workgroup per query it was a quarter of the sort at small nq.
constexpr int SCAN_GROUPS = 32, SCAN_DIGITS = 32;

__global__ __launch_bounds__(SCAN_GROUPS * SCAN_DIGITS) void sort_scan_kernel(uint32_t *__restrict__ block_hist,
                                                                              int nblk,
                                                                              uint32_t *__restrict__ digit_tot)
{
    __shared__ uint32_t part[SCAN_GROUPS][SCAN_DIGITS + 1];
    const int dl = threadIdx.x & (SCAN_DIGITS - 1), g = threadIdx.x / SCAN_DIGITS;
    const int d = blockIdx.x * SCAN_DIGITS + dl;
    const int64_t q = blockIdx.y;
    const int per = (nblk + SCAN_GROUPS - 1) / SCAN_GROUPS;
    const int b0 = g * per < nblk ? g * per : nblk, b1 = (b0 + per) < nblk ? (b0 + per) : nblk;
    uint32_t *p = block_hist + q * nblk * RADIX + d;
    uint32_t sum = 0;
    // a chain of at most 8 tiles (n <= 1 M: 246 tiles / 32 groups) stays in registers between the two passes: one round trip to
    // the table instead of two in a kernel that is nothing but latency
    constexpr int KEEP = 8;
    uint32_t kept[KEEP];
    const bool keep = per <= KEEP;          // uniform
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            kept[u] = (b0 + u) < b1 ? p[(int64_t)(b0 + u) * RADIX] : 0u;
            sum += kept[u];
        }
    } else {
        for (int b = b0; b < b1; ++b) sum += p[(int64_t)b * RADIX];
    }
    part[g][dl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < g; ++k) run += part[k][dl];
    if (g == SCAN_GROUPS - 1) digit_tot[q * RADIX + d] = run + sum;
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            if ((b0 + u) < b1) {
                p[(int64_t)(b0 + u) * RADIX] = run;
                run += kept[u];
            }
        return;
    }
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = p[(int64_t)(b + u) * RADIX];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            p[(int64_t)(b + u) * RADIX] = run;
            run += c[u];
        }
    }
    for (; b < b1; ++b) {
        const uint32_t c = p[(int64_t)b * RADIX];
        p[(int64_t)b * RADIX] = run;
        run += c;
    }
}

// stable scatter of one tile.  Element order inside a query = (tile, wave, round,
// lane); each wave ranks its 64 elements of a round with ballots (lanes with the
// same digit, lower lane first) and keeps a running per-digit count in LDS.  The
// tile is then put in digit order in LDS, so that consecutive lanes write
// consecutive global addresses inside each digit run (coalesced scatter).
// IN / OUT: the formats above (w_* = key words or packed words, h_* = id words or middle key bits).
//
// ARANK: the rank of an element among the earlier elements of its wave with the same digit is what `ds_add_rtn`
// on the wave's own counter returns, PROVIDED the LDS serves the lanes of one instruction that hit the same
// address in ascending lane order.  gfx950 does (a priority encoder per bank), but no manual promises it, so the
// library asks the device once (lds_order_probe_kernel below) and otherwise ranks with the eight ballots of the
// match-any form: ~45 vector instructions per round of 64 elements instead of one LDS instruction, and the
// scatter is bound by exactly that instruction count.
template <int IN, int OUT, bool ARANK, typename SEGT = NoSeg>
__global__ __launch_bounds__(SORT_THREADS, (SORT_WAVES >= 16 ? 8 : (3 * SORT_WAVES) / 4)) void sort_scatter_kernel(
    const float *__restrict__ scores, const uint32_t *__restrict__ w_in, const void *__restrict__ h_in,
    uint32_t *__restrict__ w_out, void *__restrict__ h_out, int64_t *__restrict__ ranks,
    float *__restrict__ top_scores, int64_t n, int64_t stride, int nblk, int shift,
    const uint32_t *__restrict__ block_hist, const uint32_t *__restrict__ digit_tot, int64_t id_offset,
    int64_t klimit, SEGT seg)
{
    constexpr bool FIRST = IN == FMT_SCORES, LAST = OUT == FMT_RANKS;
    constexpr bool VAL_IN_KEY = IN == FMT_C;        // the id is the low 24 bits of the word the digit comes from
    __shared__ uint32_t dtot[RADIX];        // per-query digit totals -> digit bases (scanned below)
    __shared__ uint32_t wcnt[SORT_WAVES][RADIX];     // per-wave digit counts, then tile-local offsets
    __shared__ uint32_t gdelta[RADIX];      // global position of a digit run minus its tile offset
    __shared__ uint32_t scan[RADIX];
    __shared__ uint32_t skey[SORT_TILE];
    __shared__ uint32_t sval[SORT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = FIRST ? (int64_t)blockIdx.y : (int64_t)gridDim.y - 1 - blockIdx.y;
    // Neighbouring tiles end their digit runs in the same cache lines.  Workgroups are dealt
    // round-robin over the 8 XCDs (speed only, never correctness), so give each XCD a
    // contiguous range of tiles: partial lines then meet in one L2 instead of two.
    int64_t b = blockIdx.x;
#ifndef MDX_SORT_NO_XCD
    {
        const int qn = nblk / 8, rn = nblk % 8, x = (int)(blockIdx.x % 8), k = (int)(blockIdx.x / 8);
        b = (x < rn ? (int64_t)x * (qn + 1) : (int64_t)rn * (qn + 1) + (int64_t)(x - rn) * qn) + k;
    }
#endif
    if (!FIRST) b = (int64_t)nblk - 1 - b;  // the histogram pass just streamed the keys forward: the tail is still in the Infinity Cache
    for (int e = tid; e < SORT_WAVES * RADIX; e += SORT_THREADS) (&wcnt[0][0])[e] = 0;
    __syncthreads();

    // Everything per element is 32-bit and tile-local: row and tile starts are folded into uniform (scalar) pointers.
    // The kernel is bound by its VALU instruction count (measured: ~130 of ~200 us per pass do not depend on the
    // bytes moved), so 64-bit per-lane address arithmetic and per-element range checks are what there is to save.
    const int64_t base = q * n;                 // row of the scores
    const int64_t row = q * stride;             // row of the intermediate arrays
    const int64_t tile0 = b * SORT_TILE;
    const int tile_n = (int)((n - tile0) < SORT_TILE ? (n - tile0) : SORT_TILE);
    const int sub = wave * SUB_TILE + lane;     // this lane's element of round 0, tile-local
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], pos[SORT_ITEMS];
    // global start of every digit run of this tile + the query's digit totals: requested first,
    // parked in LDS once the key loads are in flight (no register held across the ranking)
    uint32_t gbase = 0, qtot = 0;
    if (tid < RADIX) {
        gbase = block_hist[(q * nblk + b) * RADIX + tid];
        qtot = digit_tot[q * RADIX + tid];
    }
    if (FIRST) {
        // pass 0 reads the scores from HBM: 16-byte loads (a quarter of the memory requests: 317 -> 243 us); the keys
        // go through LDS (skey is free until the tile is staged) to reach the (wave, round, lane) order the ranking
        // is defined on.  Later passes find their input in the caches, where the same detour costs 6-30 us per pass.
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t *src = (const uint32_t *)scores + base + tile0;
        bool straddles = false;         // (uniform) the tile lies in two column blocks of segmented scores
        SegHit sh = {nullptr, 0, 0, 0};
        if constexpr (SEGT::HAS) {
            sh = seg_find(seg, tile0);
            straddles = tile0 + tile_n > sh.s1;
            src = (const uint32_t *)(sh.p + q * (sh.s1 - sh.s0) + (tile0 - sh.s0));
        }
        if (straddles) {
            if constexpr (SEGT::HAS) {
                const SegPair pr = seg_pair(seg, sh, q, tile0 + tile_n);
                if (pr.ok) {
                    const uint32_t *pa = pr.a + tile0, *pb = pr.b + tile0;
                    const int cut = (int)(pr.cut - tile0);
                    for (int e = tid; e < tile_n; e += SORT_THREADS) skey[e] = desc_key(__uint_as_float((e < cut ? pa : pb)[e]));
                } else {
                    for (int e = tid; e < tile_n; e += SORT_THREADS) skey[e] = desc_key(*seg_elem(seg, q, tile0 + e));
                }
            }
        } else if (tile_n == SORT_TILE) {      // whole tile (uniform branch): both loads in flight together
            u32x4u w[SORT_ITEMS / 4];
#pragma unroll
            for (int v = 0; v < SORT_ITEMS / 4; ++v) w[v] = *(const u32x4u *)(src + (v * SORT_THREADS + tid) * 4);
#pragma unroll
            for (int v = 0; v < SORT_ITEMS / 4; ++v)
#pragma unroll
                for (int j = 0; j < 4; ++j) skey[(v * SORT_THREADS + tid) * 4 + j] = desc_key(__uint_as_float(w[v][j]));
        } else {
            for (int e = tid; e < tile_n; e += SORT_THREADS) skey[e] = desc_key(__uint_as_float(src[e]));
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            key[r] = (sub + r * 64) < tile_n ? skey[sub + r * 64] : 0xFFFFFFFFu;
            val[r] = (uint32_t)tile0 + (uint32_t)(sub + r * 64);
        }
    } else {
        // every lane loads (past the end of the row: the row's last element again) and all loads are issued before
        // the first use: a load under `if (i < n)` makes the compiler wait for each one in turn
        const uint32_t *wt = w_in + row + tile0;
        const uint8_t *ht = (const uint8_t *)h_in + (row + tile0) * (IN == FMT_KV ? 4 : IN == FMT_A ? 2 : 1);
        uint32_t w[SORT_ITEMS], hh[SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const int e = min(sub + r * 64, tile_n - 1);
            w[r] = wt[e];
            hh[r] = 0;
            if (IN == FMT_KV) hh[r] = ((const uint32_t *)ht)[e];
            if (IN == FMT_A) hh[r] = ((const uint16_t *)ht)[e];
            if (IN == FMT_B) hh[r] = ht[e];
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const bool valid = (sub + r * 64) < tile_n;
            if (IN == FMT_KV) {
                key[r] = valid ? w[r] : 0xFFFFFFFFu;
                val[r] = hh[r];
            } else {
                const uint32_t k = IN == FMT_C ? w[r] : ((w[r] & 0xFF000000u) | (hh[r] << (IN == FMT_A ? 8 : 16)));
                key[r] = valid ? k : 0xFFFFFFFFu;
                val[r] = w[r] & 0x00FFFFFFu;
            }
        }
    }
    if (tid < RADIX) {
        gdelta[tid] = gbase;
        dtot[tid] = qtot;
    }
    uint32_t *mycnt = wcnt[wave];
    if (ARANK) {
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            pos[r] = 0;
            if ((sub + r * 64) < tile_n)
                pos[r] = __hip_atomic_fetch_add(&mycnt[(key[r] >> shift) & 255u], 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const bool valid = (sub + r * 64) < tile_n;
            const uint32_t d = (key[r] >> shift) & 255u;
            // match-any on the 8 digit bits: `dlo/dhi` collect the lanes that differ from this one in some bit.  Per bit:
            // one sign-extract (0 / ~0), one compare (= the ballot, lands in SGPRs) and one three-input bit-op per half,
            // acc | (ballot ^ sel).  Lanes past the end of the row carry digit 255 and take part; they are masked below.
            uint32_t dlo = 0, dhi = 0;
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                int32_t sel = __builtin_amdgcn_sbfe((int32_t)d, bit, 1);
                asm("" : "+v"(sel));            // compare THIS register (else: a second shift of d per bit)
                const uint64_t m = __builtin_amdgcn_ballot_w64(sel < 0);
                dlo = __builtin_amdgcn_bitop3_b32(dlo, (uint32_t)m, (uint32_t)sel, 0xF6);
                dhi = __builtin_amdgcn_bitop3_b32(dhi, (uint32_t)(m >> 32), (uint32_t)sel, 0xF6);
            }
            const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
            const uint32_t plo = (uint32_t)vmask & ~dlo, phi = (uint32_t)(vmask >> 32) & ~dhi;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t cnt = __popc(plo) + __popc(phi);
            const uint32_t old = mycnt[d];
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) mycnt[d] = old + cnt;
            __builtin_amdgcn_wave_barrier();
            pos[r] = old + rank;
        }
    }
    __syncthreads();
    // per digit: tile count and per-wave offsets inside the digit (thread = digit)
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const uint32_t c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
        scan[tid] = run;
    }
    __syncthreads();
    // exclusive scan of the 256 digit totals by ONE wave (4 digits per lane + shuffle scan):
    // two barriers instead of the sixteen of a workgroup-wide Hillis-Steele scan
    if (wave == 0) {
        const uint32_t t0 = scan[4 * lane], t1 = scan[4 * lane + 1], t2 = scan[4 * lane + 2],
                       t3 = scan[4 * lane + 3];
        const uint32_t g0 = dtot[4 * lane], g1 = dtot[4 * lane + 1], g2 = dtot[4 * lane + 2],
                       g3 = dtot[4 * lane + 3];
        const uint32_t mine = t0 + t1 + t2 + t3, gmine = g0 + g1 + g2 + g3;
        const uint32_t inc = wave_inclusive_sum(mine), ginc = wave_inclusive_sum(gmine);
        const uint32_t ex = inc - mine, gex = ginc - gmine;
        const uint32_t toff[4] = {ex, ex + t0, ex + t0 + t1, ex + t0 + t1 + t2};
        const uint32_t goff[4] = {gex, gex + g0, gex + g0 + g1, gex + g0 + g1 + g2};   // digit bases of the query
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dgt = 4 * lane + k;
            scan[dgt] = toff[k];        // now: first slot of the digit in the tile
            gdelta[dgt] += goff[k] - toff[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if ((sub + r * 64) >= tile_n) continue;
        const uint32_t d = (key[r] >> shift) & 255u;
        const uint32_t lp = scan[d] + mycnt[d] + pos[r];
        skey[lp] = key[r];
        if (!VAL_IN_KEY) sval[lp] = val[r];
    }
    __syncthreads();
    // Offsets from the row start fit 32 bits in the packed formats (n <= 2^24 elements of at most 8 bytes): scalar
    // row pointer + 32-bit lane offset, no 64-bit arithmetic per element.
    constexpr bool OFF32 = IN == FMT_C;         // (the last pass of the packed formats)
    char *const wrow = (char *)(w_out + row), *const hrow = (char *)h_out + row * (OUT == FMT_KV ? 4 : OUT == FMT_A ? 2 : 1);
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int i = r * SORT_THREADS + tid;
        if (i >= tile_n) continue;
        const uint32_t k = skey[i], v = VAL_IN_KEY ? (k & 0x00FFFFFFu) : sval[i];
        const uint32_t dst32 = gdelta[(k >> shift) & 255u] + (uint32_t)i;
        if (LAST) {
            const int64_t dst = (int64_t)dst32;
            if (dst < klimit) {
                if (ranks) {
                    // the ranking leaves the sort here and is not read by it again: a non-temporal store (MDX_SORT_NT)
                    int64_t *dstp = OFF32 ? (int64_t *)((char *)(ranks + q * klimit) + (dst32 << 3)) : ranks + q * klimit + dst;
#if MDX_SORT_NT
                    __builtin_nontemporal_store((int64_t)v + id_offset, dstp);
#else
                    *dstp = (int64_t)v + id_offset;
#endif
                }
                if (top_scores) top_scores[q * klimit + dst] = scores[base + v];
            }
        } else if (OUT == FMT_KV) {
            w_out[row + (int64_t)dst32] = k;
            ((uint32_t *)h_out)[row + (int64_t)dst32] = v;
        } else {
            // (plain stores: the intermediates are written as partial lines that neighbouring tiles complete in the L2;
            // non-temporal stores here cost 2.2x -- 1.88 ms against 0.85)
            *(uint32_t *)(wrow + (dst32 << 2)) = (k & 0xFF000000u) | v;
            if (OUT == FMT_A) *(uint16_t *)(hrow + (dst32 << 1)) = (uint16_t)(k >> 8);
            if (OUT == FMT_B) *(uint8_t *)(hrow + dst32) = (uint8_t)(k >> 16);
        }
    }
}

// ---------------------------------------------------------------------------
// counting ranks of labelled items
// ---------------------------------------------------------------------------
constexpr int CNT_TILE = 4096;
constexpr int CNT_ITEMS = CNT_TILE / 256;     // elements per lane and tile

// Instead of comparing every row with every labelled item (rows x items compares; rOxford queries
// list up to hundreds), the items of a query are SORTED (64-bit composite (key : id), the ranking
// order) and every row binary-searches its place among them: u = number of
// items that rank before-or-at the row.  A row precedes item j iff j >= u, so the per-item counts
// are the prefix sums of the histogram of u -- log2(items) compares per row.  Rows behind all items
// (the vast majority) touch nothing; the histogram lives in LDS, one atomic per item and workgroup
// at the end.
constexpr int CNTB_REFS = 256;

__global__ __launch_bounds__(256) void rank_count_bsearch_kernel(
    const float *__restrict__ scores, int64_t n, int64_t id_offset,
    const float *__restrict__ ref_scores, const int64_t *__restrict__ ref_ids,
    const int64_t *__restrict__ offsets, unsigned long long *__restrict__ cnt, int nblk)
{
    // the histogram of u in CNT_COPIES copies selected by LANE: with few labelled items (20 per query: 21 values of u) and
    // many live rows, 64 lanes of one ds_add hit a handful of addresses; row stride 257 keeps the copies in different banks
    constexpr int CNT_COPIES = 8;
    __shared__ uint64_t sref[CNTB_REFS];
    __shared__ uint32_t histc[CNT_COPIES][CNTB_REFS + 1];
    __shared__ uint32_t hist[CNTB_REFS + 1];
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, copy = tid & (CNT_COPIES - 1);
    const int64_t q = blockIdx.y;
    const int64_t lo = offsets[q], hi = offsets[q + 1];
    for (int64_t r0 = lo; r0 < hi; r0 += CNTB_REFS) {
        const int nref = (int)((hi - r0) < CNTB_REFS ? (hi - r0) : CNTB_REFS);
        int P = 2;
        while (P < nref) P <<= 1;                                   // search width (power of two)
        __syncthreads();
        uint64_t mine = ~0ull;                                      // padding ranks after everything
        if (tid < nref) {
            const int64_t id = ref_ids[r0 + tid];
            const uint32_t idc = id < 0 ? 0u : (id > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)id);
            mine = ((uint64_t)desc_key(ref_scores[r0 + tid]) << 32) | idc;
        }
        sref[tid] = mine;
        for (int e = tid; e < CNT_COPIES * (CNTB_REFS + 1); e += 256) (&histc[0][0])[e] = 0;
        __syncthreads();
        bitonic_sort_lds(sref, CNTB_REFS, tid, 256);
        for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
            const int64_t t0 = (int64_t)b * CNT_TILE;
            uint64_t el[CNT_ITEMS];
            // a lane takes 4 consecutive rows with one 16-byte load (dword-aligned: rows of an odd length start anywhere); all
            // loads of the tile are issued before the first use, whole tiles without a range check
            typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const uint32_t *row = (const uint32_t *)scores + q * n;
            if (t0 + CNT_TILE <= n) {
                u32x4u w[CNT_ITEMS / 4];
#pragma unroll
                for (int v = 0; v < CNT_ITEMS / 4; ++v) w[v] = *(const u32x4u *)(row + t0 + (v * 256 + tid) * 4);
#pragma unroll
                for (int v = 0; v < CNT_ITEMS / 4; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t i = t0 + (v * 256 + tid) * 4 + j;
                        el[v * 4 + j] = ((uint64_t)desc_key(__uint_as_float(w[v][j])) << 32) | (uint32_t)(i + id_offset);
                    }
            } else {
#pragma unroll
                for (int e = 0; e < CNT_ITEMS; ++e) {
                    const int64_t i = t0 + (e / 4 * 256 + tid) * 4 + e % 4;
                    el[e] = i < n ? ((uint64_t)desc_key(scores[q * n + i]) << 32) | (uint32_t)(i + id_offset) : ~0ull;
                }
            }
            const uint64_t last = sref[nref - 1];                   // broadcast: rows behind every item are skipped
#pragma unroll
            for (int e = 0; e < CNT_ITEMS; ++e) {
                const bool live = el[e] < last;                     // else u = nref: precedes no item
                if (__ballot(live) == 0) continue;                  // uniform: the common case for most tiles
                // u = #{j : sref[j] <= el}: branch-free binary search over the P-padded array
                uint32_t u = 0;
                for (int step = P >> 1; step > 0; step >>= 1)
                    u += (sref[u + step - 1] <= el[e]) ? (uint32_t)step : 0u;
                if (live) atomicAdd(&histc[copy][u], 1u);
            }
        }
        __syncthreads();
        // inclusive prefix over u (thread = position in the sorted order)
        uint32_t inc = 0;
#pragma unroll
        for (int c = 0; c < CNT_COPIES; ++c) inc += histc[c][tid];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        for (int w = 0; w < (tid >> 6); ++w) inc += wsum[w];
        __syncthreads();
        hist[tid] = inc;                                            // rows preceding the item at sorted position tid
        __syncthreads();
        if (tid < nref) {
            // my item's position among the sorted ones (items are distinct unless listed twice, and
            // equal items have equal counts)
            uint32_t pos = 0;
            for (int step = CNTB_REFS >> 1; step > 0; step >>= 1)
                pos += (sref[pos + step - 1] < mine) ? (uint32_t)step : 0u;
            const uint32_t v = hist[pos];
            if (v) atomicAdd(&cnt[r0 + tid], (unsigned long long)v);
        }
    }
}

__global__ void gather_scores_kernel(const float *__restrict__ scores, int64_t n, int64_t nq,
                                     const int64_t *__restrict__ ids,
                                     const int64_t *__restrict__ offsets, int64_t total,
                                     float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    // binary search the query this entry belongs to
    int64_t a = 0, b = nq;
    while (b - a > 1) {
        const int64_t m = (a + b) >> 1;
        if (offsets[m] <= t) a = m; else b = m;
    }
    const int64_t id = ids[t];
    out[t] = (id >= 0 && id < n) ? scores[a * n + id] : __uint_as_float(0x7FC00000u);
}

struct RankWs {
    uint32_t *w[2];             // key words (FMT_KV) or packed words (FMT_A/B/C), ping-pong
    void *h[2];                 // id words (FMT_KV); h[0] also holds the u16 middles (FMT_A), h[1] the u8 ones (FMT_B)
    uint32_t *block_hist;
    uint32_t *digit_tot;
    int nblk;
    int64_t stride;             // row stride of w / h, elements
};

static int64_t carve(RankWs *ws, char *base, int64_t n, int64_t nq)
{
    const int64_t nblk = ceil_div(n, SORT_TILE);
    const int64_t stride = round_up(n, 256);
    const int64_t elems = stride * nq * 4;
    int64_t off = 0;
    for (int i = 0; i < 2; ++i) {
        if (ws) ws->w[i] = (uint32_t *)(base + off);
        off += elems;
        if (ws) ws->h[i] = (void *)(base + off);
        off += elems;
    }
    if (ws) ws->block_hist = (uint32_t *)(base + off);
    off += round_up(nq * nblk * RADIX * 4, 256);
    if (ws) ws->digit_tot = (uint32_t *)(base + off);
    off += round_up(nq * RADIX * 4, 256);
    if (ws) ws->nblk = (int)nblk;
    if (ws) ws->stride = stride;
    return off;
}

// ---------------------------------------------------------------------------
// Rows of at most 8192 scores (rOxford5k alone: 4 993; 247tokyo1k: 1 125; the candidate lists of mdx_topk): the whole
// ranking of a query by ONE workgroup in LDS -- load, stable LSD radix over the key bytes that differ, write -- one launch
// instead of twelve.  Needs the ordered ds_add_rtn (atomic_rank_ok); the tiled passes above serve otherwise.
// ---------------------------------------------------------------------------
constexpr int LS_THREADS = 512, LS_WAVES = LS_THREADS / 64, LS_MAX_ITEMS = 16, LS_CAP = LS_THREADS * LS_MAX_ITEMS;

struct LsShared {
    uint32_t key[LS_CAP];
    uint32_t val[LS_CAP];
    uint32_t cnt[LS_WAVES][RADIX];
    uint32_t tot[RADIX];
    uint32_t diff;
};

// Stable sort of m <= 512 * ITEMS (key, val) pairs held in registers in (wave, round, lane) order -- element
// i = wave * 64 * ITEMS + round * 64 + lane -- by key; equal keys keep their order.  LSD radix, 8 bits per pass,
// only over the bytes in which the keys differ; a pass ranks with ds_add_rtn on per-wave digit counters (ordered:
// atomic_rank_ok), puts the elements in place in LDS and reloads them.  Three barriers per pass: a wave zeroes and
// reads only ITS OWN counters outside the scan (LDS operations of one wave stay in order), and the scan over the
// waves and over the digits is one wave's work.  The sorted pairs are left in sh.key / sh.val[0..m).
template <int ITEMS>
__device__ __forceinline__ void lds_radix_sort(uint32_t (&key)[ITEMS], uint32_t (&val)[ITEMS], int m, LsShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave * 64 * ITEMS + lane;
    uint32_t pos[ITEMS];
    uint32_t diff = 0;
    if (tid == 0) {
        sh.diff = 0;
        sh.tot[0] = key[0];
    }
    uint32_t *mycnt = sh.cnt[wave];
#pragma unroll
    for (int e = 0; e < RADIX / 64; ++e) mycnt[e * 64 + lane] = 0;
    __syncthreads();
    const uint32_t first = sh.tot[0];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
        if (sub + r * 64 < m) diff |= key[r] ^ first;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0 && diff) atomicOr(&sh.diff, diff);
    __syncthreads();
    diff = sh.diff;
    bool staged = false;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        if (((diff >> shift) & 255u) == 0) continue;        // this byte is the same in every key
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            pos[r] = 0;
            if (sub + r * 64 < m)
                pos[r] = __hip_atomic_fetch_add(&mycnt[(key[r] >> shift) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (wave == 0) {
            // lane l owns digits l, l + 64, l + 128, l + 192 (consecutive lanes = consecutive banks): per digit the
            // waves' counts become offsets, then the digit totals are scanned in digit order
            uint32_t t[4], c[4][LS_WAVES];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int w = 0; w < LS_WAVES; ++w) c[k][w] = sh.cnt[w][64 * k + lane];      // all 32 reads in flight
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < LS_WAVES; ++w) {
                    sh.cnt[w][64 * k + lane] = run;
                    run += c[k][w];
                }
                t[k] = run;
            }
            uint32_t inc[4] = {t[0], t[1], t[2], t[3]};
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t v = __shfl_up(inc[k], o, 64);
                    if (lane >= o) inc[k] += v;
                }
            uint32_t before = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sh.tot[64 * k + lane] = before + inc[k] - t[k];
                before += __shfl(inc[k], 63, 64);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (sub + r * 64 >= m) continue;
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t dst = sh.tot[d] + mycnt[d] + pos[r];
            sh.key[dst] = key[r];
            sh.val[dst] = val[r];
        }
        staged = true;
        if ((diff >> shift) >> 8) {                         // a later pass follows
#pragma unroll
            for (int e = 0; e < RADIX / 64; ++e) mycnt[e * 64 + lane] = 0;     // own counters, after this wave's own reads
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; ++r)
                if (sub + r * 64 < m) {                     // back to registers, in order
                    key[r] = sh.key[sub + r * 64];
                    val[r] = sh.val[sub + r * 64];
                }
        } else {
            __syncthreads();
        }
    }
    if (!staged) {          // all keys equal: the input order is the answer
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            if (sub + r * 64 < m) {
                sh.key[sub + r * 64] = key[r];
                sh.val[sub + r * 64] = val[r];
            }
        __syncthreads();
    }
}

template <int ITEMS>
__device__ __forceinline__ void rank_small(const float *__restrict__ scores, const SegTable &seg, int64_t q, int n, int64_t id_offset,
                                           int64_t *__restrict__ ranks, float *__restrict__ top_scores, int klimit, LsShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave * 64 * ITEMS + lane;
    uint32_t key[ITEMS], val[ITEMS];
    const float *row = scores + q * (int64_t)n;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int i = sub + r * 64;
        const int j = i < n ? i : n - 1;                // every lane loads: all loads in flight together
        key[r] = desc_key(seg.nseg > 0 ? *seg_elem(seg, q, j) : row[j]);
        val[r] = (uint32_t)j;
    }
    lds_radix_sort<ITEMS>(key, val, n, sh);
    for (int i = tid; i < klimit; i += LS_THREADS) {
        const uint32_t v = sh.val[i];
        if (ranks) ranks[q * (int64_t)klimit + i] = (int64_t)v + id_offset;
        if (top_scores) top_scores[q * (int64_t)klimit + i] = seg.nseg > 0 ? *seg_elem(seg, q, v) : row[v];
    }
}

__global__ __launch_bounds__(LS_THREADS, 2) void rank_small_kernel(const float *__restrict__ scores, SegTable seg, int n, int64_t id_offset,
                                                                   int64_t *__restrict__ ranks, float *__restrict__ top_scores, int klimit)
{
    __shared__ LsShared sh;
    const int64_t q = blockIdx.x;
    if (n <= LS_THREADS * 4) rank_small<4>(scores, seg, q, n, id_offset, ranks, top_scores, klimit, sh);
    else if (n <= LS_THREADS * 8) rank_small<8>(scores, seg, q, n, id_offset, ranks, top_scores, klimit, sh);
    else if (n <= LS_THREADS * 12) rank_small<12>(scores, seg, q, n, id_offset, ranks, top_scores, klimit, sh);
    else rank_small<16>(scores, seg, q, n, id_offset, ranks, top_scores, klimit, sh);
}

// Does the LDS serve same-address lanes of one ds_add_rtn in ascending lane order?  Eight waves at once, 64 rounds
// each, digits drawn from a few values (long conflict chains), from 256 values, and all equal; every lane checks
// that what it got back is the count of earlier rounds plus the number of LOWER lanes with its digit (ballots).
__global__ __launch_bounds__(SORT_THREADS) void lds_order_probe_kernel(uint32_t seed, int *bad)
{
    __shared__ uint32_t cnt[SORT_WAVES][RADIX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < SORT_WAVES * RADIX; e += SORT_THREADS) (&cnt[0][0])[e] = 0;
    __syncthreads();
    uint32_t x = seed ^ (uint32_t)(tid * 2654435761u) ^ (blockIdx.x * 40503u);
    int wrong = 0;
    for (int round = 0; round < 64; ++round) {
        x = x * 1664525u + 1013904223u;
        const int mode = (round + blockIdx.x) % 3;
        const uint32_t d = mode == 0 ? (x >> 24) : mode == 1 ? ((x >> 24) & 3u) * 37u : 200u;
        const bool active = ((x >> 8) & 15u) != 0;          // some lanes sit a round out
        uint32_t before = 0, got = 0;
        if (active) {
            before = cnt[wave][d];
            __builtin_amdgcn_wave_barrier();
            got = __hip_atomic_fetch_add(&cnt[wave][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t lower = 0;
        for (int l = 0; l < 64; ++l) {
            const uint32_t dl = __shfl(d, l, 64);
            const bool al = __shfl(active ? 1 : 0, l, 64) != 0;
            if (al && dl == d && l < lane) ++lower;
        }
        if (active && got != before + lower) wrong = 1;
        __builtin_amdgcn_wave_barrier();
    }
    if (wrong) atomicOr(bad, 1);
}

// Which form ranks the elements of a wave, per device: 0 unknown, 1 ballots, 2 ds_add_rtn.  The atomic form is taken only
// on the architecture it was validated on (gfx950) AND after the probe above has passed on this very device; every other
// case -- another architecture, a failed or impossible probe, a stream that is being captured before the probe has run --
// ranks with ballots, which rest on documented behaviour only.  MDX_SORT_RANK=ballot|atomic overrides both (tests run both).
static std::atomic<int> g_arank[64];

// Runs the probe on `s` and WAITS for it (a few hundred microseconds, once per device and process).  Called from
// mdx_index_create (which may synchronise) so that enqueue-only ranking calls normally find the answer; a ranking call on
// a device that never built an index runs it itself, unless its stream is capturing.
int probe_lds_order(hipStream_t s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    int known = g_arank[dev].load(std::memory_order_acquire);
    if (known > 0) return known;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;     // ask later
    int verdict = 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && !strncmp(prop.gcnArchName, "gfx950", 6)) {
        int *bad = nullptr, host = 1;
        if (hipMalloc(&bad, sizeof(int)) == hipSuccess) {
            bool ok = hipMemsetAsync(bad, 0, sizeof(int), s) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(lds_order_probe_kernel, dim3(1024), dim3(SORT_THREADS), 0, s, 0x9E3779B9u, bad);
                ok = hipMemcpyAsync(&host, bad, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
                     hipStreamSynchronize(s) == hipSuccess;
            }
            (void)hipFree(bad);
            if (ok && host == 0) verdict = 2;
        }
    }
    g_arank[dev].store(verdict, std::memory_order_release);
    return verdict;
}

static bool atomic_rank_ok(hipStream_t s)
{
    static const char *force = getenv("MDX_SORT_RANK");
    if (force && !strcmp(force, "ballot")) return false;
    if (force && !strcmp(force, "atomic")) return true;
    return probe_lds_order(s) == 2;
}

template <int IN, int OUT, typename SEGT>
static void sort_pass_on(const RankWs &ws, const float *scores, int64_t n, int64_t nq, int pass, int64_t id_offset,
                         int64_t *ranks, float *top_scores, int64_t klimit, bool arank, const SEGT &seg, hipStream_t s)
{
    const int shift = 8 * pass;
    const uint32_t *w_in = pass == 0 ? nullptr : ws.w[(pass - 1) & 1];
    const void *h_in = pass == 0 ? nullptr : ws.h[(pass - 1) & 1];
    const dim3 grid((unsigned)ws.nblk, (unsigned)nq), blk(SORT_THREADS);
    // the histogram reads only the array the digit of this pass lives in
    const void *digits = (IN == FMT_A || IN == FMT_B) ? h_in : (const void *)w_in;
    hipLaunchKernelGGL((sort_hist_kernel<IN, SEGT>), grid, blk, 0, s, scores, digits, n, ws.stride, ws.nblk, shift,
                       ws.block_hist, seg);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(RADIX / SCAN_DIGITS, (unsigned)nq), dim3(SCAN_GROUPS * SCAN_DIGITS), 0, s,
                       ws.block_hist, ws.nblk, ws.digit_tot);
    if (arank)
        hipLaunchKernelGGL((sort_scatter_kernel<IN, OUT, true, SEGT>), grid, blk, 0, s, scores, w_in, h_in, ws.w[pass & 1],
                           ws.h[pass & 1], ranks, top_scores, n, ws.stride, ws.nblk, shift, ws.block_hist, ws.digit_tot,
                           id_offset, klimit, seg);
    else
        hipLaunchKernelGGL((sort_scatter_kernel<IN, OUT, false, SEGT>), grid, blk, 0, s, scores, w_in, h_in, ws.w[pass & 1],
                           ws.h[pass & 1], ranks, top_scores, n, ws.stride, ws.nblk, shift, ws.block_hist, ws.digit_tot,
                           id_offset, klimit, seg);
}

// only pass 0 of a ranking of column blocks reads the segment table
template <int IN, int OUT>
static void sort_pass(const RankWs &ws, const float *scores, int64_t n, int64_t nq, int pass, int64_t id_offset,
                      int64_t *ranks, float *top_scores, int64_t klimit, bool arank, const SegTable &seg, hipStream_t s)
{
    if (IN == FMT_SCORES && seg.nseg > 0)
        sort_pass_on<IN, OUT, SegTable>(ws, scores, n, nq, pass, id_offset, ranks, top_scores, klimit, arank, seg, s);
    else
        sort_pass_on<IN, OUT, NoSeg>(ws, scores, n, nq, pass, id_offset, ranks, top_scores, klimit, arank, NoSeg{}, s);
}

static int rank_impl(const float *scores, int64_t n, int64_t nq, int64_t id_offset, int64_t *ranks,
                     float *top_scores, int64_t klimit, void *workspace, int64_t workspace_bytes,
                     hipStream_t s, const char *who, const SegTable *segments = nullptr)
{
    SegTable seg;
    if (segments) seg = *segments;
    else memset(&seg, 0, sizeof seg);
    if (seg.nseg > 0) {
        const uint64_t w = (uint64_t)ceil_div(n, (int64_t)seg.nseg), inv = (1ull << 32) / (w ? w : 1);
        seg.inv_w = inv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)inv;
    }
    MDX_CHECK_ARG(scores || seg.nseg > 0, "%s: NULL scores", who);
    MDX_CHECK_ARG(n > 0 && nq > 0, "%s: n=%lld nq=%lld must be positive", who, (long long)n,
                  (long long)nq);
    MDX_CHECK_ARG(n < (1ll << 32) && nq < 65536, "%s: n or nq too large", who);
    const int64_t need = carve(nullptr, nullptr, n, nq);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace %lld B < required %lld B", who, (long long)workspace_bytes,
                  (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    RankWs ws;
    carve(&ws, (char *)workspace, n, nq);
    // MDX_SORT_NO_PACK=1: the (key word, id word) layout also for small n (tests run both)
    static const bool no_pack = getenv("MDX_SORT_NO_PACK") && atoi(getenv("MDX_SORT_NO_PACK")) != 0;
    const bool arank = atomic_rank_ok(s);
    // MDX_SORT_SMALL=0: the tiled passes also for short rows (tests run both)
    static const bool no_small = getenv("MDX_SORT_SMALL") && atoi(getenv("MDX_SORT_SMALL")) == 0;
    if (arank && n <= LS_CAP && !no_small) {
        hipLaunchKernelGGL(rank_small_kernel, dim3((unsigned)nq), dim3(LS_THREADS), 0, s, scores, seg, (int)n, id_offset, ranks, top_scores,
                           (int)klimit);
        MDX_LAUNCH_CHECK();
        return MDX_OK;
    }
    if (n <= (1ll << 24) && !no_pack) {
        sort_pass<FMT_SCORES, FMT_A>(ws, scores, n, nq, 0, id_offset, ranks, top_scores, klimit, arank, seg, s);
        sort_pass<FMT_A, FMT_B>(ws, scores, n, nq, 1, id_offset, ranks, top_scores, klimit, arank, seg, s);
        sort_pass<FMT_B, FMT_C>(ws, scores, n, nq, 2, id_offset, ranks, top_scores, klimit, arank, seg, s);
        sort_pass<FMT_C, FMT_RANKS>(ws, scores, n, nq, 3, id_offset, ranks, top_scores, klimit, arank, seg, s);
    } else {
        sort_pass<FMT_SCORES, FMT_KV>(ws, scores, n, nq, 0, id_offset, ranks, top_scores, klimit, arank, seg, s);
        sort_pass<FMT_KV, FMT_KV>(ws, scores, n, nq, 1, id_offset, ranks, top_scores, klimit, arank, seg, s);
        sort_pass<FMT_KV, FMT_KV>(ws, scores, n, nq, 2, id_offset, ranks, top_scores, klimit, arank, seg, s);
        sort_pass<FMT_KV, FMT_RANKS>(ws, scores, n, nq, 3, id_offset, ranks, top_scores, klimit, arank, seg, s);
    }
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

// ---------------------------------------------------------------------------
// Row-wise top-k by radix SELECT (k << n): instead of sorting a million scores per query to
// keep the first k, find the k-th key digit by digit from the top (one histogram pass per level,
// normally ONE level: the best scores sit in the sparse tail of the distribution), compact
// the candidates (keys below the threshold prefix, then keys equal to it) in id order, and
// sort only those.  Exact, same tie rule as mdx_rank_full.
// ---------------------------------------------------------------------------
struct SelState { uint32_t prefix, mask, k_rem, resolved; };

constexpr int SEL_CAP = 4096;   // slack: a level is final once (#equal-prefix keys) <= k_rem + SEL_CAP

__global__ __launch_bounds__(SORT_THREADS) void select_hist_kernel(const float *__restrict__ scores, int64_t n,
                                                                   int nblk, int shift,
                                                                   const SelState *__restrict__ st,
                                                                   uint32_t *__restrict__ block_hist)
{
    __shared__ uint32_t h[SORT_WAVES / 2][RADIX];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t q = blockIdx.y;
    const SelState s = st[q];
    if (s.resolved) return;         // later levels of an already decided query: nothing to do
    // workgroups stride over the tiles, so that the (usual) early exit above costs few launches
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        for (int e = tid; e < (SORT_WAVES / 2) * RADIX; e += SORT_THREADS) (&h[0][0])[e] = 0;
        uint32_t k[SORT_ITEMS];
        bool ok[SORT_ITEMS];
        if ((b + 1) * SORT_TILE <= n) {
            // whole tile: a lane takes 4 consecutive rows per 16-byte load (a histogram does not care which lane counts which row)
            typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const uint32_t *row = (const uint32_t *)scores + q * n + b * SORT_TILE;
            u32x4u w[SORT_ITEMS / 4];
#pragma unroll
            for (int v = 0; v < SORT_ITEMS / 4; ++v) w[v] = *(const u32x4u *)(row + (v * SORT_THREADS + tid) * 4);
#pragma unroll
            for (int r = 0; r < SORT_ITEMS; ++r) {
                k[r] = desc_key(__uint_as_float(w[r / 4][r % 4]));
                ok[r] = true;
            }
        } else {
#pragma unroll
            for (int r = 0; r < SORT_ITEMS; ++r) {
                const int64_t i = b * SORT_TILE + r * SORT_THREADS + tid;
                ok[r] = i < n;
                k[r] = ok[r] ? desc_key(scores[q * n + i]) : 0u;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r)
            if (ok[r] && (k[r] & s.mask) == s.prefix) atomicAdd(&h[wave >> 1][(k[r] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < RADIX) {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES / 2; ++w) tot += h[w][tid];
            block_hist[(q * nblk + b) * RADIX + tid] = tot;
        }
        __syncthreads();
    }
}

// per query: the digit at which the cumulative count (best keys first) reaches k_rem; per-tile
// counts of the two candidate classes (strictly below the new prefix / equal to it).
// 1024 threads: 4 groups x 256 digits for the totals, 16 waves for the per-tile class counts.
__global__ __launch_bounds__(1024) void select_step_kernel(const uint32_t *__restrict__ block_hist, int nblk,
                                                           int shift, SelState *__restrict__ st,
                                                           uint32_t *__restrict__ tile_less,
                                                           uint32_t *__restrict__ tile_eq)
{
    __shared__ uint32_t part[4][RADIX];
    __shared__ uint32_t cum[RADIX];
    __shared__ uint32_t dstar_s;
    const int tid = threadIdx.x, d = tid & 255, g = tid >> 8;
    const int64_t q = blockIdx.x;
    SelState s = st[q];
    if (s.resolved) return;
    const uint32_t *bh = block_hist + q * nblk * RADIX;
    const int per = (nblk + 3) / 4, b0 = g * per, b1 = (b0 + per) < nblk ? (b0 + per) : nblk;
    uint32_t tot = 0;
    for (int b = b0; b < b1; ++b) tot += bh[(int64_t)b * RADIX + d];
    part[g][d] = tot;
    __syncthreads();
    if (g == 0) cum[d] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    __syncthreads();
    for (int off = 1; off < RADIX; off <<= 1) {
        const uint32_t v = (g == 0 && d >= off) ? cum[d - off] : 0u;
        __syncthreads();
        if (g == 0) cum[d] += v;
        __syncthreads();
    }
    // k_rem >= 1 and the matching keys number at least k_rem, so exactly one digit crosses
    if (g == 0 && cum[d] >= s.k_rem && (d == 0 || cum[d - 1] < s.k_rem)) dstar_s = d;
    __syncthreads();
    const uint32_t ds = dstar_s;
    const uint32_t c_eq = ds == 0 ? cum[0] : cum[ds] - cum[ds - 1];
    const uint32_t c_less = cum[ds] - c_eq;
    // one wave per tile: lane l sums digits 4l..4l+3 (one coalesced KiB), shuffle-reduce
    const int lane = tid & 63, wv = tid >> 6;
    for (int b = wv; b < nblk; b += 16) {
        const uint4 c = *(const uint4 *)(bh + (int64_t)b * RADIX + 4 * lane);
        const uint32_t e0 = 4 * lane;
        uint32_t less = (e0 < ds ? c.x : 0u) + (e0 + 1 < ds ? c.y : 0u) + (e0 + 2 < ds ? c.z : 0u) + (e0 + 3 < ds ? c.w : 0u);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) less += __shfl_xor(less, o, 64);
        if (lane == 0) {
            tile_less[q * nblk + b] += less;
            tile_eq[q * nblk + b] = bh[(int64_t)b * RADIX + ds];
        }
    }
    if (tid == 0) {
        s.prefix |= ds << shift;
        s.mask |= 255u << shift;
        s.k_rem -= c_less;
        s.resolved = (shift == 0 || c_eq <= s.k_rem + SEL_CAP) ? 1u : 0u;
        st[q] = s;
    }
}

// per query: exclusive offsets of every tile's two candidate blocks (all "less" first, then "equal")
__global__ __launch_bounds__(256) void select_offsets_kernel(uint32_t *__restrict__ tile_less,
                                                             uint32_t *__restrict__ tile_eq, int nblk)
{
    __shared__ uint32_t sa[256], sb[256];
    __shared__ uint32_t carry_a, carry_b;
    const int t = threadIdx.x;
    const int64_t q = blockIdx.x;
    uint32_t *la = tile_less + q * nblk, *lb = tile_eq + q * nblk;
    if (t == 0) { carry_a = 0; carry_b = 0; }
    __syncthreads();
    // pass 1: totals of "less" (the "equal" block starts after all of them)
    uint32_t part = 0;
    for (int b = t; b < nblk; b += 256) part += la[b];
    sa[t] = part;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sa[t] += sa[t + off]; __syncthreads(); }
    const uint32_t total_less = sa[0];
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int b = b0 + t;
        const uint32_t va = b < nblk ? la[b] : 0u, vb = b < nblk ? lb[b] : 0u;
        sa[t] = va; sb[t] = vb;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t xa = t >= off ? sa[t - off] : 0u, xb = t >= off ? sb[t - off] : 0u;
            __syncthreads();
            sa[t] += xa; sb[t] += xb;
            __syncthreads();
        }
        if (b < nblk) { la[b] = carry_a + sa[t] - va; lb[b] = total_less + carry_b + sb[t] - vb; }
        __syncthreads();
        if (t == 255) { carry_a += sa[255]; carry_b += sb[255]; }
        __syncthreads();
    }
}

// ordered compaction of the candidates of one tile: (score, id) at their block offsets, id order kept
__global__ __launch_bounds__(SORT_THREADS) void select_scatter_kernel(const float *__restrict__ scores, int64_t n,
                                                                      int nblk, const SelState *__restrict__ st,
                                                                      const uint32_t *__restrict__ off_less,
                                                                      const uint32_t *__restrict__ off_eq,
                                                                      uint32_t cap_total,
                                                                      float *__restrict__ cand_scores,
                                                                      uint32_t *__restrict__ cand_ids)
{
    __shared__ uint32_t wl[SORT_WAVES], we[SORT_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.y, b = blockIdx.x;
    const SelState s = st[q];
    const int64_t sub0 = b * SORT_TILE + wave * SUB_TILE;
    float v[SORT_ITEMS];
    uint32_t pl[SORT_ITEMS], pe[SORT_ITEMS];        // position inside the wave's less / equal sequence (or ~0)
    uint32_t nl = 0, ne = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = sub0 + r * 64 + lane;
        v[r] = i < n ? scores[q * n + i] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = sub0 + r * 64 + lane;
        const uint32_t km = desc_key(v[r]) & s.mask;
        const bool is_l = i < n && km < s.prefix, is_e = i < n && km == s.prefix;
        const uint64_t ml = __ballot(is_l), me = __ballot(is_e);
        pl[r] = is_l ? nl + __popcll(ml & lt_mask) : 0xFFFFFFFFu;
        pe[r] = is_e ? ne + __popcll(me & lt_mask) : 0xFFFFFFFFu;
        nl += __popcll(ml);
        ne += __popcll(me);
    }
    if (lane == 0) { wl[wave] = nl; we[wave] = ne; }
    __syncthreads();
    uint32_t bl = off_less[q * nblk + b], be = off_eq[q * nblk + b];
    for (int w = 0; w < wave; ++w) { bl += wl[w]; be += we[w]; }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t pos = pl[r] != 0xFFFFFFFFu ? bl + pl[r] : (pe[r] != 0xFFFFFFFFu ? be + pe[r] : 0xFFFFFFFFu);
        if (pos < cap_total) {
            cand_scores[q * cap_total + pos] = v[r];
            cand_ids[q * cap_total + pos] = (uint32_t)(sub0 + r * 64 + lane);
        }
    }
}

__global__ void select_init_kernel(SelState *st, int64_t nq, uint32_t k)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) st[q] = SelState{0u, 0u, k, 0u};
}

__global__ void select_gather_kernel(const int64_t *__restrict__ local, const float *__restrict__ cand_scores,
                                     const uint32_t *__restrict__ cand_ids, int64_t nq, int64_t k,
                                     uint32_t cap_total, int64_t id_offset, int64_t *__restrict__ top_ids,
                                     float *__restrict__ top_scores)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) return;
    const int64_t q = t / k, j = local[t];
    if (top_ids) top_ids[t] = (int64_t)cand_ids[q * cap_total + j] + id_offset;
    if (top_scores) top_scores[t] = cand_scores[q * cap_total + j];
}

// workspace carve-up of the select path (lives inside a workspace sized by mdx_rank_workspace(n, nq))
struct SelWs {
    uint32_t *block_hist, *tile_less, *tile_eq, *cand_ids;
    SelState *state;
    float *cand_scores;
    int64_t *local;
    char *sort_ws;
    int64_t sort_ws_bytes;
};

static int64_t carve_select(SelWs *ws, char *base, int64_t n, int64_t nq, int64_t k)
{
    const int64_t nblk = ceil_div(n, SORT_TILE), ct = k + SEL_CAP;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char *p = base ? base + off : nullptr; off += round_up(bytes, 256); return p; };
    char *p;
    p = take(nq * nblk * RADIX * 4); if (ws) ws->block_hist = (uint32_t *)p;
    p = take(nq * nblk * 4);         if (ws) ws->tile_less = (uint32_t *)p;
    p = take(nq * nblk * 4);         if (ws) ws->tile_eq = (uint32_t *)p;
    p = take(nq * sizeof(SelState)); if (ws) ws->state = (SelState *)p;
    p = take(nq * ct * 4);           if (ws) ws->cand_scores = (float *)p;
    p = take(nq * ct * 4);           if (ws) ws->cand_ids = (uint32_t *)p;
    p = take(nq * k * 8);            if (ws) ws->local = (int64_t *)p;
    const int64_t sw = carve(nullptr, nullptr, ct, nq);
    p = take(sw);                    if (ws) { ws->sort_ws = p; ws->sort_ws_bytes = sw; }
    return off;
}

static int topk_select(const float *scores, int64_t n, int64_t nq, int64_t k, int64_t id_offset,
                       int64_t *top_ids, float *top_scores, void *workspace, hipStream_t s)
{
    SelWs ws;
    carve_select(&ws, (char *)workspace, n, nq, k);
    const int nblk = (int)ceil_div(n, SORT_TILE);
    const uint32_t ct = (uint32_t)(k + SEL_CAP);
    const dim3 grid((unsigned)nblk, (unsigned)nq), blk(SORT_THREADS);
    hipLaunchKernelGGL(select_init_kernel, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, s, ws.state, nq, (uint32_t)k);
    MDX_HIP(hipMemsetAsync(ws.tile_less, 0, (size_t)nq * nblk * 4, s));
    MDX_HIP(hipMemsetAsync(ws.cand_scores, 0xFF, (size_t)nq * ct * 4, s));      // NaN padding sorts last
    MDX_HIP(hipMemsetAsync(ws.cand_ids, 0, (size_t)nq * ct * 4, s));
    for (int shift = 24; shift >= 0; shift -= 8) {
        // level 1 always runs: one workgroup per tile; later levels usually exit at once: fewer workgroups
        const dim3 hgrid((unsigned)((shift == 24 || nblk < 512) ? nblk : 512), (unsigned)nq);
        hipLaunchKernelGGL(select_hist_kernel, hgrid, blk, 0, s, scores, n, nblk, shift, ws.state, ws.block_hist);
        hipLaunchKernelGGL(select_step_kernel, dim3((unsigned)nq), dim3(1024), 0, s, ws.block_hist, nblk, shift,
                           ws.state, ws.tile_less, ws.tile_eq);
    }
    hipLaunchKernelGGL(select_offsets_kernel, dim3((unsigned)nq), dim3(256), 0, s, ws.tile_less, ws.tile_eq, nblk);
    hipLaunchKernelGGL(select_scatter_kernel, grid, blk, 0, s, scores, n, nblk, ws.state, ws.tile_less, ws.tile_eq,
                       ct, ws.cand_scores, ws.cand_ids);
    MDX_LAUNCH_CHECK();
    int rc = rank_impl(ws.cand_scores, ct, nq, 0, ws.local, nullptr, k, ws.sort_ws, ws.sort_ws_bytes, s, "mdx_topk");
    if (rc != MDX_OK) return rc;
    hipLaunchKernelGGL(select_gather_kernel, dim3((unsigned)ceil_div(nq * k, 256)), dim3(256), 0, s, ws.local,
                       ws.cand_scores, ws.cand_ids, nq, k, ct, id_offset, top_ids, top_scores);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

// ---------------------------------------------------------------------------
// Top-k by SAMPLED THRESHOLD (k << n, the serving case): elements are ordered by the 64-bit
// composite (descending-order key : row index) -- a strict total order that IS the ranking order,
// ties included.  (1) per query, 4096 jittered samples are sorted in LDS and the m-th smallest
// composite becomes the threshold, m chosen so that ~max(6k, 2048) elements are expected below it;
// (2) ONE pass over the scores appends every element <= threshold to a per-tile candidate list
// (order does not matter, the composite carries it); (3) one workgroup per
// query sorts its ~2 000 candidates in LDS (bitonic) and writes the first k.  If a query ends with
// fewer than k or more than TKS_CAP candidates -- astronomically unlikely with jittered samples,
// but correctness does not rest on luck -- its workgroup finds the exact k-th composite by a 64-bit
// radix select over the query's scores and collects exactly k elements.
// 1 read of the scores instead of the 3 of the radix-select path, and no 4 096-element slack to sort.
// ---------------------------------------------------------------------------
constexpr int TKS_SAMPLES = 4096;
constexpr int TKS_CAP = 16384;

__device__ __forceinline__ uint64_t tk_comp(float s, uint32_t i) { return ((uint64_t)desc_key(s) << 32) | i; }

// Bin of a 256-bin histogram that holds the `rem`-th element (1-based), by ONE wave: lane l owns
// bins 4l..4l+3, a shuffle scan gives the cumulative counts; `rem` is reduced to the rank inside
// that bin.  (A single thread walking the bins costs 256 dependent LDS reads per level.)
__device__ __forceinline__ uint32_t find_bin(const uint32_t *hist, uint32_t &rem, int lane)
{
    const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const uint32_t mine = h0 + h1 + h2 + h3;
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    const uint64_t reached = __ballot(inc >= rem);                   // lanes whose cumulative count reaches rem
    const int owner = __builtin_ctzll(reached);
    uint32_t before = __shfl(inc - mine, owner, 64);                 // elements in the bins of lower lanes
    const uint32_t a0 = __shfl(h0, owner, 64), a1 = __shfl(h1, owner, 64), a2 = __shfl(h2, owner, 64);
    uint32_t r = rem - before, b = 4u * (uint32_t)owner;
    if (r > a0) { r -= a0; ++b; if (r > a1) { r -= a1; ++b; if (r > a2) { r -= a2; ++b; } } }
    rem = r;
    return b;
}

// k-th smallest (1-based) of the distinct 64-bit values buf[0..cnt) in LDS: radix select, 8 levels of
// 8 bits from the top (a 256-bin LDS histogram per level).  All threads of the workgroup call it;
// hist / state are caller-provided shared scratch.  24 barriers instead of the ~80 of a full sort.
__device__ __forceinline__ uint64_t lds_select_kth(const uint64_t *buf, int cnt, uint32_t kth, uint32_t *hist,
                                                   uint64_t *s_prefix, uint32_t *s_remaining, int tid, int nthreads)
{
    if (tid == 0) { *s_prefix = 0; *s_remaining = kth; }
    for (int level = 0; level < 8; ++level) {
        const int shift = 56 - 8 * level;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = *s_prefix;
        for (int i = tid; i < cnt; i += nthreads) {
            const uint64_t c = buf[i];
            if (level == 0 || (c >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(c >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            uint32_t rem = *s_remaining;
            const uint32_t b = find_bin(hist, rem, tid);
            if (tid == 0) {
                *s_remaining = rem;
                *s_prefix = prefix | ((uint64_t)b << shift);
            }
        }
        __syncthreads();
    }
    return *s_prefix;
}

__global__ __launch_bounds__(1024) void tks_sample_kernel(const float *__restrict__ scores, int64_t n, int m,
                                                          uint64_t *__restrict__ thr, uint32_t *__restrict__ ovf_count)
{
    __shared__ uint64_t s[TKS_SAMPLES];
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    const uint32_t stride = (uint32_t)(n / TKS_SAMPLES);
    for (int j = tid; j < TKS_SAMPLES; j += 1024) {
        const uint32_t jitter = (((uint32_t)j * 2654435761u) ^ ((uint32_t)q * 40503u + 0x9E3779B9u)) >> 9;
        const uint32_t i = (uint32_t)j * stride + jitter % stride;
        s[j] = tk_comp(scores[q * n + i], i);
    }
    __shared__ uint32_t hist[256];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_remaining;
    __syncthreads();
    const uint64_t T = lds_select_kth(s, TKS_SAMPLES, (uint32_t)m, hist, &s_prefix, &s_remaining, tid, 1024);
    if (tid == 0) {
        thr[q] = T;
        ovf_count[q] = 0;
    }
}

// (2) no global atomics: device-scope atomics on one address serialise across the XCDs (~1 us each;
// 2 000 per query were 1.3 ms).  Every 4096-element tile owns TKS_SLOTS candidate slots and a count.
constexpr int TKS_SLOTS = 128;

__global__ __launch_bounds__(256) void tks_compact_kernel(const float *__restrict__ scores, int64_t n, int nblk,
                                                          const uint64_t *__restrict__ thr,
                                                          uint32_t *__restrict__ tile_count, uint64_t *__restrict__ cand,
                                                          uint32_t *__restrict__ ovf_count, uint64_t *__restrict__ ovf)
{
    __shared__ uint32_t lcount;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t q = blockIdx.y, t0 = (int64_t)blockIdx.x * 4096;
    const uint64_t T = thr[q];
    if (tid == 0) lcount = 0;
    uint64_t c[16];
    // a lane takes 4 consecutive rows per 16-byte load (dword-aligned); whole tiles without a range check, all loads in flight
    // together (which lane holds which row does not matter: the composite carries the order)
    if (t0 + 4096 <= n) {
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t *row = (const uint32_t *)scores + q * n + t0;
        u32x4u w[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) w[v] = *(const u32x4u *)(row + (v * 256 + tid) * 4);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[v * 4 + j] = tk_comp(__uint_as_float(w[v][j]), (uint32_t)(t0 + (v * 256 + tid) * 4 + j));
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t i = t0 + e * 256 + tid;
            c[e] = i < n ? tk_comp(scores[q * n + i], (uint32_t)i) : ~0ull;
        }
    }
    __syncthreads();
    uint64_t *mine = cand + (q * nblk + blockIdx.x) * TKS_SLOTS;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const bool pass = c[e] <= T;
        const uint64_t mask = __ballot(pass);
        if (mask) {                                                  // uniform; rare (a fraction ~E/n of the elements pass)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&lcount, (uint32_t)__popcll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (pass && pos < (uint32_t)TKS_SLOTS) mine[pos] = c[e];
            // a tile with more candidates than slots (the best rows of a query are often neighbours:
            // a labelled set stored together) spills into the query's overflow list -- global atomics,
            // but only for the excess of such tiles
            const uint64_t spill = __ballot(pass && pos >= (uint32_t)TKS_SLOTS);
            if (spill) {
                uint32_t obase = 0;
                if (lane == 0) obase = atomicAdd(&ovf_count[q], (uint32_t)__popcll(spill));
                obase = __builtin_amdgcn_readfirstlane(obase);
                const uint32_t opos = obase + __builtin_amdgcn_mbcnt_hi((uint32_t)(spill >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)spill, 0u));
                if (pass && pos >= (uint32_t)TKS_SLOTS && opos < (uint32_t)TKS_CAP) ovf[q * TKS_CAP + opos] = c[e];
            }
        }
    }
    __syncthreads();
    if (tid == 0) tile_count[q * nblk + blockIdx.x] = lcount < (uint32_t)TKS_SLOTS ? lcount : (uint32_t)TKS_SLOTS;
}

__global__ __launch_bounds__(1024) void tks_finish_kernel(const float *__restrict__ scores, int64_t n, int nblk, int k,
                                                          int64_t id_offset, const uint32_t *__restrict__ tile_count,
                                                          const uint64_t *__restrict__ cand,
                                                          const uint32_t *__restrict__ ovf_count, const uint64_t *__restrict__ ovf,
                                                          int64_t *__restrict__ top_ids, float *__restrict__ top_scores)
{
    extern __shared__ uint64_t buf[];                                // TKS_CAP composites + 1024 kept ones
    __shared__ uint32_t hist[256];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_remaining, s_fill, s_total, s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    if (tid == 0) { s_total = 0; s_bad = 0; s_fill = 0; }
    __syncthreads();
    // gather the per-tile lists (wave per tile; list order is irrelevant: the composite carries the order)
    const uint32_t novf = ovf_count[q];
    if (tid == 0) {
        if (novf > (uint32_t)TKS_CAP) s_bad = 1;
        s_total = novf;                                              // the overflow list goes first
    }
    __syncthreads();
    if (novf <= (uint32_t)TKS_CAP)
        for (uint32_t i = tid; i < novf; i += 1024) buf[i] = ovf[q * TKS_CAP + i];
    for (int t = wave; t < nblk; t += 16) {
        const uint32_t c = tile_count[q * nblk + t];
        uint32_t off = 0;
        if (lane == 0) off = atomicAdd(&s_total, c);
        off = __builtin_amdgcn_readfirstlane(off);
        if (off + c <= (uint32_t)TKS_CAP)
            for (uint32_t i = lane; i < c; i += 64) buf[off + i] = cand[(q * nblk + t) * TKS_SLOTS + i];
    }
    __syncthreads();
    uint32_t cnt = s_total;
    if (s_bad || cnt < (uint32_t)k || cnt > (uint32_t)TKS_CAP) {
        // exact fallback: k-th smallest composite by radix select (8 levels of 8 bits, MSB first)
        if (tid == 0) { s_prefix = 0; s_remaining = (uint32_t)k; }
        for (int level = 0; level < 8; ++level) {
            const int shift = 56 - 8 * level;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint64_t prefix = s_prefix;
            for (int64_t i = tid; i < n; i += 1024) {
                const uint64_t c = tk_comp(scores[q * n + i], (uint32_t)i);
                if (level == 0 || (c >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(c >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                uint32_t rem = s_remaining;
                const uint32_t b = find_bin(hist, rem, tid);        // the k-th lies in bin b
                if (tid == 0) {
                    s_remaining = rem;
                    s_prefix = prefix | ((uint64_t)b << shift);
                }
            }
            __syncthreads();
        }
        const uint64_t T = s_prefix;                                 // exactly k composites are <= T
        for (int64_t i = tid; i < n; i += 1024) {
            const uint64_t c = tk_comp(scores[q * n + i], (uint32_t)i);
            if (c <= T) buf[atomicAdd(&s_fill, 1u)] = c;
        }
        __syncthreads();
    } else if (cnt > (uint32_t)k) {
        // keep only the k smallest candidates: select the k-th, then compact them into the tail of
        // the buffer (free: cnt + k <= TKS_CAP + 1024 slots are allocated) and move them to the front
        const uint64_t T = lds_select_kth(buf, (int)cnt, (uint32_t)k, hist, &s_prefix, &s_remaining, tid, 1024);
        uint64_t *keep = buf + TKS_CAP;
        for (uint32_t i = tid; i < cnt; i += 1024) {
            const uint64_t c = buf[i];
            if (c <= T) keep[atomicAdd(&s_fill, 1u)] = c;
        }
        __syncthreads();
        for (int i = tid; i < k; i += 1024) buf[i] = keep[i];
        __syncthreads();
    }
    cnt = (uint32_t)k;
    int P = 2;
    while (P < (int)cnt) P <<= 1;
    for (int i = (int)cnt + tid; i < P; i += 1024) buf[i] = ~0ull;
    __syncthreads();
    bitonic_sort_lds(buf, P, tid, 1024);
    for (int i = tid; i < k; i += 1024) {
        const uint32_t row = (uint32_t)buf[i];
        if (top_ids) top_ids[q * k + i] = (int64_t)row + id_offset;
        if (top_scores) top_scores[q * k + i] = scores[q * n + row];
    }
}

static int64_t sampled_workspace(int64_t n, int64_t nq)
{
    const int64_t nblk = ceil_div(n, (int64_t)4096);
    return round_up(nq * 8, 256) + round_up(nq * 4, 256) + round_up(nq * nblk * 4, 256) +
           nq * nblk * (int64_t)TKS_SLOTS * 8 + nq * (int64_t)TKS_CAP * 8;
}

static int topk_sampled(const float *scores, int64_t n, int64_t nq, int64_t k, int64_t id_offset,
                        int64_t *top_ids, float *top_scores, void *workspace, hipStream_t s)
{
    const int64_t nblk = ceil_div(n, (int64_t)4096);
    char *base = (char *)workspace;
    uint64_t *thr = (uint64_t *)base;
    uint32_t *ovf_count = (uint32_t *)(base + round_up(nq * 8, 256));
    uint32_t *tile_count = (uint32_t *)((char *)ovf_count + round_up(nq * 4, 256));
    uint64_t *cand = (uint64_t *)((char *)tile_count + round_up(nq * nblk * 4, 256));
    uint64_t *ovf = cand + nq * nblk * (int64_t)TKS_SLOTS;
    const int64_t expect = 6 * k > 2048 ? 6 * k : 2048;              // candidates aimed at
    int64_t m = ceil_div(expect * TKS_SAMPLES, n);
    m = m < 9 ? 9 : m;
    auto finish = tks_finish_kernel;
    MDX_HIP(hipFuncSetAttribute((const void *)finish, hipFuncAttributeMaxDynamicSharedMemorySize, (TKS_CAP + 1024) * 8));
    hipLaunchKernelGGL(tks_sample_kernel, dim3((unsigned)nq), dim3(1024), 0, s, scores, n, (int)m, thr, ovf_count);
    hipLaunchKernelGGL(tks_compact_kernel, dim3((unsigned)nblk, (unsigned)nq), dim3(256), 0, s, scores, n, (int)nblk,
                       (const uint64_t *)thr, tile_count, cand, ovf_count, ovf);
    hipLaunchKernelGGL(finish, dim3((unsigned)nq), dim3(1024), (TKS_CAP + 1024) * 8, s, scores, n, (int)nblk, (int)k, id_offset,
                       (const uint32_t *)tile_count, (const uint64_t *)cand, (const uint32_t *)ovf_count,
                       (const uint64_t *)ovf, top_ids, top_scores);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}


}  // namespace mdx

using namespace mdx;

namespace mdx {

// ---------------------------------------------------------------------------
// Positions of labelled ids INSIDE a given ranking: pos[t] = p with ranks[q][p] == ids[t] (or -1) -- what
// `np.arange(N)[np.in1d(ranks[:, q], ids)]` (evaluate.py:80-81) yields, for all queries in one pass over the ranking.
// The host-side form of this (one torch.isin + nonzero + copy per query, protocol level and list) took 72 ms for
// 70 x 1 M -- twenty times the similarity + ranking kernels it follows (tools/eval_path_bench.py).
// A workgroup hashes its query's ids (a chunk of up to RP_CHUNK of them; longer lists take another sweep of the slice)
// into LDS -- open addressing, the table at least twice as large as the chunk -- and streams a slice of the ranking row through it with 16-byte loads: HBM-bound, 8 B per element.
// Ids are non-negative (an entry < 0 is never looked up); of an id listed twice, either entry receives the position
// and the other stays -1 (the callers pass unique lists).
// ---------------------------------------------------------------------------
constexpr int RP_CHUNK = 512, RP_TABLE = 1024, RP_ROWS_PER_BLOCK = 16384;      // 12 KiB of LDS: a dozen workgroups per CU

__global__ __launch_bounds__(256) void rank_positions_kernel(const int64_t *__restrict__ ranks, int64_t ld, int64_t n,
                                                             const int64_t *__restrict__ ids, const int64_t *__restrict__ offsets,
                                                             int64_t *__restrict__ pos)
{
    __shared__ int64_t key[RP_TABLE];
    __shared__ int32_t slot[RP_TABLE];
    const int q = blockIdx.y;
    const int64_t lo = offsets[q], hi = offsets[q + 1];
    const int64_t p0 = (int64_t)blockIdx.x * RP_ROWS_PER_BLOCK, p1 = (p0 + RP_ROWS_PER_BLOCK) < n ? (p0 + RP_ROWS_PER_BLOCK) : n;
    const int64_t *row = ranks + (int64_t)q * ld;
    for (int64_t c0 = lo; c0 < hi; c0 += RP_CHUNK) {
        const int cnt = (int)((hi - c0) < RP_CHUNK ? (hi - c0) : RP_CHUNK);
        int tsize = 64;                                                         // at least twice the chunk: short probe chains; a few
        while (tsize < 2 * cnt) tsize <<= 1;                                    // dozen ids (rOxford) leave almost nothing to clear
        const uint32_t mask = (uint32_t)tsize - 1;
        for (int i = threadIdx.x; i < tsize; i += 256) key[i] = -1;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) {
            const int64_t id = ids[c0 + i];
            if (id < 0) continue;
            uint32_t h = (uint32_t)(((uint64_t)id * 0x9E3779B97F4A7C15ull) >> 40) & mask;
            while (true) {
                const unsigned long long old = atomicCAS((unsigned long long *)&key[h], ~0ull, (unsigned long long)id);
                if (old == ~0ull) { slot[h] = i; break; }
                if ((int64_t)old == id) break;                                  // listed twice: the first entry keeps the slot
                h = (h + 1) & mask;
            }
        }
        __syncthreads();
        auto look = [&](int64_t id, int64_t p) {
            if (id < 0) return;
            uint32_t h = (uint32_t)(((uint64_t)id * 0x9E3779B97F4A7C15ull) >> 40) & mask;
            while (true) {
                const int64_t k = key[h];
                if (k == id) { pos[c0 + slot[h]] = p; return; }
                if (k == -1) return;
                h = (h + 1) & mask;
            }
        };
        int64_t p = p0 + 2 * threadIdx.x;
        if ((((uintptr_t)row) & 15) == 0) {                                     // p0 is even: pairs are 16-byte aligned with the row
            for (; p + 3 * 512 + 1 < p1; p += 4 * 512) {                       // four loads in flight per lane
                typedef long long i64x2 __attribute__((ext_vector_type(2)));
                i64x2 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const i64x2 *)(row + p + u * 512));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    look(v[u].x, p + u * 512);
                    look(v[u].y, p + u * 512 + 1);
                }
            }
            for (; p + 1 < p1; p += 512) {
                const longlong2 v = *(const longlong2 *)(row + p);
                look(v.x, p);
                look(v.y, p + 1);
            }
            if (p < p1) look(row[p], p);
        } else {
            for (; p < p1; p += 512) {
                look(row[p], p);
                if (p + 1 < p1) look(row[p + 1], p + 1);
            }
        }
        __syncthreads();
    }
}

}  // namespace mdx

extern "C" {

int64_t mdx_rank_workspace(int64_t n, int64_t nq)
{
    if (n <= 0 || nq <= 0) return 0;
    return carve(nullptr, nullptr, n, nq);
}

int mdx_rank_full(const float *scores, int64_t n, int64_t nq, int64_t id_offset, int64_t *ranks,
                  void *workspace, int64_t workspace_bytes, void *stream)
{
    MDX_CHECK_ARG(ranks, "mdx_rank_full: NULL ranks");
    return rank_impl(scores, n, nq, id_offset, ranks, nullptr, n, workspace, workspace_bytes,
                     (hipStream_t)stream, "mdx_rank_full");
}

int mdx_rank_full_segments(const float *const *blocks, const int64_t *widths, int nblocks, int64_t nq, int64_t id_offset,
                           int64_t *ranks, void *workspace, int64_t workspace_bytes, void *stream)
{
    MDX_CHECK_ARG(blocks && widths && ranks, "mdx_rank_full_segments: NULL pointer");
    MDX_CHECK_ARG(nblocks >= 1 && nblocks <= MAX_SEG, "mdx_rank_full_segments: %d blocks, 1..%d supported", nblocks, MAX_SEG);
    SegTable seg;
    memset(&seg, 0, sizeof seg);
    int64_t n = 0;
    for (int g = 0; g < nblocks; ++g) {
        MDX_CHECK_ARG(widths[g] >= 0 && (widths[g] == 0 || blocks[g]), "mdx_rank_full_segments: block %d is NULL or negative", g);
        if (widths[g] == 0) continue;          // an empty shard contributes no columns
        seg.p[seg.nseg] = blocks[g];
        seg.start[seg.nseg] = n;
        n += widths[g];
        seg.start[++seg.nseg] = n;
    }
    MDX_CHECK_ARG(n > 0, "mdx_rank_full_segments: no columns");
    return rank_impl(nullptr, n, nq, id_offset, ranks, nullptr, n, workspace, workspace_bytes, (hipStream_t)stream,
                     "mdx_rank_full_segments", &seg);
}

int mdx_topk(const float *scores, int64_t n, int64_t nq, int64_t k, int64_t id_offset,
             int64_t *top_ids, float *top_scores, void *workspace, int64_t workspace_bytes,
             void *stream)
{
    MDX_CHECK_ARG(k > 0 && k <= n, "mdx_topk: k=%lld out of range (n=%lld)", (long long)k,
                  (long long)n);
    MDX_CHECK_ARG(top_ids || top_scores, "mdx_topk: both outputs NULL");
    MDX_CHECK_ARG(scores && n > 0 && nq > 0 && n < (1ll << 32) && nq < 65536, "mdx_topk: bad sizes");
    const int64_t need = carve(nullptr, nullptr, n, nq);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_topk: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    // k <<< n (serving): sampled threshold, one pass over the scores
    if (n >= 16384 && k <= 1024 && 256 * k <= n && sampled_workspace(n, nq) <= workspace_bytes && !getenv("MDX_NO_SAMPLED_TOPK"))
        return topk_sampled(scores, n, nq, k, id_offset, top_ids, top_scores, workspace, (hipStream_t)stream);
    // k << n: radix select + sort of the candidates; otherwise the full ranking, trimmed in its last pass
    if (4 * (k + SEL_CAP) <= n && carve_select(nullptr, nullptr, n, nq, k) <= workspace_bytes)
        return topk_select(scores, n, nq, k, id_offset, top_ids, top_scores, workspace, (hipStream_t)stream);
    return rank_impl(scores, n, nq, id_offset, top_ids, top_scores, k, workspace, workspace_bytes,
                     (hipStream_t)stream, "mdx_topk");
}

int mdx_gather_scores(const float *scores, int64_t n, int64_t nq, const int64_t *ids,
                      const int64_t *offsets, int64_t total, float *out, void *stream)
{
    MDX_CHECK_ARG(scores && ids && offsets && out, "mdx_gather_scores: NULL pointer");
    MDX_CHECK_ARG(n > 0 && nq > 0 && total >= 0, "mdx_gather_scores: bad sizes");
    if (total == 0) return MDX_OK;
    hipLaunchKernelGGL(gather_scores_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, scores, n, nq, ids, offsets, total, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_rank_count(const float *scores, int64_t n, int64_t nq, int64_t id_offset,
                   const float *ref_scores, const int64_t *ref_ids, const int64_t *offsets,
                   int64_t total, int64_t *cnt, void *stream)
{
    MDX_CHECK_ARG(scores && ref_scores && ref_ids && offsets && cnt, "mdx_rank_count: NULL pointer");
    MDX_CHECK_ARG(n > 0 && nq > 0 && total >= 0 && nq < 65536, "mdx_rank_count: bad sizes");
    if (total == 0) return MDX_OK;
    MDX_CHECK_ARG(n + id_offset < (1ll << 32) && id_offset >= 0, "mdx_rank_count: ids must fit 32 bits");
    const int64_t nblk = ceil_div(n, CNT_TILE);
    // enough workgroups to fill the chip (~8 per CU), each striding over its share of the tiles
    int64_t gx = ceil_div((int64_t)2048, nq);
    gx = gx < 1 ? 1 : (gx > nblk ? nblk : gx);
    const dim3 grid((unsigned)gx, (unsigned)nq);
    hipLaunchKernelGGL(rank_count_bsearch_kernel, grid, dim3(256), 0, (hipStream_t)stream, scores, n,
                       id_offset, ref_scores, ref_ids, offsets, (unsigned long long *)cnt, (int)nblk);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_rank_of(const float *scores, int64_t n, int64_t nq, const int64_t *ids,
                const int64_t *offsets, int64_t total, float *id_scores, int64_t *pos, void *stream)
{
    MDX_CHECK_ARG(scores && ids && offsets && pos && id_scores, "mdx_rank_of: NULL pointer");
    if (total == 0) return MDX_OK;
    int rc = mdx_gather_scores(scores, n, nq, ids, offsets, total, id_scores, stream);
    if (rc != MDX_OK) return rc;
    MDX_HIP(hipMemsetAsync(pos, 0, (size_t)total * sizeof(int64_t), (hipStream_t)stream));
    return mdx_rank_count(scores, n, nq, 0, id_scores, ids, offsets, total, pos, stream);
}


int mdx_rank_positions(const int64_t *ranks, int64_t n, int64_t nq, int64_t ld, const int64_t *ids, const int64_t *offsets,
                       int64_t total, int64_t *pos, void *stream)
{
    MDX_CHECK_ARG(ranks && ids && offsets && pos, "mdx_rank_positions: NULL pointer");
    MDX_CHECK_ARG(n > 0 && nq > 0 && nq < 65536 && ld >= n && total >= 0, "mdx_rank_positions: n=%lld nq=%lld ld=%lld total=%lld", (long long)n,
                  (long long)nq, (long long)ld, (long long)total);
    if (total == 0) return MDX_OK;
    hipStream_t s = (hipStream_t)stream;
    MDX_HIP(hipMemsetAsync(pos, 0xFF, (size_t)total * sizeof(int64_t), s));        // -1: not in the ranking
    hipLaunchKernelGGL(mdx::rank_positions_kernel, dim3((unsigned)ceil_div(n, (int64_t)mdx::RP_ROWS_PER_BLOCK), (unsigned)nq), dim3(256), 0, s, ranks, ld, n, ids,
                       offsets, pos);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
