// One-sweep form of the ranking for large problems (included by mdx_rank.hip).
//
// The 4-pass LSD sort of mdx_rank.hip reads every key TWICE per pass (per-tile histogram, then scatter) and
// runs three kernels per pass.  Here the digit histograms of all four passes come from ONE read of the
// scores, and a pass is ONE kernel: a tile ranks its elements, publishes its per-digit counts and obtains
// the number of equal-digit elements in the tiles before it by a decoupled look-back over the earlier
// tiles' published words (Merrill & Garland's single-pass scan, as in "onesweep" radix sorts), then
// scatters.  Bytes per element: 4 (histograms) + 12 + 16 + 12|16 + 12|16 = 56-64 instead of 76, and 5
// launches instead of 12.
//
//  * Tile order.  Work = (query, tile) items in eight lists: list c holds the queries q = c (mod 8), tiles in
//    ascending order, its queries interleaved.  A workgroup draws the next item of a list with ONE relaxed
//    agent-scope atomic, so a tile only ever waits for tiles with smaller tickets of the same list -- held by
//    workgroups that are resident or done: no assumption on dispatch order or placement.  Which list a
//    workgroup draws from first is its XCD id (HW_REG_XCC_ID; an exhausted list sends it to the next one):
//    speed only -- all tiles of a query then run on one XCD, and neighbouring tiles, which end their digit
//    runs in the same cache lines, meet in ONE L2 (the XCD-aware tile order of the three-kernel passes was
//    worth 0.3 ms of 1.66; a plain per-query ticket lost it again: pass 1 388 us against 251 us).
//  * Hand-off.  One 32-bit word per (tile, digit): bits 31-30 = state (0 nothing, 1 tile count, 2 inclusive
//    prefix), bits 29-28 = pass, bits 27-0 = value.  The word IS the flag (MI355X guide, Guideline 16 R2):
//    relaxed agent-scope stores / loads (sc1), no fences.  A word of another pass reads as "nothing", so one
//    array serves the four passes; it and the tickets are zeroed by the histogram kernel of the same call.
//  * Keys and ids travel as 8-byte pairs (one 128-B run per digit and tile instead of two 64-B runs); when the
//    ids fit 24 bits the third pass writes (top key byte : id) in ONE word and the last pass reads 4 bytes.
//  * Every spin is bounded; a look-back that gives up raises a flag in the workspace (mdx_rank_status).
#pragma once

namespace mdx {

constexpr uint32_t OS_STATE_COUNT = 1u << 30, OS_STATE_PREFIX = 2u << 30, OS_VALUE = (1u << 28) - 1u;
constexpr int OS_LISTS = 8;                 // work lists (= XCDs of an MI355X; any number is correct)
constexpr uint32_t OS_SPIN_LIMIT = 1u << 24;
constexpr int OS_HIST_TILES = 8;            // tiles per workgroup of the histogram kernel
constexpr int OS_HIST_COPIES = 4;           // lane-striped LDS copies (the top byte of cosine scores is concentrated)

struct OsWs {
    uint2 *pairs[2];            // [nq][n] (key, id)
    uint32_t *digit_tot;        // [nq][4][256]
    uint32_t *look;             // [nq][nblk][256]
    uint32_t *ticket;           // [4][OS_LISTS]
    uint32_t *status;           // [4]: nonzero = a look-back gave up
    int nblk;
};

static int64_t os_carve(OsWs *ws, char *base, int64_t n, int64_t nq)
{
    const int64_t nblk = ceil_div(n, SORT_TILE);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char *p = base ? base + off : nullptr; off += round_up(bytes, 256); return p; };
    char *p;
    p = take(n * nq * 8);               if (ws) ws->pairs[0] = (uint2 *)p;
    p = take(n * nq * 8);               if (ws) ws->pairs[1] = (uint2 *)p;
    p = take(nq * 4 * RADIX * 4);       if (ws) ws->digit_tot = (uint32_t *)p;
    p = take(nq * nblk * RADIX * 4);    if (ws) ws->look = (uint32_t *)p;
    p = take(4 * OS_LISTS * 4);         if (ws) ws->ticket = (uint32_t *)p;
    p = take(64);                       if (ws) ws->status = (uint32_t *)p;
    if (ws) ws->nblk = (int)nblk;
    return off;
}

// One read of the scores: the query's histogram of each of the four key bytes (global atomics, one per
// workgroup and bin), and the zeroing of this call's look-back words and tickets.
__global__ __launch_bounds__(SORT_THREADS) void os_hist_kernel(const float *__restrict__ scores, int64_t n, int nblk,
                                                              uint32_t *__restrict__ digit_tot, uint32_t *__restrict__ look,
                                                              uint32_t *__restrict__ ticket, uint32_t *__restrict__ status)
{
    __shared__ uint32_t h[4][OS_HIST_COPIES][RADIX + 1];
    const int tid = threadIdx.x, copy = tid & (OS_HIST_COPIES - 1);
    // newest rows first: the similarity kernel has just written them, the tail of its output is still in the Infinity Cache
    const int64_t q = (int64_t)gridDim.y - 1 - blockIdx.y, b0 = ((int64_t)gridDim.x - 1 - blockIdx.x) * OS_HIST_TILES;
    for (int e = tid; e < 4 * OS_HIST_COPIES * (RADIX + 1); e += SORT_THREADS) (&h[0][0][0])[e] = 0;
    // look-back words of tiles b0 .. b0+OS_HIST_TILES-1 of this query, all four passes
    {
        const int nb = (int)((nblk - b0) < OS_HIST_TILES ? (nblk - b0) : OS_HIST_TILES);
        const uint4 z = {0u, 0u, 0u, 0u};
        uint4 *dst = (uint4 *)(look + (q * nblk + b0) * RADIX);
        for (int e = tid; e < nb * (RADIX / 4); e += SORT_THREADS) dst[e] = z;
        if (blockIdx.x == 0 && q == 0 && tid < 4 * OS_LISTS) ticket[tid] = 0;
        if (blockIdx.x == 0 && q == 0 && tid < 16) status[tid] = 0;
    }
    __syncthreads();
    const int64_t base = q * n;
    for (int t = 0; t < OS_HIST_TILES; ++t) {
        if (b0 + t >= nblk) break;
        uint32_t k[SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const int64_t i = (b0 + t) * SORT_TILE + r * SORT_THREADS + tid;
            k[r] = i < n ? desc_key(scores[base + i]) : 0u;
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const int64_t i = (b0 + t) * SORT_TILE + r * SORT_THREADS + tid;
            if (i < n) {
                atomicAdd(&h[0][copy][k[r] & 255u], 1u);
                atomicAdd(&h[1][copy][(k[r] >> 8) & 255u], 1u);
                atomicAdd(&h[2][copy][(k[r] >> 16) & 255u], 1u);
                atomicAdd(&h[3][copy][k[r] >> 24], 1u);
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < 4 * RADIX; e += SORT_THREADS) {
        const int p = e / RADIX, d = e % RADIX;
        uint32_t tot = 0;
#pragma unroll
        for (int c = 0; c < OS_HIST_COPIES; ++c) tot += h[p][c][d];
        if (tot) atomicAdd(&digit_tot[(q * 4 + p) * RADIX + d], tot);
    }
}

// One pass.  PASS 0 reads the scores, PASS 3 writes the ranking; IN32 / OUT32: the one-word (key byte : id) form.
// Persistent workgroups (grid = 3 per CU): each draws items until the lists are empty, the NEXT item's ticket
// is drawn while the current tile is processed, and the look-back words of the three tiles before the current one
// are requested before the tile is ranked -- both latencies (1-2 us each under load) then hide behind the tile's
// own work.  (One prefetched ticket per workgroup keeps the "waits only for earlier tickets of resident
// workgroups" argument: a cycle of waits would need ticket times to increase all the way around it.)
#ifndef MDX_OS_EARLY
#define MDX_OS_EARLY 3
#endif
constexpr int OS_EARLY = MDX_OS_EARLY;      // look-back words requested before the tile is ranked

template <int PASS>
__device__ __forceinline__ uint2 os_draw(uint32_t *ticket, int nq, int nblk, int lane)
{
    // Called by ONE whole wave.  Next item of "my" list, or of the first later list that still has one.  "My" list =
    // XCD id (HW_REG_XCC_ID): speed only.  The last pass writes long runs (few distinct top bytes) and does not gain
    // from one-XCD-per-query; it spreads every query over all XCDs instead (measured: 228 against 262 us).
    // The ticket is drawn by lane 0 and broadcast, so that everything here is scalar: left to one divergent lane
    // the compiler precomputes the eight lists' addresses and sizes in vector registers ahead of the tile loop.
    const int first = PASS == 3 ? (int)(blockIdx.x % OS_LISTS) : (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
    for (int k = 0; k < OS_LISTS; ++k) {
        const int list = (first + k) % OS_LISTS;
        const uint32_t items = (uint32_t)((nq - list + OS_LISTS - 1) / OS_LISTS) * (uint32_t)nblk;
        if (items == 0) continue;
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ticket + PASS * OS_LISTS + list, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t < items) return make_uint2((uint32_t)list, t);
    }
    return make_uint2(0u, 0xFFFFFFFFu);
}

template <int PASS, bool IN32, bool OUT32>
__global__ __launch_bounds__(SORT_THREADS, (3 * SORT_WAVES) / 4) void os_pass_kernel(
    const float *__restrict__ scores, const void *__restrict__ in, void *__restrict__ outp, int64_t *__restrict__ ranks,
    float *__restrict__ top_scores, int64_t n, int nq, int nblk, const uint32_t *__restrict__ digit_tot, uint32_t *__restrict__ look,
    uint32_t *__restrict__ ticket, uint32_t *__restrict__ status, int64_t id_offset, int64_t klimit)
{
    constexpr bool FIRST = PASS == 0, LAST = PASS == 3;
    constexpr int shift = IN32 ? 24 : 8 * PASS;
    constexpr uint32_t TAG = (uint32_t)PASS << 28;
    __shared__ uint32_t dtot[RADIX];        // per-query digit totals -> digit bases
    __shared__ uint32_t wcnt[SORT_WAVES][RADIX];
    __shared__ uint32_t gdelta[RADIX];      // global position of a digit run minus its tile offset
    __shared__ uint32_t scan[RADIX];
    __shared__ uint32_t skey[SORT_TILE];
    __shared__ uint32_t sval[SORT_TILE];
    __shared__ uint32_t s_item[2], s_next[2];
    __shared__ uint32_t s_early[OS_EARLY > 0 ? OS_EARLY : 1][RADIX];   // look-back words requested early, parked here (no registers held)
    if (PASS > 0 && (status[0] | status[1] | status[2]) != 0u) return;      // an earlier pass gave up: the ranking is void (flagged)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {
        const uint2 it = os_draw<PASS>(ticket, nq, nblk, threadIdx.x & 63);
        if (threadIdx.x == 0) {
            s_item[0] = it.x;
            s_item[1] = it.y;
        }
    }
    for (;;) {
        __syncthreads();
        // the thread index is made opaque per iteration: otherwise every lane-dependent address of the tile body is
        // hoisted out of this loop and kept in registers (the kernel then spills 20-100 VGPRs at 80)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const uint32_t list = __builtin_amdgcn_readfirstlane(s_item[0]), tk = __builtin_amdgcn_readfirstlane(s_item[1]);
        if (tk == 0xFFFFFFFFu) break;
        const int64_t nq_list = (nq - (int)list + OS_LISTS - 1) / OS_LISTS;
        const int64_t q = list + OS_LISTS * (int64_t)(tk % (uint32_t)nq_list), b = tk / (uint32_t)nq_list;
        uint32_t *my_look = look + (q * nblk) * RADIX;                              // [nblk][256] of this query
        for (int e = tid; e < SORT_WAVES * RADIX; e += SORT_THREADS) (&wcnt[0][0])[e] = 0;

        // per-query base pointers are wave-uniform; everything inside a query is indexed with 32 bits (n < 2^28)
        const int64_t base = q * n;
        const float *sc_q = scores + base;
        const uint2 *in2_q = (const uint2 *)in + base;
        const uint32_t *in1_q = (const uint32_t *)in + base;
        const uint32_t n32 = (uint32_t)n;
        const uint32_t tile0 = (uint32_t)b * SORT_TILE;
        const uint32_t sub0 = tile0 + wave * SUB_TILE;
        const int tile_n = (int)((n32 - tile0) < (uint32_t)SORT_TILE ? (n32 - tile0) : (uint32_t)SORT_TILE);
        uint32_t key[SORT_ITEMS], val[SORT_ITEMS], pos[SORT_ITEMS];
        uint32_t qtot = 0, early[OS_EARLY > 0 ? OS_EARLY : 1];
        if (tid < RADIX) {
            qtot = digit_tot[(q * 4 + PASS) * RADIX + tid];
#pragma unroll
            for (int e = 0; e < OS_EARLY; ++e)
                early[e] = b > e ? __hip_atomic_load(my_look + (b - 1 - e) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint32_t i = sub0 + r * 64 + lane;
            const bool valid = i < n32;
            if (FIRST) {
                key[r] = valid ? desc_key(sc_q[i]) : 0xFFFFFFFFu;
                val[r] = i;
            } else if (IN32) {
                const uint32_t w = valid ? in1_q[i] : 0xFFFFFFFFu;
                key[r] = w;                     // digit = top byte
                val[r] = w & 0x00FFFFFFu;
            } else {
                const uint2 w = valid ? in2_q[i] : make_uint2(0xFFFFFFFFu, 0u);
                key[r] = w.x;
                val[r] = w.y;
            }
        }
        // the NEXT item's ticket is drawn with the key loads in flight (it returns about when they do)
#ifndef MDX_OS_NOPREFETCH
        if (__builtin_amdgcn_readfirstlane(wave) == SORT_WAVES - 1) {
            const uint2 nx = os_draw<PASS>(ticket, nq, nblk, lane);
            if (lane == 0) {
                s_next[0] = nx.x;
                s_next[1] = nx.y;
            }
        }
#endif
        if (tid < RADIX) {
            dtot[tid] = qtot;
#pragma unroll
            for (int e = 0; e < OS_EARLY; ++e) s_early[e][tid] = early[e];
        }
        __syncthreads();                        // wcnt is zero
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const bool valid = (sub0 + r * 64 + lane) < n32;
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint64_t vmask = __ballot(valid);
            uint32_t plo = (uint32_t)vmask, phi = (uint32_t)(vmask >> 32);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const int32_t sel = (int32_t)(d << (31 - bit)) >> 31;       // ~0 where the bit is set
                const uint64_t m = __ballot(valid && sel != 0);
                plo &= ~((uint32_t)m ^ (uint32_t)sel);
                phi &= ~((uint32_t)(m >> 32) ^ (uint32_t)sel);
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t cnt = __popc(plo) + __popc(phi);
            const uint32_t old = wcnt[wave][d];
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) wcnt[wave][d] = old + cnt;
            __builtin_amdgcn_wave_barrier();
            pos[r] = old + rank;
        }
        __syncthreads();
        // per digit: tile count (published at once: later tiles wait for it) and per-wave offsets inside the digit
        uint32_t mine = 0;
        if (tid < RADIX) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) {
                const uint32_t c = wcnt[w][tid];
                wcnt[w][tid] = run;
                run += c;
            }
            mine = run;
            __hip_atomic_store(my_look + b * RADIX + tid, (b == 0 ? OS_STATE_PREFIX : OS_STATE_COUNT) | TAG | run, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            scan[tid] = run;
        }
        __syncthreads();
        // exclusive scan of the 256 tile counts and of the query's digit totals by ONE wave
        if (wave == 0) {
            const uint32_t t0 = scan[4 * lane], t1 = scan[4 * lane + 1], t2 = scan[4 * lane + 2], t3 = scan[4 * lane + 3];
            const uint32_t g0 = dtot[4 * lane], g1 = dtot[4 * lane + 1], g2 = dtot[4 * lane + 2], g3 = dtot[4 * lane + 3];
            const uint32_t tsum = t0 + t1 + t2 + t3, gsum = g0 + g1 + g2 + g3;
            uint32_t inc = tsum, ginc = gsum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(inc, o, 64), gv = __shfl_up(ginc, o, 64);
                if (lane >= o) { inc += v; ginc += gv; }
            }
            const uint32_t ex = inc - tsum, gex = ginc - gsum;
            const uint32_t toff[4] = {ex, ex + t0, ex + t0 + t1, ex + t0 + t1 + t2};
            const uint32_t goff[4] = {gex, gex + g0, gex + g0 + g1, gex + g0 + g1 + g2};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int dgt = 4 * lane + k;
                scan[dgt] = toff[k];                // first slot of the digit in the tile
                gdelta[dgt] = goff[k] - toff[k];    // + the look-back prefix below
            }
        }
        __syncthreads();
        // the tile in digit order in LDS (tile-local positions only: does not need the look-back)
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const bool valid = (sub0 + r * 64 + lane) < n32;
            if (!valid) continue;
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t lp = scan[d] + wcnt[wave][d] + pos[r];
            skey[lp] = key[r];
            sval[lp] = val[r];
        }
        // look-back, thread = digit: add the counts of the tiles before this one until a tile's inclusive prefix is
        // met.  The words requested at the top are snapshots (nothing / count / prefix): each is valid as it is.
        if (tid < RADIX && b > 0) {
            uint32_t sum = 0;
            bool done = false;
            // one step: wait until the word of tile j is this pass's count or prefix, add it, stop at a prefix
            auto step = [&](uint32_t v, int64_t j) {
                for (uint32_t spins = 0; (v >> 30) == 0u || (v & (3u << 28)) != TAG; ++spins) {      // nothing yet (or an earlier pass's word)
                    if (spins > OS_SPIN_LIMIT) {            // cannot happen (the tile's owner is resident): give up, flag
                        if (__hip_atomic_fetch_add(status + PASS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            status[4] = (uint32_t)q; status[5] = (uint32_t)j; status[6] = (uint32_t)b; status[7] = (uint32_t)tid;
                            status[8] = v; status[9] = list; status[10] = tk; status[11] = blockIdx.x;
                        }
                        v = OS_STATE_PREFIX | TAG;
                        break;
                    }
                    if (spins) __builtin_amdgcn_s_sleep(2);
#ifdef MDX_OS_POLL_RMW
                    v = __hip_atomic_fetch_or(my_look + j * RADIX + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                    v = __hip_atomic_load(my_look + j * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                }
                sum += v & OS_VALUE;
                done = (v >> 30) == 2u;
            };
            int64_t j = b - 1;
#pragma unroll
            for (int e = 0; e < OS_EARLY; ++e, --j)
                if (!done && j >= 0) step(s_early[e][tid], j);
            for (; !done && j >= 0; --j)
                step(__hip_atomic_load(my_look + j * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), j);
            __hip_atomic_store(my_look + b * RADIX + tid, OS_STATE_PREFIX | TAG | (sum + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gdelta[tid] += sum;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const int i = r * SORT_THREADS + tid;
            if (i >= tile_n) continue;
            const uint32_t k = skey[i], v = sval[i];
            const uint32_t dst = gdelta[(k >> shift) & 255u] + (uint32_t)i;
            if (dst >= n32) continue;           // only after a look-back that gave up (flagged): never write outside the query
            if (LAST) {
                if ((int64_t)dst < klimit) {
                    if (ranks) (ranks + q * klimit)[dst] = (int64_t)v + id_offset;
                    if (top_scores) (top_scores + q * klimit)[dst] = sc_q[v];
                }
            } else if (OUT32) {
                ((uint32_t *)outp + base)[dst] = (k & 0xFF000000u) | v;
            } else {
                ((uint2 *)outp + base)[dst] = make_uint2(k, v);
            }
        }
#ifdef MDX_OS_NOPREFETCH
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            const uint2 nx = os_draw<PASS>(ticket, nq, nblk, lane);
            if (lane == 0) {
                s_item[0] = nx.x;
                s_item[1] = nx.y;
            }
        }
#else
        if (tid == 0) {
            s_item[0] = s_next[0];
            s_item[1] = s_next[1];
        }
#endif
    }
}

static bool os_eligible(int64_t n, int64_t nq)
{
    // look-back values carry 28 bits; below ~8 tiles per query the classic passes are launch-bound anyway
    static const bool off = getenv("MDX_NO_ONESWEEP") != nullptr;
    return !off && n < (1ll << 28) && n >= 8 * SORT_TILE && nq < 65536;
}

static int os_rank(const float *scores, int64_t n, int64_t nq, int64_t id_offset, int64_t *ranks, float *top_scores,
                   int64_t klimit, void *workspace, hipStream_t s)
{
    OsWs ws;
    os_carve(&ws, (char *)workspace, n, nq);
    const bool pack = n <= (1ll << 24);
    MDX_HIP(hipMemsetAsync(ws.digit_tot, 0, (size_t)nq * 4 * RADIX * 4, s));
    hipLaunchKernelGGL(os_hist_kernel, dim3((unsigned)ceil_div(ws.nblk, OS_HIST_TILES), (unsigned)nq), dim3(SORT_THREADS), 0, s,
                       scores, n, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        MDX_HIP(hipGetDevice(&dev));
        MDX_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    static const int per_cu = getenv("MDX_OS_SLOTS") ? atoi(getenv("MDX_OS_SLOTS")) : 3;
    const int64_t items = nq * ws.nblk, slots = (int64_t)per_cu * cus;
    const dim3 grid((unsigned)(items < slots ? items : slots)), blk(SORT_THREADS);
    hipLaunchKernelGGL((os_pass_kernel<0, false, false>), grid, blk, 0, s, scores, (const void *)nullptr, (void *)ws.pairs[0], ranks,
                       top_scores, n, (int)nq, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status, id_offset, klimit);
    hipLaunchKernelGGL((os_pass_kernel<1, false, false>), grid, blk, 0, s, scores, (const void *)ws.pairs[0], (void *)ws.pairs[1], ranks,
                       top_scores, n, (int)nq, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status, id_offset, klimit);
    if (pack) {
        hipLaunchKernelGGL((os_pass_kernel<2, false, true>), grid, blk, 0, s, scores, (const void *)ws.pairs[1], (void *)ws.pairs[0], ranks,
                           top_scores, n, (int)nq, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status, id_offset, klimit);
        hipLaunchKernelGGL((os_pass_kernel<3, true, false>), grid, blk, 0, s, scores, (const void *)ws.pairs[0], (void *)nullptr, ranks,
                           top_scores, n, (int)nq, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status, id_offset, klimit);
    } else {
        hipLaunchKernelGGL((os_pass_kernel<2, false, false>), grid, blk, 0, s, scores, (const void *)ws.pairs[1], (void *)ws.pairs[0], ranks,
                           top_scores, n, (int)nq, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status, id_offset, klimit);
        hipLaunchKernelGGL((os_pass_kernel<3, false, false>), grid, blk, 0, s, scores, (const void *)ws.pairs[0], (void *)nullptr, ranks,
                           top_scores, n, (int)nq, ws.nblk, ws.digit_tot, ws.look, ws.ticket, ws.status, id_offset, klimit);
    }
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // namespace mdx
