// PARKED (round 2): MSD form of the full ranking -- correct (tools/msd_check.py: gaussian, ties, three values, all equal,
// NaN / inf / -0, ascending, periodic, column segments: bit-exact against the C oracle), NOT faster than the 4-pass LSD
// with packed intermediates, so it is not part of libmdx.so.  Measured at 70 x 1 004 993 (rocprofv3, us):
//     msd_sample 41 + msd_hist 109 (173 with a sorted splitter array: its binary search probes 8 addresses of ONE LDS bank
//     per step; the breadth-first tree below is conflict-free) + scan 11 + partition scatter 207 + local sort 455-480 = 848
//     against 855-870 for the LSD route (hist 77+47+34+29, scans 4 x 10, scatters 154+173+145+161).
// Where it loses: the partition costs 368 us against 241 for the LSD's first pass (sampling, the bucket search, 8 B per
// element written), and the in-LDS finish runs at 2 workgroups per CU (74 KB of LDS for up to 8192 pairs): with no sort pass
// at all (load, stage, store) it takes 0.20 ms, the first pass hides under that (0.21), every further pass adds ~0.1 ms
// (0.31 with two, 0.43 with the ~3.2 the 8-bit digits need; tools/msd_local_probe.hip).  Digits of up to 10 bits on
// key - min(key) (two passes for 91 % of the buckets) measured 0.41 ms: the scan over 1024 bins costs what the saved pass
// gains.  Persistent workgroups that prefetch the next bucket into registers while the current one is sorted were tried
// last (tools/msd_local_probe.hip, PERSIST=512): 0.77 ms against 0.43 -- on gfx9 loads and stores share ONE counter (vmcnt,
// in-order), so the wait for the prefetched loads is also a wait for every store of the bucket written just before: a wave
// that both stores results and loads its next input cannot overlap the two.  (The same holds for the persistent similarity
// kernels under tools/attic/.)
// To build it again: paste this block into mdx_rank.hip before the probe kernel, add FMT_BUCKETS to the scatter kernel
// (digit = bucket byte staged through sval, carried in the top byte of the id word; git history of round 2 has the
// patch) and the launch sequence at the end of this file into rank_impl.
#pragma once

// ---------------------------------------------------------------------------
// MSD form of the full ranking (n large, ids fit 24 bits, ordered ds_add_rtn available): TWO global round trips
// instead of four.
//   (1) msd_sample_kernel   per query, 8192 stratified samples are sorted in LDS; every 32nd becomes a splitter:
//                           255 (key, id) pairs cut the query's elements into 256 buckets of ~n/256 elements.  The
//                           order is the 64-bit composite (key : id) -- a strict total order that IS the ranking
//                           order -- so equal scores split too and bucket sizes do not depend on the distribution
//                           (they are Gamma(32)/32 of the mean: a bucket beyond 2.09 x the mean is a 1e-6 event);
//   (2) msd_hist_kernel     bucket of every element by binary search over the splitters (LDS), per-tile bucket
//                           histogram, the bucket byte of every element stored (1 B) for the scatter;
//   (3) sort_scan_kernel + sort_scatter_kernel<FMT_BUCKETS, FMT_KV>: the stable partition (as one LSD pass);
//   (4) msd_local_sort_kernel   one workgroup per (query, bucket) finishes its <= 8192 (key, id) pairs in LDS: stable
//                           LSD radix over the key bits that differ inside the bucket, ranking with ds_add_rtn, and
//                           writes the int64 ids.  A bucket beyond 8192 elements is sorted by its workgroup through
//                           global memory (bitonic on the composites): slow, correct, and next to never taken.
// Bytes per element: 4+1 (hist), 5+8 (partition), 8+8 (finish) = 34 instead of 53.
// ---------------------------------------------------------------------------
constexpr int MSD_SAMPLES = 8192, MSD_PER_BUCKET = MSD_SAMPLES / RADIX;
constexpr int LS_THREADS = 512, LS_WAVES = LS_THREADS / 64, LS_MAX_ITEMS = 16, LS_CAP = LS_THREADS * LS_MAX_ITEMS;
static_assert(MSD_SAMPLES == LS_CAP, "the sample sort uses the largest local sort");

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

// (1) splitters.  Sample j of a query sits at a hashed offset inside its stratum [j * step, (j + 1) * step) of
// the columns, so the samples come in ascending column order: a stable sort by key then orders them by (key, id).
__global__ __launch_bounds__(LS_THREADS) void msd_sample_kernel(const float *__restrict__ scores, int64_t n, SegTable seg,
                                                                uint32_t *__restrict__ spl_key, uint32_t *__restrict__ spl_id,
                                                                uint32_t *__restrict__ spl_tree)
{
    __shared__ LsShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const uint32_t step = (uint32_t)(n / MSD_SAMPLES);
    uint32_t key[LS_MAX_ITEMS], val[LS_MAX_ITEMS];
#pragma unroll
    for (int r = 0; r < LS_MAX_ITEMS; ++r) {
        const uint32_t j = (uint32_t)(wave * 64 * LS_MAX_ITEMS + r * 64 + lane);
        const uint32_t jitter = ((j * 2654435761u) ^ ((uint32_t)q * 40503u + 0x9E3779B9u)) >> 7;
        const uint32_t i = j * step + jitter % step;
        const float sc = seg.nseg > 0 ? *seg_elem(seg, q, i) : scores[q * n + i];
        key[r] = desc_key(sc);
        val[r] = i;
    }
    lds_radix_sort<LS_MAX_ITEMS>(key, val, MSD_SAMPLES, sh);
    if (tid < RADIX) {
        const bool last = tid == RADIX - 1;                 // the last bucket is open-ended
        spl_key[q * RADIX + tid] = last ? 0xFFFFFFFFu : sh.key[(tid + 1) * MSD_PER_BUCKET - 1];
        spl_id[q * RADIX + tid] = last ? 0xFFFFFFFFu : sh.val[(tid + 1) * MSD_PER_BUCKET - 1];
        // the 255 splitter keys as a complete search tree in breadth-first order (node 1 = the median; node i of level l,
        // position p, is splitter (2p+1) * 2^(7-l) - 1): level l of a search touches 2^l CONSECUTIVE words, where the sorted
        // array is probed at a stride of 2^(8-l) words -- eight addresses in one LDS bank
        if (tid >= 1) {
            const int level = 31 - __clz(tid), pos = tid - (1 << level);
            const int rank = (2 * pos + 1) * (128 >> level) - 1;
            spl_tree[q * RADIX + tid] = sh.key[(rank + 1) * MSD_PER_BUCKET - 1];
        } else {
            spl_tree[q * RADIX] = 0;
        }
    }
}

// bucket of composite (k : id) = number of splitters that rank before it; splitter 255 = (~0 : ~0) ranks after
// every element.  Searched on the keys alone, down the tree (8 LDS words, no bank conflicts); only an element whose
// key equals a splitter's key (an exact tie with a sampled row) goes on to compare ids inside the run of equal
// splitter keys (sk / si: the splitters in ascending order).
__device__ __forceinline__ uint32_t msd_bucket(const uint32_t *tree, const uint32_t *sk, const uint32_t *si, uint32_t k, uint32_t id)
{
    uint32_t i = 1;
#pragma unroll
    for (int l = 0; l < 8; ++l) i = 2 * i + (tree[i] < k ? 1u : 0u);
    uint32_t b = i - RADIX;                                  // splitter keys < k
    if (sk[b] == k && b < RADIX - 1) {
        uint32_t e = 1;
#pragma unroll
        for (int l = 0; l < 8; ++l) e = 2 * e + (tree[e] <= k ? 1u : 0u);
        uint32_t lo = b, hi = e - RADIX;                     // the splitters with this key: [lo, hi)
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (si[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        b = lo;
    }
    return b;
}

// (2) bucket histogram of one tile + the bucket byte of every element
__global__ __launch_bounds__(SORT_THREADS) void msd_hist_kernel(const float *__restrict__ scores, int64_t n, int64_t stride,
                                                                int nblk, SegTable seg, const uint32_t *__restrict__ spl_key,
                                                                const uint32_t *__restrict__ spl_id,
                                                                const uint32_t *__restrict__ spl_tree,
                                                                uint32_t *__restrict__ block_hist, uint8_t *__restrict__ bucket)
{
    static_assert(SORT_ITEMS == 8, "a lane takes 2 x 4 consecutive elements");
    __shared__ uint32_t h[HIST_COPIES][RADIX + 1];
    __shared__ uint32_t sk[RADIX], si[RADIX], tree[RADIX];
    const int tid = threadIdx.x, copy = tid & (HIST_COPIES - 1);
    // newest rows first, as the plain pass-0 histogram: the tail of the scores is still in the Infinity Cache
    const int64_t q = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t b0 = (int64_t)gridDim.x - 1 - blockIdx.x;
    for (int e = tid; e < HIST_COPIES * (RADIX + 1); e += SORT_THREADS) (&h[0][0])[e] = 0;
    if (tid < RADIX) {
        sk[tid] = spl_key[q * RADIX + tid];
        si[tid] = spl_id[q * RADIX + tid];
        tree[tid] = spl_tree[q * RADIX + tid];
    }
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t *p = (const uint32_t *)scores + q * n;
    u32x4u w[2];
    int64_t first[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        first[v] = b0 * SORT_TILE + (v * SORT_THREADS + tid) * 4;
        w[v] = u32x4u{0u, 0u, 0u, 0u};
    }
    bool straddles = false;
    if (seg.nseg > 0) {
        const int g = seg_of(seg, b0 * SORT_TILE);
        const int64_t tile_end = (b0 + 1) * SORT_TILE < n ? (b0 + 1) * SORT_TILE : n;
        straddles = tile_end > seg.start[g + 1];
        p = (const uint32_t *)(seg.p[g] + q * (seg.start[g + 1] - seg.start[g]) - seg.start[g]);
    }
    if (straddles) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (first[v] + j < n) w[v][j] = *(const uint32_t *)seg_elem(seg, q, first[v] + j);
    } else if ((b0 + 1) * SORT_TILE <= n) {
#pragma unroll
        for (int v = 0; v < 2; ++v) w[v] = *(const u32x4u *)(p + first[v]);
    } else {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (first[v] + j < n) w[v][j] = p[first[v] + j];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = first[v] + j;
            if (i < n) {
                const uint32_t d = msd_bucket(tree, sk, si, desc_key(__uint_as_float(w[v][j])), (uint32_t)i);
                atomicAdd(&h[copy][d], 1u);
                packed |= d << (8 * j);
            }
        }
        if (first[v] < n) *(uint32_t *)(bucket + q * stride + first[v]) = packed;       // rows are padded to the stride
    }
    __syncthreads();
    for (int e = tid; e < RADIX; e += SORT_THREADS) {
        uint32_t tot = 0;
#pragma unroll
        for (int c = 0; c < HIST_COPIES; ++c) tot += h[c][e];
        block_hist[(q * nblk + b0) * RADIX + e] = tot;
    }
}

// ascending sort of the m composites (keys[i] : ids[i]) of one bucket IN GLOBAL MEMORY by one workgroup: bitonic
// network in its all-ascending form (the first step of every merge pairs i with its mirror image in the block), so
// that the virtual +inf padding up to a power of two stays where it is -- a pair whose upper index is >= m is never
// exchanged.  The backstop for a bucket that does not fit the LDS.
__device__ void global_bitonic(uint32_t *keys, uint32_t *ids, int m)
{
    int P = 2;
    while (P < m) P <<= 1;
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const bool mirror = stride == (size >> 1);
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                int i, j;
                if (mirror) {
                    const int blk = t / stride, off = t % stride;
                    i = blk * size + off;
                    j = blk * size + size - 1 - off;
                } else {
                    i = 2 * t - (t & (stride - 1));
                    j = i + stride;
                }
                if (j >= m) continue;
                const uint64_t a = ((uint64_t)keys[i] << 32) | ids[i], b = ((uint64_t)keys[j] << 32) | ids[j];
                if (a > b) {
                    keys[i] = (uint32_t)(b >> 32); ids[i] = (uint32_t)b;
                    keys[j] = (uint32_t)(a >> 32); ids[j] = (uint32_t)a;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
}

template <int ITEMS>
__device__ __forceinline__ void msd_finish(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, int m,
                                           int64_t *__restrict__ out, int64_t id_offset, LsShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave * 64 * ITEMS + lane;
    uint32_t key[ITEMS], val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int i = sub + r * 64;
        const int j = i < m ? i : m - 1;            // every lane loads: all loads in flight together
        key[r] = kin[j];
        val[r] = vin[j];
    }
    lds_radix_sort<ITEMS>(key, val, m, sh);
    for (int i = tid; i < m; i += LS_THREADS) out[i] = (int64_t)sh.val[i] + id_offset;
}

// (4) one workgroup per (bucket, query)
__global__ __launch_bounds__(LS_THREADS, 2) void msd_local_sort_kernel(uint32_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                                                       int64_t stride, const uint32_t *__restrict__ digit_tot,
                                                                       int64_t n, int64_t id_offset, int64_t *__restrict__ ranks)
{
    __shared__ LsShared sh;
    __shared__ uint32_t s_start;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t q = blockIdx.y;
    const int b = blockIdx.x;
    if (tid < 64) {         // first element of bucket b = sum of the sizes of the buckets before it
        const uint32_t *t = digit_tot + q * RADIX + 4 * lane;
        const uint32_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
        uint32_t inc = t0 + t1 + t2 + t3;
        const uint32_t mine = inc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == (b >> 2)) {
            const uint32_t ex = inc - mine;
            s_start = ex + ((b & 3) > 0 ? t0 : 0u) + ((b & 3) > 1 ? t1 : 0u) + ((b & 3) > 2 ? t2 : 0u);
        }
    }
    __syncthreads();
    const int m = (int)digit_tot[q * RADIX + b];
    if (m == 0) return;
    const uint32_t start = s_start;
    uint32_t *kin = keys + q * stride + start, *vin = ids + q * stride + start;
    int64_t *out = ranks + q * n + start;
    if (m <= LS_THREADS * 8) msd_finish<8>(kin, vin, m, out, id_offset, sh);
    else if (m <= LS_THREADS * 12) msd_finish<12>(kin, vin, m, out, id_offset, sh);
    else if (m <= LS_CAP) msd_finish<16>(kin, vin, m, out, id_offset, sh);
    else {
        global_bitonic(kin, vin, m);
        for (int i = tid; i < m; i += LS_THREADS) out[i] = (int64_t)vin[i] + id_offset;
    }
}


/* launch sequence (rank_impl):
    // MDX_SORT_MSD=0 / 1: never / whenever possible (tests run both); default: from 256 k columns on
    static const int msd_env = getenv("MDX_SORT_MSD") ? atoi(getenv("MDX_SORT_MSD")) : -1;
    const bool msd_fits = arank && ranks && !top_scores && klimit == n && n <= (1ll << 24) && n >= 8 * MSD_SAMPLES;
    if (msd_fits && (msd_env == 1 || (msd_env != 0 && n >= (1ll << 18)))) {
        uint32_t *spl_key = (uint32_t *)ws.h[1], *spl_id = spl_key + nq * RADIX, *spl_tree = spl_id + nq * RADIX;   // h[1], w[1]: free here
        uint8_t *bucket = (uint8_t *)ws.w[1];
        const dim3 grid((unsigned)ws.nblk, (unsigned)nq), blk(SORT_THREADS);
        hipLaunchKernelGGL(msd_sample_kernel, dim3((unsigned)nq), dim3(LS_THREADS), 0, s, scores, n, seg, spl_key, spl_id, spl_tree);
        hipLaunchKernelGGL(msd_hist_kernel, grid, blk, 0, s, scores, n, ws.stride, ws.nblk, seg, (const uint32_t *)spl_key,
                           (const uint32_t *)spl_id, (const uint32_t *)spl_tree, ws.block_hist, bucket);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(RADIX / SCAN_DIGITS, (unsigned)nq), dim3(SCAN_GROUPS * SCAN_DIGITS), 0, s,
                           ws.block_hist, ws.nblk, ws.digit_tot);
        hipLaunchKernelGGL((sort_scatter_kernel<FMT_BUCKETS, FMT_KV, true>), grid, blk, 0, s, scores, (const uint32_t *)nullptr,
                           (const void *)bucket, ws.w[0], ws.h[0], (int64_t *)nullptr, (float *)nullptr, n, ws.stride, ws.nblk, 0,
                           (const uint32_t *)ws.block_hist, (const uint32_t *)ws.digit_tot, (int64_t)0, n, seg);
        hipLaunchKernelGGL(msd_local_sort_kernel, dim3(RADIX, (unsigned)nq), dim3(LS_THREADS), 0, s, ws.w[0], (uint32_t *)ws.h[0],
                           ws.stride, (const uint32_t *)ws.digit_tot, n, id_offset, ranks);
        MDX_LAUNCH_CHECK();
        return MDX_OK;
    }
*/
