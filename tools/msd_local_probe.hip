// Prototype of the in-LDS finish of an MSD-partitioned ranking: one workgroup per (query, bucket) sorts its
// ~3.9 k (key, id) pairs with a stable LSD radix whose ranking is ds_add_rtn on per-wave digit counters,
// skipping the key bytes that are constant inside the bucket.  Synthetic buckets of the real shape
// (70 queries x 1 004 993 gaussian scores, 256 buckets per query of random size ~N(mean, 17 %)), checked against
// std::sort.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/msd_local_probe.hip -o tools/msd_local_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <random>

constexpr int LS_THREADS = 512, LS_WAVES = 8;

template <int ITEMS>
__device__ __forceinline__ void local_sort(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, int m,
                                           int64_t *__restrict__ out, uint32_t *skey, uint32_t *sval, uint32_t (*cnt)[256],
                                           uint32_t *tot, uint32_t *sdiff)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave * 64 * ITEMS + lane;
    uint32_t key[ITEMS], val[ITEMS], pos[ITEMS];
    uint32_t diff = 0;
    const uint32_t k0 = kin[0];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int i = sub + r * 64;
        const int j = i < m ? i : m - 1;
        key[r] = kin[j];
        val[r] = vin[j];
        diff |= key[r] ^ k0;
    }
    if (tid == 0) *sdiff = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0) atomicOr(sdiff, diff);
    __syncthreads();
    diff = *sdiff;
#ifdef NOPASS
    diff &= NOPASS;
#endif
    bool staged = false;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        if (((diff >> shift) & 255u) == 0) continue;        // this byte is the same in every key of the bucket
        for (int e = tid; e < LS_WAVES * 256; e += LS_THREADS) (&cnt[0][0])[e] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            pos[r] = 0;
            if (sub + r * 64 < m)
                pos[r] = __hip_atomic_fetch_add(&cnt[wave][(key[r] >> shift) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < LS_WAVES; ++w) {
                const uint32_t c = cnt[w][tid];
                cnt[w][tid] = run;
                run += c;
            }
            tot[tid] = run;
        }
        __syncthreads();
        if (wave == 0) {
            const uint32_t t0 = tot[4 * lane], t1 = tot[4 * lane + 1], t2 = tot[4 * lane + 2], t3 = tot[4 * lane + 3];
            const uint32_t mine = t0 + t1 + t2 + t3;
            uint32_t inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            const uint32_t ex = inc - mine;
            tot[4 * lane] = ex; tot[4 * lane + 1] = ex + t0; tot[4 * lane + 2] = ex + t0 + t1; tot[4 * lane + 3] = ex + t0 + t1 + t2;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (sub + r * 64 >= m) continue;
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t dst = tot[d] + cnt[wave][d] + pos[r];
            skey[dst] = key[r];
            sval[dst] = val[r];
        }
        __syncthreads();
        staged = true;
        bool more = false;
        for (int p2 = pass + 1; p2 < 4; ++p2) more = more || (((diff >> (8 * p2)) & 255u) != 0);
        if (more) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int i = sub + r * 64;
                if (i < m) { key[r] = skey[i]; val[r] = sval[i]; }
            }
        }
    }
    if (!staged) {          // all keys equal: input order (ascending id) is the answer
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) if (sub + r * 64 < m) sval[sub + r * 64] = val[r];
        __syncthreads();
    }
    for (int i = tid; i < m; i += LS_THREADS) out[i] = (int64_t)sval[i];
}


// ---- variant: key - min(key), digits of up to 10 bits (u16 counters packed in pairs), so that most buckets take 2 passes ----
constexpr int WD_BINS = 1024, WD_WORDS = WD_BINS / 2;
struct WdShared {
    uint32_t key[LS_THREADS * 12];
    uint32_t val[LS_THREADS * 12];
    uint32_t cnt[LS_WAVES][WD_WORDS];
    uint32_t base[WD_WORDS];
    uint32_t kmin, kmax;
};

template <int ITEMS>
__device__ __forceinline__ void local_sort_wide(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, int m,
                                                int64_t *__restrict__ out, WdShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave * 64 * ITEMS + lane;
    uint32_t key[ITEMS], val[ITEMS], pos[ITEMS];
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int i = sub + r * 64;
        const int j = i < m ? i : m - 1;
        key[r] = kin[j];
        val[r] = vin[j];
        lo = min(lo, key[r]);
        hi = max(hi, key[r]);
    }
    if (tid == 0) { sh.kmin = 0xFFFFFFFFu; sh.kmax = 0; }
    uint32_t *mycnt = sh.cnt[wave];
#pragma unroll
    for (int e = 0; e < WD_WORDS / 64; ++e) mycnt[e * 64 + lane] = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor(lo, o, 64)); hi = max(hi, (uint32_t)__shfl_xor(hi, o, 64)); }
    if (lane == 0) { atomicMin(&sh.kmin, lo); atomicMax(&sh.kmax, hi); }
    __syncthreads();
    const uint32_t kmin = sh.kmin, range = sh.kmax - kmin;
    const int bits = 32 - __clz(range | 0u) - (range == 0 ? 0 : 0);
    const int nbits = range == 0 ? 0 : bits;
    const int passes = (nbits + 9) / 10;
    const int w = passes ? (nbits + passes - 1) / passes : 0;
    const uint32_t mask = (1u << w) - 1u;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) key[r] -= kmin;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * w;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            pos[r] = 0;
            if (sub + r * 64 < m) {
                const uint32_t d = (key[r] >> shift) & mask;
                const uint32_t old = __hip_atomic_fetch_add(&mycnt[d >> 1], 1u << ((d & 1u) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pos[r] = (old >> ((d & 1u) * 16)) & 0xFFFFu;
            }
        }
        __syncthreads();
        {   // thread = counter word (two digits): the waves' counts become offsets (both halves at once: no carry, counts <= 6144)
            uint32_t c[LS_WAVES];
#pragma unroll
            for (int wv = 0; wv < LS_WAVES; ++wv) c[wv] = sh.cnt[wv][tid];
            uint32_t run = 0;
#pragma unroll
            for (int wv = 0; wv < LS_WAVES; ++wv) { sh.cnt[wv][tid] = run; run += c[wv]; }
            sh.base[tid] = run;                     // (total of the even digit | total of the odd digit << 16)
        }
        __syncthreads();
        if (wave == 0) {    // exclusive scan of the 1024 digit totals, lane l owns words l + 64 k
            uint32_t t[8], s2[8], inc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { t[k] = sh.base[64 * k + lane]; s2[k] = (t[k] & 0xFFFFu) + (t[k] >> 16); inc[k] = s2[k]; }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const uint32_t v = __shfl_up(inc[k], o, 64); if (lane >= o) inc[k] += v; }
            uint32_t before = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t ex = before + inc[k] - s2[k];
                sh.base[64 * k + lane] = ex | ((ex + (t[k] & 0xFFFFu)) << 16);
                before += __shfl(inc[k], 63, 64);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (sub + r * 64 >= m) continue;
            const uint32_t d = (key[r] >> shift) & mask, sh16 = (d & 1u) * 16;
            const uint32_t dst = ((sh.base[d >> 1] >> sh16) & 0xFFFFu) + ((mycnt[d >> 1] >> sh16) & 0xFFFFu) + pos[r];
            sh.key[dst] = key[r];
            sh.val[dst] = val[r];
        }
        if (pass + 1 < passes) {
#pragma unroll
            for (int e = 0; e < WD_WORDS / 64; ++e) mycnt[e * 64 + lane] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; ++r)
                if (sub + r * 64 < m) { key[r] = sh.key[sub + r * 64]; val[r] = sh.val[sub + r * 64]; }
        } else {
            __syncthreads();
        }
    }
    if (passes == 0) {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) if (sub + r * 64 < m) sh.val[sub + r * 64] = val[r];
        __syncthreads();
    }
    for (int i = tid; i < m; i += LS_THREADS) out[i] = (int64_t)sh.val[i];
}

__global__ __launch_bounds__(LS_THREADS, 2) void msd_local_sort_wide_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                                            const uint32_t *__restrict__ bucket_start, int64_t n,
                                                                            int64_t *__restrict__ out, int *__restrict__ oversize)
{
    __shared__ WdShared sh;
    const int64_t q = blockIdx.y;
    const int b = blockIdx.x;
    const uint32_t s0 = bucket_start[q * 257 + b], s1 = bucket_start[q * 257 + b + 1];
    const int m = (int)(s1 - s0);
    if (m == 0) return;
    const uint32_t *kin = keys + q * n + s0, *vin = vals + q * n + s0;
    int64_t *o = out + q * n + s0;
    if (m <= LS_THREADS * 8) local_sort_wide<8>(kin, vin, m, o, sh);
    else if (m <= LS_THREADS * 12) local_sort_wide<12>(kin, vin, m, o, sh);
    else if (threadIdx.x == 0) atomicAdd(oversize, 1);
}


// ---- variant: persistent workgroups; the next bucket's pairs are prefetched into registers while the current one is sorted ----
template <int ITEMS>
__device__ __forceinline__ void sort_regs(uint32_t (&key)[ITEMS], uint32_t (&val)[ITEMS], int m, uint32_t *skey, uint32_t *sval,
                                          uint32_t (*cnt)[256], uint32_t *tot, uint32_t *sdiff)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave * 64 * ITEMS + lane;
    uint32_t pos[ITEMS];
    uint32_t diff = 0;
    if (tid == 0) { *sdiff = 0; tot[0] = key[0]; }
    uint32_t *mycnt = cnt[wave];
#pragma unroll
    for (int e = 0; e < 4; ++e) mycnt[e * 64 + lane] = 0;
    __syncthreads();
    const uint32_t first = tot[0];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) if (sub + r * 64 < m) diff |= key[r] ^ first;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0 && diff) atomicOr(sdiff, diff);
    __syncthreads();
    diff = *sdiff;
    bool staged = false;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        if (((diff >> shift) & 255u) == 0) continue;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            pos[r] = 0;
            if (sub + r * 64 < m) pos[r] = __hip_atomic_fetch_add(&mycnt[(key[r] >> shift) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < LS_WAVES; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
            tot[tid] = run;
        }
        __syncthreads();
        if (wave == 0) {
            const uint32_t t0 = tot[4 * lane], t1 = tot[4 * lane + 1], t2 = tot[4 * lane + 2], t3 = tot[4 * lane + 3];
            const uint32_t mine = t0 + t1 + t2 + t3;
            uint32_t inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
            const uint32_t ex = inc - mine;
            tot[4 * lane] = ex; tot[4 * lane + 1] = ex + t0; tot[4 * lane + 2] = ex + t0 + t1; tot[4 * lane + 3] = ex + t0 + t1 + t2;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (sub + r * 64 >= m) continue;
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t dst = tot[d] + mycnt[d] + pos[r];
            skey[dst] = key[r];
            sval[dst] = val[r];
        }
        staged = true;
        if ((diff >> shift) >> 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) mycnt[e * 64 + lane] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) if (sub + r * 64 < m) { key[r] = skey[sub + r * 64]; val[r] = sval[sub + r * 64]; }
        } else {
            __syncthreads();
        }
    }
    if (!staged) {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) if (sub + r * 64 < m) sval[sub + r * 64] = val[r];
        __syncthreads();
    }
}

__global__ __launch_bounds__(LS_THREADS, 2) void msd_local_sort_persistent_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                                                  const uint32_t *__restrict__ bucket_start, int64_t n, int njobs,
                                                                                  int64_t *__restrict__ out, int *__restrict__ oversize)
{
    __shared__ uint32_t skey[LS_THREADS * 16];
    __shared__ uint32_t sval[LS_THREADS * 16];
    __shared__ uint32_t cnt[LS_WAVES][256];
    __shared__ uint32_t tot[256];
    __shared__ uint32_t sdiff;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t nk[16], nv[16];
    int job = blockIdx.x;
    auto geometry = [&](int j, int64_t &q, uint32_t &s0, int &m, int &items) {
        q = j / 256;
        const int b = j % 256;
        s0 = bucket_start[q * 257 + b];
        m = (int)(bucket_start[q * 257 + b + 1] - s0);
        items = m <= LS_THREADS * 8 ? 8 : (m <= LS_THREADS * 12 ? 12 : 16);
    };
    auto prefetch = [&](int j) {
        int64_t q; uint32_t s0; int m, items;
        geometry(j, q, s0, m, items);
        const uint32_t *kin = keys + q * n + s0, *vin = vals + q * n + s0;
        const int sub = wave * 64 * items + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r < items && m > 0) {
                const int i = sub + r * 64;
                const int jj = i < m ? i : m - 1;
                nk[r] = kin[jj];
                nv[r] = vin[jj];
            }
    };
    if (job < njobs) prefetch(job);
    for (; job < njobs; job += gridDim.x) {
        int64_t q; uint32_t s0; int m, items;
        geometry(job, q, s0, m, items);
        int64_t *o = out + q * n + s0;
        if (m > LS_THREADS * 16) { if (tid == 0) atomicAdd(oversize, 1); if (job + (int)gridDim.x < njobs) prefetch(job + gridDim.x); continue; }
        if (items == 8) {
            uint32_t k8[8], v8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { k8[r] = nk[r]; v8[r] = nv[r]; }
            if (job + (int)gridDim.x < njobs) prefetch(job + gridDim.x);
            if (m > 0) sort_regs<8>(k8, v8, m, skey, sval, cnt, tot, &sdiff);
        } else if (items == 12) {
            uint32_t k12[12], v12[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) { k12[r] = nk[r]; v12[r] = nv[r]; }
            if (job + (int)gridDim.x < njobs) prefetch(job + gridDim.x);
            sort_regs<12>(k12, v12, m, skey, sval, cnt, tot, &sdiff);
        } else {
            uint32_t k16[16], v16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { k16[r] = nk[r]; v16[r] = nv[r]; }
            if (job + (int)gridDim.x < njobs) prefetch(job + gridDim.x);
            sort_regs<16>(k16, v16, m, skey, sval, cnt, tot, &sdiff);
        }
        for (int i = tid; i < m; i += LS_THREADS) o[i] = (int64_t)sval[i];
        __syncthreads();        // the next sort writes the staging arrays
    }
}

constexpr int LS_MAX_ITEMS = 16;

__global__ __launch_bounds__(LS_THREADS, 2) void msd_local_sort_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                                       const uint32_t *__restrict__ bucket_start, int64_t n,
                                                                       int64_t *__restrict__ out, int *__restrict__ oversize)
{
    __shared__ uint32_t skey[LS_THREADS * LS_MAX_ITEMS];
    __shared__ uint32_t sval[LS_THREADS * LS_MAX_ITEMS];
    __shared__ uint32_t cnt[LS_WAVES][256];
    __shared__ uint32_t tot[256];
    __shared__ uint32_t sdiff;
    const int64_t q = blockIdx.y;
    const int b = blockIdx.x;
    const uint32_t s0 = bucket_start[q * 257 + b], s1 = bucket_start[q * 257 + b + 1];
    const int m = (int)(s1 - s0);
    if (m == 0) return;
    const uint32_t *kin = keys + q * n + s0, *vin = vals + q * n + s0;
    int64_t *o = out + q * n + s0;
    if (m <= LS_THREADS * 8) local_sort<8>(kin, vin, m, o, skey, sval, cnt, tot, &sdiff);
    else if (m <= LS_THREADS * 12) local_sort<12>(kin, vin, m, o, skey, sval, cnt, tot, &sdiff);
    else if (m <= LS_THREADS * 16) local_sort<16>(kin, vin, m, o, skey, sval, cnt, tot, &sdiff);
    else if (threadIdx.x == 0) atomicAdd(oversize, 1);
}

static uint32_t desc_key(float s)
{
    uint32_t u; memcpy(&u, &s, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;
    if ((u << 1) == 0) u = 0;
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1004993; const int nq = argc > 2 ? atoi(argv[2]) : 70;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;        // 1: heavy ties
    std::vector<uint32_t> keys((size_t)nq * n), vals((size_t)nq * n), starts((size_t)nq * 257);
    std::vector<int64_t> want((size_t)nq * n);
    std::mt19937_64 rng(7);
    std::normal_distribution<float> gauss(0.f, 0.0221f);
    std::normal_distribution<double> bsz(1.0, 0.17);
    for (int q = 0; q < nq; ++q) {
        std::vector<uint64_t> c(n);
        for (int64_t i = 0; i < n; ++i) {
            float s = gauss(rng);
            if (mode == 1) s = roundf(s * 200.f) / 200.f;
            c[i] = ((uint64_t)desc_key(s) << 32) | (uint32_t)i;
        }
        std::sort(c.begin(), c.end());
        // bucket borders: random sizes, renormalised to n
        std::vector<double> w(256); double tot = 0;
        for (auto &x : w) { x = std::max(0.3, bsz(rng)); tot += x; }
        int64_t at = 0;
        for (int b = 0; b < 256; ++b) {
            starts[(size_t)q * 257 + b] = (uint32_t)at;
            int64_t end = b == 255 ? n : std::min<int64_t>(n, at + (int64_t)(w[b] / tot * n));
            for (int64_t i = at; i < end; ++i) want[(size_t)q * n + i] = (int64_t)(uint32_t)c[i];
            std::vector<uint64_t> byid(c.begin() + at, c.begin() + end);
            std::sort(byid.begin(), byid.end(), [](uint64_t a, uint64_t b2) { return (uint32_t)a < (uint32_t)b2; });
            for (int64_t i = at; i < end; ++i) { keys[(size_t)q * n + i] = (uint32_t)(byid[i - at] >> 32); vals[(size_t)q * n + i] = (uint32_t)byid[i - at]; }
            at = end;
        }
        starts[(size_t)q * 257 + 256] = (uint32_t)n;
    }
    uint32_t *dk, *dv, *ds; int64_t *dout; int *dover;
    hipMalloc(&dk, keys.size() * 4); hipMalloc(&dv, vals.size() * 4); hipMalloc(&ds, starts.size() * 4); hipMalloc(&dout, want.size() * 8); hipMalloc(&dover, 4);
    hipMemcpy(dk, keys.data(), keys.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dv, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, starts.data(), starts.size() * 4, hipMemcpyHostToDevice); hipMemset(dover, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) {
            if (getenv("PERSIST")) hipLaunchKernelGGL(msd_local_sort_persistent_kernel, dim3(atoi(getenv("PERSIST"))), dim3(LS_THREADS), 0, 0, dk, dv, ds, n, 256 * nq, dout, dover);
            else if (getenv("WIDE")) hipLaunchKernelGGL(msd_local_sort_wide_kernel, dim3(256, nq), dim3(LS_THREADS), 0, 0, dk, dv, ds, n, dout, dover);
            else hipLaunchKernelGGL(msd_local_sort_kernel, dim3(256, nq), dim3(LS_THREADS), 0, 0, dk, dv, ds, n, dout, dover);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("local sort of %d x 256 buckets (n=%lld): %.4f ms   (%s)\n", nq, (long long)n, ms / 5, hipGetErrorString(hipGetLastError()));
    }
    std::vector<int64_t> got(want.size()); int over;
    hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(&over, dover, 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int q = 0; q < nq; ++q)
        for (int b2 = 0; b2 < 256; ++b2) {
            const size_t s0 = starts[(size_t)q * 257 + b2], s1 = starts[(size_t)q * 257 + b2 + 1];
            if (getenv("WIDE") && s1 - s0 > 6144) continue;
            for (size_t i = s0; i < s1; ++i) bad += got[(size_t)q * n + i] != want[(size_t)q * n + i];
        }
    printf("mismatches %zu of %zu, oversize buckets %d\n", bad, got.size(), over);
    return 0;
}
