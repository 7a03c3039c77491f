// Feasibility probe: in-LDS stable LSD sort of one bucket per workgroup with THREAD-PRIVATE packed
// digit counters (4 bits per pass, no ballots).  Synthetic pre-partitioned input: NBKT buckets per
// query, ~3926 elements each, keys sharing all but NB low bits.  Prints time for 70 x 1M elements
// and checks a few buckets on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/local_sort_probe.hip -o tools/local_sort_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

constexpr int LS_THREADS = 512, LS_ITEMS = 16, LS_CAP = LS_THREADS * LS_ITEMS;   // 8192
constexpr int LS_WAVES = LS_THREADS / 64;
__device__ __forceinline__ int pad(int j) { return j + (j >> 4); }                // blocked reads conflict-free

__global__ __launch_bounds__(LS_THREADS) void local_sort_kernel(const uint32_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ vals,
                                                               const uint32_t *__restrict__ bbase,
                                                               const uint32_t *__restrict__ bcnt,
                                                               const uint32_t *__restrict__ bbits, int nbkt,
                                                               int64_t n, int64_t *__restrict__ ranks)
{
    extern __shared__ uint32_t lds[];
    uint32_t *skey = lds, *sval = lds + pad(LS_CAP) + 1, *cntr = lds;       // counters alias the exchange area
    __shared__ uint32_t wtot[8][LS_WAVES];
    __shared__ uint32_t rowtot[8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t q = blockIdx.y;
    const int b = blockIdx.x;
    const uint32_t cnt = bcnt[q * nbkt + b], base = bbase[q * nbkt + b];
    if (cnt == 0) return;
    const int nbits = (int)bbits[q * nbkt + b];
    const uint32_t *kin = keys + q * n + base, *vin = vals + q * n + base;
    // coalesced (striped) load, transposed through LDS into the blocked arrangement
    for (int i = 0; i < LS_ITEMS; ++i) {
        const int j = i * LS_THREADS + t;
        skey[pad(j)] = j < (int)cnt ? kin[j] : 0xFFFFFFFFu;
        sval[pad(j)] = j < (int)cnt ? vin[j] : 0u;
    }
    __syncthreads();
    uint32_t key[LS_ITEMS], val[LS_ITEMS];
#pragma unroll
    for (int i = 0; i < LS_ITEMS; ++i) {
        key[i] = skey[pad(t * LS_ITEMS + i)];
        val[i] = sval[pad(t * LS_ITEMS + i)];
    }
    for (int shift = 0; shift < nbits; shift += 4) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) cntr[r * LS_THREADS + t] = 0;
        // (private slots: no barrier needed between zeroing and the adds of the same thread)
        uint32_t rnk[LS_ITEMS];
#pragma unroll
        for (int i = 0; i < LS_ITEMS; ++i) {
            const uint32_t d = (key[i] >> shift) & 15u, sh = (d >> 3) * 16;
            const uint32_t old = atomicAdd(&cntr[(d & 7) * LS_THREADS + t], 1u << sh);
            rnk[i] = (old >> sh) & 0xFFFFu;
        }
        // exclusive scan across threads of every packed row
        uint32_t ex[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t v = cntr[r * LS_THREADS + t];
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(inc, o, 64);
                if (lane >= o) inc += u;
            }
            if (lane == 63) wtot[r][wave] = inc;
            ex[r] = inc - v;
        }
        __syncthreads();
        if (t < 8) {
            uint32_t run = 0;
            for (int w = 0; w < LS_WAVES; ++w) { const uint32_t v = wtot[t][w]; wtot[t][w] = run; run += v; }
            rowtot[t] = run;
        }
        __syncthreads();
        // bin bases: bins 0..7 = low halves of rows 0..7, bins 8..15 = high halves
        uint32_t binbase[16];
        {
            uint32_t run = 0;
#pragma unroll
            for (int d = 0; d < 16; ++d) { binbase[d] = run; run += (rowtot[d & 7] >> ((d >> 3) * 16)) & 0xFFFFu; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) cntr[r * LS_THREADS + t] = ex[r] + wtot[r][wave];     // packed exclusive prefix
        uint32_t pos[LS_ITEMS];
#pragma unroll
        for (int i = 0; i < LS_ITEMS; ++i) {
            const uint32_t d = (key[i] >> shift) & 15u, sh = (d >> 3) * 16;
            const uint32_t pre = (cntr[(d & 7) * LS_THREADS + t] >> sh) & 0xFFFFu;
            uint32_t bb = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) bb = d == (uint32_t)e ? binbase[e] : bb;
            pos[i] = bb + pre + rnk[i];
        }
        __syncthreads();                        // everyone has read its counters: the area becomes the exchange buffer
#pragma unroll
        for (int i = 0; i < LS_ITEMS; ++i) {
            skey[pad(pos[i])] = key[i];
            sval[pad(pos[i])] = val[i];
        }
        __syncthreads();
        if (shift + 4 < nbits) {
#pragma unroll
            for (int i = 0; i < LS_ITEMS; ++i) {
                key[i] = skey[pad(t * LS_ITEMS + i)];
                val[i] = sval[pad(t * LS_ITEMS + i)];
            }
        }
    }
    __syncthreads();
    int64_t *out = ranks + q * n + base;
    for (int j = t; j < (int)cnt; j += LS_THREADS) out[j] = (int64_t)sval[pad(j)];
}

int main()
{
    const int NQ = 70, NB = 256; const int64_t n = 1004993;
    std::vector<uint32_t> hk(NQ * n), hv(NQ * n), hbase(NQ * NB), hcnt(NQ * NB), hbits(NQ * NB);
    srand(1);
    for (int q = 0; q < NQ; ++q) {
        uint32_t off = 0;
        for (int b = 0; b < NB; ++b) {
            static int delta = 0;
            if ((b & 1) == 0) delta = rand() % 1800 - 900;
            uint32_t c = (uint32_t)(n / NB + (b < (int)(n % NB) ? 1 : 0)) + ((b & 1) == 0 ? delta : -delta);
            hbase[q * NB + b] = off; hcnt[q * NB + b] = c;
            const int bits = 17 + rand() % 3;
            hbits[q * NB + b] = bits;
            const uint32_t prefix = ((uint32_t)b << 24) | 0x00800000u;
            for (uint32_t i = 0; i < c; ++i) {
                uint32_t r = ((uint32_t)rand() << 12) ^ (uint32_t)rand();
                hk[q * n + off + i] = (prefix & ~((1u << bits) - 1)) | (r & ((1u << bits) - 1));
                if (i % 97 == 5 && i) hk[q * n + off + i] = hk[q * n + off + i - 1];     // some exact duplicates
                hv[q * n + off + i] = off + i;
            }
            off += c;
        }
    }
    uint32_t *dk, *dv, *dbase, *dcnt, *dbits; int64_t *dr;
    hipMalloc(&dk, hk.size() * 4); hipMalloc(&dv, hv.size() * 4); hipMalloc(&dbase, hbase.size() * 4);
    hipMalloc(&dcnt, hcnt.size() * 4); hipMalloc(&dbits, hbits.size() * 4); hipMalloc(&dr, (size_t)NQ * n * 8);
    hipMemcpy(dk, hk.data(), hk.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbase, hbase.data(), hbase.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dcnt, hcnt.data(), hcnt.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbits, hbits.data(), hbits.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)(2 * (LS_CAP + LS_CAP / 16) + 2) * 4;
    hipFuncSetAttribute((const void *)local_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(local_sort_kernel, dim3(NB, NQ), dim3(LS_THREADS), ldsb, 0, dk, dv, dbase, dcnt, dbits, NB, n, dr);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(local_sort_kernel, dim3(NB, NQ), dim3(LS_THREADS), ldsb, 0, dk, dv, dbase, dcnt, dbits, NB, n, dr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("local sort of %d x %lld elements in %d buckets/query: %.3f ms (lds %zu B, err %s)\n", NQ, (long long)n, NB, ms / 10, ldsb, hipGetErrorString(hipGetLastError()));
    std::vector<int64_t> hr((size_t)NQ * n);
    hipMemcpy(hr.data(), dr, hr.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int q = 0; q < NQ; q += 23)
        for (int b = 0; b < NB; b += 51) {
            const uint32_t off = hbase[q * NB + b], c = hcnt[q * NB + b];
            std::vector<uint32_t> idx(c);
            for (uint32_t i = 0; i < c; ++i) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return hk[q * n + off + x] < hk[q * n + off + y]; });
            for (uint32_t i = 0; i < c; ++i) bad += hr[q * n + off + i] != (int64_t)hv[q * n + off + idx[i]];
        }
    printf("mismatches in checked buckets: %ld\n", bad);
    return bad != 0;
}
