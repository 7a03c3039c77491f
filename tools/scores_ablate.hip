// Timing-only ablations of the similarity kernel (results are wrong for ABL != 0).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mdir_amd/csrc -I tools tools/scores_ablate.hip -o tools/scores_ablate
#include <stdarg.h>
#include <vector>
#include <math.h>
#include <stdlib.h>
#include "mdx_scores_kernel.h"
#include "attic/scores_pb_kernel.h"      // persistent variant, measured slower: kept for the record
#include "scores_v1_kernel.h"
namespace mdx { void set_error(const char *, ...) {} }
using namespace mdx;
static int g_kbs = 0;

template <int ABL, int R, bool CM = false, bool NT = false, int NS = 2, int WPS = 2, bool SP = false, int NW = 4, int KC = 4>
static float run(const f32x4 *db, const f32x4 *q, float *out, int64_t n, int64_t RT, int KB, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int64_t blocks = (RT + NW * R - 1) / (NW * R);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((scores_kernel<5, R, ABL, CM, NT, NS, WPS, SP, NW, KC>), dim3(blocks), dim3(NW * 64), 0, 0, db, q, out, n, KB, 70, (RT + 7) / 8 * 8);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((scores_kernel<5, R, ABL, CM, NT, NS, WPS, SP, NW, KC>), dim3(blocks), dim3(NW * 64), 0, 0, db, q, out, n, KB, 70, (RT + 7) / 8 * 8);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1004993, d = 2048; const int KB = d / 16;
    const int64_t RT = (n + 15) / 16, RTp = (RT + 7) / 8 * 8;
    f32x4 *db, *q; float *out;
    hipMalloc(&db, (RTp + 256) * (KB + 8) * 1024); hipMalloc(&q, 5 * KB * 1024); hipMalloc(&out, 70 * n * 4);
    std::vector<float> h(RTp * (KB + 8) * 256);
    {   // full-mantissa gaussian data of the real magnitude (unit rows in 2048-d): power/clock as in production
        unsigned long long st = 88172645463325252ull;
        auto u = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)); };
        for (size_t i = 0; i < h.size(); i += 2) {
            const float r = sqrtf(-2.0f * logf(u() + 1e-12f)) * 0.0221f, a = 6.2831853f * u();
            h[i] = r * cosf(a); if (i + 1 < h.size()) h[i + 1] = r * sinf(a);
        }
    }
    hipMemcpy(db, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(q, h.data(), 5 * KB * 1024, hipMemcpyHostToDevice);
    {
        auto lc = [&](auto kern, int R_, int KC_, int NST, int QT_, int CW_ = 4) {
            const size_t lds = (size_t)NST * (QT_ + CW_ * R_) * KC_ * 1024;
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int64_t blocks = (RT + CW_ * R_ - 1) / (CW_ * R_);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, q, out, n, KB, 70, (unsigned long long *)nullptr);
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, q, out, n, KB, 70, (unsigned long long *)nullptr);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
            return ms / 10;
        };
        float *out2; hipMalloc(&out2, 70 * n * 4);
        auto pk = [&](auto kern, int lds, float *o, int nq, int grid = 512) {
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, db, q, o, n, RT, KB, nq);
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, db, q, o, n, RT, KB, nq);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
            return ms / 10;
        };
        if (getenv("STAMPS")) {     // in-kernel stamps of the consumer waves: barrier-wait share, loop cycles, clock
            auto kern = scores_lc_kernel<4, 2, 2, 3, 2, true, MmaF32, 1>;
            const size_t lds = (size_t)3 * (5 + 8) * 2 * 1024;
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int64_t blocks = (RT + 7) / 8;
            unsigned long long *dbg; hipMalloc(&dbg, blocks * 4 * 8 * 8);
            std::vector<unsigned long long> hd(blocks * 4 * 8);
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(dbg, 0, blocks * 4 * 8 * 8);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, db, q, out, n, KB, 70, dbg);
                hipDeviceSynchronize();
                hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
                double wait = 0, work = 0, cyc = 0, ticks = 0, life = 0, epi = 0; unsigned long long t0 = ~0ull, t1 = 0;
                for (int64_t i = 0; i < blocks * 4; ++i) {
                    const unsigned long long *d = &hd[i * 8];
                    wait += d[0]; work += d[1]; cyc += d[2]; ticks += d[3]; life += d[6] - d[4]; epi += d[6] - (d[5] + d[3]);
                    t0 = d[4] < t0 ? d[4] : t0; t1 = d[6] > t1 ? d[6] : t1;
                }
                const double nw = blocks * 4;
                printf("stamps: barrier-wait share %.3f of the loop, loop %.1f us at %.3f GHz, workgroup life %.1f us (prologue %.1f, epilogue %.1f), launch %.3f ms\n",
                       wait / (wait + work), ticks / nw / 100.0, cyc / ticks / 10.0, life / nw / 100.0, (life - epi) / nw / 100.0 - ticks / nw / 100.0,
                       epi / nw / 100.0, (t1 - t0) / 1e5);
            }
            return 0;
        }
        if (getenv("R4")) {         // 64 queries, no leftover tile: 2 WG/CU x 4 consumers x 2 row tiles against 1 WG/CU x 8 consumers x 4 row tiles
            auto lc64 = [&](auto kern, int R_, int KC_, int NST, int CW_) {
                const size_t lds = (size_t)NST * (4 + CW_ * R_) * KC_ * 1024;
                hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                const int64_t blocks = (RT + CW_ * R_ - 1) / (CW_ * R_);
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, q, out, n, KB, 64, (unsigned long long *)nullptr);
                hipEventRecord(a);
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, q, out, n, KB, 64, (unsigned long long *)nullptr);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
                return ms / 10;
            };
            for (int rep = 0; rep < 3; ++rep)
                printf("64 queries: 2 WG/CU x 4 cons x R2 KC2 NST3 %.4f | 1 WG/CU x 8 cons x R2 KC2 NST3 %.4f | 1 WG/CU x 8 cons x R4 KC1 NST4 %.4f | 1 WG/CU x 4 cons x R4 KC2 NST3 %.4f ms\n",
                       lc64(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 0, 4>, 2, 2, 3, 4),
                       lc64(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 0, 8>, 2, 2, 3, 8),
                       lc64(scores_lc_kernel<4, 4, 1, 4, 2, false, MmaF32, 0, 8>, 4, 1, 4, 8),
                       lc64(scores_lc_kernel<4, 4, 2, 3, 2, false, MmaF32, 0, 4>, 4, 2, 3, 4));
            return 0;
        }
        if (getenv("CW8")) {        // one workgroup per CU with 8 consumer waves sharing the query stage
            for (int rep = 0; rep < 3; ++rep)
                printf("2 WG/CU x 4 consumers %.4f | 1 WG/CU x 8 consumers KC2 NST3 %.4f | KC1 NST6 %.4f | KC2 NST2 %.4f ms\n",
                       lc(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 1>, 2, 2, 3, 5),
                       lc(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 1, 8>, 2, 2, 3, 5, 8),
                       lc(scores_lc_kernel<4, 2, 1, 6, 2, false, MmaF32, 1, 8>, 2, 1, 6, 5, 8),
                       lc(scores_lc_kernel<4, 2, 2, 2, 2, false, MmaF32, 1, 8>, 2, 2, 2, 5, 8));
            {   // bitwise: 8 consumers against 4
                std::vector<float> ha((size_t)70 * n), hb((size_t)70 * n);
                hipMemset(out, 0xFF, (size_t)70 * n * 4);
                lc(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 1>, 2, 2, 3, 5);
                hipMemcpy(ha.data(), out, ha.size() * 4, hipMemcpyDeviceToHost);
                hipMemset(out, 0xFF, (size_t)70 * n * 4);
                lc(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 1, 8>, 2, 2, 3, 5, 8);
                hipMemcpy(hb.data(), out, hb.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0;
                for (size_t i = 0; i < ha.size(); ++i) bad += memcmp(&ha[i], &hb[i], 4) != 0;
                printf("8 consumers vs 4 consumers: %zu of %zu scores differ\n", bad, ha.size());
            }
            return 0;
        }
        // interleaved rounds in one process
        for (int rep = 0; rep < 3; ++rep)
            printf("n=%lld: QT5 KC2 NST3 %.4f | QT5 KC1 NST4 (3 WG/CU) %.4f | QT5 KC1 NST6 %.4f | QT4+leftover KC2 NST3 %.4f | QT4+leftover KC1 NST6 %.4f | persistent QT5 %.4f ms\n", (long long)n,
                   lc(scores_lc_kernel<5, 2, 2, 3, 2>, 2, 2, 3, 5), lc(scores_lc_kernel<5, 2, 1, 4, 2>, 2, 1, 4, 5), lc(scores_lc_kernel<5, 2, 1, 6, 2>, 2, 1, 6, 5),
                   lc(scores_lc_kernel<4, 2, 2, 3, 2, false, MmaF32, 1>, 2, 2, 3, 5), lc(scores_lc_kernel<4, 2, 1, 6, 2, false, MmaF32, 1>, 2, 1, 6, 5),
                   pk(scores_pb_kernel<5, 0, 2, 3>, pb_lds_bytes<5, 0, 2, 3>(), out2, 70));
        {   // bitwise: persistent kernels against the per-block padded kernel
            lc(scores_lc_kernel<5, 2, 2, 3, 2>, 2, 2, 3, 5);
            std::vector<float> ha((size_t)70 * n), hb((size_t)70 * n);
            hipMemcpy(ha.data(), out, ha.size() * 4, hipMemcpyDeviceToHost);
            for (int v = 0; v < 2; ++v) {
                hipMemset(out2, 0xFF, (size_t)70 * n * 4);
                if (v == 0) pk(scores_pb_kernel<5, 0, 2, 3>, pb_lds_bytes<5, 0, 2, 3>(), out2, 70);
                else pk(scores_pb_kernel<4, 1, 2, 3>, pb_lds_bytes<4, 1, 2, 3>(), out2, 70);
                hipMemcpy(hb.data(), out2, hb.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0, first = 0;
                for (size_t i = 0; i < ha.size(); ++i) if (memcmp(&ha[i], &hb[i], 4)) { if (!bad) first = i; ++bad; }
                printf("persistent %s vs per-block kernel: %zu of %zu scores differ (first at q=%zu row=%zu)\n",
                       v == 0 ? "QT5" : "QT4+leftover", bad, ha.size(), first / n, first % n);
            }
        }
    }
    return 0;
}
