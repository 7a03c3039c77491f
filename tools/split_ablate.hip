// Timing harness for the split-precision similarity kernel (mdx_scores_split_kernel.h): shapes of the workgroup and
// timing-only ablations (ABL != 0: results wrong), all variants interleaved in one process on gaussian data of the real
// magnitude (low-entropy data runs faster through DVFS and misleads).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMDX_XCD_BLOCKS_RUNTIME -I mdir_amd/csrc -I tools/attic tools/split_ablate.hip -o tools/split_ablate_bin
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
#include "scores_direct_kernel.h"       // tools/attic: the parked register-streaming form
#include "mdx_scores_stream_kernel.h"
#include "scores_stream_persistent_kernel.h"   // tools/attic
namespace mdx { void set_error(const char *, ...) {} }
using namespace mdx;

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1004993, d = 2048;
    const int KB = d / 16, NC = KB / 2, QTT = 5, NQ = 70;
    const int64_t RT = (n + 15) / 16, RTp = (RT + 31) / 32 * 32;
    f32x4 *db; u32x4 *qp; float *out;
    hipMalloc(&db, (size_t)RTp * KB * 1024); hipMalloc(&qp, (size_t)3 * QTT * NC * 1024); hipMalloc(&out, (size_t)80 * n * 4);
    {
        std::vector<float> h((size_t)RTp * KB * 256);
        unsigned long long st = 88172645463325252ull;
        auto u = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)); };
        for (size_t i = 0; i < h.size(); i += 2) {
            const float r = sqrtf(-2.0f * logf(u() + 1e-12f)) * 0.0221f, a = 6.2831853f * u();
            h[i] = r * cosf(a); h[i + 1] = r * sinf(a);
        }
        hipMemcpy(db, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        // query pieces: bf16 pairs of gaussian values (piece 1, 2 scaled down as real residuals are)
        std::vector<uint32_t> q((size_t)3 * QTT * NC * 256);
        for (size_t i = 0; i < q.size(); ++i) {
            const int piece = (int)(i / ((size_t)QTT * NC * 256));
            const float sc = piece == 0 ? 1.0f : (piece == 1 ? 1.0f / 256 : 1.0f / 65536);
            uint32_t a, b; float fa = h[2 * i] * sc, fb = h[2 * i + 1] * sc;
            memcpy(&a, &fa, 4); memcpy(&b, &fb, 4);
            q[i] = (a >> 16) | (b & 0xFFFF0000u);
        }
        hipMemcpy(qp, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    }
    if (getenv("ZERO")) {           // all-zero operands: the same instruction stream at a fraction of the switching power (DVFS check)
        hipMemset(db, 0, (size_t)RTp * KB * 1024);
        hipMemset(qp, 0, (size_t)3 * QTT * NC * 1024);
        printf("ZERO operands\n");
    }
    auto go = [&](auto kern, int QT_, int R_, int NST, int CW_) {
        const size_t lds = (size_t)NST * (3 * QT_ + 2 * CW_ * R_) * 1024;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int64_t blocks = (RT + CW_ * R_ - 1) / (CW_ * R_);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, qp, out, n, KB, QTT, 0, NQ);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, qp, out, n, KB, QTT, 0, NQ);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
        return ms / 10;
    };
    auto direct = [&](auto kern, int lds, int R_, int CW_) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int64_t blocks = (RT + CW_ * R_ - 1) / (CW_ * R_);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64), lds, 0, db, qp, out, n, KB, QTT, 0, NQ);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64), lds, 0, db, qp, out, n, KB, QTT, 0, NQ);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
        return ms / 10;
    };
    // ---- the fp16 shard (configs[4]): half the bytes, one MFMA per tile pair.  KB = 64 tiles of 32 k per row tile.
    auto direct16 = [&](auto kern, int lds, int R_, int CW_) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int64_t blocks = (RT + CW_ * R_ - 1) / (CW_ * R_);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64), lds, 0, db, qp, out, n, KB / 2, QTT, 0, NQ);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64), lds, 0, db, qp, out, n, KB / 2, QTT, 0, NQ);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
        return ms / 10;
    };
    auto ring16 = [&](auto kern, int R_, int KC_, int NST) {
        const size_t lds = (size_t)NST * (5 + 4 * R_) * KC_ * 1024;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int64_t blocks = (RT + 4 * R_ - 1) / (4 * R_);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, db, (const f32x4 *)qp, out, n, KB / 2, NQ, (unsigned long long *)nullptr);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, db, (const f32x4 *)qp, out, n, KB / 2, NQ, (unsigned long long *)nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms / 10;
    };
    auto stream16 = [&](auto kern, int lds, int R_, int CW_, int wgs, bool check = false) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int nblocks = (int)((RT + CW_ * R_ - 1) / (CW_ * R_));
        const int grid = nblocks < 256 * wgs ? nblocks : 256 * wgs;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(CW_ * 64), lds, 0, db, qp, out, n, KB / 2, QTT, 0, NQ, nblocks);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(CW_ * 64), lds, 0, db, qp, out, n, KB / 2, QTT, 0, NQ, nblocks);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
        return ms / 10;
    };
    auto shipped16 = [&](auto kern, int lds, int R_) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int64_t blocks = (RT + 4 * R_ - 1) / (4 * R_);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, db, (const f32x4 *)qp, out, n, KB / 2, NQ);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, db, (const f32x4 *)qp, out, n, KB / 2, NQ);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
        return ms / 10;
    };
    if (getenv("SPLIT2")) {         // the two-piece mode: ring (shipped) against the register-streaming form
        uint32_t *cell; hipMalloc(&cell, 256);
        { float one = 0.02f; uint32_t bits; memcpy(&bits, &one, 4); hipMemcpy(cell, &bits, 4, hipMemcpyHostToDevice); }
        auto ring2 = [&](auto kern, int R_, int CW_) {
            const size_t lds = (size_t)3 * (2 * 5 + 2 * CW_ * R_) * 1024;
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int64_t blocks = (RT + CW_ * R_ - 1) / (CW_ * R_);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, qp, out, n, KB, QTT, 0, NQ, 131072.0f, (const uint32_t *)cell);
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW_ * 64 + 256), lds, 0, db, qp, out, n, KB, QTT, 0, NQ, 131072.0f, (const uint32_t *)cell);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            return ms / 10;
        };
        for (int rep = 0; rep < 3; ++rep) {
            printf("split2  ring CW8 R2 NST3 (shipped) %.4f | direct CW8 R2 x1 %.4f | direct CW4 R2 x2 %.4f | direct CW8 R1 x1 %.4f ms\n",
                   ring2(scores_split2_kernel<5, 2, 3, 8>, 2, 8),
                   direct(scores_direct_kernel<5, 2, 8, DirectSplit2>, direct_lds_bytes<5, 2, 8, DirectSplit2>(), 2, 8),
                   direct(scores_direct_kernel<5, 2, 4, DirectSplit2, 0, 2>, direct_lds_bytes<5, 2, 4, DirectSplit2>(), 2, 4),
                   direct(scores_direct_kernel<5, 1, 8, DirectSplit2>, direct_lds_bytes<5, 1, 8, DirectSplit2>(), 1, 8));
            fflush(stdout);
        }
        return 0;
    }
    if (getenv("XCD")) {            // row-block order: plain (block = workgroup id) against XCD-contiguous, alternating in one process
        for (int rep = 0; rep < 6; ++rep) {
            const int flag = rep & 1;
            hipMemcpyToSymbol(HIP_SYMBOL(mdx_xcd_blocks_flag), &flag, sizeof flag);
            printf("%s  fp16: ring %.4f | SHIPPED stream x3 %.4f | x2 %.4f | x4 %.4f | direct CW4 R2 x3 %.4f | persistent CW4 R2 x3 %.4f | CW4 R4 x2 %.4f || split3 ring %.4f ms\n",
                   flag ? "XCD-contiguous blocks" : "block = workgroup id ",
                   ring16(scores_lc_kernel<5, 2, 2, 3, 2, false, MmaF16>, 2, 2, 3),
                   shipped16(scores_f16_stream_kernel<5, 2, 3>, stream_lds_bytes<5, 2>(), 2),
                   shipped16(scores_f16_stream_kernel<5, 2, 2>, stream_lds_bytes<5, 2>(), 2),
                   shipped16(scores_f16_stream_kernel<5, 2, 4>, stream_lds_bytes<5, 2>(), 2),
                   direct16(scores_direct_kernel<5, 2, 4, DirectF16, 0, 3>, direct_lds_bytes<5, 2, 4, DirectF16>(), 2, 4),
                   stream16(scores_stream_persistent_kernel<5, 2, 4, StreamF16, 3>, stream_persistent_lds_bytes<5, 2, 4, StreamF16>(), 2, 4, 3),
                   stream16(scores_stream_persistent_kernel<5, 4, 4, StreamF16, 2>, stream_persistent_lds_bytes<5, 4, 4, StreamF16>(), 4, 4, 2),
                   go(scores_split3_kernel<5, 2, 3, 8>, 5, 2, 3, 8));
            fflush(stdout);
        }
        return 0;
    }
    if (getenv("F16")) {
        {   // bitwise: the persistent form against the ring kernel
            std::vector<float> ha((size_t)70 * n), hb((size_t)70 * n);
            hipMemset(out, 0xFF, (size_t)70 * n * 4);
            ring16(scores_lc_kernel<5, 2, 2, 3, 2, false, MmaF16>, 2, 2, 3);
            hipMemcpy(ha.data(), out, ha.size() * 4, hipMemcpyDeviceToHost);
            hipMemset(out, 0xFF, (size_t)70 * n * 4);
            stream16(scores_stream_persistent_kernel<5, 2, 4, StreamF16, 3>, stream_persistent_lds_bytes<5, 2, 4, StreamF16>(), 2, 4, 3);
            hipMemcpy(hb.data(), out, hb.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < ha.size(); ++i) if (memcmp(&ha[i], &hb[i], 4)) { if (!bad) first = i; ++bad; }
            printf("persistent stream kernel vs ring kernel: %zu of %zu scores differ (first at q=%zu row=%zu)\n", bad, ha.size(), first / n, first % n);
        }
        for (int rep = 0; rep < 3; ++rep) {
            printf("fp16 shard (4.4 GB)  ring (shipped) %.4f | direct CW8 R2 x1 %.4f | CW4 R2 x2 %.4f | CW4 R2 x3 %.4f | CW4 R4 x2 %.4f | CW8 R4 x1 %.4f ms\n",
                   ring16(scores_lc_kernel<5, 2, 2, 3, 2, false, MmaF16>, 2, 2, 3),
                   direct16(scores_direct_kernel<5, 2, 8, DirectF16>, direct_lds_bytes<5, 2, 8, DirectF16>(), 2, 8),
                   direct16(scores_direct_kernel<5, 2, 4, DirectF16, 0, 2>, direct_lds_bytes<5, 2, 4, DirectF16>(), 2, 4),
                   direct16(scores_direct_kernel<5, 2, 4, DirectF16, 0, 3>, direct_lds_bytes<5, 2, 4, DirectF16>(), 2, 4),
                   direct16(scores_direct_kernel<5, 4, 4, DirectF16, 0, 2>, direct_lds_bytes<5, 4, 4, DirectF16>(), 4, 4),
                   direct16(scores_direct_kernel<5, 4, 8, DirectF16>, direct_lds_bytes<5, 4, 8, DirectF16>(), 4, 8));
            printf("fp16 PERSISTENT      CW4 R2 x3 %.4f | CW4 R2 x2 %.4f | CW4 R2 x4 %.4f | CW8 R2 x1 %.4f | CW8 R1 x2 %.4f | CW4 R4 x2 %.4f ms\n",
                   stream16(scores_stream_persistent_kernel<5, 2, 4, StreamF16, 3>, stream_persistent_lds_bytes<5, 2, 4, StreamF16>(), 2, 4, 3),
                   stream16(scores_stream_persistent_kernel<5, 2, 4, StreamF16, 2>, stream_persistent_lds_bytes<5, 2, 4, StreamF16>(), 2, 4, 2),
                   stream16(scores_stream_persistent_kernel<5, 2, 4, StreamF16, 4>, stream_persistent_lds_bytes<5, 2, 4, StreamF16>(), 2, 4, 4),
                   stream16(scores_stream_persistent_kernel<5, 2, 8, StreamF16, 1>, stream_persistent_lds_bytes<5, 2, 8, StreamF16>(), 2, 8, 1),
                   stream16(scores_stream_persistent_kernel<5, 1, 8, StreamF16, 2>, stream_persistent_lds_bytes<5, 1, 8, StreamF16>(), 1, 8, 2),
                   stream16(scores_stream_persistent_kernel<5, 4, 4, StreamF16, 2>, stream_persistent_lds_bytes<5, 4, 4, StreamF16>(), 4, 4, 2));
            printf("fp16 stream only     with barriers CW8 R2 %.4f | no barriers %.4f | no barriers, no epilogue %.4f | CW4 R2 x3: barriers %.4f | none %.4f | none, no epilogue %.4f ms\n",
                   direct16(scores_direct_kernel<5, 2, 8, DirectF16, 3>, direct_lds_bytes<5, 2, 8, DirectF16>(), 2, 8),
                   direct16(scores_direct_kernel<5, 2, 8, DirectF16, 4>, direct_lds_bytes<5, 2, 8, DirectF16>(), 2, 8),
                   direct16(scores_direct_kernel<5, 2, 8, DirectF16, 5>, direct_lds_bytes<5, 2, 8, DirectF16>(), 2, 8),
                   direct16(scores_direct_kernel<5, 2, 4, DirectF16, 3, 3>, direct_lds_bytes<5, 2, 4, DirectF16>(), 2, 4),
                   direct16(scores_direct_kernel<5, 2, 4, DirectF16, 4, 3>, direct_lds_bytes<5, 2, 4, DirectF16>(), 2, 4),
                   direct16(scores_direct_kernel<5, 2, 4, DirectF16, 5, 3>, direct_lds_bytes<5, 2, 4, DirectF16>(), 2, 4));
            fflush(stdout);
        }
        return 0;
    }
    for (int rep = 0; rep < 3; ++rep) {
        printf("direct  split3 CW8 R2 %.4f | CW8 R1 %.4f | CW4 R2 %.4f | CW8 R2 stream only %.4f | CW8 R2 no split %.4f || fp16 tiles (half the bytes) CW8 R2 %.4f | R4 %.4f ms\n",
               direct(scores_direct_kernel<5, 2, 8, DirectSplit3>, direct_lds_bytes<5, 2, 8, DirectSplit3>(), 2, 8),
               direct(scores_direct_kernel<5, 1, 8, DirectSplit3>, direct_lds_bytes<5, 1, 8, DirectSplit3>(), 1, 8),
               direct(scores_direct_kernel<5, 2, 4, DirectSplit3>, direct_lds_bytes<5, 2, 4, DirectSplit3>(), 2, 4),
               direct(scores_direct_kernel<5, 2, 8, DirectSplit3, 3>, direct_lds_bytes<5, 2, 8, DirectSplit3>(), 2, 8),
               direct(scores_direct_kernel<5, 2, 8, DirectSplit3, 1>, direct_lds_bytes<5, 2, 8, DirectSplit3>(), 2, 8),
               direct(scores_direct_kernel<5, 2, 8, DirectF16>, direct_lds_bytes<5, 2, 8, DirectF16>(), 2, 8), 0.0f);
        printf("direct  stream only, no query staging, no barriers: split3 tiles CW8 R2 %.4f | R1 %.4f | CW4 R2 %.4f | fp16 tiles stream + barriers %.4f ms\n",
               direct(scores_direct_kernel<5, 2, 8, DirectSplit3, 4>, direct_lds_bytes<5, 2, 8, DirectSplit3>(), 2, 8),
               direct(scores_direct_kernel<5, 1, 8, DirectSplit3, 4>, direct_lds_bytes<5, 1, 8, DirectSplit3>(), 1, 8),
               direct(scores_direct_kernel<5, 2, 4, DirectSplit3, 4>, direct_lds_bytes<5, 2, 4, DirectSplit3>(), 2, 4),
               direct(scores_direct_kernel<5, 2, 8, DirectF16, 3>, direct_lds_bytes<5, 2, 8, DirectF16>(), 2, 8));
        printf("shapes  CW8 R2 NST3 %.4f | CW4 R4 NST3 %.4f | CW8 R1 NST4 %.4f | CW8 R1 NST5 %.4f | CW4 R2 NST4 %.4f | CW8 R2 NST2 %.4f | default-policy db CW8 R2 NST3 %.4f ms\n",
               go(scores_split3_kernel<5, 2, 3, 8>, 5, 2, 3, 8), go(scores_split3_kernel<5, 4, 3, 4>, 5, 4, 3, 4),
               go(scores_split3_kernel<5, 1, 4, 8>, 5, 1, 4, 8), go(scores_split3_kernel<5, 1, 5, 8>, 5, 1, 5, 8),
               go(scores_split3_kernel<5, 2, 4, 4>, 5, 2, 4, 4), go(scores_split3_kernel<5, 2, 2, 8>, 5, 2, 2, 8),
               go(scores_split3_kernel<5, 2, 3, 8, 0>, 5, 2, 3, 8));
        printf("ablate  CW8 R2 NST3: full %.4f | two-piece cost model (3 products) %.4f | no split %.4f | no MFMA %.4f | stream + barriers only %.4f ms\n",
               go(scores_split3_kernel<5, 2, 3, 8>, 5, 2, 3, 8), go(scores_split3_kernel<5, 2, 3, 8, 2, 4>, 5, 2, 3, 8), go(scores_split3_kernel<5, 2, 3, 8, 2, 1>, 5, 2, 3, 8),
               go(scores_split3_kernel<5, 2, 3, 8, 2, 2>, 5, 2, 3, 8), go(scores_split3_kernel<5, 2, 3, 8, 2, 3>, 5, 2, 3, 8));
        fflush(stdout);
    }
    return 0;
}
