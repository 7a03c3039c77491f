// Issue-rate probe for the fp32 MFMA shapes used by the similarity kernel (cycles per instruction,
// one wave per SIMD, s_memtime around an unrolled loop).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float *out, unsigned long long *cyc, float a0, float b0)
{
    f32x4 big[8], sm[4];
    for (int i = 0; i < 8; ++i) big[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) sm[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {            // 8 independent 16x16x4
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
            } else if (MODE == 1) {     // 8 independent-ish 4x4x1 (4 accumulators round robin)
#pragma unroll
                for (int i = 0; i < 8; ++i) sm[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[i & 3], 0, 0, 0);
            } else if (MODE == 2) {     // 8 dependent 4x4x1 (one chain)
#pragma unroll
                for (int i = 0; i < 8; ++i) sm[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[0], 0, 0, 0);
            } else if (MODE == 3) {     // 8 big + 4 small interleaved (2 big, 1 small), small = one chain
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                    if (i & 1) { sm[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[0], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                }
            } else if (MODE == 4) {     // 8 big then 4 small (one chain) grouped
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) sm[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE == 5) {     // 8 big then 4 small on 4 independent accumulators
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) sm[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE == 6) {     // 8 big then 16 small on 4 accumulators
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) sm[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE == 7) {     // 8 big then 16 small on 2 accumulators
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) sm[i & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[i & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE == 8) {     // two waves per SIMD: waves 0-3 issue 8 big, waves 4-7 issue 8 small (independent)
                if (threadIdx.x < 256) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) sm[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[i & 3], 0, 0, 0);
                }
            } else if (MODE == 9) {     // two waves per SIMD, both 8 big
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
            } else if (MODE == 10) {    // 8 big + 16 small: one small (2 accumulators alternating) after every big... 2 after each
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    big[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, big[i], 0, 0, 0);
                    sm[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[0], 0, 0, 0);
                    sm[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += big[i][0] + big[i][3];
    for (int i = 0; i < 4; ++i) s += sm[i][0] + sm[i][2];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 255) == 0) cyc[blockIdx.x * 2 + (threadIdx.x >> 8)] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int per_iter_big, int per_iter_small, int blocks, int threads = 256)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 16);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f, 2.0f);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 16, hipMemcpyDeviceToHost);
    double c = (double)h[(blocks / 2) * 2] / (64.0 * 4.0), c2 = (double)h[(blocks / 2) * 2 + 1] / (64.0 * 4.0);
    printf("%-52s %7.1f (waves 4-7: %7.1f) cycles per unrolled group (%d big + %d small)  [%d blocks x %d]\n", what, c, threads > 256 ? c2 : 0.0, per_iter_big, per_iter_small, blocks, threads);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : {1, 256}) {
        run<0>("8 x 16x16x4 independent", 8, 0, blocks);
        run<1>("8 x 4x4x1 on 4 accumulators", 0, 8, blocks);
        run<2>("8 x 4x4x1 one dependent chain", 0, 8, blocks);
        run<3>("8 x 16x16x4 + 4 x 4x4x1 interleaved (chain)", 8, 4, blocks);
        run<4>("8 x 16x16x4 then 4 x 4x4x1 chain", 8, 4, blocks);
        run<5>("8 x 16x16x4 then 4 x 4x4x1 independent", 8, 4, blocks);
        run<6>("8 big then 16 small on 4 accumulators", 8, 16, blocks);
        run<7>("8 big then 16 small on 2 accumulators", 8, 16, blocks);
        run<10>("8 x (big, small, small) 2 chains", 8, 16, blocks);
        run<8>("2 waves/SIMD: one 8 big, other 8 small", 8, 8, blocks, 512);
        run<9>("2 waves/SIMD: both 8 big", 8, 0, blocks, 512);
    }
    return 0;
}
