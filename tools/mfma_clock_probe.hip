// Sustained fp32 MFMA rate by shape (wall time of a long bare loop on random operands, every CU busy):
// the chip lowers its clock under MFMA load, and the clock it holds can depend on the shape
// (MI355X_MICROARCH.md, DVFS give-back item 7).  hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float *in, float *out, int iters, unsigned long long *clk)
{
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 977) & 4095]; }
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if (SHAPE == 0) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + u) & 7], b[i], acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 7], b[u], acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main()
{
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&clk, 2048 * 16);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = ((rand() % 20001) - 10000) * 2.21e-6f;   // ~ +-0.0221: unit rows in 2048-d
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape)
            for (int wgs : {256, 512}) {
                hipEventRecord(e0);
                for (int k = 0; k < 4; ++k) {
                    if (shape == 0) hipLaunchKernelGGL(loop<0>, dim3(wgs), dim3(256), 0, 0, in, out, iters, clk);
                    else hipLaunchKernelGGL(loop<1>, dim3(wgs), dim3(256), 0, 0, in, out, iters, clk);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long c[2]; hipMemcpy(c, clk + 2 * (wgs / 2), 16, hipMemcpyDeviceToHost);
                // flops: per wave per iter: shape0 32 MFMA x 2048 flop; shape1 16 MFMA x 4096 flop = 65536 either way
                const double flops = 4.0 * wgs * 4.0 * iters * 65536.0;
                printf("%s, %d workgroups x 4 waves: %.2f ms, %.1f TFLOP/s, in-kernel clock %.0f MHz\n",
                       shape == 0 ? "16x16x4" : "32x32x2", wgs, ms, flops / ms / 1e9, (double)c[0] / (double)c[1] * 100.0);
            }
    return 0;
}
