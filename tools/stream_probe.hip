// What read bandwidth does HBM give a once-read 8.2 GB stream on this box, by access pattern and by bytes in flight?
// (Sets the floor of every HBM-bound similarity kernel.)  Each variant sums the stream into one float per wave so that
// the loads stay alive; non-temporal 16-B loads, one coalesced KiB per wave instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_probe.hip -o tools/stream_probe_bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// pattern 0: a wave owns a contiguous run of `run` KiB and walks it front to back, DEPTH loads in flight
// pattern 1: a workgroup owns a contiguous run; its waves interleave KiB by KiB
template <int DEPTH, int PATTERN>
__global__ __launch_bounds__(512) void stream_kernel(const f32x4 *__restrict__ src, float *__restrict__ sink, int64_t kib_total, int run)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    const int64_t runs = kib_total / run;
    if (PATTERN == 0) {
        for (int64_t r = (int64_t)blockIdx.x * nw + wave; r < runs; r += (int64_t)gridDim.x * nw) {
            const f32x4 *p = src + r * run * 64 + lane;
            for (int k = 0; k < run; k += DEPTH) {
                f32x4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(p + (int64_t)(k + d) * 64);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc += v[d];
            }
        }
    } else {
        for (int64_t r = blockIdx.x; r < runs; r += gridDim.x) {
            const f32x4 *p = src + r * run * 64 + lane;
            for (int k = wave * DEPTH; k < run; k += nw * DEPTH) {
                f32x4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(p + (int64_t)(k + d) * 64);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc += v[d];
            }
        }
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) sink[blockIdx.x] = s;
}

// pattern 2: the direct similarity kernel's shape, nothing but the loads: one workgroup per 16 row tiles (NOT persistent),
// a wave owns two 128-KiB runs and walks them in step, 2 KiB of each per chunk, 4 chunks (16 loads) in flight, rolling
__global__ __launch_bounds__(512) void shaped_kernel(const f32x4 *__restrict__ src, float *__restrict__ sink, int64_t kib_total, int run)
{
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 *p0 = src + ((int64_t)blockIdx.x * 16 + wave * 2) * run * 64 + lane, *p1 = p0 + (int64_t)run * 64;
    f32x4 v[4][4], acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j][0] = __builtin_nontemporal_load(p0 + (2 * j) * 64); v[j][1] = __builtin_nontemporal_load(p0 + (2 * j + 1) * 64);
        v[j][2] = __builtin_nontemporal_load(p1 + (2 * j) * 64); v[j][3] = __builtin_nontemporal_load(p1 + (2 * j + 1) * 64);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int nit = run / 8;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc += v[j][0] + v[j][1] + v[j][2] + v[j][3];
            if (it + 1 < nit) {
                const int c = (it + 1) * 4 + j;
                v[j][0] = __builtin_nontemporal_load(p0 + (2 * c) * 64); v[j][1] = __builtin_nontemporal_load(p0 + (2 * c + 1) * 64);
                v[j][2] = __builtin_nontemporal_load(p1 + (2 * c) * 64); v[j][3] = __builtin_nontemporal_load(p1 + (2 * c + 1) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) { lds[threadIdx.x] = acc; sink[blockIdx.x] = s + lds[0][0]; }
}

template <typename K>
static void go(const char *name, K kern, const f32x4 *src, float *sink, int64_t kib, int run, int grid, int block)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, src, sink, kib, run);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, src, sink, kib, run);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-58s run %4d KiB grid %5d x %3d: %.4f ms = %.2f TB/s\n", name, run, grid, block, ms, kib * 1024.0 / ms / 1e9);
}

int main()
{
    const int64_t kib = (int64_t)62816 * 128;          // 1 005 056 rows x 2048 fp32 = 8.23 GB
    f32x4 *src; float *sink;
    hipMalloc(&src, kib * 1024); hipMalloc(&sink, 1 << 20);
    hipMemset(src, 0x3c, kib * 1024);
    auto shaped = [&](const char *name, int lds, int grid_div) {
        hipFuncSetAttribute((const void *)shaped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int grid = (int)(kib / 128 / 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(shaped_kernel, dim3(grid), dim3(512), lds, 0, src, sink, kib, 128);
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(shaped_kernel, dim3(grid), dim3(512), lds, 0, src, sink, kib, 128);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-58s grid %5d x 512, %3d KiB LDS: %.4f ms = %.2f TB/s\n", name, grid, lds / 1024, ms, kib * 1024.0 / ms / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        shaped("direct-kernel shape, 1 WG/CU (120 KiB LDS)", 120 * 1024, 1);
        shaped("direct-kernel shape, 2 WG/CU (64 KiB LDS)", 64 * 1024, 1);
        shaped("direct-kernel shape, 4 WG/CU (no LDS)", 0, 1);
        go("wave-owned runs, 4 in flight", stream_kernel<4, 0>, src, sink, kib, 128, 256 * 2, 512);
        go("wave-owned runs, 8 in flight", stream_kernel<8, 0>, src, sink, kib, 128, 256 * 2, 512);
        go("wave-owned runs, 16 in flight", stream_kernel<16, 0>, src, sink, kib, 128, 256 * 2, 512);
        go("wave-owned runs, 16 in flight, 1 WG/CU", stream_kernel<16, 0>, src, sink, kib, 128, 256, 512);
        go("wave-owned runs, 16 in flight, 4 WG/CU x 256", stream_kernel<16, 0>, src, sink, kib, 128, 256 * 4, 256);
        go("wave-owned runs, 32 in flight", stream_kernel<32, 0>, src, sink, kib, 128, 256 * 2, 512);
        go("wave-owned 16-KiB runs, 16 in flight", stream_kernel<16, 0>, src, sink, kib, 16, 256 * 2, 512);
        go("workgroup-owned runs, waves interleaved, 4 in flight", stream_kernel<4, 1>, src, sink, kib, 2048, 256 * 2, 512);
        go("workgroup-owned runs, waves interleaved, 8 in flight", stream_kernel<8, 1>, src, sink, kib, 2048, 256 * 2, 512);
        go("workgroup-owned runs, waves interleaved, 16 in flight", stream_kernel<16, 1>, src, sink, kib, 2048, 256 * 2, 512);
        go("workgroup-owned 256-KiB runs, interleaved, 8 in flight", stream_kernel<8, 1>, src, sink, kib, 256, 256 * 2, 512);
        go("workgroup-owned 256-KiB runs, interleaved, 8, 8 WG/CU", stream_kernel<8, 1>, src, sink, kib, 256, 256 * 8, 256);
    }
    return 0;
}
