// NOT SHIPPED (tools/attic): GeM + L2N in one launch with a last-arriver hand-off.  Correct (bit-identical to the
// two-launch form under the race screen of round 2) and slower: 29 us against 12-20 us.  Fragments of
// mdir_amd/csrc/mdx_pool.hip as they were; not built by anything.

// GeM / MAC / SPoC pooling AND the L2 normalisation over channels in ONE launch (layers/functional.py:21-22 then
// :130-131).  grid = (ceil(C/4), B): a wave pools one plane; the workgroup that finishes LAST for an image (told by
// the value its arrival ticket returns) reads the image's C pooled values back and normalises them.
// Hand-off as MI355X guide, Guideline 16 (R1, counter form): the pooled values are stored WRITE-THROUGH (sc1: relaxed
// agent-scope stores), so no release fence is needed -- a release (buffer_wbl2) would write back every dirty line the
// trunk's convolutions left in the L2, once per workgroup: measured 42 us per launch against 7 for two launches --
// every storing wave drains (vmcnt 0) -> barrier -> one lane takes a relaxed agent-scope ticket; the last arriver:
// agent-scope acquire (drops this CU's stale L1 lines), drain, barrier, plain loads.  The ticket word is re-armed (0)
// by the last arriver; the caller provides it zeroed once (mdx_pool_l2n_fused).
template <int KIND, int MODE>
__global__ __launch_bounds__(256) void pool_l2n_fused_kernel(const float *__restrict__ feat, int C, int HW, float p,
                                                             float inv_p, float eps, float l2n_eps,
                                                             float *__restrict__ out, unsigned *__restrict__ tickets)
{
    __shared__ float part[4];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    // a wave takes planes c, c + 4*gridDim.x, ...: at most 64 workgroups per image, so that few tickets are drawn
    // (2048 channels as 512 workgroups = 512 atomics on one word, ~12 ns each, all at the same moment)
    for (int c = blockIdx.x * 4 + wave; c < C; c += 4 * gridDim.x) {
        const float *src = feat + ((int64_t)b * C + c) * HW;
        float acc = KIND == MDX_POOL_MAC ? -INFINITY : 0.0f;
        if ((HW & 3) == 0 && ((uintptr_t)feat & 15) == 0) {
            const float4 *s4 = (const float4 *)src;
            const int n4 = HW >> 2;
            for (int i = lane; i < n4; i += 64) {
                const float4 v = s4[i];
                const float a0 = pool_elem<KIND, MODE>(v.x, p, eps), a1 = pool_elem<KIND, MODE>(v.y, p, eps);
                const float a2 = pool_elem<KIND, MODE>(v.z, p, eps), a3 = pool_elem<KIND, MODE>(v.w, p, eps);
                if (KIND == MDX_POOL_MAC) acc = fmaxf(acc, fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
                else acc += (a0 + a1) + (a2 + a3);
            }
        } else {
            for (int i = lane; i < HW; i += 64) {
                const float a = pool_elem<KIND, MODE>(src[i], p, eps);
                if (KIND == MDX_POOL_MAC) acc = fmaxf(acc, a);
                else acc += a;
            }
        }
        acc = KIND == MDX_POOL_MAC ? wave_max(acc) : wave_sum(acc);
        if (lane == 0) {
            float r = acc;
            if (KIND != MDX_POOL_MAC) r = acc / (float)HW;
            if (KIND == MDX_POOL_GEM && MODE != 1) r = powf(r, inv_p);
            __hip_atomic_store(out + (int64_t)b * C + c, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sc1
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch
        }
    }
    __syncthreads();
    if (!s_last) return;
    float *row = out + (int64_t)b * C;
    float ss = 0.0f;
    for (int k = tid; k < C; k += 256) {
        const float v = row[k];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 0) part[wave] = ss;
    __syncthreads();
    const float den = sqrtf((part[0] + part[1]) + (part[2] + part[3])) + l2n_eps;
    for (int k = tid; k < C; k += 256) row[k] = row[k] / den;
}

template <int KIND>
static void launch_pool_fused(int mode, const float *feat, int B, int C, int HW, float p, float eps, float l2n_eps,
                              float *out, unsigned *tickets, hipStream_t s)
{
    const int64_t per_image = ceil_div(C, 4);
    const dim3 grid((unsigned)(per_image < 64 ? per_image : 64), (unsigned)B), blk(256);
    const float inv_p = 1.0f / p;
    switch (mode) {
        case 1: hipLaunchKernelGGL((pool_l2n_fused_kernel<KIND, 1>), grid, blk, 0, s, feat, C, HW, p, inv_p, eps, l2n_eps, out, tickets); break;
        case 2: hipLaunchKernelGGL((pool_l2n_fused_kernel<KIND, 2>), grid, blk, 0, s, feat, C, HW, p, inv_p, eps, l2n_eps, out, tickets); break;
        case 3: hipLaunchKernelGGL((pool_l2n_fused_kernel<KIND, 3>), grid, blk, 0, s, feat, C, HW, p, inv_p, eps, l2n_eps, out, tickets); break;
        default: hipLaunchKernelGGL((pool_l2n_fused_kernel<KIND, 0>), grid, blk, 0, s, feat, C, HW, p, inv_p, eps, l2n_eps, out, tickets); break;
    }
}

int mdx_pool_l2n_fused(const float *feat, int B, int C, int H, int W, int kind, float p, float pool_eps,
                       float l2n_eps, float *out, uint32_t *tickets, void *stream)
{
    MDX_CHECK_ARG(feat && out && tickets, "mdx_pool_l2n_fused: NULL pointer");
    MDX_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && B < 65536, "mdx_pool_l2n_fused: bad shape [%d,%d,%d,%d]", B, C, H, W);
    MDX_CHECK_ARG((int64_t)H * W < (1ll << 31), "mdx_pool_l2n_fused: H*W too large");
    MDX_CHECK_ARG(l2n_eps >= 0.0f, "mdx_pool_l2n_fused: l2n_eps must be >= 0 (use mdx_pool_l2n to skip the normalisation)");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    switch (kind) {
        case MDX_POOL_GEM: {
            MDX_CHECK_ARG(p > 0.0f && pool_eps > 0.0f, "mdx_pool_l2n_fused: gem needs p > 0 and eps > 0");
            const int mode = p == 1.0f ? 1 : p == 2.0f ? 2 : p == 3.0f ? 3 : 0;
            launch_pool_fused<MDX_POOL_GEM>(mode, feat, B, C, HW, p, pool_eps, l2n_eps, out, tickets, s);
            break;
        }
        case MDX_POOL_MAC: launch_pool_fused<MDX_POOL_MAC>(1, feat, B, C, HW, 1.0f, 0.0f, l2n_eps, out, tickets, s); break;
        case MDX_POOL_SPOC: launch_pool_fused<MDX_POOL_SPOC>(1, feat, B, C, HW, 1.0f, 0.0f, l2n_eps, out, tickets, s); break;
        default: MDX_CHECK_ARG(false, "mdx_pool_l2n_fused: unknown pooling kind %d", kind);
    }
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

