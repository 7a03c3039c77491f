// Shared helpers for libmdx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mdx.h"

namespace mdx {

void set_error(const char *fmt, ...);

// mdx_p2p.hip: the row pointers of the open step of a connected direct-store exchange (false: not connected / another nq)
bool p2p_route(const mdx_p2p *p, int64_t nq, float *const **rows);

// mdx_rank.hip: decides once per device how the sort ranks inside a wave (may synchronise `s`); 0 = not decided (capturing)
int probe_lds_order(hipStream_t s);

#define MDX_CHECK_ARG(cond, ...)                  \
    do {                                          \
        if (!(cond)) {                            \
            ::mdx::set_error(__VA_ARGS__);        \
            return MDX_ERR_INVALID;               \
        }                                         \
    } while (0)

#define MDX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::mdx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                             __FILE__, __LINE__);                                       \
            return e_ == hipErrorOutOfMemory ? MDX_ERR_NOMEM : MDX_ERR_RUNTIME;         \
        }                                                                               \
    } while (0)

#define MDX_LAUNCH_CHECK() MDX_HIP(hipGetLastError())

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

constexpr int WAVE = 64;

// Order-preserving key of the ranking (include/mdx.h, mdx_rank_full):
// smaller key = ranked earlier.  Larger score first, -0 == +0, NaN last.
// Same function as desc_key() in oracle/chain.c.
__device__ __forceinline__ uint32_t desc_key(float s)
{
    uint32_t u = __float_as_uint(s);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;  // NaN
    if ((u << 1) == 0) u = 0;                                   // -0 -> +0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum over the 64 lanes with DPP moves (row shifts inside the rows of 16, then row broadcasts): a dozen
// vector instructions, where __shfl_up is an LDS round trip per step
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace mdx
