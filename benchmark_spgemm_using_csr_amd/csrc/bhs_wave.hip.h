// bhs_wave.hip.h — wave64 cross-lane primitives for gfx950 built on DPP and the
// CDNA4 permlane-swap instructions (VALU latency) instead of ds_bpermute (LDS
// pipe latency).  Semantics verified on hardware by tools/dpp_probe.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace bhs {

template <int CTRL, int ROWM, int BANKM, bool BC>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWM, BANKM, BC);
}

// value of lane (lane ^ X) for X in {1,2,4,8,16,32}
template <int X>
__device__ __forceinline__ unsigned lane_xor(unsigned x, int lane)
{
    static_assert(X == 1 || X == 2 || X == 4 || X == 8 || X == 16 || X == 32, "power of two < 64");
    if constexpr (X == 1) return dpp_u32<0xB1, 0xf, 0xf, true>(x, x);          // quad_perm [1,0,3,2]
    else if constexpr (X == 2) return dpp_u32<0x4E, 0xf, 0xf, true>(x, x);     // quad_perm [2,3,0,1]
    else if constexpr (X == 4) {
        const unsigned y = dpp_u32<0x104, 0xf, 0x5, false>(x, x);               // row_shl:4 -> banks 0,2
        return dpp_u32<0x114, 0xf, 0xa, false>(y, x);                           // row_shr:4 -> banks 1,3
    } else if constexpr (X == 8) {
        const unsigned y = dpp_u32<0x108, 0xf, 0x3, false>(x, x);               // row_shl:8 -> banks 0,1
        return dpp_u32<0x118, 0xf, 0xc, false>(y, x);                           // row_shr:8 -> banks 2,3
    } else if constexpr (X == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

template <int X>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long x, int lane)
{
    const unsigned lo = lane_xor<X>((unsigned)x, lane);
    const unsigned hi = lane_xor<X>((unsigned)(x >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// inclusive prefix sum across the 64 lanes (6 DPP adds)
__device__ __forceinline__ int wave_incl_scan_dpp(int v)
{
    unsigned s = (unsigned)v;
    s += dpp_u32<0x111, 0xf, 0xf, true>(0, s);    // row_shr:1
    s += dpp_u32<0x112, 0xf, 0xf, true>(0, s);    // row_shr:2
    s += dpp_u32<0x114, 0xf, 0xf, true>(0, s);    // row_shr:4
    s += dpp_u32<0x118, 0xf, 0xf, true>(0, s);    // row_shr:8
    s += dpp_u32<0x142, 0xa, 0xf, false>(0, s);   // row_bcast:15 -> rows 1,3
    s += dpp_u32<0x143, 0xc, 0xf, false>(0, s);   // row_bcast:31 -> rows 2,3
    return (int)s;
}

// sum over the 64 lanes, result in every lane
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    return __builtin_amdgcn_readlane(wave_incl_scan_dpp(v), 63);
}

}  // namespace bhs
