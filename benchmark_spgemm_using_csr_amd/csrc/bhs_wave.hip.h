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

// ---------------------------------------------------------------------------
// Ascending sorting network for 32-bit keys held E per lane (element index
// i = lane*E + e), GW lanes per independent sort (64: the wave, 16: one DPP row).
// "Flip" form of the bitonic network: level kk first compares i with
// i ^ (kk-1) (two ascending runs -> two bitonic halves, min to the lower index),
// then i with i ^ j for j = kk/4 .. 1.  Every compare-exchange is ascending, so
//   * in-register pairs cost v_min + v_max (no direction flags);
//   * lane partners whose deciding bit is >= 4 cost TWO instructions: the
//     lower lanes of a pair form whole DPP banks, so a bank-masked
//     v_min_u32_dpp writes them and a bank-masked v_max_u32_dpp the others
//     (row_shl/shr:4, row_ror:8, row_half_mirror, row_mirror);
//   * lane partners 1,2,3 cost three (quad_perm min, max, select), 16..63 go
//     through v_permlane16/32_swap.
// 128 keys (E = 2): 140 VALU instructions against 225 for the direction-flag
// network with separate DPP moves.  The kernels that use it are VALU-issue bound.
// Inline asm is needed for the bank-masked forms; it carries its own s_nop for
// the "VALU write -> DPP read" hazard (2 wait states) on both sides because the
// compiler's hazard recogniser does not look inside asm blocks.
// ---------------------------------------------------------------------------
#define BHS_DPP_MIN(D, S, X, CTRL, BANK) "v_min_u32_dpp " D ", " S ", " X " " CTRL " row_mask:0xf bank_mask:" BANK "\n\t"
#define BHS_DPP_MAX(D, S, X, CTRL, BANK) "v_max_u32_dpp " D ", " S ", " X " " CTRL " row_mask:0xf bank_mask:" BANK "\n\t"
// one compare-exchange step for 1, 2 or 4 independent keys per asm block (one pair of s_nop for the group)
#define BHS_CX_BANK1(CL, BL, CH, BH)                                                                          \
    asm("s_nop 1\n\t" BHS_DPP_MIN("%0", "%1", "%2", CL, BL) BHS_DPP_MAX("%0", "%1", "%2", CH, BH) "s_nop 1"     \
        : "=&v"(r[0]) : "v"(s[0]), "v"(x[0]))
#define BHS_CX_BANK2(CL, BL, CH, BH)                                                                          \
    asm("s_nop 1\n\t" BHS_DPP_MIN("%0", "%2", "%4", CL, BL) BHS_DPP_MIN("%1", "%3", "%5", CL, BL)               \
        BHS_DPP_MAX("%0", "%2", "%4", CH, BH) BHS_DPP_MAX("%1", "%3", "%5", CH, BH) "s_nop 1"                   \
        : "=&v"(r[0]), "=&v"(r[1]) : "v"(s[0]), "v"(s[1]), "v"(x[0]), "v"(x[1]))
#define BHS_CX_BANK4(CL, BL, CH, BH)                                                                          \
    asm("s_nop 1\n\t" BHS_DPP_MIN("%0", "%4", "%8", CL, BL) BHS_DPP_MIN("%1", "%5", "%9", CL, BL)               \
        BHS_DPP_MIN("%2", "%6", "%10", CL, BL) BHS_DPP_MIN("%3", "%7", "%11", CL, BL)                           \
        BHS_DPP_MAX("%0", "%4", "%8", CH, BH) BHS_DPP_MAX("%1", "%5", "%9", CH, BH)                             \
        BHS_DPP_MAX("%2", "%6", "%10", CH, BH) BHS_DPP_MAX("%3", "%7", "%11", CH, BH) "s_nop 1"                 \
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])                                                  \
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]))
#define BHS_CX_BANKN(CL, BL, CH, BH)                                      \
    do {                                                                  \
        if constexpr (N == 1) BHS_CX_BANK1(CL, BL, CH, BH);               \
        else if constexpr (N == 2) BHS_CX_BANK2(CL, BL, CH, BH);          \
        else BHS_CX_BANK4(CL, BL, CH, BH);                                \
    } while (0)

// r[i] = min (lanes whose deciding bit of LM is 0) / max (the others) of x[i] and s[i][lane ^ LM], LM in {4,8,7,15}
template <int LM, int N>
__device__ __forceinline__ void cx_bank(unsigned* __restrict__ r, const unsigned* s, const unsigned* x)
{
    static_assert(N == 1 || N == 2 || N == 4, "group size");
    if constexpr (LM == 4) BHS_CX_BANKN("row_shl:4", "0x5", "row_shr:4", "0xa");
    else if constexpr (LM == 8) BHS_CX_BANKN("row_ror:8", "0x3", "row_ror:8", "0xc");
    else if constexpr (LM == 7) BHS_CX_BANKN("row_half_mirror", "0x5", "row_half_mirror", "0xa");
    else { static_assert(LM == 15, "bank-masked partner"); BHS_CX_BANKN("row_mirror", "0x3", "row_mirror", "0xc"); }
}

// min (lane whose deciding bit of LM is 0) / max (the other) of x and src[lane ^ LM], LM in {1,2,3,16,32,31,63}
template <int LM>
__device__ __forceinline__ unsigned cx_asc(unsigned x, unsigned src, int lane)
{
    unsigned y;
    int bit;
    if constexpr (LM == 1) { y = dpp_u32<0xB1, 0xf, 0xf, true>(src, src); bit = 1; }          // quad_perm [1,0,3,2]
    else if constexpr (LM == 2) { y = dpp_u32<0x4E, 0xf, 0xf, true>(src, src); bit = 2; }     // quad_perm [2,3,0,1]
    else if constexpr (LM == 3) { y = dpp_u32<0x1B, 0xf, 0xf, true>(src, src); bit = 2; }     // quad_perm [3,2,1,0]
    else if constexpr (LM == 16) { y = lane_xor<16>(src, lane); bit = 16; }
    else if constexpr (LM == 32) { y = lane_xor<32>(src, lane); bit = 32; }
    else if constexpr (LM == 31) {
        y = lane_xor<16>(dpp_u32<0x140, 0xf, 0xf, true>(src, src), lane);                      // row_mirror, then ^16
        bit = 16;
    } else {
        static_assert(LM == 63, "unsupported lane mask");
        y = lane_xor<16>(dpp_u32<0x140, 0xf, 0xf, true>(src, src), lane);
        y = lane_xor<32>(y, lane);
        bit = 32;
    }
    const unsigned lo = x < y ? x : y, hi = x < y ? y : x;
    return (lane & bit) ? hi : lo;
}

// one lane-partner step over the lane's E keys: x[e] against s[e] of lane ^ LM
template <int LM, int E>
__device__ __forceinline__ void cx_step(unsigned (&x)[E], const unsigned (&s)[E], int lane)
{
    unsigned r[E];
    if constexpr (LM == 4 || LM == 8 || LM == 7 || LM == 15) {
        constexpr int G = E >= 4 ? 4 : E;
#pragma unroll
        for (int e = 0; e < E; e += G) cx_bank<LM, G>(&r[e], &s[e], &x[e]);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = cx_asc<LM>(x[e], s[e], lane);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = r[e];
}

template <int E, int J>
__device__ __forceinline__ void flip_sort_cleaners(unsigned (&x)[E], int lane)
{
    if constexpr (J >= 1) {
        if constexpr (J >= E) {
            cx_step<J / E, E>(x, x, lane);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & J) == 0) {
                    const unsigned a = x[e], b = x[e | J];
                    x[e] = a < b ? a : b;
                    x[e | J] = a < b ? b : a;
                }
        }
        flip_sort_cleaners<E, J / 2>(x, lane);
    }
}

template <int E, int N, int KK>
__device__ __forceinline__ void flip_sort_levels(unsigned (&x)[E], int lane)
{
    if constexpr (KK <= N) {
        if constexpr (KK <= E) {                       // flip inside the lane's own elements
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int q = e ^ (KK - 1);
                if (e < q) {
                    const unsigned a = x[e], b = x[q];
                    x[e] = a < b ? a : b;
                    x[q] = a < b ? b : a;
                }
            }
        } else {                                       // flip across lanes: partner lane ^ (KK/E - 1), element E-1-e
            unsigned rev[E];
#pragma unroll
            for (int e = 0; e < E; ++e) rev[e] = x[E - 1 - e];
            cx_step<KK / E - 1, E>(x, rev, lane);
        }
        flip_sort_cleaners<E, KK / 4>(x, lane);
        flip_sort_levels<E, N, KK * 2>(x, lane);
    }
}

// all lanes of the wave must be active
template <int E, int GW = 64>
__device__ __forceinline__ void wave_flip_sort_u32(unsigned (&x)[E], int lane)
{
    static_assert(GW == 64 || GW == 16, "whole wave or one DPP row");
    asm volatile("s_nop 4");                           // an EXEC write may precede us: DPP wants 5 wait states after it
    flip_sort_levels<E, GW * E, 2>(x, lane & (GW - 1));
}

}  // namespace bhs
