// bhs_row_quad.hip.h -- the quarter-wave accumulator k_row_quad: four tiny rows per wavefront.  (Split from bhs_kernels.hip.h in round 4.)
#pragma once

namespace bhs {

// ===========================================================================
// Quarter-wave accumulator for tiny rows (the reference's ESC_2heap territory,
// bhsparse_cuda.h:653-722: poisson5pt rows have 25 products -> 13 entries).
// FOUR rows per wavefront, 16 lanes each: a DPP "row" is 16 lanes, so the
// segmented scan of the B row lengths, the count reduction and the bitonic sort
// (64 keys per row = 4 per lane, strides <= 8 lanes) never leave the VALU.
// Each quarter owns a 64-slot LDS table; product -> A entry mapping is a 64-bit
// mark word per quarter.  Rows qualify with <= 16 A entries and <= 48 products
// (symbolic) / <= 48 entries (numeric); products beyond 64 are walked in windows.
// ===========================================================================
template <bool NUM, bool PACK32>
struct QuadSmem {
    using packed_t = typename std::conditional<PACK32, unsigned, unsigned long long>::type;
    int keys[4][64];
    acc_t vals[NUM ? 4 : 1][NUM ? 64 : 1];
    packed_t packed[NUM ? 4 : 1][NUM ? 64 : 2];
    value_t sAv[NUM ? 4 : 1][NUM ? 16 : 1];
    int sBase[4][16];
    unsigned long long marks[4];
};

template <bool NUM, bool PACK32>
__global__ __launch_bounds__(64) void k_row_quad(
    const int4* __restrict__ desc, int qn, const int* __restrict__ Ap,
    const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    int* __restrict__ cntOut, int* __restrict__ Cj, value_t* __restrict__ Cx, int* __restrict__ errFlag)
{
    using Smem = QuadSmem<NUM, PACK32>;
    using packed_t = typename Smem::packed_t;
    __shared__ Smem sm;
    constexpr int LOG2TS = 6, TS = 64;
    const int lane = threadIdx.x, g = lane >> 4, l16 = lane & 15;

    // XCD-aware persistent schedule over groups of 4 queue entries
    const int nGroups = (qn + 3) >> 2;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, perX = gridDim.x >> 3;
    const int region = (nGroups + 7) >> 3;
    const int gBeg = xcd * region;
    const int gEnd = gBeg + region < nGroups ? gBeg + region : nGroups;

    // Row pipeline (same shape as k_row_wave): descriptor of group i+3, A entries of group i+2 and
    // B extents of group i+1 are in flight while group i is accumulated, so the three dependent
    // global round trips of a row never sit on the critical path.
    auto load_desc = [&](int grp_) {
        int4 r = make_int4(-1, 0, 0, 0);
        if (grp_ < gEnd && grp_ * 4 + g < qn) {
            const int q = grp_ * 4 + g;
            if (desc) r = desc[q];
            else r = make_int4(q, Ap[q], Ap[q + 1], NUM ? cntOut[q] : 0);   // direct: no queue, entry q is row q (see k_row_lane)
        }
        return r;
    };
    auto load_a = [&](const int4& dd, int& c_, value_t& av_) {
        c_ = -1;
        av_ = 0.0;
        const int nA_ = dd.x >= 0 ? dd.z - dd.y : 0;
        if (l16 < nA_) {
            c_ = Aj[dd.y + l16];
            if (NUM) av_ = Ax[dd.y + l16];
        }
    };
    auto load_b = [&](int c_, int2& be_) {                  // raw (begin, end): the length is formed where it is used
        be_ = make_int2(0, 0);
        if (c_ >= 0) __builtin_memcpy(&be_, Bp + c_, sizeof(be_));
    };
    const int g0 = gBeg + lb;
    int4 dC = load_desc(g0), d1 = load_desc(g0 + perX), d2 = load_desc(g0 + 2 * perX);
    int cC, c1;
    int2 beC;
    value_t avC, av1;
    load_a(dC, cC, avC);
    load_a(d1, c1, av1);
    load_b(cC, beC);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                     // prologue loads complete (see k_row_wave)
    for (int grp = g0; grp < gEnd; grp += perX) {
        const int4 d = dC;                                     // this quarter's row (row < 0: idle quarter)
        const int4 d3 = load_desc(grp + 3 * perX);
        int c2;
        int2 be1;
        value_t av2;
        load_a(d2, c2, av2);
        load_b(c1, be1);
        // ---- one A entry per lane of the quarter
        const int b0 = beC.x, len = beC.y - beC.x;
        const value_t av = avC;
        // ---- clear the four tables (64 lanes x 4 slots = 256 slots)
        *reinterpret_cast<int4*>(&sm.keys[0][lane * 4]) = make_int4(kEmpty, kEmpty, kEmpty, kEmpty);
        if (NUM) {
            *reinterpret_cast<double2*>(&sm.vals[0][lane * 4]) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(&sm.vals[0][lane * 4 + 2]) = make_double2(0.0, 0.0);
        }
        // segmented inclusive scan inside each 16-lane DPP row
        unsigned sc = (unsigned)len;
        sc += dpp_u32<0x111, 0xf, 0xf, true>(0, sc);
        sc += dpp_u32<0x112, 0xf, 0xf, true>(0, sc);
        sc += dpp_u32<0x114, 0xf, 0xf, true>(0, sc);
        sc += dpp_u32<0x118, 0xf, 0xf, true>(0, sc);
        const int incl = (int)sc;
        const int total = __shfl(incl, (lane & 48) | 15, 64);   // products of this quarter's row
        int maxTotal = __builtin_amdgcn_readlane(incl, 15);
        maxTotal = max(maxTotal, __builtin_amdgcn_readlane(incl, 31));
        maxTotal = max(maxTotal, __builtin_amdgcn_readlane(incl, 47));
        maxTotal = max(maxTotal, __builtin_amdgcn_readlane(incl, 63));
        const int last = incl - 1;
        const unsigned long long nz = __ballot(len > 0);
        const unsigned gmaskNz = (unsigned)(nz >> (g * 16)) & 0xffffu;
        const int jc = __popc(gmaskNz & ((1u << l16) - 1u));     // compacted index among the quarter's non-empty entries
        wave_sync();
        if (len > 0) {
            sm.sBase[g][jc] = b0 - (incl - len);
            if (NUM) sm.sAv[g][jc] = av;
        }
        int myNew = 0;
        int done = 0;
        for (int w0 = 0; w0 < maxTotal; w0 += 64) {
            if (l16 == 0) sm.marks[g] = 0ull;
            wave_sync();
            const int rel = last - w0;
            if (len > 0 && rel >= 0 && rel < 64) atomicOr(&sm.marks[g], 1ull << rel);
            wave_sync();
            const unsigned long long mk = sm.marks[g];
            int col[4];
            value_t bxq[4], avq[4];                                 // multiplied at insert time: no wait behind each load
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                col[u] = kEmpty;
                bxq[u] = 0.0;                                       // (left uninitialised the kernel gets slower: measured)
                avq[u] = 0.0;
                const int pr = u * 16 + l16;                        // product index inside the window
                const int p = w0 + pr;
                if (p < total) {
                    const int j = done + __popcll(mk & ((1ull << pr) - 1ull));
                    const long long idx = (long long)sm.sBase[g][j] + p;
                    col[u] = Bj[idx];
                    if (NUM) { avq[u] = sm.sAv[g][j]; bxq[u] = Bx[idx]; }
                }
            }
            done += __popcll(mk);
            unsigned hh[4];
            int cur[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                hh[u] = hash_col(col[u], LOG2TS);
                cur[u] = kEmpty;
                if (col[u] != kEmpty) cur[u] = atomicCAS(&sm.keys[g][hh[u]], kEmpty, col[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cv = col[u];
                if (cv != kEmpty) {
                    bool ok = cur[u] == cv;
                    if (cur[u] == kEmpty) { ++myNew; ok = true; }
                    if (!ok) {
                        unsigned h = hh[u];
                        int left = TS;                        // bounded probing (see k_row_wave)
                        for (;;) {
                            h = (h + 1) & (TS - 1);
                            const int c2 = atomicCAS(&sm.keys[g][h], kEmpty, cv);
                            if (c2 == kEmpty) { ++myNew; break; }
                            if (c2 == cv) break;
                            if (--left == 0) { atomicOr(errFlag, 1); break; }
                        }
                        hh[u] = h;
                    }
                    if (NUM) unsafeAtomicAdd(&sm.vals[g][hh[u]], (acc_t)avq[u] * (acc_t)bxq[u]);
                }
            }
            __builtin_amdgcn_s_waitcnt(kWaitVm0);             // the window's loads are consumed (see k_row_wave)
        }
        wave_sync();
        // ---- rotate the pipeline ahead of the stores of C (see k_row_wave)
        dC = d1; d1 = d2; d2 = d3;
        avC = av1; av1 = av2;
        c1 = c2;
        beC = be1;
        asm volatile("" : "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(c1), "+v"(beC.x), "+v"(beC.y));
        if (NUM) asm volatile("" : "+v"(av1));
        if (!NUM) {
            // per-quarter sum of myNew: DPP row reduction, lane 15 of the row holds it
            unsigned r = (unsigned)myNew;
            r += dpp_u32<0x111, 0xf, 0xf, true>(0, r);
            r += dpp_u32<0x112, 0xf, 0xf, true>(0, r);
            r += dpp_u32<0x114, 0xf, 0xf, true>(0, r);
            r += dpp_u32<0x118, 0xf, 0xf, true>(0, r);
            if (l16 == 15 && d.x >= 0) cntOut[d.x] = (int)r;
        } else {
            // ---- compact each quarter's 64 slots, 16 at a time
            int run = 0;
#pragma unroll
            for (int s0 = 0; s0 < 64; s0 += 16) {
                const int s = s0 + l16;
                const int key = sm.keys[g][s];
                const bool valid = key != kEmpty;
                const unsigned long long bal = __ballot(valid);
                const unsigned gm = (unsigned)(bal >> (g * 16)) & 0xffffu;
                if (valid) {
                    packed_t pk;
                    if constexpr (PACK32) pk = ((unsigned)key << LOG2TS) | (unsigned)s;
                    else pk = ((unsigned long long)(unsigned)key << 32) | (unsigned)s;
                    sm.packed[g][run + __popc(gm & ((1u << l16) - 1u))] = pk;
                }
                run += __popc(gm);
            }
            const int uniq = run;
            wave_sync();
            const long long outBase = d.w;
            if (__ballot(uniq > 16) == 0ull) {
                // all four rows have <= 16 entries (poisson5pt: 13): one key per lane, 10 DPP stages
                packed_t x1[1];
                x1[0] = l16 < uniq ? sm.packed[g][l16] : (packed_t)~(packed_t)0;
                wave_bitonic_sort<packed_t, 1, 16>(x1, lane);
                if (l16 < uniq) {
                    int c;
                    unsigned slot;
                    if constexpr (PACK32) { c = (int)(x1[0] >> LOG2TS); slot = x1[0] & 63u; }
                    else { c = (int)(x1[0] >> 32); slot = (unsigned)x1[0]; }
                    gen_store_c(&Cj[outBase + l16], c);
                    gen_store_c(&Cx[outBase + l16], (value_t)sm.vals[g][slot]);
                }
            } else {
                packed_t x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = l16 * 4 + e;
                    x[e] = i < uniq ? sm.packed[g][i] : (packed_t)~(packed_t)0;
                }
                wave_bitonic_sort<packed_t, 4, 16>(x, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = l16 * 4 + e;
                    if (r < uniq) {
                        int c;
                        unsigned slot;
                        if constexpr (PACK32) { c = (int)(x[e] >> LOG2TS); slot = x[e] & 63u; }
                        else { c = (int)(x[e] >> 32); slot = (unsigned)x[e]; }
                        gen_store_c(&Cj[outBase + r], c);
                        gen_store_c(&Cx[outBase + r], (value_t)sm.vals[g][slot]);
                    }
                }
            }
        }
        wave_sync();
    }
}

}  // namespace bhs
