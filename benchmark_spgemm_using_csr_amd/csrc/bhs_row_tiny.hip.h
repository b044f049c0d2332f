// bhs_row_tiny.hip.h -- one row per lane, ALL of the row's products in registers: rows of <= 32 products (round 5).
// (Included after bhs_row_lane.hip.h.)
//
// k_row_lane merges a row's <= K sorted B rows head by head: every step of the merge ends in a load of the next entry
// of the heads that advanced -- thirteen dependent round trips for a row of poisson5pt (25 products, 13 results), at the
// three waves per SIMD its staging buffers leave.  Where every row of A has <= KA entries and every row of B <= LB with
// KA * LB <= 32 (the 5-point stencil, a road-like grid graph: BASELINE's configs[1]) a lane can hold the whole row: two
// rounds of independent loads (A's entries and their B extents, then every B entry), a 32-key bitonic sorting network
// on (column, product) pairs in registers -- 240 compare-exchanges of 7 instructions, shared by 64 rows --, equal
// neighbours added up, the results staged in LDS and written S lanes per row like k_row_lane's.  Replaces
// ESC_2heap_noncoalesced (SpGEMM_cuda/bhsparse_cuda.h:520-722) for such inputs.  The bounds come from the hand-over's
// scans and are verified here for every row (bit 1 of the error word: the host repeats the multiply elsewhere).
#pragma once

namespace bhs {

template <int KA, int LB, bool NUM>
__global__ __launch_bounds__(256) void k_row_tiny(const int4* __restrict__ desc, int qn, const int* __restrict__ Ap,
                                                  const int* __restrict__ Aj, const value_t* __restrict__ Ax,
                                                  const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                  const value_t* __restrict__ Bx, int* __restrict__ cntOut,
                                                  int* __restrict__ Cj, value_t* __restrict__ Cx, int* __restrict__ ubOut,
                                                  unsigned long long* __restrict__ ctSlots, int* __restrict__ errFlag)
{
    static_assert(KA * LB <= 32, "a row's products fit 32 registers");
    constexpr int NP = 32, kEnd = 0x7fffffff, kDead = -1;
    constexpr int S = 16, SP = S + 1, RPP = 64 / S;                // staging: S results per row and round, written by S lanes per row
    __shared__ int sCol[NUM ? 4 : 1][NUM ? 64 * SP : 1];
    __shared__ value_t sVal[NUM ? 4 : 1][NUM ? 64 * SP : 1];
    __shared__ int sN[NUM ? 4 : 1][NUM ? 64 : 1];
    __shared__ int sOut[NUM ? 4 : 1][NUM ? 64 : 1];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool has = q < qn;
    int4 d = make_int4(0, 0, 0, 0);
    if (has) d = desc ? desc[q] : make_int4(q, Ap[q], Ap[q + 1], NUM ? cntOut[q] : 0);
    const int row = d.x, a0 = d.y;
    int nA = d.z - d.y;
    bool bad = has && nA > KA;
    nA = min(nA, KA);
    // round 1: the A entries and the extents of their B rows
    int b0[KA], ln[KA];
    acc_t av[KA];
#pragma unroll
    for (int j = 0; j < KA; ++j) {
        b0[j] = ln[j] = 0;
        av[j] = 0.0;
        if (j < nA) {
            const int c = Aj[a0 + j];
            if (NUM) av[j] = (acc_t)Ax[a0 + j];
            int2 be;
            __builtin_memcpy(&be, Bp + c, sizeof(be));
            b0[j] = be.x;
            ln[j] = be.y - be.x;
        }
    }
    long long prods = 0;
#pragma unroll
    for (int j = 0; j < KA; ++j) {
        prods += ln[j];
        bad = bad || ln[j] > LB;
        ln[j] = min(ln[j], LB);
    }
    if (__any(bad) && lane == 0) atomicOr(errFlag, 2);             // (a row beyond the bounds of the hand-over: the host repeats the multiply)
    // round 2: every product
    int key[NP];
    acc_t val[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { key[i] = kEnd; val[i] = 0.0; }
#pragma unroll
    for (int j = 0; j < KA; ++j)
#pragma unroll
        for (int e = 0; e < LB; ++e)
            if (e < ln[j]) {
                key[j * LB + e] = Bj[b0[j] + e];
                if (NUM) val[j * LB + e] = av[j] * (acc_t)Bx[b0[j] + e];
            }
    // bitonic sorting network, ascending by column
#pragma unroll
    for (int kk = 2; kk <= NP; kk <<= 1)
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int l = i ^ jj;
                if (l > i) {
                    const bool up = (i & kk) == 0;
                    const bool sw = up ? key[i] > key[l] : key[i] < key[l];
                    const int ki = key[i], kl = key[l];
                    key[i] = sw ? kl : ki;
                    key[l] = sw ? ki : kl;
                    if (NUM) {
                        const acc_t vi = val[i], vl = val[l];
                        val[i] = sw ? vl : vi;
                        val[l] = sw ? vi : vl;
                    }
                }
            }
    // equal neighbours: the later one takes the sum, the earlier one dies
    int cnt = 0;
#pragma unroll
    for (int i = 1; i < NP; ++i) {
        if (key[i] == key[i - 1] && key[i] != kEnd) {
            if (NUM) val[i] += val[i - 1];
            key[i - 1] = kDead;
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) cnt += (key[i] != kEnd && key[i] != kDead) ? 1 : 0;
    if constexpr (!NUM) {
        if (has) cntOut[row] = cnt;
        if (ubOut) {
            __shared__ unsigned long long bsum;
            if (threadIdx.x == 0) bsum = 0;
            __syncthreads();
            if (has) ubOut[row] = (int)prods;
            unsigned long long t = has ? (unsigned long long)prods : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0 && t) atomicAdd(&bsum, t);
            __syncthreads();
            if (threadIdx.x == 0 && bsum) atomicAdd(&ctSlots[blockIdx.x & 63], bsum);
        }
    } else {
        int out = d.w;
#pragma unroll
        for (int round = 0; round < NP / S; ++round) {
            if (!__any(has && cnt > round * S)) break;
            int pos = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool live = has && key[i] != kEnd && key[i] != kDead;
                if (live && pos >= round * S && pos < (round + 1) * S) {
                    sCol[w][lane * SP + pos - round * S] = key[i];
                    sVal[w][lane * SP + pos - round * S] = (value_t)val[i];
                }
                pos += live ? 1 : 0;
            }
            const int nst = has ? max(0, min(S, cnt - round * S)) : 0;
            sN[w][lane] = nst;
            sOut[w][lane] = out;
            out += nst;
            wave_sync();
#pragma unroll
            for (int pass = 0; pass < S; ++pass) {
                const int r = pass * RPP + lane / S, e = lane % S;
                if (e < sN[w][r]) {
                    const long long o = (long long)sOut[w][r] + e;
                    gen_store_c(&Cj[o], sCol[w][r * SP + e]);
                    gen_store_c(&Cx[o], sVal[w][r * SP + e]);
                }
            }
            wave_sync();
        }
    }
}

}  // namespace bhs
