// bhs_row_span.hip.h -- one wave per row, the row of C accumulated over its COLUMN SPAN instead of a hash table (round 5).
// (Included after bhs_row_wave.hip.h: same queue schedule, same row pipeline, same flat product mapping.)
//
// k_row_wave finds a row's distinct columns by hashing every product into an LDS table, then compacts and sorts the
// table: 330 (symbolic) / 607 (numeric) VALU instructions per row of poisson27pt.  Where the rows of C are NARROW -- a
// banded or block-diagonal matrix, a 2-D grid, a mesh numbered along its geometry: every column of row i within a few
// thousand of i -- the set of columns is a short bitmap over [base, base + span):
//   symbolic   one ds_or_b32 per product; the row's count is the popcount of 64 * WPL words (a word or a few per lane);
//   numeric    the same marks, a wave scan of the words' popcounts (rank of every word's first entry), then every product
//              once more: its entry is rank[word] + popcount(bits below it) -- one LDS read, one ds_add_f64 into a COMPACT
//              array of nnz(row) sums -- and the row leaves already in ascending order: no probe, no compaction, no sort.
// The span is not searched for: base = (smallest column of the A row) - (how far LEFT of its own number any row of B
// reaches), and likewise to the right; both reaches are scanned once per data set (k_b_reach, at bhs_set_data time),
// the A row's smallest / largest column is a wave reduction.  The bound is checked for every row HERE -- a row whose span
// is beyond the bitmap raises bit 1 of the error word and the host repeats the multiply on the hash kernels, like every
// other launch chosen from hand-over hints.  Replaces, for such inputs, ESC_bitonic / EM_mergepath
// (SpGEMM_cuda/bhsparse_cuda.h:1400-1518, 1902-2157) as k_row_wave does.
#pragma once

namespace bhs {

// reach of B's rows: out[0] = max over non-empty rows j of (j - first column), out[1] = max of (last column - j); both
// preset to INT_MIN by the caller (signed: a strictly upper triangular B has a negative left reach)
__global__ __launch_bounds__(256) void k_b_reach(int k, const int* __restrict__ Bp, const int* __restrict__ Bj, int* __restrict__ out)
{
    __shared__ int sl[4], sr[4];
    int left = -0x7fffffff - 1, right = -0x7fffffff - 1;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < k; j += (long long)gridDim.x * 256) {
        const int b0 = Bp[j], b1 = Bp[j + 1];
        if (b1 > b0) {
            left = max(left, (int)j - Bj[b0]);
            right = max(right, Bj[b1 - 1] - (int)j);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { left = max(left, __shfl_xor(left, o, 64)); right = max(right, __shfl_xor(right, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = left; sr[threadIdx.x >> 6] = right; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&out[0], max(max(sl[0], sl[1]), max(sl[2], sl[3])));
        atomicMax(&out[1], max(max(sr[0], sr[1]), max(sr[2], sr[3])));
    }
}
// width of A's rows as their first and last entries say (a hint: the multiply takes the true smallest / largest column)
__global__ __launch_bounds__(256) void k_a_width(int m, const int* __restrict__ Ap, const int* __restrict__ Aj, int* __restrict__ out)
{
    __shared__ int sw[4];
    int w = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256) {
        const int a0 = Ap[i], a1 = Ap[i + 1];
        if (a1 > a0) w = max(w, abs(Aj[a1 - 1] - Aj[a0]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o, 64));
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(sw[0], sw[1]), max(sw[2], sw[3])));
}

constexpr int kSpanMaxBSym = 12, kSpanMaxBNum = 6;                 // product batches per window
template <int WPL, int VCAP, bool NUM>
struct SpanSmem {
    unsigned bits[64 * WPL];                                      // the row's columns: bit (col - base); lane L owns words [L * WPL, (L + 1) * WPL)
    int rank[NUM ? 64 * WPL : 1];                                 // entries of the row in front of a word's
    acc_t vals[NUM ? VCAP : 1];                                   // the row's sums, in order
    int cols[NUM ? VCAP : 1];                                     // ... and columns
    value_t sAv[NUM ? 64 : 1];
    int sBase[64];
    alignas(8) unsigned marks[2 * kSpanMaxBSym];
};

// WPL: bitmap words per lane (2048 * WPL columns); VCAP: entries of the compact accumulator (numeric bins: <= 3/4 of it)
template <int WPL, int VCAP, bool NUM>
__global__ __launch_bounds__(64) void k_row_span(
    const int4* __restrict__ desc, int qn, int chunkLog2, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx, int* __restrict__ cntOut,
    int* __restrict__ Cj, value_t* __restrict__ Cx, const int* __restrict__ Ap, int* __restrict__ ubOut,
    unsigned long long* __restrict__ ctSlots, int* __restrict__ errFlag, int reachL, int reachR)
{
    using Smem = SpanSmem<WPL, VCAP, NUM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
    const int lane = threadIdx.x;
    constexpr int MAXB = NUM ? kSpanMaxBNum : kSpanMaxBSym;
    constexpr int SPANBITS = 2048 * WPL;

    // the XCD-aware persistent schedule and the row pipeline of k_row_wave (bhs_row_wave.hip.h), one wave per workgroup
    const int chunk = 1 << chunkLog2;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, perX = gridDim.x >> 3;
    const int nChunks = (qn + chunk - 1) >> chunkLog2;
    int positions = 0;
    if (nChunks > xcd) {
        positions = ((nChunks - xcd + 7) >> 3) << chunkLog2;
        if (((nChunks - 1) & 7) == xcd) positions -= (nChunks << chunkLog2) - qn;
    }
    const int nIt = lb < positions ? (positions - lb + perX - 1) / perX : 0;
    auto q_of = [&](int it) {
        const int t = lb + it * perX;
        return ((((t >> chunkLog2) << 3) + xcd) << chunkLog2) + (t & (chunk - 1));
    };
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    auto load_desc = [&](int it_) {
        const bool has = it_ < nIt;
        int4 r;
        if (desc) r = desc[q_of(has ? it_ : 0) + vzero];
        else {
            const int q = q_of(has ? it_ : 0) + vzero;
            int2 aa;
            __builtin_memcpy(&aa, Ap + q, 8);
            r = make_int4(q, aa.x, aa.y, NUM ? cntOut[q] : 0);    // (numeric pass: cntOut is rowPtrC)
        }
        if (!has) r = make_int4(-1, 0, 0, 0);
        return r;
    };
    unsigned long long prodSum = 0;
    if (nIt == 0) return;
#pragma unroll
    for (int i = 0; i < WPL; ++i) sm.bits[lane * WPL + i] = 0u;
    int4 dC = load_desc(0), d1 = load_desc(1), d2 = load_desc(2);
    int cC = 0, c1 = 0;
    value_t avC = 0.0, av1 = 0.0;
    if (lane < dC.z - dC.y) { cC = Aj[dC.y + lane]; if (NUM) avC = Ax[dC.y + lane]; }
    if (lane < d1.z - d1.y) { c1 = Aj[d1.y + lane]; if (NUM) av1 = Ax[d1.y + lane]; }
    int2 beC = make_int2(0, 0);
    if (lane < dC.z - dC.y) __builtin_memcpy(&beC, Bp + cC, 8);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                          // (nothing of the prologue pending into the loop: bhs_row_wave.hip.h)
    wave_sync();
    for (int it = 0; it < nIt; ++it) {
        const int4 d3 = load_desc(it + 3);
        int c2 = 0;
        value_t av2 = 0.0;
        if (lane < d2.z - d2.y) { c2 = Aj[d2.y + lane]; if (NUM) av2 = Ax[d2.y + lane]; }
        int2 be1 = make_int2(0, 0);
        if (lane < d1.z - d1.y) __builtin_memcpy(&be1, Bp + c1, 8);

        const int row = dC.x, a0 = dC.y, a1 = dC.z;
        const int nAr = a1 - a0;
        const bool inA = lane < nAr;
        // the span: [smallest column of the A row - B's left reach, largest + right reach]
        int cmin = inA ? cC : 0x7fffffff, cmax = inA ? cC : -0x7fffffff - 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cmin = min(cmin, __shfl_xor(cmin, o, 64)); cmax = max(cmax, __shfl_xor(cmax, o, 64)); }
        const int base = cmin - reachL;
        const long long need = nAr > 0 ? (long long)cmax + reachR - base + 1 : 0;
        const bool fits = nAr <= 64 && need <= SPANBITS;
        if (!fits && lane == 0) atomicOr(errFlag, 2);              // (the hand-over's hints are refuted: the host repeats the multiply on the hash kernels)
        int b0 = beC.x, len = fits && inA ? beC.y - beC.x : 0;
        const value_t av = avC;
        const int incl = wave_incl_scan_dpp(len);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int last = incl - 1;
        const unsigned long long nz = __ballot(len > 0);
        const int jc = mbcnt64(nz);
        wave_sync();
        if (len > 0) {
            sm.sBase[jc] = b0 - (incl - len);
            if (NUM) sm.sAv[jc] = av;
        }
        int nnzRow = 0;
        for (int pass = 0; pass < (NUM ? 2 : 1); ++pass) {
            int done = 0;
            for (int w0 = 0; w0 < total; w0 += 64 * MAXB) {
                const int nb = (total - w0 + 63) >> 6;
                if (lane < 2 * MAXB) sm.marks[lane] = 0;
                wave_sync();
                const int rel = last - w0;
                if (len > 0 && rel >= 0 && rel < 64 * MAXB) atomicOr(&sm.marks[rel >> 5], 1u << (rel & 31));
                wave_sync();
                int col[MAXB];
                value_t bxv[MAXB], avv[MAXB];
#pragma unroll
                for (int u = 0; u < MAXB; ++u) { bxv[u] = (value_t)0; avv[u] = (value_t)0; }
                int cum = done;
#pragma unroll
                for (int u = 0; u < MAXB; ++u) {
                    col[u] = kEmpty;
                    if (u < nb) {
                        const int p = w0 + u * 64 + lane;
                        const unsigned long long mk = *reinterpret_cast<const unsigned long long*>(&sm.marks[2 * u]);
                        const int j = cum + mbcnt64(mk);
                        cum += __popcll(mk);
                        if (p < total) {
                            const long long idx = (long long)sm.sBase[j] + p;
                            col[u] = Bj[idx];
                            if (NUM && pass == 1) { avv[u] = sm.sAv[j]; bxv[u] = Bx[idx]; }
                        }
                    }
                }
                done = cum;
#pragma unroll
                for (int u = 0; u < MAXB; ++u) {
                    if (col[u] != kEmpty) {
                        const unsigned bit = (unsigned)(col[u] - base);
                        if (bit < (unsigned)SPANBITS) {             // (always, by the bound above -- unless the arrays changed under us)
                            if (pass == 0) atomicOr(&sm.bits[bit >> 5], 1u << (bit & 31));
                            else {
                                const unsigned wv = sm.bits[bit >> 5];
                                const int pos = sm.rank[NUM ? (bit >> 5) : 0] + __popc(wv & ((1u << (bit & 31)) - 1u));
                                if (pos < VCAP) unsafeAtomicAdd(&sm.vals[NUM ? pos : 0], (acc_t)avv[u] * (acc_t)bxv[u]);
                            }
                        } else atomicOr(errFlag, 2);
                    }
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);              // (every load of the window consumed: bhs_row_wave.hip.h)
            }
            wave_sync();
            if (pass == 0) {
                // the words' popcounts, their exclusive scan over the wave
                unsigned w[WPL];
                int mine = 0;
#pragma unroll
                for (int i = 0; i < WPL; ++i) { w[i] = sm.bits[lane * WPL + i]; mine += __popc(w[i]); }
                const int inclN = wave_incl_scan_dpp(mine);
                nnzRow = __builtin_amdgcn_readlane(inclN, 63);
                if (NUM) {
                    int r = inclN - mine;
#pragma unroll
                    for (int i = 0; i < WPL; ++i) {
                        sm.rank[NUM ? lane * WPL + i : 0] = r;
                        // the row's columns, in order (a lane's words hold consecutive columns)
                        unsigned ww = w[i];
                        while (ww) {
                            const int b = __ffs((int)ww) - 1;
                            ww &= ww - 1;
                            if (r < VCAP) sm.cols[NUM ? r : 0] = base + 32 * (lane * WPL + i) + b;
                            ++r;
                        }
                    }
                    for (int s = lane; s < nnzRow && s < VCAP; s += 64) sm.vals[NUM ? s : 0] = 0.0;
                    if (nnzRow > VCAP && lane == 0) atomicOr(errFlag, 1);
                } else {
#pragma unroll
                    for (int i = 0; i < WPL; ++i) sm.bits[lane * WPL + i] = 0u;
                }
                wave_sync();
            }
        }
        // ---- rotate the pipeline here, in front of the stores of C (bhs_row_wave.hip.h)
        const int outW = dC.w;
        dC = d1; d1 = d2; d2 = d3;
        avC = av1; av1 = av2;
        cC = c1; c1 = c2;
        beC = be1;
        asm volatile("" : "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(c1), "+v"(beC.x), "+v"(beC.y));
        if (NUM) asm volatile("" : "+v"(av1));
        if (!NUM) {
            if (lane == 0) cntOut[row] = nnzRow;
            if (ubOut) {
                if (lane == 0) ubOut[row] = total;
                prodSum += (unsigned long long)total;
            }
        } else {
            const long long outBase = outW;
            for (int s = lane; s < nnzRow && s < VCAP; s += 64) {
                gen_store_c(&Cj[outBase + s], sm.cols[NUM ? s : 0]);
                gen_store_c(&Cx[outBase + s], (value_t)sm.vals[NUM ? s : 0]);
            }
#pragma unroll
            for (int i = 0; i < WPL; ++i) sm.bits[lane * WPL + i] = 0u;
        }
        wave_sync();
    }
    if (!NUM && ubOut && lane == 0 && prodSum) atomicAdd(&ctSlots[blockIdx.x & 63], prodSum);
}

}  // namespace bhs
