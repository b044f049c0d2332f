// bhs_row_lane.hip.h -- the lane-per-row kernel k_row_lane: K-way merge of sorted B rows in registers, for matrices whose rows are all tiny.  (Split from bhs_kernels.hip.h in round 4.)
#pragma once

namespace bhs {

// ===========================================================================
// Lane-per-row kernel for matrices whose rows are ALL tiny (stencils in their
// natural ordering: poisson5pt has 5 entries per row, 25 products, 13 results).
// The reference gives such rows a thread each and a heap in shared memory
// (ESC_2heap_noncoalesced, bhsparse_cuda.h:520-722); here a lane merges the
// <= K sorted B rows of its row directly: K heads (position, end, column, A
// value, B value) live in registers, every step emits the smallest head column
// with the sum of the heads that carry it and advances those heads -- no table,
// no sort, no LDS, results leave in ascending order.  64 rows share every
// instruction, so the per-row cost of the wave kernels (scan, marks, compaction,
// sort: ~125 VALU instructions per row in k_row_quad) shrinks to the merge steps
// themselves (~1 instruction per product and head).  Adjacent lanes hold adjacent
// rows, whose B rows are adjacent in memory, so the per-lane loads coalesce for
// banded matrices.  Needs strictly ascending B rows (checked at set_data time).
//
// Measured on MI355X (poisson5pt 1024^2 / poisson7pt 128^3 / poisson9pt 1024^2): the symbolic pass drops from
// 0.081 / 0.476 / 0.273 ms (quarter-wave and wave kernels) to 0.030 / 0.12 / 0.10 ms.  The numeric pass gains
// while K <= 8 (poisson5pt 0.187 -> 0.116 ms, 7pt 0.66 -> 0.51 ms; 9pt loses, 0.36 -> 0.42 ms: two more loads per
// advancing head and 10 heads in registers), and only with its stores staged through LDS (see S below); the host
// routes the numeric stage here for K <= 8 (option "lane_numeric").
// ===========================================================================
// SMALLB: nnz(B) < 2^29, so byte offsets into colIndB / valB fit 32 bits and the loads take the scalar base +
// 32-bit lane offset form: no 64-bit address pair per head.
// waves per SIMD asked of the register allocator (left alone it keeps both arms of every predicated load live:
// 118 VGPRs for K = 6); the numeric pass is bounded by its LDS staging buffers (S = 16: 52 KB per workgroup)
constexpr int kLaneFromCountsBlocks = 8192;       // "rowPtrC on the way": every block of the numeric kernel sums all the blocks' sums -- up to this many (2 M rows)
constexpr int lane_waves(int K, bool NUM) { return !NUM ? (K <= 8 ? 8 : K <= 10 ? 6 : 5) : (BHS_LANE_S == 16 ? 3 : BHS_LANE_S == 8 ? (K <= 10 ? 5 : 4) : (K <= 10 ? 7 : 4)); }

template <int K, bool NUM, bool SMALLB>
__global__ __launch_bounds__(256, lane_waves(K, NUM)) void k_row_lane(const int4* __restrict__ desc, int qn,
                                                  const int* __restrict__ Ap,
                                                  const int* __restrict__ Aj, const value_t* __restrict__ Ax,
                                                  const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                  const value_t* __restrict__ Bx, int* __restrict__ cntOut,
                                                  int* __restrict__ Cj, value_t* __restrict__ Cx,
                                                  int* __restrict__ ubOut, unsigned long long* __restrict__ ctSlots,
                                                  int* __restrict__ errFlag,
                                                  const int* __restrict__ specWord = nullptr,   // launched before the host saw this multiply's counts (k_lane_spec_check): go on only if 1
                                                  // round 6, "rowPtrC on the way": symbolic pass -- blockSums[block] = entries of the block's 256 rows;
                                                  // numeric pass, direct -- cntOut holds the COUNTS: this block sums the blocks' sums before it, scans
                                                  // its own rows and writes rowPtrC in place; nothing is written (specOut = 2) unless the sums of all
                                                  // blocks are the nnz(C) the host assumed when it sized C
                                                  int* __restrict__ blockSums = nullptr, int nBlocks = 0, long long assumedNnzC = 0,
                                                  int* __restrict__ specOut = nullptr)
{
    if (specWord != nullptr && *specWord != 1) return;
    // ubOut != nullptr (symbolic pass of a "lane-first" multiply, where no upper-bound pass ran): the row's product
    // count is written to ubOut and added into one of 64 counters (ctSlots; the host sums them)
    constexpr int kEnd = 0x7fffffff;                       // exhausted head (column indices are < 2^31 - 1)
    // numeric pass: S results per row are staged in LDS (row-major, stride S + 1) and then written by S lanes per
    // row, so that C receives runs of up to S consecutive entries instead of one entry per lane at a stride of a
    // whole row (those 4-byte stores left the L2 as partially written lines: 0.41 ms on poisson5pt, 0.09 ms
    // without the stores).  Longer runs beat occupancy: S = 4 / 8 / 16 -> 0.23 / 0.15 / 0.12 ms on poisson5pt
    // (7 / 5 / 3 waves per SIMD; 13 results per row, so S = 16 writes every row in one piece).
    constexpr int S = BHS_LANE_S, SP = S + 1, RPP = 64 / S;   // RPP rows per flush pass, S lanes each
    __shared__ int sCol[NUM ? 4 : 1][NUM ? 64 * SP : 1];
    __shared__ value_t sVal[NUM ? 4 : 1][NUM ? 64 * SP : 1];
    __shared__ int sN[NUM ? 4 : 1][NUM ? 64 : 1];
    __shared__ int sOut[NUM ? 4 : 1][NUM ? 64 : 1];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool more = q < qn;
    // desc == nullptr ("direct"): every row of the matrix is in this bin, so the queue was never built and entry q
    // is row q (its descriptor comes from rowPtrA and, for the numeric pass, rowPtrC in cntOut)
    int4 d = make_int4(0, 0, 0, 0);
    if (more) d = desc ? desc[q] : make_int4(q, Ap[q], Ap[q + 1], NUM ? cntOut[q] : 0);
    if constexpr (NUM) {
        if (blockSums != nullptr) {                        // (block-uniform: cntOut holds counts, rowPtrC is made here)
            // (blockSums[nBlocks + 1 + b]: entries in the blocks before block b, [2 nBlocks + 1]: in all -- the symbolic kernel's last block)
            __shared__ int sCnt[4];
            const long long* pfx = reinterpret_cast<const long long*>(blockSums + kLaneFromCountsBlocks);
            const long long pre = pfx[blockIdx.x], total = pfx[nBlocks];
            if (total != assumedNnzC) {                        // nothing of C is written: the host runs the multiply again the slow way
                if (blockIdx.x == 0 && threadIdx.x == 0) *specOut = 2;
                return;
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) *specOut = 1;
            const int cnt = more ? d.w : 0;
            const int incl = wave_incl_scan_dpp(cnt);
            if (lane == 63) sCnt[w] = incl;
            __syncthreads();
            long long at = pre + incl - cnt;
            for (int ww = 0; ww < w; ++ww) at += sCnt[ww];
            if (more) {
                cntOut[q] = (int)at;
                if (q == qn - 1) cntOut[qn] = (int)(at + cnt);
            }
            d.w = (int)at;
        }
    }
    const int row = d.x, a0 = d.y, nA = d.z - d.y;
    // lane-first / direct launches rest on the longest row of A seen at bhs_set_data time: verified here (bit 1 of the
    // error word sends the host back through the general pipeline)
    if (!desc && more && nA > K) atomicOr(errFlag, 2);
    auto ld_col = [&](int p) {
        if constexpr (SMALLB) return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(Bj) + ((unsigned)p << 2));
        else return Bj[p];
    };
    auto ld_val = [&](int p) {
        if constexpr (SMALLB)
            return *reinterpret_cast<const value_t*>(reinterpret_cast<const char*>(Bx) + (unsigned)p * (unsigned)sizeof(value_t));
        else return Bx[p];
    };
    int pos[K], end[K], col[K];
    acc_t av[K], bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        pos[j] = end[j] = 0;
        av[j] = 0.0;
        if (j < nA) {
            const int c = Aj[a0 + j];
            if (NUM) av[j] = (acc_t)Ax[a0 + j];
            int2 be;
            __builtin_memcpy(&be, Bp + c, sizeof(be));
            pos[j] = be.x;
            end[j] = be.y;
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        col[j] = kEnd;
        bv[j] = 0.0;
        if (pos[j] < end[j]) { col[j] = ld_col(pos[j]); if (NUM) bv[j] = (acc_t)ld_val(pos[j]); }
    }
    // one merge step: smallest head column, sum of the heads that carry it, those heads advance
    auto step = [&](int& mn, acc_t& sum) {
        mn = col[0];
#pragma unroll
        for (int j = 1; j < K; ++j) mn = min(mn, col[j]);
        sum = 0.0;
        if (mn == kEnd) return false;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (col[j] == mn) {
                if (NUM) sum += av[j] * bv[j];
                ++pos[j];
                col[j] = kEnd;
                if (pos[j] < end[j]) { col[j] = ld_col(pos[j]); if (NUM) bv[j] = (acc_t)ld_val(pos[j]); }
            }
        }
        return true;
    };
    if constexpr (!NUM) {
        long long prods = 0;
        if (ubOut) {
#pragma unroll
            for (int j = 0; j < K; ++j) prods += end[j] - pos[j];
        }
        int cnt = 0, mn;
        acc_t sum;
        while (more) { more = step(mn, sum); cnt += more ? 1 : 0; }
        if (q < qn) cntOut[row] = cnt;
        if (blockSums != nullptr) {                          // (direct launches: block b holds the rows 256 b ..)
            __shared__ int csum[4];
            const int ws = wave_sum_dpp(q < qn ? cnt : 0);
            if (lane == 63) csum[w] = ws;
            __syncthreads();
            if (threadIdx.x == 0) blockSums[blockIdx.x] = csum[0] + csum[1] + csum[2] + csum[3];
        }
        if (ubOut) {
            __shared__ unsigned long long bsum;
            if (threadIdx.x == 0) bsum = 0;
            __syncthreads();
            if (q < qn) ubOut[row] = prods > 0x7fffffffLL ? 0x7fffffff : (int)prods;
            unsigned long long t = (unsigned long long)prods;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0 && t) atomicAdd(&bsum, t);
            __syncthreads();
            if (threadIdx.x == 0 && bsum) atomicAdd(&ctSlots[blockIdx.x & 63], bsum);
        }
    } else {
        int out = d.w;                                      // (nnz(C) < 2^31)
        while (__any(more)) {
            int nst = 0;
#pragma unroll
            for (int e = 0; e < S; ++e) {
                if (more) {
                    int mn;
                    acc_t sum;
                    more = step(mn, sum);
                    if (more) {
                        sCol[w][lane * SP + e] = mn;
                        sVal[w][lane * SP + e] = (value_t)sum;
                        ++nst;
                    }
                }
            }
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

// "rowPtrC on the way" (round 6): the exclusive scan of the symbolic lane kernel's block sums (64-bit) and their total behind it --
// one workgroup, a few microseconds, between the symbolic and the numeric kernel.  (As the symbolic kernel's last block to finish
// it cost 100 us: that block's device-coherent loads of the other blocks' sums come one at a time.)
__global__ __launch_bounds__(1024) void k_lane_block_prefix(int nb, const int* __restrict__ sums, long long* __restrict__ pfx)
{
    __shared__ long long wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (nb + 1023) / 1024;
    long long mine = 0;
    for (int i = tid * per; i < min(nb, (tid + 1) * per); ++i) mine += sums[i];
    long long x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wtot[w] = x;
    __syncthreads();
    long long at = x - mine, total = 0;
    for (int ww = 0; ww < 16; ++ww) { at += ww < w ? wtot[ww] : 0; total += wtot[ww]; }
    for (int i = tid * per; i < min(nb, (tid + 1) * per); ++i) { pfx[i] = at; at += sums[i]; }
    if (tid == 0) pfx[nb] = total;
}

// Speculative numeric launch of a lane-first multiply (round 6; the class path has had one since round 5, bhs_class.hip.h):
// between the scan and the numeric kernel the host reads nnz(C) and the bins' counts back -- 30 us of a 0.2 ms poisson5pt
// multiply.  From a data set's second multiply on the host assumes the last multiply's nnz(C) and "every row in the lane bin",
// launches at once, and this kernel compares on the device; the numeric kernel returns before its first load unless word == 1.
__global__ __launch_bounds__(64) void k_lane_spec_check(long long nnzC, int m, const long long* __restrict__ total, const int* __restrict__ err,
                                                        const int* __restrict__ numCount, int* __restrict__ word)
{
    if (threadIdx.x == 0) *word = (*err == 0 && *total == nnzC && numCount[kLaneBin] == m) ? 1 : 2;
}

}  // namespace bhs
