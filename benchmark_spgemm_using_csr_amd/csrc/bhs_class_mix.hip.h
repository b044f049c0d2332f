// bhs_class_mix.hip.h -- the class path per ROW instead of per data set (round 6).  (Included after bhs_class.hip.h.)
//
// Until round 5 one row without a class sent the whole multiply -- and the data set, for good -- to the general pipeline:
// poisson27pt 128^3 with a single odd row fell from 1.6 to 5.5 ms.  The reference bins every row for itself
// (SpGEMM_cuda/bhsparse.h:483-586: a row's bin follows ITS upper bound, whatever its neighbours look like); so does this:
//   * a row the classifier gives no class (longer than the classifier's lanes take, a B row without a class behind one of
//     its entries, a full table), a row whose class is beyond the tables (k_class_patterns: z < 0) and a row of a class
//     with fewer than kMixMinRows rows (a pattern per single row would cost more than the row) is IRREGULAR;
//   * k_mix_collect lists the irregular rows; k_mix_upper_bound gives them their product count and symbolic bin;
//     the general pipeline's symbolic and numeric kernels (k_row_wave, k_row_block, hub rows ...) run on queues of THOSE
//     rows only -- built by the general pipeline's own k_fill_queues / k_bin_hist from the list;
//   * k_class_scan takes an irregular row's count from the symbolic kernels, every other row's from its class: ONE scan,
//     one rowPtrC;
//   * the ring kernel (bhs_class_ring.hip.h) skips the irregular rows -- a skipped row ends a stretch -- and runs beside
//     the general numeric kernels.
// A perturbed row of B makes every row of A that points at it irregular (its products are not its neighbours' shifted),
// and each of them would claim a class of its own: the classes of B are counted (k_mix_class_hist) and the single-row
// ones pruned (k_mix_prune) BEFORE A is classified, so that the rows of A behind them never reach A's table.
#pragma once

namespace bhs {

constexpr int kMixMinRows = kClassMixMinRows;   // a class with fewer rows than this is not worked out: its rows are irregular

// rows per class: hist[c] += 1 for every row with class c >= 0.  A wave's 64 consecutive rows are mostly one class (rows
// come in stretches): one LDS atomic per run of equal classes in the wave, not per lane.
constexpr int kMixHistBlock = 1024;
__global__ __launch_bounds__(kMixHistBlock) void k_mix_class_hist(int n, const int* __restrict__ cls, int* __restrict__ hist,
                                                                  const int* __restrict__ range)   // rows [range[0], range[1]] only (nullptr: all)
{
    __shared__ int sh[kClassSlots];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < kClassSlots; s += kMixHistBlock) sh[s] = 0;
    __syncthreads();
    long long first = 0, last = n;
    if (range != nullptr) {
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        last = lo <= hi ? min((long long)n, (long long)hi + 1) : 0;
    }
    const long long span = first + (last - first + 63) / 64 * 64;                 // (whole waves take part in the ballots)
    for (long long i = first + (long long)blockIdx.x * kMixHistBlock + tid; i < span; i += (long long)gridDim.x * kMixHistBlock) {
        const int c = i < last ? cls[i] : -1;
        const int prev = __shfl_up(c, 1, 64);
        const unsigned long long starts = __ballot(lane == 0 || c != prev);       // a run of equal classes begins in these lanes
        if ((starts >> lane) & 1ull) {
            const unsigned long long later = lane == 63 ? 0ull : (starts >> (lane + 1));
            const int len = later ? __ffsll((long long)later) : 64 - lane;
            if (c >= 0) atomicAdd(&sh[c], len);
        }
    }
    __syncthreads();
    for (int s = tid; s < kClassSlots; s += kMixHistBlock)
        if (sh[s]) atomicAdd(&hist[s], sh[s]);
}

// the rows of classes with fewer than kMixMinRows rows lose their class (rows of B: the rows of A behind them then find
// none either)
__global__ __launch_bounds__(256) void k_mix_prune(int n, int* __restrict__ cls, const int* __restrict__ hist, const int* __restrict__ range)
{
    long long first = 0, last = n;
    if (range != nullptr) {
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        last = lo <= hi ? min((long long)n, (long long)hi + 1) : 0;
    }
    for (long long i = first + (long long)blockIdx.x * 256 + threadIdx.x; i < last; i += (long long)gridDim.x * 256) {
        const int c = cls[i];
        if (c >= 0 && hist[c] < kMixMinRows) cls[i] = -1;
    }
}

// The irregular rows of A, listed (in no particular order: where a row of C goes is the scan's business): no class, or a
// class k_class_patterns did not work out.  Their classC becomes kClassDummy -- the class without entries or products: the
// kernels behind this one have ONE test, and the ring kernel's row loop none (bhs_class_ring.hip.h).  count[0]: rows listed.
constexpr int kMixCollectPer = 16, kMixCollectTile = 256 * kMixCollectPer;    // rows per thread and per reservation of list space
__global__ __launch_bounds__(256) void k_mix_collect(int m, int* __restrict__ classC, const int4* __restrict__ classInfo,
                                                     int* __restrict__ list, int* __restrict__ count)
{
    // (one reservation of list space per tile of 4096 rows: with one per wave -- 27 k same-address atomics on poisson27pt 128^3
    // with 0.1 % of its rows perturbed -- the kernel took 0.27 ms, all of it atomics queueing on one L2 word)
    __shared__ int wtot[4], sBase;
    __shared__ unsigned char sNoPattern[kClassSlots];                // class -> not worked out (one pass over classInfo per block instead of a gather per row)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = tid; c < kClassSlots; c += 256) sNoPattern[c] = classInfo[c].z < 0 ? 1 : 0;
    __syncthreads();
    for (long long t0 = (long long)blockIdx.x * kMixCollectTile; t0 < m; t0 += (long long)gridDim.x * kMixCollectTile) {
        unsigned badBits = 0;
        int pos[kMixCollectPer], mine = 0;
#pragma unroll
        for (int j = 0; j < kMixCollectPer; ++j) {
            const long long i = t0 + j * 256 + tid;
            bool bad = false;
            if (i < m) {
                const int c = classC[i];
                bad = c < 0 || c >= kClassSlots || sNoPattern[c] != 0;
                if (bad) classC[i] = kClassDummy;
            }
            const unsigned long long mask = __ballot(bad);
            pos[j] = mine + mbcnt64(mask);                           // (place among this wave's rows of the tile)
            mine += __popcll(mask);
            badBits |= bad ? 1u << j : 0u;
        }
        if (lane == 0) wtot[wv] = mine;
        __syncthreads();
        if (tid == 0) {
            const int all = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            sBase = all ? atomicAdd(count, all) : 0;
        }
        __syncthreads();
        int base = sBase;
        for (int w = 0; w < wv; ++w) base += wtot[w];
#pragma unroll
        for (int j = 0; j < kMixCollectPer; ++j)
            if ((badBits >> j) & 1u) list[base + pos[j]] = (int)(t0 + j * 256 + tid);
        __syncthreads();
    }
}

// Product count (upper bound of the row's entries) of every listed row, as k_upper_bound delivers it for all rows: ub[row],
// cnt[row] = 0 where there is no product, the symbolic bins' histogram, the products' total.  Sixteen lanes per row; a
// row of A beyond 256 entries is walked by its whole wave afterwards.
__global__ __launch_bounds__(256) void k_mix_upper_bound(const int* __restrict__ list, const int* __restrict__ count,
                                                         const int* __restrict__ Ap, const int* __restrict__ Aj, const int* __restrict__ Bp,
                                                         int* __restrict__ ub, int* __restrict__ cnt, unsigned long long* __restrict__ total,
                                                         int* __restrict__ binCount, BinSpec spec,
                                                         int* __restrict__ binCount2, BinSpec spec2)   // the same on a second ladder: the host chooses when it knows how many rows there are
{
    __shared__ int hist[kMaxBins], hist2[kMaxBins];
    __shared__ unsigned long long bsum;
    const int tid = threadIdx.x, lane = tid & 63, g = lane & 15, grp = lane >> 4;
    if (tid < kMaxBins) hist[tid] = hist2[tid] = 0;
    if (tid == 0) bsum = 0;
    __syncthreads();
    const int n = *count;
    unsigned long long mySum = 0;
    auto finish = [&](int row, int nA, long long tot) {              // (one lane per row)
        const int v = tot > 0x7fffffffLL ? 0x7fffffff : (int)tot;
        ub[row] = v;
        if (v == 0) cnt[row] = 0;
        atomicAdd(&hist[bin_of(spec, v, nA, v, v)], 1);
        atomicAdd(&hist2[bin_of(spec2, v, nA, v, v)], 1);
        mySum += (unsigned long long)tot;
    };
    const int nPad = (n + 15) / 16 * 16;                             // whole waves take part in the ballots
    for (long long i0 = ((long long)blockIdx.x * 4 + (tid >> 6)) * 4; i0 < nPad; i0 += (long long)gridDim.x * 16) {
        const long long i = i0 + grp;
        const int row = i < n ? list[i] : -1;
        int a0 = 0, a1 = 0;
        if (row >= 0) { a0 = Ap[row]; a1 = Ap[row + 1]; }
        const bool isLong = a1 - a0 > 256;
        long long s = 0;
        if (!isLong)
            for (int j = a0 + g; j < a1; j += 16) {
                const int c = Aj[j];
                s += Bp[c + 1] - Bp[c];
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (row >= 0 && !isLong && g == 0) finish(row, a1 - a0, s);
        unsigned long long longs = __ballot(isLong && g == 0);
        while (longs) {                                              // (wave-uniform)
            const int src = __ffsll((long long)longs) - 1;
            longs &= longs - 1;
            const int r = __shfl(row, src, 64), b0 = __shfl(a0, src, 64), b1 = __shfl(a1, src, 64);
            long long t = 0;
            for (int j = b0 + lane; j < b1; j += 64) {
                const int c = Aj[j];
                t += Bp[c + 1] - Bp[c];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) finish(r, b1 - b0, t);
        }
    }
    if (mySum) atomicAdd(&bsum, mySum);
    __syncthreads();
    if (tid < kMaxBins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
    if (tid < kMaxBins && hist2[tid]) atomicAdd(&binCount2[tid], hist2[tid]);
    if (tid == 0 && bsum) atomicAdd(total, bsum);
}

// The queues of the listed rows, bin by bin (what k_fill_queues does for all rows of a matrix, 4096 rows per block: here a
// few thousand rows want many small blocks).  FROM_ROWPTR: numeric bins -- key = rowPtrC, v = the row's entries, outBase --,
// else symbolic bins by ub.  The bins' starts are summed from their counts by every block for itself (count[]: complete
// when this kernel starts).  A row per thread, 256 rows per reservation; rows outside [rlo, rhi) are left out.
template <bool FROM_ROWPTR>
__global__ __launch_bounds__(256) void k_mix_fill(const int* __restrict__ list, const int* __restrict__ listCount, int rlo, int rhi,
                                                  const int* __restrict__ key, const int* __restrict__ Ap, const int* __restrict__ ub,
                                                  const int* __restrict__ binCount, int* __restrict__ binCursor, int4* __restrict__ queue,
                                                  BinSpec spec, unsigned long long* __restrict__ binSums)
{
    __shared__ int hist[kMaxBins], base[kMaxBins], start[kMaxBins];
    __shared__ unsigned long long sums[kMaxBins * 3];
    const int tid = threadIdx.x;
    if (tid < kMaxBins * 3) sums[tid] = 0;
    if (tid == 0) {
        int at = 0;
        for (int b = 0; b < kMaxBins; ++b) { start[b] = at; at += b == 0 ? 0 : binCount[b]; }
    }
    const int n = *listCount;
    for (long long i0 = (long long)blockIdx.x * 256; i0 < n; i0 += (long long)gridDim.x * 256) {
        if (tid < kMaxBins) hist[tid] = 0;
        __syncthreads();
        const long long idx = i0 + tid;
        int row = idx < n ? list[idx] : -1;
        if (row < rlo || row >= rhi) row = -1;
        int b = 0, pos = 0, a0 = 0, a1 = 0, outBase = 0, v = 0, ubv = 0;
        if (row >= 0) {
            if (FROM_ROWPTR) { outBase = key[row]; v = key[row + 1] - outBase; } else v = key[row];
            a0 = Ap[row];
            a1 = Ap[row + 1];
            ubv = ub[row];
            b = bin_of(spec, v, a1 - a0, FROM_ROWPTR ? v : ubv, ubv);
            if (b > 0) {
                pos = atomicAdd(&hist[b], 1);
                atomicAdd(&sums[b * 3 + 0], (unsigned long long)(unsigned)ubv);
                if (FROM_ROWPTR) atomicAdd(&sums[b * 3 + 1], (unsigned long long)(unsigned)v);
                atomicAdd(&sums[b * 3 + 2], (unsigned long long)(unsigned)(a1 - a0));
            }
        }
        __syncthreads();
        if (tid < kMaxBins && hist[tid]) base[tid] = start[tid] + atomicAdd(&binCursor[tid], hist[tid]);
        __syncthreads();
        if (b > 0) queue[base[b] + pos] = make_int4(row, a0, a1, outBase);
        __syncthreads();
    }
    if (tid < kMaxBins * 3 && sums[tid]) atomicAdd(&binSums[tid], sums[tid]);
}

}  // namespace bhs
