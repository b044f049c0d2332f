// bhs_kernels.hip.h — gfx950 (MI355X, CDNA4) device kernels of the CSR SpGEMM
// hot path.  Written for 64-lane wavefronts, LDS-resident per-row hash
// accumulators and ballot/shuffle wave primitives; no MFMA (irregular
// gather/merge), no CUDA-compat layer.
//
// Reference functions these kernels replace (SpGEMM_cuda/bhsparse_cuda.h):
//   k_upper_bound      <- compute_nnzCt_cudakernel            :210-237
//   k_fill_queues      <- bhsparse::statistics (host)         bhsparse.h:365-481
//   k_row_hash<..,0>   <- (symbolic) no counterpart: the reference sizes Ct by
//                         upper bound and compacts later (create_Ct :285-301,
//                         copyCt2C_* :2813-2911); here an exact count replaces both
//   k_row_hash<..,1>   <- ESC_0/ESC_1 :1582-1640, ESC_2heap_noncoalesced :653-722,
//                         ESC_bitonic_scan :1400-1518, EM_mergepath :1902-2157,
//                         EM_mergepath_global :2270-2525 (all numeric families)
//   k_scan_*           <- create_C's host exclusive scan      :2783-2811
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bhs {

constexpr int kEmpty = -1;          // empty hash slot (column indices are >= 0)
constexpr int kMaxBins = 16;

struct BinSpec {                    // bin b holds rows with upper[b-1] < v <= upper[b]; bin 0: v == 0
    int nbins;
    int upper[kMaxBins];
};

__device__ __forceinline__ int bin_of(const BinSpec& s, int v)
{
    int b = 0;
#pragma unroll
    for (int i = 0; i < kMaxBins; ++i) b += (i < s.nbins - 1 && v > s.upper[i]) ? 1 : 0;
    return b;
}

__device__ __forceinline__ unsigned hash_col(int col, int log2ts)
{
    return ((unsigned)col * 2654435761u) >> (32 - log2ts);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------
// Stage 1a: per-row upper bound ub[i] = sum_{j in A(i,:)} len(B(j,:)), with G
// lanes cooperating on one row (G chosen from the average row length of A so
// that colIndA reads are coalesced and lanes are busy).  Also: total product
// count (int64), histogram of symbolic bins, and rowCnt[i] = 0 for empty rows.
// ---------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void k_upper_bound(int m, const int* __restrict__ Ap,
                                                     const int* __restrict__ Aj,
                                                     const int* __restrict__ Bp, int* __restrict__ ub,
                                                     int* __restrict__ cnt,
                                                     unsigned long long* __restrict__ total,
                                                     int* __restrict__ binCount, BinSpec spec)
{
    __shared__ int hist[kMaxBins];
    __shared__ unsigned long long bsum;
    const int tid = threadIdx.x;
    if (tid < kMaxBins) hist[tid] = 0;
    if (tid == 0) bsum = 0;
    __syncthreads();
    const int rows_per_block = 256 / G;
    const int g = tid % G;
    for (long long rbase = (long long)blockIdx.x * rows_per_block; rbase < m;
         rbase += (long long)gridDim.x * rows_per_block) {
        const int row = (int)rbase + tid / G;
        long long s = 0;
        if (row < m) {
            const int a1 = Ap[row + 1];
            for (int j = Ap[row] + g; j < a1; j += G) {
                const int c = Aj[j];
                s += Bp[c + 1] - Bp[c];
            }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (row < m && g == 0) {
            const int v = s > 0x7fffffffLL ? 0x7fffffff : (int)s;
            ub[row] = v;
            if (v == 0) cnt[row] = 0;          // ESC_0 (bhsparse_cuda.h:1582-1595): nothing else to do
            atomicAdd(&hist[bin_of(spec, v)], 1);
            atomicAdd(&bsum, (unsigned long long)s);
        }
    }
    __syncthreads();
    if (tid < spec.nbins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
    if (tid == 0 && bsum) atomicAdd(total, bsum);
}

// ---------------------------------------------------------------------------
// Stage 1b / 3b: scatter row ids into per-bin queues.  key[] is ub (symbolic
// bins) or the per-row nnz (numeric bins; given as rowPtrC so v = Cp[i+1]-Cp[i]).
// One global atomic per (block, bin); rows of a block stay together so queue
// order stays close to row order (L2 locality of the B rows they touch).
// ---------------------------------------------------------------------------
template <bool FROM_ROWPTR>
__global__ __launch_bounds__(256) void k_fill_queues(int m, const int* __restrict__ key,
                                                     const int* __restrict__ Ap, const int* __restrict__ ub,
                                                     const int* __restrict__ binStart,
                                                     int* __restrict__ binCursor, int* __restrict__ queue,
                                                     BinSpec spec, unsigned long long* __restrict__ binSums)
{
    __shared__ int hist[kMaxBins];
    __shared__ int base[kMaxBins];
    __shared__ unsigned long long sums[kMaxBins * 3];   // per bin: products, nnz(C rows), nnz(A rows)
    const int tid = threadIdx.x;
    if (tid < kMaxBins * 3) sums[tid] = 0;
    for (long long r0 = (long long)blockIdx.x * 256; r0 < m; r0 += (long long)gridDim.x * 256) {
        if (tid < kMaxBins) hist[tid] = 0;
        __syncthreads();
        const int row = (int)r0 + tid;
        int b = -1, pos = 0;
        if (row < m) {
            const int v = FROM_ROWPTR ? key[row + 1] - key[row] : key[row];
            b = bin_of(spec, v);
            if (b > 0) {
                pos = atomicAdd(&hist[b], 1);
                atomicAdd(&sums[b * 3 + 0], (unsigned long long)(unsigned)ub[row]);
                if (FROM_ROWPTR) atomicAdd(&sums[b * 3 + 1], (unsigned long long)v);
                atomicAdd(&sums[b * 3 + 2], (unsigned long long)(Ap[row + 1] - Ap[row]));
            } else b = -1;                                       // bin 0 (empty rows) has no queue
        }
        __syncthreads();
        if (tid < spec.nbins && hist[tid]) base[tid] = binStart[tid] + atomicAdd(&binCursor[tid], hist[tid]);
        __syncthreads();
        if (b > 0) queue[base[b] + pos] = row;
        __syncthreads();
    }
    if (tid < kMaxBins * 3 && sums[tid]) atomicAdd(&binSums[tid], sums[tid]);
}

// ---------------------------------------------------------------------------
// Stage 3a: exclusive scan of the per-row counts into rowPtrC (int32), total in
// int64 so that nnz(C) >= 2^31 is detected instead of wrapping, plus the
// histogram of numeric bins.  Reduce -> scan of block sums -> apply.
// ---------------------------------------------------------------------------
constexpr int kScanItems = 16;                  // per thread
constexpr int kScanTile = 256 * kScanItems;     // per block

__global__ __launch_bounds__(256) void k_scan_reduce(int m, const int* __restrict__ cnt,
                                                     long long* __restrict__ blockSum,
                                                     int* __restrict__ binCount, BinSpec spec)
{
    __shared__ int hist[kMaxBins];
    __shared__ long long wsum[4];
    const int tid = threadIdx.x;
    if (tid < kMaxBins) hist[tid] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kScanTile;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const long long idx = base + (long long)i * 256 + tid;
        if (idx < m) {
            const int v = cnt[idx];
            s += v;
            const int b = bin_of(spec, v);
            if (b > 0) atomicAdd(&hist[b], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) blockSum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tid < spec.nbins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
}

__global__ __launch_bounds__(1024) void k_scan_blocksums(int nb, long long* __restrict__ blockSum,
                                                         long long* __restrict__ total)
{
    __shared__ long long wtot[16];
    __shared__ long long carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + tid;
        const long long v = i < nb ? blockSum[i] : 0;
        long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wtot[w] = x;
        __syncthreads();
        long long woff = 0;
        for (int q = 0; q < w; ++q) woff += wtot[q];
        const long long c = carry;
        if (i < nb) blockSum[i] = c + woff + x - v;          // exclusive
        __syncthreads();
        if (tid == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(int m, int* __restrict__ cnt_to_ptr,
                                                    const long long* __restrict__ blockOff)
{
    // covers indices 0..m (element m counts as 0, so rowPtrC[m] = total falls out)
    __shared__ long long wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long base = (long long)blockIdx.x * kScanTile + (long long)tid * kScanItems;
    int v[kScanItems];
    long long s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < m) ? cnt_to_ptr[base + i] : 0; s += v[i]; }
    long long x = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wtot[w] = x;
    __syncthreads();
    long long off = blockOff[blockIdx.x] + x - s;
    for (int q = 0; q < w; ++q) off += wtot[q];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i <= m) cnt_to_ptr[base + i] = (int)off;
        off += v[i];
    }
}

// B-row sortedness check (reference precondition for EM_mergepath,
// bhsparse_cuda.h:1902ff; here only the column-window path relies on it).
__global__ __launch_bounds__(256) void k_check_sorted(int k, const int* __restrict__ Bp,
                                                      const int* __restrict__ Bj, int* __restrict__ flag)
{
    const long long nnz = Bp[k];
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i + 1 < nnz; i += (long long)gridDim.x * 256)
        if (Bj[i] >= Bj[i + 1]) {
            // unsorted only if i and i+1 are in the same row: find via binary search of i+1 in Bp
            int lo = 0, hi = k;                     // largest r with Bp[r] <= i+1
            while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (Bp[mid] <= i + 1) lo = mid; else hi = mid - 1; }
            if (Bp[lo] != i + 1) bad = 1;           // i+1 is not the first entry of its row
        }
    if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------
// The accumulator kernel.  One thread group of BLOCK lanes per row of C
// (BLOCK = 64: one wavefront per row, the workhorse; BLOCK >= 256: one
// workgroup per long row).  Per row:
//   1. clear an LDS open-addressing table of TS slots (keys int32 [+ fp64 vals])
//   2. expand the products: sub-groups of L = 2^logL lanes walk one B row each
//      (coalesced colIndB/valB segments), insert by multiplicative hash +
//      linear probing; a plain ds_read first (duplicates dominate: 83% of the
//      products on poisson27pt), ds_cmpst only on an empty slot, ds_add_f64 to
//      accumulate
//   3a. SYMBOLIC (NUM=0): wave-reduce the number of successful inserts -> cnt[row]
//   3b. NUMERIC (NUM=1): compact the occupied slots as packed (col<<32 | slot)
//      into LDS, bitonic-sort them there, and stream the row out once, in final
//      CSR position, ascending by column
// WIN=1 (only with BLOCK >= 256) adds the column-window loop for rows whose
// accumulator does not fit the table: the row is produced in successive column
// ranges [lo,hi), each range small enough for the table; an overflowing range
// is halved and retried (replaces the reference's progressive re-allocation
// rounds, bhsparse_cuda.h:2527-2780).  Windows come out in ascending column
// order, so the concatenation is sorted.
// ---------------------------------------------------------------------------
template <int BLOCK>
__device__ __forceinline__ void group_sync()
{
    __syncthreads();   // BLOCK == 64: one wave per workgroup, lowers to a waitcnt (no s_barrier)
}

template <int TS, int BLOCK, bool NUM>
struct RowHashSmem {
    int keys[TS];
    double vals[NUM ? TS : 1];
    unsigned long long sorted[NUM ? TS : 1];
    int counter[4];      // [0] unique count, [1] overflow flag, [2] compaction cursor
};

template <int TS, int LOG2TS, int BLOCK, bool NUM, bool WIN>
__global__ __launch_bounds__(BLOCK) void k_row_hash(
    const int* __restrict__ queue, int qn, int ncolsB, int logL, int bSorted,
    const int* __restrict__ Ap, const int* __restrict__ Aj, const double* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const double* __restrict__ Bx,
    const int* __restrict__ ubArr,          // symbolic + WIN: per-row upper bound (first window guess)
    int* __restrict__ CpOrCnt, int* __restrict__ Cj, double* __restrict__ Cx,
    int* __restrict__ errFlag)
{
    static_assert((1 << LOG2TS) == TS, "table size must be 2^LOG2TS");
    static_assert(!WIN || BLOCK > 64, "column windows are a workgroup-per-row feature");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    RowHashSmem<TS, BLOCK, NUM>& sm = *reinterpret_cast<RowHashSmem<TS, BLOCK, NUM>*>(smem_raw);
    constexpr int CAP = TS - TS / 4;      // max unique keys admitted per table fill

    const int tid = threadIdx.x;
    const int L = 1 << logL;
    const int sub = tid >> logL, t = tid & (L - 1), nsub = BLOCK >> logL;

    for (int q = blockIdx.x; q < qn; q += gridDim.x) {
        const int row = queue[q];
        const int a0 = Ap[row], a1 = Ap[row + 1];
        long long outBase = 0;
        if (NUM) outBase = CpOrCnt[row];
        int rowTotal = 0;                 // symbolic: unique count over all windows

        // column window [lo, hi); without WIN a single window covers everything
        long long lo = 0, width = 0x7fffffffLL;
        if (WIN) {
            // first guess: split the column range uniformly by the expected load
            const long long need = NUM ? (long long)(CpOrCnt[row + 1] - CpOrCnt[row]) : (long long)ubArr[row];
            const long long nwin = (need + CAP / 2 - 1) / (CAP / 2);
            width = ncolsB / (nwin > 0 ? nwin : 1);
            if (width < 1) width = 1;
        }
        for (;;) {
            if (WIN && lo >= ncolsB) break;
            const long long hi = WIN ? (lo + width < ncolsB ? lo + width : (long long)ncolsB) : 0x7fffffffLL;
            // ---- 1. clear
            for (int s = tid; s < TS; s += BLOCK) {
                sm.keys[s] = kEmpty;
                if (NUM) sm.vals[s] = 0.0;
            }
            if (tid < 4) sm.counter[tid] = 0;
            group_sync<BLOCK>();

            // ---- 2. expand + insert
            int myNew = 0;
            for (int ja = a0 + sub; ja < a1; ja += nsub) {
                if (WIN && __atomic_load_n(&sm.counter[1], __ATOMIC_RELAXED)) break;
                const int c = Aj[ja];
                int b0 = Bp[c];
                const int b1 = Bp[c + 1];
                double av = 0.0;
                if (NUM) av = Ax[ja];
                if (WIN && bSorted && lo > 0) {              // lower_bound(lo) in the sorted B row
                    int l = b0, r = b1;
                    while (l < r) { const int mid = (l + r) >> 1; if (Bj[mid] < (int)lo) l = mid + 1; else r = mid; }
                    b0 = l;
                }
                for (int jb = b0 + t; jb < b1; jb += L) {
                    const int col = Bj[jb];
                    if (WIN) {
                        if (col >= hi) { if (bSorted) break; else continue; }
                        if (col < lo) continue;
                    }
                    unsigned h = hash_col(col, LOG2TS);
                    bool overflow = false;
                    int probes = 0;
                    for (;;) {
                        int cur = __atomic_load_n(&sm.keys[h], __ATOMIC_RELAXED);
                        if (cur == kEmpty) {
                            cur = atomicCAS(&sm.keys[h], kEmpty, col);
                            if (cur == kEmpty) {
                                ++myNew;
                                if (WIN) {
                                    const int u = atomicAdd(&sm.counter[0], 1);
                                    if (u + 1 > CAP) { atomicOr(&sm.counter[1], 1); overflow = true; }
                                }
                                break;
                            }
                        }
                        if (cur == col) break;
                        h = (h + 1) & (TS - 1);
                        if (WIN && ++probes >= TS) { atomicOr(&sm.counter[1], 1); overflow = true; break; }
                    }
                    if (NUM && !overflow) unsafeAtomicAdd(&sm.vals[h], av * Bx[jb]);
                }
            }
            if (!WIN) {
                myNew = wave_sum(myNew);
                if (BLOCK == 64) { if (tid == 0) sm.counter[0] = myNew; }
                else if ((tid & 63) == 0 && myNew) atomicAdd(&sm.counter[0], myNew);
            }
            group_sync<BLOCK>();
            const int uniq = sm.counter[0];
            const int ovf = WIN ? sm.counter[1] : 0;
            group_sync<BLOCK>();
            if (WIN && ovf) {                      // halve the window and retry the same lo
                if (width <= 1) { if (tid == 0) atomicOr(errFlag, 1); lo = hi; }
                else width = (width + 1) >> 1;
                continue;
            }

            if (!NUM) {
                rowTotal += uniq;
            } else if (uniq > 0) {
                // ---- 3b. compact occupied slots -> packed (col<<32 | slot)
                int P = 2;
                while (P < uniq) P <<= 1;
                int run = 0;                       // BLOCK == 64: running output cursor (wave-uniform)
                for (int s0 = 0; s0 < TS; s0 += BLOCK) {
                    const int s = s0 + tid;
                    const int key = sm.keys[s];
                    const bool valid = key != kEmpty;
                    const unsigned long long bal = __ballot(valid);
                    const int lanePos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                         __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                    int wbase;
                    if (BLOCK == 64) {
                        wbase = run;
                        run += __popcll(bal);
                    } else {
                        wbase = 0;
                        if ((tid & 63) == 0) wbase = atomicAdd(&sm.counter[2], __popcll(bal));
                        wbase = __shfl(wbase, 0, 64);
                    }
                    if (valid)
                        sm.sorted[wbase + lanePos] = ((unsigned long long)(unsigned)key << 32) | (unsigned)s;
                }
                for (int s = uniq + tid; s < P; s += BLOCK) sm.sorted[s] = ~0ull;
                group_sync<BLOCK>();
                // ---- bitonic sort of P packed keys in LDS
                for (int kk = 2; kk <= P; kk <<= 1) {
                    for (int j = kk >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < (P >> 1); i += BLOCK) {
                            const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                            const int b = a | j;
                            const bool up = (a & kk) == 0;
                            const unsigned long long x = sm.sorted[a], y = sm.sorted[b];
                            if ((x > y) == up) { sm.sorted[a] = y; sm.sorted[b] = x; }
                        }
                        group_sync<BLOCK>();
                    }
                }
                // ---- stream the window out at its final CSR position
                for (int r = tid; r < uniq; r += BLOCK) {
                    const unsigned long long e = sm.sorted[r];
                    Cj[outBase + r] = (int)(e >> 32);
                    Cx[outBase + r] = sm.vals[(unsigned)e];
                }
                outBase += uniq;
                group_sync<BLOCK>();
            }
            if (!WIN) break;
            lo = hi;
            if (uniq < CAP / 4 && width < ncolsB) width <<= 1;   // sparse window: grow the next one
        }
        if (!NUM && tid == 0) CpOrCnt[row] = rowTotal;
    }
}

}  // namespace bhs
