// bhs_kernels.hip.h — gfx950 (MI355X, CDNA4) device kernels of the CSR SpGEMM
// hot path.  Written for 64-lane wavefronts, LDS-resident per-row hash
// accumulators and ballot/shuffle wave primitives; no MFMA (irregular
// gather/merge), no CUDA-compat layer.
//
// Reference functions these kernels replace (SpGEMM_cuda/bhsparse_cuda.h):
//   k_upper_bound      <- compute_nnzCt_cudakernel            :210-237
//   k_fill_queues      <- bhsparse::statistics (host)         bhsparse.h:365-481
//   k_row_{quad,wave,block,spa}<..,NUM=0>
//                      <- (symbolic) no counterpart: the reference sizes Ct by
//                         upper bound and compacts later (create_Ct :285-301,
//                         copyCt2C_* :2813-2911); here an exact count replaces both
//   k_row_*<..,NUM=1>  <- ESC_0/ESC_1 :1582-1640, ESC_2heap_noncoalesced :653-722,
//                         ESC_bitonic_scan :1400-1518, EM_mergepath :1902-2157,
//                         EM_mergepath_global :2270-2525 (all numeric families)
//   k_scan_*           <- create_C's host exclusive scan      :2783-2811
#pragma once
#include <hip/hip_runtime.h>
#include "bhs_wave.hip.h"
#include "bhs_lab.hip.h"
#include <stdint.h>
#include <type_traits>

namespace bhs {

// value_type of the matrices: double (default, SpGEMM_cuda/common.h:31) or float when the library is built
// with -DBHS_VALUE_FLOAT (the reference's other supported build, README.md:84-86)
#ifdef BHS_VALUE_FLOAT
using value_t = float;
#else
using value_t = double;
#endif

// LDS accumulators are fp64 in BOTH builds: on gfx950 ds_add_f32 is a slow path (measured: the float build's
// numeric pass took 8.9 ms with fp32 LDS atomics against 4.0 ms without the adds, while ds_add_f64 costs next to
// nothing), so float values are widened on load, accumulated in double and narrowed when the row is stored.
using acc_t = double;

constexpr int kEmpty = -1;          // empty hash slot (column indices are >= 0)
constexpr int kMaxBins = 16;

struct BinSpec {                    // bin b >= 2 holds rows with upper[b-1] < v <= upper[b]; bin 0: v == 0;
    int nbins;                      // bin 1 ("quad" bin): 0 < v <= quadMax and at most kQuadMaxA entries in the A row
    int quadMax;                    // 0 disables the quad bin
    int laneMax, laneMaxA;          // lane bin (kLaneBin, k_row_lane): 0 < v <= laneMax and at most laneMaxA entries in the A row
    int hubMin;                     // hub bin (kHubBin, bhs_hub.hip.h): rows with at least hubMin products; 0 disables
    int upper[kMaxBins];            // upper[1] is 0: the size ladder starts at bin 2
};
constexpr int kLaneBin = kMaxBins - 1;   // outside every size ladder (ladders have at most 12 bins)
constexpr int kHubBin = kMaxBins - 2;    // rows split across workgroups, whatever their size-ladder bin would be
constexpr int kQuadMaxA = 16;       // a 16-lane quarter wave holds one A entry per lane

// qv: the quantity the quad bin's 64-slot quarter tables are sized by (products / entries).  It differs from v
// only for symbolic bins keyed by the compressed pair count, where the quad kernel still walks plain products.
// prod: the row's product count (upper bound), the key of the hub bin in both stages.
__device__ __forceinline__ int bin_of(const BinSpec& s, int v, int nA, int qv, int prod)
{
    if (s.hubMin > 0 && v > 0 && prod >= s.hubMin) return kHubBin;
    if (qv > 0 && qv <= s.laneMax && nA <= s.laneMaxA) return kLaneBin;
    if (qv > 0 && qv <= s.quadMax && nA <= kQuadMaxA) return 1;
    int b = 0;
#pragma unroll
    for (int i = 0; i < kMaxBins; ++i) b += (i < s.nbins - 1 && v > s.upper[i]) ? 1 : 0;
    return (b == 1) ? 2 : b;        // 0 < v <= upper[2] that did not qualify for the quad bin
}

// s_waitcnt immediate (gfx9 encoding): vmcnt(0), expcnt and lgkmcnt left at their maxima (no wait)
constexpr int kWaitVm0 = 0x0F70;

__device__ __forceinline__ unsigned hash_col(int col, int log2ts)
{
    return ((unsigned)col * 2654435761u) >> (32 - log2ts);
}

__device__ __forceinline__ int mbcnt64(unsigned long long m)   // number of set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}


__device__ __forceinline__ int wave_sum(int v) { return wave_sum_dpp(v); }

// ---------------------------------------------------------------------------
// Stage 1a: per-row upper bound ub[i] = sum_{j in A(i,:)} len(B(j,:)), with G
// lanes cooperating on one row (G chosen from the average row length of A so
// that colIndA reads are coalesced and lanes are busy).  Also: total product
// count (int64), histogram of symbolic bins, and rowCnt[i] = 0 for empty rows.
// ---------------------------------------------------------------------------
// rows each lane group keeps in flight: short rows (G <= 8 lanes per row: poisson5pt, web graphs) are pure
// latency, 8 rows hide it (poisson5pt 1024^2: 0.109 -> 0.066 ms); longer rows are gather-bound and 4 is best
__host__ __device__ constexpr int ub_rows_in_flight(int G) { return G <= 8 ? 8 : 4; }
constexpr int kUbLongA = 512;       // A rows beyond this go to k_upper_bound_long (when the launch provides the list); less where rows are short: 32 passes of the row's lane group
constexpr int kLongParts = 16;      // workgroups that share one such row (also: one long B row in k_check_sorted_long)
__host__ __device__ constexpr int long_parts(long long len)     // parts of >= 2048 entries
{
    return len >= 2048LL * kLongParts ? kLongParts : (len < 2048 ? 1 : (int)(len / 2048));
}

// CMP (B's pattern has been compressed, k_compress_b): the gather reads (len, pairs) of the B row from cLen
// instead of the two row pointers, and the symbolic bin of a row is chosen by its PAIR count -- the key the
// compressed symbolic kernel sizes its table by -- as long as that stays within the wave-per-row tables
// (keyMax); longer rows keep the product count as their key (their kernels walk the uncompressed pattern).
template <int G, bool CMP>
__global__ __launch_bounds__(256) void k_upper_bound(int m, const int* __restrict__ Ap,
                                                     const int* __restrict__ Aj,
                                                     const int* __restrict__ Bp, int* __restrict__ ub,
                                                     int* __restrict__ cnt,
                                                     unsigned long long* __restrict__ total,
                                                     int* __restrict__ binCount, BinSpec spec,
                                                     const int2* __restrict__ cLen, int* __restrict__ keyOut,
                                                     int keyMax, int2* __restrict__ longList,
                                                     int* __restrict__ longCount, int longThresh)
{
    // Rows of A with more than longThresh (<= kUbLongA) entries would be walked by their G lanes alone while the rest of the device
    // idles (a 180 k-entry row: 7 ms); with a list to put them on (longList != nullptr) they are left to
    // k_upper_bound_long, which spreads every such row over 16 workgroups.
    // The per-row work is a chain of three dependent loads (rowPtrA -> colIndA -> rowPtrB) and little
    // else, so every lane group keeps R rows in flight: all rowPtrA pairs, then all colIndA, then all
    // rowPtrB gathers are issued before the first sum is needed.
    constexpr int R = ub_rows_in_flight(G);
    __shared__ int hist[kMaxBins];
    __shared__ unsigned long long bsum;
    const int tid = threadIdx.x;
    if (tid < kMaxBins) hist[tid] = 0;
    if (tid == 0) bsum = 0;
    __syncthreads();
    constexpr int rows_per_block = 256 / G;
    const int g = tid % G;
    unsigned long long mySum = 0;                       // leaders only
    // Software pipeline over the block's passes: the rowPtrA pairs of pass t+2 and the colIndA entries of pass
    // t+1 are in flight while the rowPtrB gathers of pass t are issued and reduced, so a pass costs one
    // memory round trip instead of three (poisson7pt 128^3: 0.133 -> 0.095 ms).
    const long long stride = (long long)gridDim.x * rows_per_block * R;
    auto load_ap = [&](long long rbase, int (&a0_)[R], int (&a1_)[R]) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const long long r = rbase + k * rows_per_block + tid / G;
            a0_[k] = a1_[k] = 0;
            if (r < m) { a0_[k] = Ap[r]; a1_[k] = Ap[r + 1]; }
        }
    };
    auto deferred = [&](int a0_, int a1_) { return longList != nullptr && a1_ - a0_ > longThresh; };
    auto load_aj = [&](const int (&a0_)[R], const int (&a1_)[R], int (&c_)[R]) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            c_[k] = -1;
            if (a0_[k] + g < a1_[k] && !deferred(a0_[k], a1_[k])) c_[k] = Aj[a0_[k] + g];
        }
    };
    long long rbase = (long long)blockIdx.x * rows_per_block * R;
    int a0[R], a1[R], c[R], a0n[R], a1n[R];
    load_ap(rbase, a0, a1);
    load_ap(rbase + stride, a0n, a1n);
    load_aj(a0, a1, c);
    for (; rbase < m; rbase += stride) {
        int cn[R], a0nn[R], a1nn[R];
        long long s[R];
        int cs[R];                                           // CMP: pairs of the compressed pattern (saturating)
        // ---- this pass: gathers; next pass: colIndA; the one after: rowPtrA
#pragma unroll
        for (int k = 0; k < R; ++k) {
            s[k] = 0;
            cs[k] = 0;
            if (c[k] >= 0) {
                int2 be;                                     // rowPtrB[c], rowPtrB[c+1] in one 8-byte gather
                if constexpr (CMP) { be = cLen[c[k]]; s[k] = be.x; cs[k] = be.y; }
                else { __builtin_memcpy(&be, Bp + c[k], sizeof(be)); s[k] = be.y - be.x; }
            }
        }
        load_aj(a0n, a1n, cn);
        load_ap(rbase + 2 * stride, a0nn, a1nn);
#pragma unroll
        for (int k = 0; k < R; ++k)                          // rows longer than G entries: the rest, plainly
            for (int j = a0[k] + g + G; j < (deferred(a0[k], a1[k]) ? a0[k] : a1[k]); j += G) {
                const int cc = Aj[j];
                int2 be;
                if constexpr (CMP) {
                    be = cLen[cc];
                    s[k] += be.x;
                    cs[k] = min(cs[k] + be.y, 1 << 24);      // only compared with keyMax: saturate, never wrap
                } else {
                    __builtin_memcpy(&be, Bp + cc, sizeof(be));
                    s[k] += be.y - be.x;
                }
            }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            // group sum: one DPP wave scan + one cross-lane read while the partial sums are small (the
            // common case); 64-bit butterfly otherwise.  The LAST lane of a group is its leader.
            long long tot;
            if (!__any(s[k] >= (1LL << 24))) {
                const int incl = wave_incl_scan_dpp((int)s[k]);
                const int before = __shfl(incl, (int)(threadIdx.x & 63) - G, 64);   // end of the previous group
                tot = incl - ((threadIdx.x & 63) >= G ? before : 0);
            } else {
                long long t = s[k];
#pragma unroll
                for (int o = G / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                tot = t;
            }
            int ctot = 0;
            if constexpr (CMP) {                             // <= 64 lanes x 2^24: no overflow
                const int incl = wave_incl_scan_dpp(cs[k]);
                const int before = __shfl(incl, (int)(threadIdx.x & 63) - G, 64);
                ctot = incl - ((threadIdx.x & 63) >= G ? before : 0);
            }
            const long long r = rbase + k * rows_per_block + tid / G;
            if (r < m && g == G - 1) {
                if (deferred(a0[k], a1[k])) {
                    const int np = long_parts(a1[k] - a0[k]);           // one list entry {row, part, parts} per part
                    const int at = atomicAdd(longCount, np);
                    for (int p = 0; p < np; ++p) longList[at + p] = make_int2((int)r, p | (np << 8));
                } else {
                    const int v = tot > 0x7fffffffLL ? 0x7fffffff : (int)tot;
                    ub[r] = v;
                    if (v == 0) cnt[r] = 0;   // ESC_0 (bhsparse_cuda.h:1582-1595): nothing else to do
                    int key = v;
                    if constexpr (CMP) { if (ctot <= keyMax) key = ctot; keyOut[r] = key; }
                    atomicAdd(&hist[bin_of(spec, key, a1[k] - a0[k], v, v)], 1);
                    mySum += (unsigned long long)tot;
                }
            }
        }
        // ---- rotate the pipeline
#pragma unroll
        for (int k = 0; k < R; ++k) { a0[k] = a0n[k]; a1[k] = a1n[k]; c[k] = cn[k]; a0n[k] = a0nn[k]; a1n[k] = a1nn[k]; }
    }
    if (mySum) atomicAdd(&bsum, mySum);
    __syncthreads();
    if (tid < kMaxBins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
    if (tid == 0 && bsum) atomicAdd(total, bsum);
}

// The rows k_upper_bound left on its list, one entry per (row, part): a workgroup sums one part of one row into
// part[]; k_upper_bound_long_finish adds a row's parts up and does what the leader lane does for a short row.
template <bool CMP>
__global__ __launch_bounds__(256) void k_upper_bound_long(const int2* __restrict__ longList,
                                                          const int* __restrict__ longCount,
                                                          const int* __restrict__ Ap, const int* __restrict__ Aj,
                                                          const int* __restrict__ Bp, const int2* __restrict__ cLen,
                                                          long long* __restrict__ part)
{
    __shared__ long long ws[4], wc[4];
    const int tid = threadIdx.x;
    const int items = *longCount;
    for (int v = blockIdx.x; v < items; v += gridDim.x) {
        const int2 it = longList[v];
        const int r = it.x, p = it.y & 255, np = it.y >> 8;
        const long long a0 = Ap[r], len = Ap[r + 1] - a0;
        const long long j0 = a0 + len * p / np, j1 = a0 + len * (p + 1) / np;
        long long s = 0, cs = 0;
        for (long long j = j0 + tid; j < j1; j += 256) {
            const int cc = Aj[j];
            int2 be;
            if constexpr (CMP) { be = cLen[cc]; s += be.x; cs += be.y; }
            else { __builtin_memcpy(&be, Bp + cc, sizeof(be)); s += be.y - be.x; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); cs += __shfl_xor(cs, o, 64); }
        if ((tid & 63) == 0) { ws[tid >> 6] = s; wc[tid >> 6] = cs; }
        __syncthreads();
        if (tid == 0) {
            part[2 * (long long)v] = ws[0] + ws[1] + ws[2] + ws[3];
            part[2 * (long long)v + 1] = wc[0] + wc[1] + wc[2] + wc[3];
        }
        __syncthreads();
    }
}

template <bool CMP>
__global__ __launch_bounds__(256) void k_upper_bound_long_finish(const int2* __restrict__ longList,
                                                                 const int* __restrict__ longCount,
                                                                 const int* __restrict__ Ap,
                                                                 const long long* __restrict__ part,
                                                                 int* __restrict__ ub, int* __restrict__ cnt,
                                                                 unsigned long long* __restrict__ total,
                                                                 int* __restrict__ binCount, BinSpec spec,
                                                                 int* __restrict__ keyOut, int keyMax)
{
    const int n = *longCount;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int2 it = longList[i];
        if (it.y & 255) continue;                              // the entry of part 0 speaks for the row
        const int r = it.x, np = it.y >> 8;
        long long tot = 0, ctot = 0;
        for (int p = 0; p < np; ++p) { tot += part[2 * (long long)(i + p)]; ctot += part[2 * (long long)(i + p) + 1]; }
        const int v = tot > 0x7fffffffLL ? 0x7fffffff : (int)tot;
        ub[r] = v;
        if (v == 0) cnt[r] = 0;
        int key = v;
        if constexpr (CMP) { if (ctot <= keyMax) key = (int)ctot; keyOut[r] = key; }
        atomicAdd(&binCount[bin_of(spec, key, Ap[r + 1] - Ap[r], v, v)], 1);
        atomicAdd(total, (unsigned long long)tot);
    }
}

// ---------------------------------------------------------------------------
// Stage 1b / 3b: scatter 16-byte row descriptors {row, a0, a1, outBase} into
// per-bin queues (replaces the reference's 6-int host-built queue tuples,
// bhsparse.h:365-481).  key[] is ub (symbolic bins) or rowPtrC (numeric bins,
// v = Cp[i+1]-Cp[i], outBase = Cp[i]).  Counting is wave-aggregated: one LDS
// atomic per (wave, distinct bin), one global atomic per (block, bin); rows of
// a block stay together so queue order stays close to row order (L2 locality
// of the B rows they touch).
// ---------------------------------------------------------------------------
constexpr int kFillRounds = BHS_FILL_ROUNDS;     // rows per thread per reservation
constexpr int kFillTile = 256 * kFillRounds;     // rows per block per global reservation

template <bool FROM_ROWPTR>
__global__ __launch_bounds__(256) void k_fill_queues(int m, const int* __restrict__ key,
                                                     const int* __restrict__ Ap, const int* __restrict__ ub,
                                                     const int* __restrict__ binStart,
                                                     int* __restrict__ binCursor, int4* __restrict__ queue,
                                                     BinSpec spec, unsigned long long* __restrict__ binSums)
{
    __shared__ int hist[kMaxBins];
    __shared__ int base[kMaxBins];
    __shared__ unsigned long long sums[kMaxBins * 3];   // per bin: products, nnz(C rows), nnz(A rows)
    const int tid = threadIdx.x;
    if (tid < kMaxBins * 3) sums[tid] = 0;
    for (long long r0 = (long long)blockIdx.x * kFillTile; r0 < m; r0 += (long long)gridDim.x * kFillTile) {
        if (tid < kMaxBins) hist[tid] = 0;
        __syncthreads();
        int bb[kFillRounds], pp[kFillRounds];
        int4 dd[kFillRounds];
#pragma unroll
        for (int r = 0; r < kFillRounds; ++r) {
            const long long row = r0 + (long long)r * 256 + tid;
            int b = 0, pos = 0, a0 = 0, a1 = 0, outBase = 0, v = 0, ubv = 0;
            if (row < m) {
                if (FROM_ROWPTR) { outBase = key[row]; v = key[row + 1] - outBase; } else v = key[row];
                a0 = Ap[row];
                a1 = Ap[row + 1];
                ubv = ub[row];
                b = bin_of(spec, v, a1 - a0, FROM_ROWPTR ? v : ubv, ubv);   // (symbolic keys may be pair counts: quad bin by products)
            }
            // all 64 lanes take part (rows past m carry b == 0 and never match a leader's bin)
            unsigned long long todo = __ballot(b > 0);
            while (todo) {                                   // one pass per distinct bin in this wave
                const int leader = __ffsll((long long)todo) - 1;
                const int lbin = __shfl(b, leader, 64);
                const bool mine = (b == lbin);
                const unsigned long long peers = __ballot(mine);
                // per-bin statistics of this wave's rows (bhs_get_kernel_stats): DPP sums while the addends are
                // small (the usual case), 64-bit butterflies otherwise
                unsigned long long s0, s1 = 0ull, s2;
                const int x0 = mine ? ubv : 0, x1 = (mine && FROM_ROWPTR) ? v : 0, x2 = mine ? a1 - a0 : 0;
                if (BHS_FILL_NOSTATS) { s0 = s2 = 0ull; }
                else if (!__any((x0 | x1 | x2) >= (1 << 24))) {
                    s0 = (unsigned long long)(unsigned)wave_sum_dpp(x0);
                    if (FROM_ROWPTR) s1 = (unsigned long long)(unsigned)wave_sum_dpp(x1);
                    s2 = (unsigned long long)(unsigned)wave_sum_dpp(x2);
                } else {
                    s0 = (unsigned long long)(unsigned)x0;
                    s1 = (unsigned long long)(unsigned)x1;
                    s2 = (unsigned long long)(unsigned)x2;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        s0 += __shfl_xor(s0, o, 64);
                        s1 += __shfl_xor(s1, o, 64);
                        s2 += __shfl_xor(s2, o, 64);
                    }
                }
                int wbase = 0;
                if ((tid & 63) == leader) {
                    wbase = atomicAdd(&hist[lbin], __popcll(peers));
                    atomicAdd(&sums[lbin * 3 + 0], s0);
                    if (FROM_ROWPTR) atomicAdd(&sums[lbin * 3 + 1], s1);
                    atomicAdd(&sums[lbin * 3 + 2], s2);
                }
                wbase = __shfl(wbase, leader, 64);
                if (mine) pos = wbase + mbcnt64(peers);
                todo &= ~peers;
            }
            bb[r] = b;
            pp[r] = pos;
            dd[r] = make_int4((int)row, a0, a1, outBase);
        }
        __syncthreads();
        if (tid < kMaxBins && hist[tid]) base[tid] = binStart[tid] + atomicAdd(&binCursor[tid], hist[tid]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kFillRounds; ++r)
            if (bb[r] > 0) queue[base[bb[r]] + pp[r]] = dd[r];   // bin 0 (empty rows): no queue
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
                                                     const int* __restrict__ Ap,
                                                     long long* __restrict__ blockSum,
                                                     int* __restrict__ binCount, BinSpec spec,
                                                     int* __restrict__ maxCnt, const int* __restrict__ ub)
{
    __shared__ int hist[kMaxBins];
    __shared__ long long wsum[4];
    __shared__ int wmax[4];
    int mx = 0;
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
            mx = max(mx, v);
            const int b = bin_of(spec, v, Ap[idx + 1] - Ap[idx], v, spec.hubMin > 0 ? ub[idx] : 0);
            if (b > 0) atomicAdd(&hist[b], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); mx = max(mx, __shfl_xor(mx, o, 64)); }
    if ((tid & 63) == 0) { wsum[tid >> 6] = s; wmax[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        blockSum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        atomicMax(maxCnt, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));   // longest row of C (numeric-first test)
    }
    if (tid < kMaxBins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
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

// numeric-bin histogram and longest row of a ROW RANGE of C (rowPtrC is final): the scan delivers these for the
// whole matrix, bhs_spgemm_numeric needs them per range
__global__ __launch_bounds__(256) void k_bin_hist(int m, const int* __restrict__ Cp, const int* __restrict__ Ap,
                                                  BinSpec spec, int* __restrict__ binCount, int* __restrict__ maxCnt,
                                                  const int* __restrict__ ub)
{
    __shared__ int hist[kMaxBins];
    __shared__ int wmax[4];
    const int tid = threadIdx.x;
    if (tid < kMaxBins) hist[tid] = 0;
    __syncthreads();
    int mx = 0;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < m; i += (long long)gridDim.x * 256) {
        const int v = Cp[i + 1] - Cp[i];
        mx = max(mx, v);
        const int b = bin_of(spec, v, Ap[i + 1] - Ap[i], v, spec.hubMin > 0 ? ub[i] : 0);
        if (b > 0) atomicAdd(&hist[b], 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = mx;
    __syncthreads();
    if (tid < kMaxBins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
    if (tid == 0) {
        mx = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (mx) atomicMax(maxCnt, mx);
    }
}

// longest row of a CSR matrix (chooses the lanes-per-row of k_upper_bound for skewed inputs).  One same-address
// atomic per BLOCK: with one per wave (8192 of them on a 2 M-row matrix) the kernel took 98 us for 8 MB, all of
// it atomics queueing on one L2 word.
__global__ __launch_bounds__(256) void k_max_row(int m, const int* __restrict__ Ap, int* __restrict__ out)
{
    __shared__ int wmax[4];
    int mx = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256)
        mx = max(mx, Ap[i + 1] - Ap[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (mx) atomicMax(out, mx);
    }
}

// B-row sortedness check (reference precondition for EM_mergepath, bhsparse_cuda.h:1902ff; here the lane kernels,
// the compressed symbolic pass and the column-window path rely on strictly ascending rows).  G = 2^logG lanes per
// row walk it with coalesced loads; neighbours are compared inside a row only, so no search for row boundaries
// (the element-parallel version with a binary search at every row end took 0.93 ms on poisson27pt 128^3).
constexpr int kSortedLongB = 4096;
__global__ __launch_bounds__(256) void k_check_sorted(int k, int logG, const int* __restrict__ Bp,
                                                      const int* __restrict__ Bj, int* __restrict__ flag,
                                                      int2* __restrict__ longList, int* __restrict__ longCount)
{
    const int G = 1 << logG, g = threadIdx.x & (G - 1), rpb = 256 >> logG;
    int bad = 0;
    for (long long r = (long long)blockIdx.x * rpb + (threadIdx.x >> logG); r < k; r += (long long)gridDim.x * rpb) {
        const int a0 = Bp[r], a1 = Bp[r + 1];
        if (longList != nullptr && a1 - a0 > kSortedLongB) {
            if (g == 0) {                                                // left to k_check_sorted_long, in parts
                const int np = long_parts(a1 - a0);
                const int at = atomicAdd(longCount, np);
                for (int p = 0; p < np; ++p) longList[at + p] = make_int2((int)r, p | (np << 8));
            }
            continue;
        }
        for (int e = a0 + g; e + 1 < a1; e += G) bad |= Bj[e] >= Bj[e + 1] ? 1 : 0;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// the rows k_check_sorted left on its list: one workgroup per (row, part)
__global__ __launch_bounds__(256) void k_check_sorted_long(const int2* __restrict__ longList,
                                                           const int* __restrict__ longCount,
                                                           const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                           int* __restrict__ flag)
{
    int bad = 0;
    const int items = *longCount;
    for (int v = blockIdx.x; v < items; v += gridDim.x) {
        const int2 it = longList[v];
        const int r = it.x, p = it.y & 255, np = it.y >> 8;
        const long long a0 = Bp[r], len = Bp[r + 1] - a0;
        const long long e0 = a0 + len * p / np, e1 = a0 + len * (p + 1) / np;   // pairs (e, e+1), e in [e0, e1)
        for (long long e = e0 + threadIdx.x; e < e1 && e + 1 < a0 + len; e += 256) bad |= Bj[e] >= Bj[e + 1] ? 1 : 0;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------
// Workgroup-per-row accumulator for long rows (the reference's EM_mergepath /
// EM_mergepath_global territory, bhsparse_cuda.h:1902-2525, and its progressive
// re-allocation rounds :2527-2780).  One workgroup of BLOCK lanes per row:
//   * A entries are taken BLOCK at a time, one per lane; a block-wide scan of
//     the B row lengths gives a flat product index space, and every lane finds
//     the A entry of its product by binary search in the LDS prefix array
//     (U products per lane in flight);
//   * LDS open-addressing table of TS slots, first probe = ds_cmpst_rtn;
//     new keys are counted per wave (ballot) so the fill level is known after
//     every batch;
//   * COLUMN WINDOWS: a row whose accumulator does not fit the table is produced
//     in successive column ranges [lo,hi).  With column-sorted B rows each lane
//     restricts its B row to the range by two binary searches; an overflowing
//     range is halved and retried, a sparse one doubles the next.  Ranges come
//     out in ascending column order, so the concatenation is the sorted row;
//   * numeric: every lane packs TS/BLOCK slots as (column << 32 | slot) and the
//     workgroup sorts them in REGISTERS: a DPP bitonic sort per wave, then
//     flip-merges across waves that exchange through the (no longer needed)
//     key array -- a dozen barriers instead of one per network stage, and no
//     second copy of the table, which lets an 8192-slot fp64 table fit the
//     160 KiB LDS.  Values never move: they are read by slot when C is written.
// ---------------------------------------------------------------------------
// sorts the occupied slots of a workgroup's table by column and streams (column, value) to C; defined below
template <int TS, int BLOCK>
__device__ __forceinline__ void block_sort_and_store(int* keys, const acc_t* vals, int uniq, int tid,
                                                     int* __restrict__ Cj, value_t* __restrict__ Cx, long long outBase);

template <int TS, int BLOCK, bool NUM>
struct BlockSmem {
    int keys[TS];
    acc_t vals[NUM ? TS : 1];
    value_t sAv[NUM ? BLOCK : 1];
    int sIncl[BLOCK];
    int sBase[BLOCK];
    int wtot[BLOCK / 64];
    int counter[4];      // [0] unique keys in the table, [1] overflow flag
};

template <int TS, int LOG2TS, int BLOCK, bool NUM>
__global__ __launch_bounds__(BLOCK) void k_row_block(
    const int4* __restrict__ desc, int qn, int ncolsB, int bSorted,
    const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    const int* __restrict__ ubArr,          // symbolic: per-row upper bound (first window guess)
    int* __restrict__ CpOrCnt, int* __restrict__ Cj, value_t* __restrict__ Cx,
    int* __restrict__ errFlag, int* __restrict__ ticket, const int* __restrict__ qnPtr)
{
    static_assert((1 << LOG2TS) == TS, "table size must be 2^LOG2TS");
    if (qnPtr) qn = *qnPtr;               // queue filled on the device (overflow rows of k_sym_blocks): length read here
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Smem = BlockSmem<TS, BLOCK, NUM>;
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
    constexpr int CAP = TS - TS / 4;      // max unique keys admitted per table fill
    constexpr int U = 4;                  // products per lane per batch
    constexpr int NW = BLOCK / 64;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // rows differ in cost by orders of magnitude: workgroups pull the next row from a device ticket
    for (;;) {
        if (tid == 0) sm.counter[2] = atomicAdd(ticket, 1);
        __syncthreads();
        const int q = sm.counter[2];
        __syncthreads();
        if (q >= qn) break;
        const int4 d = desc[q];
        const int row = d.x, a0 = d.y, a1 = d.z;
        long long outBase = NUM ? CpOrCnt[row] : 0;   // (numeric: CpOrCnt is rowPtrC; equals d.w where the queue carries it)
        int rowTotal = 0;                 // symbolic: unique count over all windows
        const long long need = NUM ? (long long)(CpOrCnt[row + 1] - CpOrCnt[row]) : (long long)ubArr[row];
        long long lo = 0, width = ncolsB;
        if (need > CAP) {                 // first guess: split the column range uniformly by the expected load
            const long long nwin = (need + CAP / 2 - 1) / (CAP / 2);
            width = ncolsB / nwin;
            if (width < 1) width = 1;
        }
        while (lo < ncolsB) {
            const long long hi = lo + width < ncolsB ? lo + width : (long long)ncolsB;
            const bool full = (lo == 0 && hi >= ncolsB);
            // ---- clear
            for (int s = tid; s < TS; s += BLOCK) {
                sm.keys[s] = kEmpty;
                if (NUM) sm.vals[s] = 0.0;
            }
            if (tid < 2) sm.counter[tid] = 0;
            __syncthreads();
            // The whole row in one window whose table cannot overflow (need = exact nnz / upper bound <= CAP): the fill
            // level needs no watching, so the per-batch wave reduction, LDS atomic and block barrier go away and the
            // new keys are added up once at the end of the row (most rows of the workgroup bins are of this kind).
            const bool fits = full && need <= CAP;
            int accNew = 0;

            for (int ca = a0; ca < a1; ca += BLOCK) {
                if (sm.counter[1]) break;                       // uniform: read after a barrier
                // ---- one A entry per lane, restricted to the column window
                const int e = ca + tid;
                int b0 = 0, len = 0;
                value_t av = 0.0;
                if (e < a1) {
                    const int c = Aj[e];
                    if (NUM) av = Ax[e];
                    int2 be;
                    __builtin_memcpy(&be, Bp + c, sizeof(be));
                    b0 = be.x;
                    int b1 = be.y;
                    if (!full && bSorted) {
                        int l = b0, r = b1;                      // lower_bound(lo)
                        while (l < r) { const int mid = (l + r) >> 1; if (Bj[mid] < (int)lo) l = mid + 1; else r = mid; }
                        b0 = l;
                        r = b1;                                  // lower_bound(hi)
                        while (l < r) { const int mid = (l + r) >> 1; if ((long long)Bj[mid] < hi) l = mid + 1; else r = mid; }
                        b1 = l;
                    }
                    len = b1 - b0;
                }
                // ---- block-wide inclusive scan of len
                int incl = wave_incl_scan_dpp(len);
                if (lane == 63) sm.wtot[wv] = incl;
                __syncthreads();
                int woff = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int t = sm.wtot[w];
                    if (w < wv) woff += t;
                    total += t;
                }
                incl += woff;
                sm.sIncl[tid] = incl;
                sm.sBase[tid] = b0 - (incl - len);
                if (NUM) sm.sAv[tid] = av;
                __syncthreads();

                for (int p0 = 0; p0 < total; p0 += BLOCK * U) {
                    int col[U];
                    value_t bxu[U], avu[U];                          // multiplied at insert time: no wait behind each load
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int p = p0 + u * BLOCK + tid;
                        col[u] = kEmpty;
                        if (p < total) {
                            int l = 0, r = BLOCK - 1;            // first entry j with sIncl[j] > p
                            while (l < r) { const int mid = (l + r) >> 1; if (sm.sIncl[mid] > p) r = mid; else l = mid + 1; }
                            const long long idx = (long long)sm.sBase[l] + p;
                            const int c = Bj[idx];
                            if (full || bSorted || ((long long)c >= lo && (long long)c < hi)) {
                                col[u] = c;
                                if (NUM) { avu[u] = sm.sAv[l]; bxu[u] = Bx[idx]; }
                            }
                        }
                    }
                    int myNew = 0;
                    bool ovf = false;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int cv = col[u];
                        if (cv != kEmpty) {
                            unsigned h = hash_col(cv, LOG2TS);
                            int probes = 0;
                            for (;;) {
                                const int c2 = atomicCAS(&sm.keys[h], kEmpty, cv);
                                if (c2 == kEmpty) { ++myNew; break; }
                                if (c2 == cv) break;
                                h = (h + 1) & (TS - 1);
                                if (++probes >= TS) { ovf = true; break; }
                            }
                            if (NUM && !ovf) unsafeAtomicAdd(&sm.vals[h], (acc_t)avu[u] * (acc_t)bxu[u]);
                        }
                    }
                    if (fits) {
                        accNew += myNew;
                        if (ovf) sm.counter[1] = 1;                 // (cannot happen while need <= CAP holds; ends in S_ERR)
                        continue;
                    }
                    // ---- fill level after this batch (one LDS atomic per wave)
                    const int wNew = wave_sum_dpp(myNew);
                    const unsigned long long anyOvf = __ballot(ovf);
                    if (lane == 0) {
                        if (wNew) { const int before = atomicAdd(&sm.counter[0], wNew); if (before + wNew > CAP) sm.counter[1] = 1; }
                        if (anyOvf) sm.counter[1] = 1;
                    }
                    __syncthreads();
                    if (sm.counter[1]) break;                       // uniform
                }
                __syncthreads();                                    // sIncl/sBase are rewritten by the next chunk
            }
            if (fits) {
                const int wNew = wave_sum_dpp(accNew);
                if (lane == 0 && wNew) atomicAdd(&sm.counter[0], wNew);
            }
            __syncthreads();
            const int uniq = sm.counter[0];
            const int ovfl = sm.counter[1];
            __syncthreads();
            if (ovfl) {                                             // halve the window and retry the same lo
                if (width <= 1) { if (tid == 0) atomicOr(errFlag, 1); lo = hi; }
                else width = (width + 1) >> 1;
                continue;
            }
            if constexpr (!NUM) {
                rowTotal += uniq;
            } else if (uniq > 0) {
                // ---- sort by column in registers (no second copy of the table) and stream the row out
                block_sort_and_store<TS, BLOCK>(sm.keys, sm.vals, uniq, tid, Cj, Cx, outBase);
                outBase += uniq;
                __syncthreads();
            }
            lo = hi;
            if (uniq < CAP / 4 && width < ncolsB) width <<= 1;      // sparse window: grow the next one
        }
        if (!NUM && tid == 0) CpOrCnt[row] = rowTotal;
    }
}

// ---------------------------------------------------------------------------
// Bitmap accumulator for rows whose result does not fit the LDS table of
// k_row_block (hub rows of power-law matrices: webbase-1M has C rows with
// ~100 k entries).  Each resident workgroup owns one slot: an n-bit occupancy
// bitmap plus (numeric) one rank word per 32 columns.
//   pass 1  every product sets its column's bit (global_atomic_or; the slot is
//           private to the workgroup and lives in this XCD's L2);
//   scan    the bitmap yields the row's columns in ascending order -- no column
//           windows, no sort -- and the prefix popcounts (rank) map a column to
//           its position in the row; Cj is written and Cx zeroed here;
//   pass 2  (numeric) every product is added straight into its final place,
//           Cx[rowBase + rank[c/32] + popc(bits[c/32] below c)], with
//           global_atomic_add_f64: the accumulation target is the row of C
//           itself (compact, cache resident), not an n-entry dense vector whose
//           random 8-byte updates would each move a whole line to and from HBM.
// Replaces, for those rows, the reference's EM_mergepath_global rounds
// (bhsparse_cuda.h:2270-2525) and their progressive re-allocation (:2527-2780).
// ---------------------------------------------------------------------------
template <int BLOCK, bool NUM>
__global__ __launch_bounds__(BLOCK) void k_row_spa(
    const int4* __restrict__ desc, int qn, int ncolsB,
    const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    int* __restrict__ cntOut, int* __restrict__ Cj, value_t* __restrict__ Cx,
    int* __restrict__ ticket, int* __restrict__ rankBase, unsigned* __restrict__ bitsBase)
{
    __shared__ value_t sAv[NUM ? BLOCK : 1];
    __shared__ int sIncl[BLOCK];
    __shared__ int sBase[BLOCK];
    __shared__ int wtot[BLOCK / 64];
    __shared__ int bcast;
    constexpr int U = BHS_SPA_U, NW = BLOCK / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nWords = (int)((((long long)ncolsB + 31) >> 5) + 3) & ~3;   // slot stride: whole 16-byte groups
    int* rank = NUM ? rankBase + (size_t)blockIdx.x * (size_t)nWords : nullptr;
    unsigned* bits = bitsBase + (size_t)blockIdx.x * (size_t)nWords;

    // flat product space per chunk of BLOCK A entries; f(column, product index in B, A entry slot)
    auto expand = [&](int a0, int a1, auto&& f) {
        for (int ca = a0; ca < a1; ca += BLOCK) {
            const int e = ca + tid;
            int b0 = 0, len = 0;
            value_t av = 0.0;
            if (e < a1) {
                const int c = Aj[e];
                if (NUM) av = Ax[e];
                int2 be;
                __builtin_memcpy(&be, Bp + c, sizeof(be));
                b0 = be.x;
                len = be.y - be.x;
            }
            int incl = wave_incl_scan_dpp(len);
            if (lane == 63) wtot[wv] = incl;
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int t = wtot[w];
                if (w < wv) woff += t;
                total += t;
            }
            incl += woff;
            sIncl[tid] = incl;
            sBase[tid] = b0 - (incl - len);
            if (NUM) sAv[tid] = av;
            __syncthreads();
            for (int p0 = 0; p0 < total; p0 += BLOCK * U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * BLOCK + tid;
                    if (p < total) {
                        int l = 0, r = BLOCK - 1;                // first entry j with sIncl[j] > p
                        while (l < r) { const int mid = (l + r) >> 1; if (sIncl[mid] > p) r = mid; else l = mid + 1; }
                        const long long idx = (long long)sBase[l] + p;
                        f(Bj[idx], idx, l);
                    }
                }
            }
            __syncthreads();
        }
    };

    for (;;) {
        if (tid == 0) bcast = atomicAdd(ticket, 1);
        __syncthreads();
        const int q = bcast;
        __syncthreads();
        if (q >= qn) break;
        const int4 d = desc[q];
        const int row = d.x, a0 = d.y, a1 = d.z;
#if BHS_PHASES_SPA
        unsigned long long tSpa = __builtin_readcyclecounter();
#endif
        // ---- pass 1: occupancy bits
        expand(a0, a1, [&](int c, long long, int) {
            atomicOr(&bits[c >> 5], 1u << (c & 31));
        });
        // The slot is private to this workgroup and every access to it is served by this XCD's L2
        // (device-scope atomics, sc1 loads, write-through stores), so a workgroup barrier (which drains
        // each wave's vmcnt) orders them; an agent-scope fence would write back the whole L2 (buffer_wbl2).
        __syncthreads();
        BHS_TICK_SPA(8);
        // ---- scan the bitmap: thread t owns the words [t*per, (t+1)*per), per a multiple of 4.  The bits were
        // set by atomics in L2, so stale L1 lines are dropped first (acquire = buffer_inv, no write-back);
        // then plain 16-byte loads.  Up to kWC words per thread stay in registers for all three sweeps
        // (count, expand, clear): one memory round trip instead of a dozen on this latency-bound path.
        constexpr int kWC = 32;
        const int per = (((nWords + BLOCK - 1) / BLOCK) + 3) & ~3;
        const int wBeg = tid * per < nWords ? tid * per : nWords;
        const int wEnd = wBeg + per < nWords ? wBeg + per : nWords;
        const bool cached = per <= kWC;                       // workgroup-uniform
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned wc[kWC];
        int mine = 0;
        if (cached) {
#pragma unroll
            for (int t = 0; t < kWC; t += 4) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (wBeg + t < wEnd) v = *reinterpret_cast<const uint4*>(&bits[wBeg + t]);
                wc[t] = v.x; wc[t + 1] = v.y; wc[t + 2] = v.z; wc[t + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < kWC; ++t) mine += __popc(wc[t]);
        } else {
            for (int w = wBeg; w < wEnd; w += 16) {
                uint4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    v[t] = (w + 4 * t < wEnd) ? *reinterpret_cast<const uint4*>(&bits[w + 4 * t]) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int t = 0; t < 4; ++t) mine += __popc(v[t].x) + __popc(v[t].y) + __popc(v[t].z) + __popc(v[t].w);
            }
        }
        int inc2 = wave_incl_scan_dpp(mine);
        if (lane == 63) wtot[wv] = inc2;
        __syncthreads();
        int off = 0, rowCount = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int t = wtot[w];
            if (w < wv) off += t;
            rowCount += t;
        }
        off += inc2 - mine;                                   // entries of this row before this thread's words
        BHS_TICK_SPA(9);
        const long long base = d.w;
        if (NUM) {
            // each thread expands its words: rank per occupied word, columns in order, values zeroed
            int run = off;
            auto emit = [&](int w, unsigned mm) {
                if (mm) rank[w] = run;                          // only occupied words are ever looked up
                while (mm) {
                    const int b = __ffs((int)mm) - 1;
                    mm &= mm - 1;
                    Cj[base + run] = (w << 5) + b;
                    Cx[base + run] = (value_t)0;
                    ++run;
                }
            };
            if (cached) {
#pragma unroll
                for (int t = 0; t < kWC; ++t) emit(wBeg + t, wc[t]);
            } else {
                for (int w = wBeg; w < wEnd; w += 4) {
                    const uint4 v = *reinterpret_cast<const uint4*>(&bits[w]);
                    emit(w, v.x); emit(w + 1, v.y); emit(w + 2, v.z); emit(w + 3, v.w);
                }
            }
            __syncthreads();
            BHS_TICK_SPA(10);
            // ---- pass 2: every product lands in its final place
            expand(a0, a1, [&](int c, long long idx, int l) {
                const int w = c >> 5;
                const unsigned word = __hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int pos = __hip_atomic_load(&rank[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                                __popc(word & ((1u << (c & 31)) - 1u));
                unsafeAtomicAdd(&Cx[base + pos], (value_t)((acc_t)sAv[l] * (acc_t)Bx[idx]));
            });
            __syncthreads();
            BHS_TICK_SPA(11);
        } else if (tid == 0) {
            cntOut[row] = rowCount;
        }
        // ---- leave the slot clean
        if (cached) {
#pragma unroll
            for (int t = 0; t < kWC; ++t)
                if (wc[t]) bits[wBeg + t] = 0u;
        } else {
            for (int w = wBeg; w < wEnd; w += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(&bits[w]);
                if (v.x | v.y | v.z | v.w) *reinterpret_cast<uint4*>(&bits[w]) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        __syncthreads();
        BHS_TICK_SPA(12);
    }
}

// ---------------------------------------------------------------------------
// The same bitmap accumulator with the bitmap in LDS, for matrices with up to
// kLdsBitmapCols (2^20) columns: 128 KB of occupancy bits + 16 KB of rank words
// (one per 256 columns) fill the CU's 160 KB, so one 1024-lane workgroup per CU.
// Bit sets, the ordered sweep, the rank lookups of pass 2 and the final clear
// are all LDS traffic; HBM/L2 see only the B rows (twice), the row of C and the
// fp64 adds into it.  The sweep gives each lane one bitmap word per step, so a
// wave's stores of Cj/Cx land on one contiguous run of the row.
// ---------------------------------------------------------------------------
constexpr int kLdsBitmapCols = 1 << 20;
constexpr int kLdsBitmapBlock = 1024, kLdsBitmapChunk = 512;
constexpr int kLdsBitmapEntryMajor = 256;      // average B row of a chunk from which the products are taken entry by entry

template <bool NUM>
constexpr size_t lds_bitmap_smem(int nWords)
{
    // bitmap + (numeric) rank per 8 words + duplicate flags per 16 columns + A-chunk arrays + wave totals
    return (size_t)nWords * 4 + (NUM ? (size_t)(nWords / 8) * 4 + (size_t)(nWords / 16) * 4 : 0) +
           (size_t)kLdsBitmapChunk * 2 * sizeof(int) + 32 * sizeof(int);
}

template <bool NUM>
__global__ __launch_bounds__(kLdsBitmapBlock) void k_row_bitmap_lds(
    const int4* __restrict__ desc, int qn, int nWords,       // nWords: bitmap words, a multiple of 1024
    const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    int* __restrict__ cntOut, int* __restrict__ Cj, value_t* __restrict__ Cx, int* __restrict__ ticket,
    int reverse)                                               // 1: queue taken from its end (longest rows there)
{
    constexpr int BLOCK = kLdsBitmapBlock, CH = kLdsBitmapChunk, U = 4, NW = BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    unsigned* bm = reinterpret_cast<unsigned*>(smemRaw);
    int* rank8 = reinterpret_cast<int*>(bm + nWords);
    // dup: one flag per 16 columns, set when a column of the group is hit twice.  Only those entries need the
    // zero + atomic-add treatment; everything else (98.7 % of the products of a web graph) is a plain store.
    unsigned* dup = reinterpret_cast<unsigned*>(rank8 + (NUM ? nWords / 8 : 0));
    int* sIncl = reinterpret_cast<int*>(dup + (NUM ? nWords / 16 : 0));
    int* sBase = sIncl + CH;
    int* wtot = sBase + CH;                                   // [NW] + broadcast word
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nClear = (nWords + (NUM ? nWords / 8 + nWords / 16 : 0)) / 4;   // bitmap .. dup are contiguous

    for (int i = tid; i < nClear; i += BLOCK) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // flat product space per chunk of CH A entries; f(column, product index in B, A entry index)
    auto expand = [&](int a0, int a1, auto&& f) {
        for (int ca = a0; ca < a1; ca += CH) {
            const int e = ca + tid;
            int b0 = 0, len = 0;
            if (tid < CH && e < a1) {
                const int c = Aj[e];
                int2 be;
                __builtin_memcpy(&be, Bp + c, sizeof(be));
                b0 = be.x;
                len = be.y - be.x;
            }
            int incl = wave_incl_scan_dpp(len);
            if (lane == 63) wtot[wv] = incl;
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < CH / 64; ++w) {
                const int t = wtot[w];
                if (w < wv) woff += t;
                total += t;
            }
            incl += woff;
            if (tid < CH) {
                sIncl[tid] = incl;
                sBase[tid] = b0 - (incl - len);
            }
            __syncthreads();
            const int nE = min(CH, a1 - ca);
            if ((long long)total >= (long long)nE * kLdsBitmapEntryMajor) {
                // long B rows behind this chunk (a portal row of a web graph: a handful of directory pages): entry by
                // entry, the whole workgroup along one B row -- coalesced loads, no search for the product's entry
                for (int l = 0; l < nE; ++l) {
                    const int end = sIncl[l], beg = l ? sIncl[l - 1] : 0;
                    const long long bb = sBase[l];
                    for (int p = beg + tid; p < end; p += BLOCK * 2) {
                        const long long i0 = bb + p, i1 = i0 + BLOCK;
                        const bool two = p + BLOCK < end;
                        const int c0 = Bj[i0], c1 = Bj[two ? i1 : i0];
                        f(c0, i0, ca + l);
                        if (two) f(c1, i1, ca + l);
                    }
                }
            } else {
            for (int p0 = 0; p0 < total; p0 += BLOCK * U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * BLOCK + tid;
                    if (p < total) {
                        int l = 0, r = CH - 1;                   // first entry j with sIncl[j] > p
                        while (l < r) { const int mid = (l + r) >> 1; if (sIncl[mid] > p) r = mid; else l = mid + 1; }
                        const long long idx = (long long)sBase[l] + p;
                        f(Bj[idx], idx, ca + l);
                    }
                }
            }
            }
            __syncthreads();
        }
    };

    const int steps = nWords / BLOCK;                         // bitmap words per lane; wave wv owns words [wv*steps*64, ..)
    for (;;) {
#if BHS_PHASES_SPA
        unsigned long long tSpa = __builtin_readcyclecounter();
#endif
        if (tid == 0) wtot[NW] = atomicAdd(ticket, 1);
        __syncthreads();
        const int q = wtot[NW];
        __syncthreads();
        if (q >= qn) break;
        const int4 d = desc[reverse ? qn - 1 - q : q];
        const int row = d.x, a0 = d.y, a1 = d.z;
        BHS_TICK_SPA(8);
        // ---- pass 1: occupancy bits
        expand(a0, a1, [&](int c, long long, int) {
            const unsigned bit = 1u << (c & 31);
            const unsigned old = atomicOr(&bm[c >> 5], bit);
            if (NUM && (old & bit)) atomicOr(&dup[c >> 9], 1u << ((c >> 4) & 31));
        });
        BHS_TICK_SPA(9);
        // ---- entries before each wave's words
        const int w0 = wv * steps * 64;
        int mine = 0;
        for (int i = 0; i < steps; ++i) mine += __popc(bm[w0 + i * 64 + lane]);
        const int waveCount = wave_sum_dpp(mine);
        if (lane == 0) wtot[wv] = waveCount;
        __syncthreads();
        int run = 0, rowCount = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int t = wtot[w];
            if (w < wv) run += t;
            rowCount += t;
        }
        BHS_TICK_SPA(10);
        if (!NUM) {
            if (tid == 0) cntOut[row] = rowCount;
        } else {
            // ---- ordered sweep: one word per lane per step; rank of every 8-word group, columns, zeroed values
            const long long base = d.w;
            for (int i = 0; i < steps; ++i) {
                const int w = w0 + i * 64 + lane;
                unsigned mm = bm[w];
                const unsigned dd = (dup[w >> 4] >> ((w & 15) * 2)) & 3u;   // flags of this word's two 16-column halves
                const int cnt = __popc(mm);
                const int incl = wave_incl_scan_dpp(cnt);
                int r = run + incl - cnt;
                if ((lane & 7) == 0) rank8[w >> 3] = r;
                while (mm) {
                    const int b = __ffs((int)mm) - 1;
                    mm &= mm - 1;
                    Cj[base + r] = (w << 5) + b;
                    if ((dd >> (b >> 4)) & 1u) Cx[base + r] = (value_t)0;
                    ++r;
                }
                run += __builtin_amdgcn_readlane(incl, 63);
            }
            // the zeroed values must be in L2 before any wave adds to them: the barrier drains every wave's stores
            __syncthreads();
            BHS_TICK_SPA(11);
            // ---- pass 2: every product is added straight into its place in the row of C
            expand(a0, a1, [&](int c, long long idx, int e) {
                const int w = c >> 5;
                const uint4 lo = *reinterpret_cast<const uint4*>(&bm[w & ~7]);
                const uint4 hi = *reinterpret_cast<const uint4*>(&bm[(w & ~7) + 4]);
                const unsigned g[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const int k = w & 7;
                int pos = rank8[w >> 3];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned mask = t < k ? 0xffffffffu : (t == k ? (1u << (c & 31)) - 1u : 0u);
                    pos += __popc(g[t] & mask);
                }
                const value_t v = (value_t)((acc_t)Ax[e] * (acc_t)Bx[idx]);   // product formed in acc_t, narrowed once
                if ((dup[c >> 9] >> ((c >> 4) & 31)) & 1u) unsafeAtomicAdd(&Cx[base + pos], v);
                else Cx[base + pos] = v;                       // the only product of this column
            });
        }
        __syncthreads();
        BHS_TICK_SPA(12);
        // ---- leave the bitmap clean
        for (int i = tid; i < nClear; i += BLOCK) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        BHS_TICK_SPA(13);
    }
}

// ===========================================================================
// Wavefront-per-row accumulator (the workhorse; one 64-lane workgroup per row
// in flight, persistent over an XCD-aware slice of the row queue).
//
// What matters on CDNA4:
//  * no dependent load chain per A entry: the whole A row (<= 64 entries per
//    pass) is fetched by one coalesced load, the B row extents by one gather,
//    and a wave scan turns the B row lengths into a flat product index space;
//  * flat product mapping: lane l of batch u owns product p = w0 + 64u + l; its
//    A entry is found with ONE v_mbcnt on a 64-bit mask of "last product of an
//    entry" marks kept in LDS (ds_or_b32 by the entry lanes), so all 64 lanes
//    are busy whatever the B row lengths are (27-entry rows: 11.4 passes
//    instead of 14);
//  * U = 4 batches of colIndB/valB loads are issued back to back before the
//    first LDS insert (256 independent loads in flight per wave);
//  * numeric: occupied slots are compacted to packed (col<<32 | slot) words and
//    sorted in REGISTERS by a wave-wide bitonic network (cross-lane exchange by
//    DPP/ds_bpermute, no LDS round trip per stage), then streamed out;
//  * XCD-aware persistent schedule: workgroup b runs on XCD b%8 (observed
//    dispatch order), so each XCD walks one contiguous eighth of the queue and
//    neighbouring rows share B rows through that XCD's private L2.
// ===========================================================================
// numeric loads: valB and the A value of a batch stay in registers and are multiplied when the batch is
// inserted (1; 2 keeps the A entry index instead of its value; 0 = multiply behind the load, which makes every
// valB load wait for its data before the next batch's loads are issued: measured 3.92 -> 3.80 ms on p27 128^3)
// ask the register allocator for >= 5 waves per SIMD (<= 96 VGPRs).  With the deferred multiply the window
// holds 6 x (col, valB, av) in registers; 6 waves (80 VGPRs) spill, measured 3.80 vs 3.49 ms.
// first probe = one ds_cmpst_rtn (claims an empty slot or returns the resident key) instead of
// ds_read + conditional ds_cmpst: measured -19 % symbolic / -8 % numeric on poisson27pt
constexpr int kWavesPerBlock = BHS_WPB;   // independent row-waves per workgroup (co-located on one CU)

// Orders LDS traffic between the lanes of ONE wave: the LDS pipe executes a wave's DS
// instructions in order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// product batches per window (MAXB x 64 products with their loads in flight): deeper for the symbolic pass
// (one register per product), shallower for the numeric pass (three) so that it keeps 8 waves per SIMD
constexpr int kMaxBSym = BHS_MAXB_SYM, kMaxBNum = BHS_MAXB_NUM, kMaxBNumLong = BHS_MAXB_LONG;
constexpr int kMaxB = kMaxBSym > kMaxBNumLong ? kMaxBSym : kMaxBNumLong;   // sizes the LDS mark words
// Every window of a row costs one exposed memory round trip (~3 us on a loaded chip).  Rows of the 256-slot numeric
// tables live on occupancy: 5 batches and 6 waves per SIMD (80 VGPRs, no spills) since round 3 -- poisson27pt 128^3
// numeric_wave<256> 3.43 -> 3.36 ms, 72^3 1.04 -> 0.98 (6 batches at 5 waves: the round-2 setting; 6 at 6: 3.78; 4 at 6:
// 3.33; anything at 7 or 8 waves spills and takes 4.8 - 6 ms: the kernel does not fit 64 VGPRs); rows of the larger tables have
// thousands of products -- a 3-dof FEM row: 6561, i.e. 18 windows of 6 -- and do better with 12 batches in flight
// and 4 waves per SIMD (numeric_wave<512> on that matrix: 4.88 -> 3.14 ms; poisson27pt would lose 4 %).
constexpr int wave_window_batches(int TS, bool NUM) { return !NUM ? kMaxBSym : (TS >= 512 ? kMaxBNumLong : kMaxBNum); }
constexpr int wave_min_waves(int TS, bool NUM) { return !NUM ? BHS_SYM_WAVES : (TS < 512 ? BHS_NUM_WAVES : (TS >= 1024 ? 3 : BHS_LONG_WAVES)); }   // 1024 slots: 3 waves, no spills (3.59 -> 3.48 ms on the 4-dof case)

// PACK32: sort keys are (col << LOG2TS | slot) in 32 bits (legal when every column < 2^(32-LOG2TS));
// otherwise (col << 32 | slot) in 64 bits.
template <int TS, bool NUM, bool PACK32>
struct WaveSmem {
    using packed_t = typename std::conditional<PACK32, unsigned, unsigned long long>::type;
    int keys[TS];
    acc_t vals[NUM ? TS : 1];
    packed_t packed[NUM ? TS : 2];
    value_t sAv[NUM ? 64 : 1];
    int sBase[64];
    alignas(8) unsigned marks[2 * kMaxB];   // read as 64-bit words
    unsigned magic[BHS_UNIFORM ? 64 : 1];   // ceil(2^32 / L), L = 1..64: product index -> A entry when all B rows have L entries
};

template <typename T>
__device__ __forceinline__ T lane_xor_any(T x, int lj, int lane)
{
    if constexpr (sizeof(T) == 8) {
        switch (lj) {
            case 1: return lane_xor64<1>(x, lane);
            case 2: return lane_xor64<2>(x, lane);
            case 4: return lane_xor64<4>(x, lane);
            case 8: return lane_xor64<8>(x, lane);
            case 16: return lane_xor64<16>(x, lane);
            default: return lane_xor64<32>(x, lane);
        }
    } else {
        switch (lj) {
            case 1: return lane_xor<1>(x, lane);
            case 2: return lane_xor<2>(x, lane);
            case 4: return lane_xor<4>(x, lane);
            case 8: return lane_xor<8>(x, lane);
            case 16: return lane_xor<16>(x, lane);
            default: return lane_xor<32>(x, lane);
        }
    }
}

// wave-wide bitonic sort of 64*E keys (u32 or u64), ascending; element index
// i = lane*E + e, so each lane ends with E consecutive sorted keys.  Cross-lane
// exchanges are DPP / permlane-swap moves (bhs_wave.hip.h): no LDS round trips.
template <typename T, int E, int GW = 64>
__device__ __forceinline__ void wave_bitonic_sort(T (&x)[E], int lane)
{
    // GW = lanes per independent sort (64: whole wave; 16: four quarter-wave sorts side by side, DPP only)
    if constexpr (sizeof(T) == 4) {        // 32-bit keys: the cheaper ascending-only network (bhs_wave.hip.h)
        wave_flip_sort_u32<E, GW>(x, lane);
        return;
    }
    lane &= GW - 1;
#pragma unroll
    for (int k = 2; k <= GW * E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                const int lj = j / E;
                const bool lower = (lane & lj) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool up = (((lane * E + e) & k) == 0);
                    const T y = lane_xor_any<T>(x[e], lj, lane);
                    const T lo = x[e] < y ? x[e] : y;
                    const T hi = x[e] < y ? y : x[e];
                    x[e] = (lower == up) ? lo : hi;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) == 0) {
                        const bool up = (((lane * E + e) & k) == 0);
                        const T a = x[e], b = x[e | j];
                        const bool sw = (a > b) == up;
                        x[e] = sw ? b : a;
                        x[e | j] = sw ? a : b;
                    }
                }
            }
        }
    }
}

// ascending merge of a wave's 64*E keys that form a bitonic sequence (element index i = lane*E + e)
template <typename T, int E>
__device__ __forceinline__ void wave_merge_asc(T (&x)[E], int lane)
{
#pragma unroll
    for (int j = 32 * E; j > 0; j >>= 1) {
        if (j >= E) {
            const int lj = j / E;
            const bool lower = (lane & lj) == 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T y = lane_xor_any<T>(x[e], lj, lane);
                const T lo = x[e] < y ? x[e] : y;
                const T hi = x[e] < y ? y : x[e];
                x[e] = lower ? lo : hi;
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((e & j) == 0) {
                    const T a = x[e], b = x[e | j];
                    const bool sw = a > b;
                    x[e] = sw ? b : a;
                    x[e | j] = sw ? a : b;
                }
            }
        }
    }
}

template <int TS, int BLOCK>
__device__ __forceinline__ void block_sort_and_store(int* keys, const acc_t* vals, int uniq, int tid,
                                                     int* __restrict__ Cj, value_t* __restrict__ Cx, long long outBase)
{
    constexpr int E = TS / BLOCK;                 // slots per lane
    constexpr int SEG = 64 * E;                   // keys per wave
    using T = unsigned long long;
    const int lane = tid & 63;
    const int i0 = tid * E;                       // element index of x[0]: wave w owns [w*SEG, (w+1)*SEG)
    T x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = keys[i0 + e];
        x[e] = k == kEmpty ? ~0ull : (((T)(unsigned)k << 32) | (unsigned)(i0 + e));   // empty slots sort last
    }
    wave_bitonic_sort<T, E>(x, lane);             // every wave: its SEG keys ascending
    __syncthreads();                              // all lanes have read their keys: the array is free
    unsigned* xch = reinterpret_cast<unsigned*>(keys);
    // partner exchange across waves: high words, then low words, through the key array
    auto exchange = [&](int mask) {
        T y[E];
#pragma unroll
        for (int e = 0; e < E; ++e) xch[i0 + e] = (unsigned)(x[e] >> 32);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) y[e] = (T)xch[(i0 + e) ^ mask] << 32;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) xch[i0 + e] = (unsigned)x[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) y[e] |= (T)xch[(i0 + e) ^ mask];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool lower = (i0 + e) < ((i0 + e) ^ mask);
            const T lo = x[e] < y[e] ? x[e] : y[e];
            const T hi = x[e] < y[e] ? y[e] : x[e];
            x[e] = lower ? lo : hi;
        }
    };
#pragma unroll
    for (int kk = 2 * SEG; kk <= TS; kk <<= 1) {
        exchange(kk - 1);                         // flip: two ascending runs -> two bitonic halves
#pragma unroll
        for (int j = kk >> 2; j >= SEG; j >>= 1) exchange(j);
        wave_merge_asc<T, E>(x, lane);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int r = i0 + e;
        if (r < uniq) {
            Cj[outBase + r] = (int)(x[e] >> 32);
            Cx[outBase + r] = (value_t)vals[(unsigned)x[e]];
        }
    }
}

template <int LOG2TS, bool PACK32, int E, typename T>
__device__ __forceinline__ void wave_sort_and_store(const T* packed, const acc_t* vals, int uniq, int lane,
                                                    int* __restrict__ Cj, value_t* __restrict__ Cx,
                                                    long long outBase)
{
    T x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        x[e] = i < uniq ? packed[i] : (T)~(T)0;
    }
    wave_bitonic_sort<T, E>(x, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int r = lane * E + e;
        if (r < uniq) {
            int col;
            unsigned slot;
            if constexpr (PACK32) { col = (int)(x[e] >> LOG2TS); slot = x[e] & ((1u << LOG2TS) - 1); }
            else { col = (int)(x[e] >> 32); slot = (unsigned)x[e]; }
#if BHS_NT_STORES
            __builtin_nontemporal_store(col, &Cj[outBase + r]);
            __builtin_nontemporal_store((value_t)vals[slot], &Cx[outBase + r]);
#else
            Cj[outBase + r] = col;
            Cx[outBase + r] = (value_t)vals[slot];
#endif
        }
    }
}

// SMALLB: nnz(B) < 2^29, byte offsets into colIndB / valB fit 32 bits
template <int TS, int LOG2TS, bool NUM, bool PACK32, bool SMALLB>
__global__ __launch_bounds__(64 * kWavesPerBlock, wave_min_waves(TS, NUM)) void k_row_wave(
    const int4* __restrict__ desc, int qn, int chunkLog2,
    const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    int* __restrict__ cntOut, int* __restrict__ Cj, value_t* __restrict__ Cx,
    const int* __restrict__ Ap, int* __restrict__ ubOut, unsigned long long* __restrict__ ctSlots,
    int* __restrict__ errFlag)
{
    // desc == nullptr ("wave-first" symbolic pass: maxRow(A) x maxRow(B) fits this table for EVERY row, so no
    // upper-bound pass ran and no queue exists): queue entry q is row q, its descriptor comes from rowPtrA, and the
    // row's product count goes to ubOut[row] and into one of 64 spread counters (ctSlots), which is all the
    // upper-bound pass would have delivered.
    using Smem = WaveSmem<TS, NUM, PACK32>;
    using packed_t = typename Smem::packed_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = kWavesPerBlock;
    // the wave index is wave-uniform by construction: saying so (readfirstlane) moves the whole queue-index arithmetic
    // of the row pipeline to the scalar unit
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Smem& sm = reinterpret_cast<Smem*>(smem_raw)[wave];
    constexpr int MAXB = wave_window_batches(TS, NUM);
    constexpr int GRP = (MAXB % 4 == 0) ? 4 : (MAXB % 3 == 0 ? 3 : 2);                   // probes in flight per insert group

    // XCD-aware persistent schedule (gridDim.x is a multiple of 8; block b runs on XCD b % 8, tools/xcc_probe.hip).
    // The queue is cut into chunks of 2^chunkLog2 consecutive entries and chunk k belongs to XCD k % 8: inside a
    // chunk neighbouring rows share B rows through that XCD's private L2, while all eight XCDs stay within the
    // same few thousand rows of the matrix, so the B rows reused across grid planes form ONE working set in the
    // 256 MB Infinity Cache instead of eight.  The host picks 2048-entry chunks for long queues and smaller ones
    // for short queues, so that every XCD still gets an equal share of a bin with only a few thousand rows.
    const int chunk = 1 << chunkLog2;
    const int xcd = blockIdx.x & 7, lb = (blockIdx.x >> 3) * WPB + wave, perX = (gridDim.x >> 3) * WPB;
    const int nChunks = (qn + chunk - 1) >> chunkLog2;
    int positions = 0;                                   // queue entries that belong to this XCD
    if (nChunks > xcd) {
        positions = ((nChunks - xcd + 7) >> 3) << chunkLog2;
        if (((nChunks - 1) & 7) == xcd) positions -= (nChunks << chunkLog2) - qn;
    }
    const int nIt = lb < positions ? (positions - lb + perX - 1) / perX : 0;
    auto q_of = [&](int it) {                             // it-th entry of this wave (it < nIt)
        const int t = lb + it * perX;
        return ((((t >> chunkLog2) << 3) + xcd) << chunkLog2) + (t & (chunk - 1));
    };
    // Descriptor of this wave's it-th row, (-1,0,0,0) past the end.  Always a load from the queue (a clamped
    // index, then a select of the VALUES): "cond ? desc[q] : constant" becomes a select of two ADDRESSES, one of
    // them a stack copy of the constant, and the load a FLAT load -- which counts on lgkmcnt as well as vmcnt, so
    // the next wait for any LDS read would also wait for this prefetch to come back from memory.  The laundered
    // zero keeps the address in VGPRs: a global (vmcnt-only) load, not a scalar one (lgkmcnt again).
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
    unsigned long long prodSum = 0;                       // wave-first: products of this wave's rows

    // ---- software pipeline over rows: descriptor (i+3) -> A entries (i+2) -> B extents (i+1) -> work (i)
    if (nIt == 0) return;                                // (wave-uniform; there is no barrier in this kernel)
    if (BHS_UNIFORM && NUM && TS <= 256) sm.magic[lane] = 0xffffffffu / (unsigned)(lane + 1) + 1u;
    int4 dC = load_desc(0);
    int4 d1 = load_desc(1);
    int4 d2 = load_desc(2);
    int cC = 0, c1 = 0;
    value_t avC = 0.0, av1 = 0.0;
    if (lane < dC.z - dC.y) { cC = Aj[dC.y + lane]; if (NUM) avC = Ax[dC.y + lane]; }
    if (lane < d1.z - d1.y) { c1 = Aj[d1.y + lane]; if (NUM) av1 = Ax[d1.y + lane]; }
    // B row extents travel through the pipeline as the raw (begin, end) pair: forming the length where the
    // gather is issued puts an s_waitcnt vmcnt(0) right behind it, i.e. one exposed round trip per row
    int2 beC = make_int2(0, 0);
    if (lane < dC.z - dC.y) __builtin_memcpy(&beC, Bp + cC, 8);

#if BHS_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tPrev = __builtin_readcyclecounter();
#endif
    // The prologue's loads are complete before the loop is entered.  Without this the compiler's wait-count
    // analysis merges "pending since the prologue" into the loop header and guards the first use of every
    // rotated register with vmcnt(0/1) -- which, the counter being in-order, waits for the prefetches the
    // iteration has just issued.
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    for (int it = 0; it < nIt; ++it) {
        // ---- prefetch for the rows behind this one
        const int4 d3 = load_desc(it + 3);
        int c2 = 0;
        value_t av2 = 0.0;
        if (lane < d2.z - d2.y) { c2 = Aj[d2.y + lane]; if (NUM) av2 = Ax[d2.y + lane]; }
        int2 be1 = make_int2(0, 0);
        if (lane < d1.z - d1.y) __builtin_memcpy(&be1, Bp + c1, 8);   // one 8-byte gather, consumed by the next row

        // (making the descriptor fields scalars as well -- readfirstlane -- was measured SLOWER: the reads need a
        // wait the compiler can only place conservatively, on top of the fresh prefetches)
        const int row = dC.x, a0 = dC.y, a1 = dC.z;
        // ---- clear the table
#pragma unroll
        for (int k = 0; k < (TS + 255) / 256; ++k) {
            const int s = k * 256 + lane * 4;
            if (TS >= 256 || s < TS) {
                *reinterpret_cast<int4*>(&sm.keys[s]) = make_int4(kEmpty, kEmpty, kEmpty, kEmpty);
                if (NUM) {
                    *reinterpret_cast<double2*>(&sm.vals[s]) = make_double2(0.0, 0.0);
                    *reinterpret_cast<double2*>(&sm.vals[s + 2]) = make_double2(0.0, 0.0);
                }
            }
        }
        int myNew = 0;
        int rowProducts = 0;
        // Rows with more than 64 A entries (power-law matrices: hundreds of short B rows per row) walk them in
        // chunks of 64.  In the larger-table instantiations, where such rows live, the chunks are pipelined
        // like the rows are: the B extents of chunk i+1 and the A entries of chunk i+2 are in flight while
        // chunk i is accumulated.
        constexpr bool kChunkPipe = NUM ? (TS >= 512) : (TS >= 2048);
        const bool multi = kChunkPipe && (a1 - a0 > 64);
        int cA = 0, cB = 0, b0N = 0, lenN = 0;
        value_t avA = 0.0, avB = 0.0, avN = 0.0;
        auto load_a = [&](int ea, int& c_, value_t& av_) {
            c_ = 0; av_ = 0.0;
            if (ea < a1) { c_ = Aj[ea]; if (NUM) av_ = Ax[ea]; }
        };
        auto gather_b = [&](int ea, int c_) {                 // extents of the chunk whose entries start at ea - lane
            b0N = 0; lenN = 0;
            if (ea < a1) { int2 be; __builtin_memcpy(&be, Bp + c_, 8); b0N = be.x; lenN = be.y - be.x; }
        };
        if (multi) load_a(a0 + 64 + lane, cA, avA);
        for (int ca = a0; ca < a1; ca += 64) {
            // ---- one A entry per lane: B row extent, flat product offsets
            int b0 = beC.x, len = beC.y - beC.x;
            value_t av = avC;
            if (multi) {
                if (ca == a0) {
                    load_a(ca + 128 + lane, cB, avB);
                } else {
                    b0 = b0N; len = lenN; av = avN;
                    gather_b(ca + 64 + lane, cA);
                    avN = avA;
                    load_a(ca + 128 + lane, cA, avA);
                }
            } else if (ca != a0) {                            // small-table instantiations: later chunks, unpipelined
                const int ea = ca + lane;
                b0 = 0; len = 0; av = 0.0;
                if (ea < a1) {
                    const int c = Aj[ea];
                    if (NUM) av = Ax[ea];
                    int2 be;
                    __builtin_memcpy(&be, Bp + c, 8);
                    b0 = be.x;
                    len = be.y - be.x;
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);         // nothing of this (rare) path stays pending at the join
            }
            const int incl = wave_incl_scan_dpp(len);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            rowProducts += total;
            const int last = incl - 1;                      // flat index of this entry's last product
            const unsigned long long nz = __ballot(len > 0);
            const int jc = mbcnt64(nz);                      // compacted index among non-empty entries
            wave_sync();                                 // previous chunk's readers are done
            if (len > 0) {
                sm.sBase[jc] = b0 - (incl - len);
                if (NUM) sm.sAv[jc] = av;
            }
            int done = 0;                                    // entries completed before the window
            // Uniform chunk (BHS_UNIFORM): every B row it touches has the same number L of entries (stencil
            // interiors, block matrices).  Product p then belongs to entry p / L -- one v_mul_hi with
            // ceil(2^32 / L), exact for p * L < 2^32 -- and the mark words, their two LDS round trips per window
            // and the mbcnt / popcount per batch are not needed.  Numeric pass only: same-box A/B on poisson27pt 128^3,
            // three runs each: numeric 3.33 -> 3.26 ms, symbolic 1.385 -> 1.40 ms (the branch costs it more than the
            // marks did).
            const int L0 = __builtin_amdgcn_readfirstlane(len);
            const int nAc = a1 - ca < 64 ? a1 - ca : 64;
            constexpr bool kUni = BHS_UNIFORM && NUM && TS <= 256;   // (compiled out of the large-table kernels: its branches cost the 12-batch windows 8 %)
            const bool uni = kUni && L0 >= 2 && L0 <= 64 && __ballot(lane < nAc && len != L0) == 0ull;
            unsigned magic = 0;
            if (uni) { wave_sync(); magic = sm.magic[BHS_UNIFORM ? L0 - 1 : 0]; }
            BHS_TICK(0);
            for (int w0 = 0; w0 < total; w0 += 64 * MAXB) {
                const int nb = (total - w0 + 63) >> 6;       // batches in this window (wave-uniform)
                if (!uni) {
                    if (lane < 2 * MAXB) sm.marks[lane] = 0;
                    wave_sync();
                    const int rel = last - w0;
                    if (len > 0 && rel >= 0 && rel < 64 * MAXB) atomicOr(&sm.marks[rel >> 5], 1u << (rel & 31));
                }
                wave_sync();
                int col[MAXB];
                acc_t pv[MAXB];
#if BHS_DEFER_MUL == 1
                value_t bxv[MAXB], avv[MAXB];
#elif BHS_DEFER_MUL == 2
                value_t bxv[MAXB];
                int jjv[MAXB];
#endif
                int cum = done;
                // ---- all loads of the window first.  The product av * valB is formed only when the batch is
                // inserted: multiplying here would put an s_waitcnt on every valB load right behind its issue
                // and serialise the window's loads.
#pragma unroll
                for (int u = 0; u < MAXB; ++u) {
                    col[u] = kEmpty;                          // (valB / A value registers are only read where col is valid)
#if !BHS_DEFER_MUL
                    pv[u] = 0.0;
#endif
                    if (u < nb) {
                        const int p = w0 + u * 64 + lane;
                        int j;
                        if (uni) j = (int)__umulhi((unsigned)p, magic);
                        else {
                            const unsigned long long mk = *reinterpret_cast<const unsigned long long*>(&sm.marks[2 * u]);
                            j = cum + mbcnt64(mk);
                            cum += __popcll(mk);
                        }
                        if (p < total) {
                            if constexpr (SMALLB && BHS_DEFER_MUL == 1) {
                                // nnz(B) < 2^29: byte offsets fit 32 bits, so the loads use SGPR base + 32-bit VGPR
                                // offset addressing and the 64-bit address arithmetic per product disappears
                                const unsigned idx32 = (unsigned)(sm.sBase[j] + p);
                                col[u] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(Bj) + (idx32 << 2));
#if BHS_DEFER_MUL == 1
                                if (NUM) {
                                    avv[u] = sm.sAv[j];
                                    bxv[u] = *reinterpret_cast<const value_t*>(reinterpret_cast<const char*>(Bx) +
                                                                               idx32 * (unsigned)sizeof(value_t));
                                }
#endif
                                continue;
                            }
                            const long long idx = (long long)sm.sBase[j] + p;
                            col[u] = Bj[idx];
                            if (NUM) {
#if BHS_DEFER_MUL == 1
                                avv[u] = sm.sAv[j];
                                bxv[u] = Bx[idx];
#elif BHS_DEFER_MUL == 2
                                jjv[u] = j;
                                bxv[u] = Bx[idx];
#else
                                pv[u] = (acc_t)sm.sAv[j] * (acc_t)Bx[idx];
#endif
                            }
                        }
                    }
                }
                done = cum;
                BHS_TICK(1);
#if BHS_PHASES
                if (NUM) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                BHS_TICK(2);
#endif
                // ---- inserts, GRP batches at a time: first probes of a group are read back to back
#pragma unroll
                for (int g = 0; g < MAXB; g += GRP) {
                    if (g < nb) {
                        unsigned hh[GRP];
                        int cur[GRP];
#pragma unroll
                        for (int v = 0; v < GRP; ++v) {
                            if (g + v >= MAXB) continue;             // (a last group of fewer batches: folded at compile time)
                            hh[v] = hash_col(col[g + v], LOG2TS);
                            cur[v] = kEmpty;
#if BHS_CAS_ONLY
                            if (col[g + v] != kEmpty) cur[v] = atomicCAS(&sm.keys[hh[v]], kEmpty, col[g + v]);
#else
                            if (col[g + v] != kEmpty) cur[v] = __atomic_load_n(&sm.keys[hh[v]], __ATOMIC_RELAXED);
#endif
                        }
#pragma unroll
                        for (int v = 0; v < GRP; ++v) {
                            if (g + v >= MAXB) continue;
                            const int cv = col[g + v];
                            if (cv != kEmpty) {
                                bool ok = cur[v] == cv;
#if BHS_CAS_ONLY
                                if (cur[v] == kEmpty) { ++myNew; ok = true; }     // this lane's CAS claimed the slot
#else
                                if (cur[v] == kEmpty) {
                                    const int old = atomicCAS(&sm.keys[hh[v]], kEmpty, cv);
                                    if (old == kEmpty) { ++myNew; ok = true; }
                                    else if (old == cv) ok = true;
                                }
#endif
                                if (!ok) {                           // collision: linear probing
                                    // bounded: the host's binning keeps every table under 75 % full, but borrowed
                                    // arrays may change under us -- a full table must end in S_ERR, not in a hang
                                    unsigned h = hh[v];
                                    int left = TS;
                                    for (;;) {
                                        h = (h + 1) & (TS - 1);
                                        const int c2 = atomicCAS(&sm.keys[h], kEmpty, cv);
                                        if (c2 == kEmpty) { ++myNew; break; }
                                        if (c2 == cv) break;
                                        if (--left == 0) { atomicOr(errFlag, 1); break; }
                                    }
                                    hh[v] = h;
                                }
#if BHS_DEFER_MUL == 1
                                if (NUM) pv[g + v] = (acc_t)avv[g + v] * (acc_t)bxv[g + v];
#elif BHS_DEFER_MUL == 2
                                if (NUM) pv[g + v] = (acc_t)sm.sAv[jjv[g + v]] * (acc_t)bxv[g + v];
#endif
                                if (NUM) unsafeAtomicAdd(&sm.vals[hh[v]], pv[g + v]);
                            }
                        }
                    }
                }
                // Every load of the window has been consumed by now, but under conditions the compiler cannot
                // match up with the ones they were issued under (u < nb vs g < nb): left alone it guards the
                // loop header with vmcnt(0) against writes into "possibly pending" registers, and on the first
                // window that wait lands on the row prefetches issued a moment ago.  Free at run time.
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
            }
            if (multi && ca == a0) {                          // first chunk done: its successor's extents (entries loaded at row start)
                gather_b(ca + 64 + lane, cA);
                avN = avA;
                cA = cB; avA = avB;
            }
        }
        wave_sync();
        BHS_TICK(3);
        // ---- rotate the pipeline HERE, not behind the stores of C: the moves need the prefetched registers, and
        // a wait placed after the stores would be a vmcnt(0) that also waits for the stores to be acknowledged.
        // At this point every load older than the last window's is back, so the moves cost nothing.
        const int outW = dC.w;
        dC = d1; d1 = d2; d2 = d3;
        avC = av1; av1 = av2;
        c1 = c2;
        beC = be1;
        // (pinned: otherwise the select inside load_desc and the moves sink to the loop latch, behind the stores)
        asm volatile("" : "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(c1), "+v"(beC.x), "+v"(beC.y));
        if (NUM) asm volatile("" : "+v"(av1));
        if (!NUM) {
            myNew = wave_sum_dpp(myNew);
            if (lane == 0) cntOut[row] = myNew;
            if (ubOut) {
                // wave-first: the host launched this table size on the strength of the row bounds it saw at
                // bhs_set_data time; the multiply itself checks them -- a row that could have overfilled the table
                // raises bit 1 of the error word and the host repeats the multiply through the general pipeline
                if (lane == 0) { ubOut[row] = rowProducts; if (rowProducts > TS - TS / 4) atomicOr(errFlag, 2); }
                prodSum += (unsigned long long)rowProducts;
            }
        } else {
            const long long outBase = outW;
            // ---- compact occupied slots -> packed sort keys
            int run = 0;
#pragma unroll
            for (int s0 = 0; s0 < TS; s0 += 64) {
                const int s = s0 + lane;
                const int key = sm.keys[s];
                const bool valid = key != kEmpty;
                const unsigned long long bal = __ballot(valid);
                if (valid) {
                    packed_t pk;
                    if constexpr (PACK32) pk = ((unsigned)key << LOG2TS) | (unsigned)s;
                    else pk = ((unsigned long long)(unsigned)key << 32) | (unsigned)s;
                    sm.packed[run + mbcnt64(bal)] = pk;
                }
                run += __popcll(bal);
            }
            const int uniq = run;
            wave_sync();
            BHS_TICK(4);
            if (uniq <= 64)
                wave_sort_and_store<LOG2TS, PACK32, 1>(sm.packed, sm.vals, uniq, lane, Cj, Cx, outBase);
            else if (TS >= 128 && uniq <= 128)
                wave_sort_and_store<LOG2TS, PACK32, 2>(sm.packed, sm.vals, uniq, lane, Cj, Cx, outBase);
            else if (TS >= 256 && uniq <= 256)
                wave_sort_and_store<LOG2TS, PACK32, 4>(sm.packed, sm.vals, uniq, lane, Cj, Cx, outBase);
            else if (TS >= 512 && uniq <= 512)
                wave_sort_and_store<LOG2TS, PACK32, 8>(sm.packed, sm.vals, uniq, lane, Cj, Cx, outBase);
            else if (TS >= 1024 && uniq <= 1024)
                wave_sort_and_store<LOG2TS, PACK32, 16>(sm.packed, sm.vals, uniq, lane, Cj, Cx, outBase);
            else if (TS >= 2048) {
                // tables beyond 1024 slots (only reachable with forced options): bitonic network in LDS
                int P = 2048;
                while (P < uniq) P <<= 1;
                for (int s = uniq + lane; s < P; s += 64) sm.packed[s] = (packed_t)~(packed_t)0;
                wave_sync();
                for (int kk = 2; kk <= P; kk <<= 1) {
                    for (int j = kk >> 1; j > 0; j >>= 1) {
                        for (int i = lane; i < (P >> 1); i += 64) {
                            const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                            const int b = a | j;
                            const bool up = (a & kk) == 0;
                            const packed_t x = sm.packed[a], y = sm.packed[b];
                            if ((x > y) == up) { sm.packed[a] = y; sm.packed[b] = x; }
                        }
                        wave_sync();
                    }
                }
                for (int r = lane; r < uniq; r += 64) {
                    const packed_t e = sm.packed[r];
                    Cj[outBase + r] = PACK32 ? (int)(e >> LOG2TS) : (int)((unsigned long long)e >> 32);
                    Cx[outBase + r] = (value_t)sm.vals[PACK32 ? (unsigned)(e & ((1u << LOG2TS) - 1)) : (unsigned)e];
                }
            }
        }
        wave_sync();
        BHS_TICK(5);
    }
    if (!NUM && ubOut && lane == 0 && prodSum) atomicAdd(&ctSlots[blockIdx.x & 63], prodSum);
#if BHS_PHASES
    if (NUM && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_phase_cycles[i], ph[i]);
        atomicAdd(&g_phase_cycles[7], (unsigned long long)nIt);
    }
#endif
}

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
                    Cj[outBase + l16] = c;
                    Cx[outBase + l16] = (value_t)sm.vals[g][slot];
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
                        Cj[outBase + r] = c;
                        Cx[outBase + r] = (value_t)sm.vals[g][slot];
                    }
                }
            }
        }
        wave_sync();
    }
}

// ===========================================================================
// Compressed pattern of B for the symbolic pass.  The symbolic pass only needs
// the SET of columns of every C row, so B's pattern is first rewritten as
// (column >> 5, 32-bit occupancy mask) pairs: a run of adjacent columns (stencils,
// FEM blocks, bands) collapses into one pair -- a poisson27pt row of 27 entries
// becomes 9.6 pairs -- and the hash table sees 2.8x fewer inserts, OR-ing masks
// instead of counting keys; a row's nnz is the popcount of its table's masks.
// (Same idea as the compression step of KokkosKernels' KKMEM symbolic phase; the
// reference has no counterpart: it over-allocates by the upper bound instead,
// bhsparse.h:365-481.)  Matrices whose rows have no adjacent columns gain
// nothing; the host checks the pair count and falls back to the plain pass.
//
// k_compress_b<G>: G <= 16 lanes per row of B.  cLen[j] = (entries, pairs) of row j;
// ext[j] = (first pair, one past the last pair) in `pair`, which reuses rowPtrB's offsets (a row never has more pairs
// than entries), so no scan is needed.  Requires strictly ascending columns inside
// a row (the host only enables it for sorted B); a block that straddles a chunk of
// G entries simply appears twice, which the OR-accumulation absorbs.
// ===========================================================================
template <int G>
__global__ __launch_bounds__(256) void k_compress_b(int k, const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                    int2* __restrict__ ext, int2* __restrict__ cLen,
                                                    int2* __restrict__ pair,
                                                    unsigned long long* __restrict__ totalPairs)
{
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "a lane group lives inside one 16-lane DPP row");
    constexpr int R = 4;                                      // rows in flight per lane group: the loads are a dependent chain
    constexpr int rows_per_block = 256 / G;
    const int tid = threadIdx.x, lane = tid & 63, g = tid & (G - 1);
    unsigned long long mine = 0;
    // one chunk of <= G entries of a row: run heads, segmented OR of the bits towards the head (DPP row shifts,
    // no LDS), compaction of the heads by ballot
    auto chunk = [&](int c, int s, int& cnt) {
        const bool v = c >= 0;
        const int blk = c >> 5;                               // idle lanes: -1, never equal to a real block
        unsigned acc = v ? 1u << (c & 31) : 0u;
        const int prev = (int)dpp_u32<0x111, 0xf, 0xf, false>((unsigned)-2, (unsigned)blk);     // lane - 1
        const bool head = v && (g == 0 || blk != prev);
#define BHS_SEG_OR(D)                                                                                           \
        if (D < G) {                                                                                            \
            const unsigned o = dpp_u32<0x100 + D, 0xf, 0xf, true>(0u, acc);                    /* lane + D */   \
            const int ob = (int)dpp_u32<0x100 + D, 0xf, 0xf, false>((unsigned)-2, (unsigned)blk);               \
            if (g + D < G && ob == blk) acc |= o;                                                               \
        }
        BHS_SEG_OR(1) BHS_SEG_OR(2) BHS_SEG_OR(4) BHS_SEG_OR(8)
#undef BHS_SEG_OR
        const unsigned long long hb = __ballot(head);
        const unsigned gm = (unsigned)(hb >> (lane - g)) & ((1u << G) - 1u);
        if (head) pair[(long long)s + cnt + __popc(gm & ((1u << g) - 1u))] = make_int2(blk, (int)acc);
        cnt += __popc(gm);
    };
    const long long stride = (long long)gridDim.x * rows_per_block * R;
    for (long long rb = (long long)blockIdx.x * rows_per_block * R; rb < k; rb += stride) {
        int s[R], e[R], c[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long long r = rb + q * rows_per_block + tid / G;
            s[q] = e[q] = 0;
            if (r < k) { s[q] = Bp[r]; e[q] = Bp[r + 1]; }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) { c[q] = -1; if (s[q] + g < e[q]) c[q] = Bj[s[q] + g]; }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            int cnt = 0;
            chunk(c[q], s[q], cnt);
            for (int base = s[q] + G; __any(base < e[q]); base += G) {       // rows longer than G entries
                int cc = -1;
                if (base + g < e[q]) cc = Bj[base + g];
                chunk(cc, s[q], cnt);
            }
            const long long r = rb + q * rows_per_block + tid / G;
            if (r < k && g == 0) {
                ext[r] = make_int2(s[q], s[q] + cnt);
                cLen[r] = make_int2(e[q] - s[q], cnt);           // (entries, pairs): what k_upper_bound<.., CMP> gathers
                mine += (unsigned long long)cnt;
            }
        }
    }
    // one same-address global atomic per block (they serialise in L2: one per wave cost 0.3 ms on 8192 blocks)
    __shared__ unsigned long long bsum;
    if (tid == 0) bsum = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(&bsum, mine);
    __syncthreads();
    if (tid == 0 && bsum) atomicAdd(totalPairs, bsum);
}

// ---------------------------------------------------------------------------
// Symbolic wave-per-row kernel on the compressed pattern: the structure of
// k_row_wave<.., NUM = false> (XCD-aware persistent schedule, row-pipelined
// metadata, flat product mapping) with (block, mask) pairs as the products:
// CAS on the block key, ds_or on the slot's mask, nnz = sum of popcounts.
// ---------------------------------------------------------------------------
constexpr int kMaxBCsym = 6;
template <int TS>
struct CsymSmem {
    int keys[TS];
    unsigned masks[TS];
    int sBase[64];
    alignas(8) unsigned marks[2 * kMaxBCsym];
};

template <int TS, int LOG2TS>
__global__ __launch_bounds__(64 * kWavesPerBlock) BHS_WAVE_ATTR void k_row_wave_csym(
    const int4* __restrict__ desc, int qn, int chunkLog2, const int* __restrict__ Aj,
    const int2* __restrict__ cExt, const int2* __restrict__ cPair, int* __restrict__ cntOut,
    int* __restrict__ errFlag)
{
    using Smem = CsymSmem<TS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = kWavesPerBlock;
    constexpr int MAXB = kMaxBCsym, GRP = 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Smem& sm = reinterpret_cast<Smem*>(smem_raw)[wave];
    const unsigned long long* __restrict__ cPair64 = reinterpret_cast<const unsigned long long*>(cPair);

    // schedule and row pipeline: see k_row_wave
    const int chunk = 1 << chunkLog2;
    const int xcd = blockIdx.x & 7, lb = (blockIdx.x >> 3) * WPB + wave, perX = (gridDim.x >> 3) * WPB;
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
        int4 r = desc[q_of(has ? it_ : 0) + vzero];
        if (!has) r = make_int4(-1, 0, 0, 0);
        return r;
    };
    if (nIt == 0) return;
    int4 dC = load_desc(0);
    int4 d1 = load_desc(1);
    int4 d2 = load_desc(2);
    int cC = 0, c1 = 0;
    if (lane < dC.z - dC.y) cC = Aj[dC.y + lane];
    if (lane < d1.z - d1.y) c1 = Aj[d1.y + lane];
    int2 beC = make_int2(0, 0);
    if (lane < dC.z - dC.y) beC = cExt[cC];
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    for (int it = 0; it < nIt; ++it) {
        const int4 d3 = load_desc(it + 3);
        int c2 = 0;
        if (lane < d2.z - d2.y) c2 = Aj[d2.y + lane];
        int2 be1 = make_int2(0, 0);
        if (lane < d1.z - d1.y) be1 = cExt[c1];
        const int row = dC.x, a0 = dC.y, a1 = dC.z;
#pragma unroll
        for (int k = 0; k < (TS + 255) / 256; ++k) {
            const int s = k * 256 + lane * 4;
            if (TS >= 256 || s < TS) {
                *reinterpret_cast<int4*>(&sm.keys[s]) = make_int4(kEmpty, kEmpty, kEmpty, kEmpty);
                *reinterpret_cast<int4*>(&sm.masks[s]) = make_int4(0, 0, 0, 0);
            }
        }
        for (int ca = a0; ca < a1; ca += 64) {
            int b0 = beC.x, len = beC.y - beC.x;
            if (ca != a0) {                                   // rows with more than 64 A entries: later chunks, unpipelined
                const int ea = ca + lane;
                b0 = 0; len = 0;
                if (ea < a1) {
                    const int2 be = cExt[Aj[ea]];
                    b0 = be.x;
                    len = be.y - be.x;
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
            }
            const int incl = wave_incl_scan_dpp(len);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int last = incl - 1;
            const unsigned long long nz = __ballot(len > 0);
            const int jc = mbcnt64(nz);
            wave_sync();
            if (len > 0) sm.sBase[jc] = b0 - (incl - len);
            int done = 0;
            for (int w0 = 0; w0 < total; w0 += 64 * MAXB) {
                const int nb = (total - w0 + 63) >> 6;
                if (lane < 2 * MAXB) sm.marks[lane] = 0;
                wave_sync();
                const int rel = last - w0;
                if (len > 0 && rel >= 0 && rel < 64 * MAXB) atomicOr(&sm.marks[rel >> 5], 1u << (rel & 31));
                wave_sync();
                // a pair stays ONE 64-bit register tuple until it is inserted: splitting it where it is loaded
                // puts a v_mov -- and with it an s_waitcnt vmcnt(0) -- right behind every load
                unsigned long long pr[MAXB];
                int cum = done;
#pragma unroll
                for (int u = 0; u < MAXB; ++u) {
                    pr[u] = 0x00000000ffffffffull;                // (kEmpty, no bits)
                    if (u < nb) {
                        const unsigned long long mk = *reinterpret_cast<const unsigned long long*>(&sm.marks[2 * u]);
                        const int p = w0 + u * 64 + lane;
                        const int j = cum + mbcnt64(mk);
                        cum += __popcll(mk);
                        if (p < total) pr[u] = cPair64[(long long)sm.sBase[j] + p];
                    }
                }
                done = cum;
#pragma unroll
                for (int g = 0; g < MAXB; g += GRP) {
                    if (g < nb) {
                        unsigned hh[GRP];
                        int cur[GRP];
#pragma unroll
                        for (int v = 0; v < GRP; ++v) {
                            const int bk = (int)(unsigned)pr[g + v];
                            hh[v] = hash_col(bk, LOG2TS);
                            cur[v] = kEmpty;
                            if (bk != kEmpty) cur[v] = atomicCAS(&sm.keys[hh[v]], kEmpty, bk);
                        }
#pragma unroll
                        for (int v = 0; v < GRP; ++v) {
                            const int cv = (int)(unsigned)pr[g + v];
                            if (cv != kEmpty) {
                                if (cur[v] != kEmpty && cur[v] != cv) {        // collision: linear probing
                                    unsigned h = hh[v];
                                    int left = TS;             // bounded probing (see k_row_wave)
                                    for (;;) {
                                        h = (h + 1) & (TS - 1);
                                        const int c2 = atomicCAS(&sm.keys[h], kEmpty, cv);
                                        if (c2 == kEmpty || c2 == cv) break;
                                        if (--left == 0) { atomicOr(errFlag, 1); break; }
                                    }
                                    hh[v] = h;
                                }
                                atomicOr(&sm.masks[hh[v]], (unsigned)(pr[g + v] >> 32));
                            }
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
            }
        }
        wave_sync();
        dC = d1; d1 = d2; d2 = d3;
        c1 = c2;
        beC = be1;
        asm volatile("" : "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(c1), "+v"(beC.x), "+v"(beC.y));
        // ---- nnz of the row: popcount of every mask
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < (TS + 255) / 256; ++k) {
            const int s = k * 256 + lane * 4;
            if (TS >= 256 || s < TS) {
                const int4 mk = *reinterpret_cast<const int4*>(&sm.masks[s]);
                cnt += __popc((unsigned)mk.x) + __popc((unsigned)mk.y) + __popc((unsigned)mk.z) + __popc((unsigned)mk.w);
            }
        }
        cnt = wave_sum_dpp(cnt);
        if (lane == 0) cntOut[row] = cnt;
        wave_sync();
    }
}

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
constexpr int lane_waves(int K, bool NUM) { return !NUM ? (K <= 8 ? 8 : K <= 10 ? 6 : 5) : (BHS_LANE_S == 16 ? 3 : BHS_LANE_S == 8 ? (K <= 10 ? 5 : 4) : (K <= 10 ? 7 : 4)); }

template <int K, bool NUM, bool SMALLB>
__global__ __launch_bounds__(256, lane_waves(K, NUM)) void k_row_lane(const int4* __restrict__ desc, int qn,
                                                  const int* __restrict__ Ap,
                                                  const int* __restrict__ Aj, const value_t* __restrict__ Ax,
                                                  const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                  const value_t* __restrict__ Bx, int* __restrict__ cntOut,
                                                  int* __restrict__ Cj, value_t* __restrict__ Cx,
                                                  int* __restrict__ ubOut, unsigned long long* __restrict__ ctSlots,
                                                  int* __restrict__ errFlag)
{
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
                    Cj[o] = sCol[w][r * SP + e];
                    Cx[o] = sVal[w][r * SP + e];
                }
            }
            wave_sync();
        }
    }
}

// ===========================================================================
// Per-row sort of a CSR matrix by column, in place and stable: the device
// counterpart of ref_spgemm::csr_sort_indices (SpGEMM_cuda/ref_spgemm.h:37-62),
// which the reference's driver runs on the host over every Matrix Market input
// (main.cu:62-64) because the long-row kernels want ascending B rows.
//   k_sort_rows_wave : one wavefront per row; rows of <= 1024 entries are sorted
//                      in registers as (column << 32 | position) keys -- the
//                      position makes the order stable and tells where the value
//                      comes from; rows already in order are left alone; longer
//                      rows are appended to a list
//   k_sort_rows_block: one workgroup per listed row, "flip" bitonic network for
//                      any length (partners past the end are +inf and never move
//                      down), keys in LDS up to 4096 entries, in a scratch array
//                      in HBM beyond
// ===========================================================================
constexpr int kSortLdsMax = 4096;

template <int E>
__device__ __forceinline__ void sort_row_wave(long long start, int len, int lane, int* __restrict__ Aj,
                                              value_t* __restrict__ Ax)
{
    using T = unsigned long long;
    T x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        x[e] = i < len ? (((T)(unsigned)Aj[start + i] << 32) | (unsigned)i) : ~0ull;
    }
    wave_bitonic_sort<T, E>(x, lane);
    value_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        v[e] = 0;
        if (lane * E + e < len) v[e] = Ax[start + (unsigned)x[e]];
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                      // every value of the row is in registers before one is overwritten
    wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        if (i < len) { Aj[start + i] = (int)(x[e] >> 32); Ax[start + i] = v[e]; }
    }
}

__global__ __launch_bounds__(256) void k_sort_rows_wave(int m, const int* __restrict__ Ap, int* __restrict__ Aj,
                                                        value_t* __restrict__ Ax, int* __restrict__ longList,
                                                        int* __restrict__ longCount)
{
    const int lane = threadIdx.x & 63;
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < m; row += (long long)gridDim.x * 4) {
        const long long start = Ap[row];
        const int len = Ap[row + 1] - Ap[row];
        bool bad = false;
        for (int i = lane; i + 1 < len; i += 64) bad |= Aj[start + i] > Aj[start + i + 1];
        if (!__any(bad)) continue;                             // (non-decreasing: a stable sort would not move anything)
        if (len <= 64) sort_row_wave<1>(start, len, lane, Aj, Ax);
        else if (len <= 128) sort_row_wave<2>(start, len, lane, Aj, Ax);
        else if (len <= 256) sort_row_wave<4>(start, len, lane, Aj, Ax);
        else if (len <= 512) sort_row_wave<8>(start, len, lane, Aj, Ax);
        else if (len <= 1024) sort_row_wave<16>(start, len, lane, Aj, Ax);
        else if (lane == 0) longList[atomicAdd(longCount, 1)] = (int)row;
    }
}

__global__ __launch_bounds__(256) void k_sort_rows_block(const int* __restrict__ Ap, int* __restrict__ Aj,
                                                         value_t* __restrict__ Ax, const int* __restrict__ longList,
                                                         const int* __restrict__ longCount,
                                                         unsigned long long* __restrict__ scratchK,
                                                         value_t* __restrict__ scratchV)
{
    using T = unsigned long long;
    __shared__ T ldsK[kSortLdsMax];
    __shared__ value_t ldsV[kSortLdsMax];
    const int tid = threadIdx.x;
    const int nLong = *longCount;
    for (int li = blockIdx.x; li < nLong; li += gridDim.x) {
        const int row = longList[li];
        const long long start = Ap[row];
        const int len = Ap[row + 1] - Ap[row];
        int P = 1;
        while (P < len) P <<= 1;
        const bool inLds = len <= kSortLdsMax;
        T* buf = inLds ? ldsK : scratchK + start;
        value_t* vb = inLds ? ldsV : scratchV + start;
        for (int i = tid; i < len; i += 256) buf[i] = ((T)(unsigned)Aj[start + i] << 32) | (unsigned)i;
        __syncthreads();
        auto cmpx = [&](int a, int b) {
            const T x = buf[a], y = buf[b];
            if (x > y) { buf[a] = y; buf[b] = x; }
        };
        for (int k = 2; k <= P; k <<= 1) {
            const int hk = k >> 1;
            for (int i = tid; i < (P >> 1); i += 256) {          // flip: o-th element of a block with its mirror image
                const int blk = i / hk, o = i - blk * hk;
                const int a = blk * k + o, b = blk * k + k - 1 - o;
                if (b < len) cmpx(a, b);
            }
            __syncthreads();
            for (int j = k >> 2; j > 0; j >>= 1) {
                for (int i = tid; i < (P >> 1); i += 256) {
                    const int a = (i / j) * 2 * j + (i % j), b = a + j;
                    if (b < len) cmpx(a, b);
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < len; i += 256) vb[i] = Ax[start + (unsigned)buf[i]];
        __syncthreads();
        for (int i = tid; i < len; i += 256) { Aj[start + i] = (int)(buf[i] >> 32); Ax[start + i] = vb[i]; }
        __syncthreads();
    }
}

}  // namespace bhs
