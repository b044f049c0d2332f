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
//
// Round 4: this header keeps the shared constants and the streaming kernels of stages 1 and 3 (upper bound, queues,
// scans, the hand-over scans); the row kernels live in bhs_row_wg.hip.h (k_row_block, k_row_spa, k_row_bitmap_lds),
// bhs_row_wave.hip.h (k_row_wave), bhs_row_quad.hip.h, bhs_row_lane.hip.h, bhs_compress.hip.h (k_compress_b,
// k_row_wave_csym) and bhs_sort.hip.h -- one translation unit, included by bhsparse_hip.hip in this order.
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
    int laneCost;                   // ... and v^2 <= laneCost x (entries in the A row); 0: no such limit (see kLaneCost)
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
    if (qv > 0 && qv <= s.laneMax && nA <= s.laneMaxA && (s.laneCost == 0 || (long long)qv * qv <= (long long)s.laneCost * nA)) return kLaneBin;
    if (qv > 0 && qv <= s.quadMax && nA <= kQuadMaxA) return 1;
    int b = 0;
#pragma unroll
    for (int i = 0; i < kMaxBins; ++i) b += (i < s.nbins - 1 && v > s.upper[i]) ? 1 : 0;
    return (b == 1) ? 2 : b;        // 0 < v <= upper[2] that did not qualify for the quad bin
}

// Stores of C in the general pipeline's kernels: written once, read by nobody here.  BHS_GEN_NT 1: non-temporal.
template <typename T>
__device__ __forceinline__ void gen_store_c(T* p, T v)
{
#if BHS_GEN_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
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
// The bins' queue starts from their counts, on the device: start[0] = 0, start[b + 1] = start[b] + (b ? count[b] : 0) -- what
// the host computes from the same counts once it has read them back; with this the queues can be filled WHILE the host
// waits for that read-back (round 5: the general pipeline's host-visible gaps, ~40 us each on a web graph).
__global__ void k_bin_starts(const int* __restrict__ count, int* __restrict__ start)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int at = 0;
        for (int b = 0; b < kMaxBins; ++b) { start[b] = at; at += b == 0 ? 0 : count[b]; }
    }
}

constexpr int kFillRounds = BHS_FILL_ROUNDS;     // rows per thread per reservation
constexpr int kFillTile = 256 * kFillRounds;     // rows per block per global reservation

template <bool FROM_ROWPTR>
__global__ __launch_bounds__(256) void k_fill_queues(int m, const int* __restrict__ key,
                                                     const int* __restrict__ Ap, const int* __restrict__ ub,
                                                     const int* __restrict__ binStart,
                                                     int* __restrict__ binCursor, int4* __restrict__ queue,
                                                     BinSpec spec, unsigned long long* __restrict__ binSums,
                                                     // round 6 (bhs_class_mix.hip.h): the rows list[0 .. *listCount) that lie in
                                                     // [rlo, rhi) instead of the rows 0 .. m - 1 (nullptr: all rows)
                                                     const int* __restrict__ list = nullptr, const int* __restrict__ listCount = nullptr,
                                                     int rlo = 0, int rhi = 0x7fffffff)
{
    __shared__ int hist[kMaxBins];
    __shared__ int base[kMaxBins];
    __shared__ unsigned long long sums[kMaxBins * 3];   // per bin: products, nnz(C rows), nnz(A rows)
    const int tid = threadIdx.x;
    if (tid < kMaxBins * 3) sums[tid] = 0;
    if (list != nullptr) m = *listCount;
    for (long long r0 = (long long)blockIdx.x * kFillTile; r0 < m; r0 += (long long)gridDim.x * kFillTile) {
        if (tid < kMaxBins) hist[tid] = 0;
        __syncthreads();
        int bb[kFillRounds], pp[kFillRounds];
        int4 dd[kFillRounds];
#pragma unroll
        for (int r = 0; r < kFillRounds; ++r) {
            const long long idx = r0 + (long long)r * 256 + tid;
            long long row = idx;
            if (list != nullptr) {
                row = idx < m ? list[idx] : 0x7fffffff;
                if (row < rlo || row >= rhi) row = 0x7fffffff;       // (outside the range: as if beyond the last row)
            }
            int b = 0, pos = 0, a0 = 0, a1 = 0, outBase = 0, v = 0, ubv = 0;
            if (row < (list != nullptr ? 0x7fffffffLL : (long long)m)) {
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
// Round 4: the same in ONE pass (the class path's k_class_scan showed the way: 53 -> 18 us there).  Tiles of 8192 rows in
// ticket order; a tile publishes its sum, its first wave looks back over its predecessors 64 at a time -- a sum (flag 1) is
// added, a running total (flag 2) ends the walk -- and publishes its own running total.  state[tile] = flag << 62 | epoch
// << 44 | value, one word written and read with relaxed device-scope atomics; the epoch (a per-handle multiply counter)
// makes last multiply's words read as "not published", so the array is never cleared.  Counts in, row pointers out, in
// place; nnz(C), the numeric bins' histogram and the longest row of C on the side, as k_scan_reduce delivers them.
constexpr int kScan1Block = 1024, kScan1Per = 8, kScan1Tile = kScan1Block * kScan1Per;
__global__ __launch_bounds__(kScan1Block) void k_scan_onepass(int m, int* __restrict__ cnt_to_ptr, const int* __restrict__ Ap,
                                                              unsigned long long* __restrict__ state, unsigned epoch,
                                                              int* __restrict__ ticket, long long* __restrict__ totalOut,
                                                              int* __restrict__ binCount, BinSpec spec, int* __restrict__ maxCnt,
                                                              const int* __restrict__ ub)
{
    constexpr int NW = kScan1Block / 64;
    __shared__ int hist[kMaxBins], sTile, wsum[NW], wmax[NW];
    __shared__ long long sPrefix;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < kMaxBins) hist[tid] = 0;
    if (tid == 0) sTile = atomicAdd(ticket, 1);
    __syncthreads();
    const int tile = sTile;
    const long long base = (long long)tile * kScan1Tile + (long long)tid * kScan1Per;
    int v[kScan1Per], mine = 0, mx = 0;
#pragma unroll
    for (int j = 0; j < kScan1Per; ++j) {
        v[j] = 0;
        const long long idx = base + j;
        if (idx < m) {
            v[j] = cnt_to_ptr[idx];
            mx = max(mx, v[j]);
            const int b = bin_of(spec, v[j], Ap[idx + 1] - Ap[idx], v[j], spec.hubMin > 0 ? ub[idx] : 0);
            if (b > 0) atomicAdd(&hist[b], 1);
        }
        mine += v[j];
    }
    const int incl = wave_incl_scan_dpp(mine);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if (lane == 63) wsum[wv] = incl;
    if (lane == 0) wmax[wv] = mx;
    __syncthreads();
    int before = 0, tileSum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { before += w < wv ? wsum[w] : 0; tileSum += wsum[w]; }
    constexpr unsigned long long kVal = (1ull << 44) - 1ull;
    const unsigned long long tag = (unsigned long long)(epoch & 0x3FFFFu) << 44;
    if (wv == 0) {
        long long run = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&state[tile], (1ull << 62) | tag | (unsigned long long)tileSum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int back = tile - 1; back >= 0;) {               // (wave-uniform)
                const int p = back - lane;
                unsigned long long st = (2ull << 62) | tag;       // (before the first tile: a running total of 0)
                if (p >= 0) st = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool mineEpoch = (st & (0x3FFFFull << 44)) == tag;
                const int flag = mineEpoch ? (int)(st >> 62) : 0;
                const unsigned long long unset = __ballot(flag == 0), total = __ballot(flag == 2);
                const int firstTotal = total ? __ffsll((long long)total) - 1 : 64;      // nearest predecessor with a running total
                if (unset & ((firstTotal < 64 ? (2ull << firstTotal) : 0ull) - 1ull)) continue;      // a nearer one has not published: again
                long long part = lane <= firstTotal ? (long long)(st & kVal) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                run += part;
                if (firstTotal < 64) break;
                back -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&state[tile], (2ull << 62) | tag | ((unsigned long long)(run + tileSum) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sPrefix = run;
        }
    }
    __syncthreads();
    long long at = sPrefix + before + incl - mine;
#pragma unroll
    for (int j = 0; j < kScan1Per; ++j) {
        if (base + j < m) cnt_to_ptr[base + j] = (int)at;
        at += v[j];
    }
    if (base <= (long long)m - 1 && (long long)m - 1 < base + kScan1Per) { cnt_to_ptr[m] = (int)at; *totalOut = at; }   // (the thread of the last row)
    if (tid == 0) {
        int mm = 0;
        for (int w = 0; w < NW; ++w) mm = max(mm, wmax[w]);
        if (mm) atomicMax(maxCnt, mm);                              // longest row of C (numeric-first test)
    }
    if (tid < kMaxBins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);
}

__global__ __launch_bounds__(256) void k_bin_hist(int m, const int* __restrict__ Cp, const int* __restrict__ Ap,
                                                  BinSpec spec, int* __restrict__ binCount, int* __restrict__ maxCnt,
                                                  const int* __restrict__ ub,
                                                  // round 6 (bhs_class_mix.hip.h): the listed rows inside [rlo, rhi) only
                                                  const int* __restrict__ list = nullptr, const int* __restrict__ listCount = nullptr,
                                                  int rlo = 0, int rhi = 0x7fffffff)
{
    __shared__ int hist[kMaxBins];
    __shared__ int wmax[4];
    const int tid = threadIdx.x;
    if (tid < kMaxBins) hist[tid] = 0;
    __syncthreads();
    int mx = 0;
    if (list != nullptr) m = *listCount;
    for (long long idx = (long long)blockIdx.x * 256 + tid; idx < m; idx += (long long)gridDim.x * 256) {
        long long i = idx;
        if (list != nullptr) {
            i = list[idx];
            if (i < rlo || i >= rhi) continue;
        }
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
// lenStats (round 6, the classifier's sizes where a FEW rows are long -- bhs_class_mix.hip.h): [0] rows beyond 64 entries, [1] the
// longest row within 64, [2] rows beyond 256, [3] the longest within 256
__global__ __launch_bounds__(256) void k_max_row(int m, const int* __restrict__ Ap, int* __restrict__ out, int* __restrict__ lenStats = nullptr)
{
    __shared__ int wmax[4], w64[4], w256[4], n64[4], n256[4];
    int mx = 0, mx64 = 0, mx256 = 0, c64 = 0, c256 = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256) {
        const int len = Ap[i + 1] - Ap[i];
        mx = max(mx, len);
        if (len <= 64) mx64 = max(mx64, len); else ++c64;
        if (len <= 256) mx256 = max(mx256, len); else ++c256;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, __shfl_xor(mx, o, 64));
        mx64 = max(mx64, __shfl_xor(mx64, o, 64));
        mx256 = max(mx256, __shfl_xor(mx256, o, 64));
        c64 += __shfl_xor(c64, o, 64);
        c256 += __shfl_xor(c256, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; wmax[w] = mx; w64[w] = mx64; w256[w] = mx256; n64[w] = c64; n256[w] = c256; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (mx) atomicMax(out, mx);
        if (lenStats != nullptr) {
            const int a = n64[0] + n64[1] + n64[2] + n64[3], b = n256[0] + n256[1] + n256[2] + n256[3];
            if (a) atomicAdd(&lenStats[0], a);
            atomicMax(&lenStats[1], max(max(w64[0], w64[1]), max(w64[2], w64[3])));
            if (b) atomicAdd(&lenStats[2], b);
            atomicMax(&lenStats[3], max(max(w256[0], w256[1]), max(w256[2], w256[3])));
        }
    }
}

// B-row sortedness check, element-parallel and without a search for row boundaries (round 5): k_sorted_flat counts the
// positions f >= 1 with Bj[f - 1] >= Bj[f] over the whole index array as one stream (16-byte loads), row boundaries
// included; k_sorted_starts counts the same test at the first entry of every non-empty row but the first.  The rows are
// strictly ascending iff the two counts are equal.  (k_check_sorted below walks row by row with a lane group per row:
// 0.14 ms for poisson27pt 128^3's 223 MB, 1.6 TB/s; these two: one stream of colIndB and one of rowPtrB with a gather.)
__global__ __launch_bounds__(256) void k_sorted_flat(long long nnz, const int* __restrict__ Bj, int* __restrict__ flat)
{
    int bad = 0;
    const bool aligned = ((size_t)Bj & 15) == 0;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < nnz; i += (long long)gridDim.x * 1024) {
        int v[4];
        if (aligned && i + 3 < nnz) {
            const int4 q = *reinterpret_cast<const int4*>(Bj + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = i + e < nnz ? Bj[i + e] : 0x7fffffff;
        }
        const int prev = i > 0 ? Bj[i - 1] : -1;                  // (columns are >= 0)
        bad += (prev >= v[0] ? 1 : 0) + (i + 1 < nnz && v[0] >= v[1] ? 1 : 0) + (i + 2 < nnz && v[1] >= v[2] ? 1 : 0) +
               (i + 3 < nnz && v[2] >= v[3] ? 1 : 0);
    }
    // (one atomic per block, on one of 32 counters: a matrix with ascending rows has a "violation" at almost every row
    // start, and thousands of same-address atomics queue for longer than the scan takes)
    __shared__ int part[4];
    bad = wave_sum_dpp(bad);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0 && part[0] + part[1] + part[2] + part[3]) atomicAdd(&flat[blockIdx.x & 31], part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void k_sorted_starts(int k, const int* __restrict__ Bp, const int* __restrict__ Bj, int* __restrict__ atStarts)
{
    int bad = 0;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < k; r += (long long)gridDim.x * 256) {
        const int a0 = Bp[r], a1 = Bp[r + 1];
        if (a1 > a0 && a0 > 0) bad += Bj[a0 - 1] >= Bj[a0] ? 1 : 0;
    }
    __shared__ int part[4];
    bad = wave_sum_dpp(bad);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0 && part[0] + part[1] + part[2] + part[3]) atomicAdd(&atStarts[blockIdx.x & 31], part[0] + part[1] + part[2] + part[3]);
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

}  // namespace bhs
