// bhs_class.hip.h -- row classes: the structure of one row of C, worked out once per CLASS of rows.
//
// On the matrices this benchmark is about (the gallery's finite-difference stencils, SpGEMM_cuda/main.cu:56-64 and
// README.md:40-54, and anything else assembled on a regular grid) almost every row looks like its neighbours, shifted:
// the columns of row i of A are i + a_0, i + a_1, ..  and the row j of B behind each of them has columns j + b_0, ..
// Two rows of A whose offset lists a_k are equal, and whose B rows are pairwise equal as offset lists too, produce the
// same intermediate products at the same relative columns: the same sorted column list (relative to the row), the
// same number of entries, and the same map from product number to position in the row.  poisson27pt on a 128^3 grid has
// 2 097 152 rows and 125 such classes.
//
// The general pipeline re-discovers that structure for every row (hash inserts, compaction, sort: ~900 wavefront
// instructions per row).  Here it is found once per class:
//   k_class_rows<false>  class id of every row of B: hash of its relative pattern -> slot of a small hash table whose
//                        entry names the class's representative row; the row is COMPARED with the representative
//                        entry by entry, so a hash collision costs a probe, never a wrong answer
//   k_class_rows<true>   class id of every row of A: the same over the list of (relative column, class of that B row)
//   k_class_patterns     per class: the products of the representative row, sorted and made unique -> relative column
//                        list, entry count, and for every product its {A entry, B entry, position} triple
//   k_class_scan         rowPtrC: the class's entry count of every row, scanned in one pass (look-back over tiles)
//   k_class_numeric      per row: 1 load of A's entries and rowPtrB, then every product is one load of B's value and
//                        one fma into a REGISTER: a lane holds consecutive products of the class's position-sorted
//                        product list, so the products of one entry of C meet in one lane; the running sums go to
//                        the entry's slot with plain LDS stores and the row is written out with its columns (class
//                        list + row number) in ascending order.  No column of B is read, nothing is hashed,
//                        compacted or sorted, and no LDS atomic runs per product.
// Rows the tables cannot take (more than kClassMaxRow entries in a row of A or B, more than kClassMaxP products or
// kClassMaxNnz entries per row, a full table) send the whole multiply back to the general pipeline, and the data set
// stays there.  Replaces, for the matrices that qualify, all of SpGEMM_cuda/bhsparse_cuda.h:210-2780.
#pragma once

namespace bhs {

constexpr int kClassEpl = 16 / (int)sizeof(value_t);   // values of B per lane of a 16-byte LDS-direct load
constexpr int kClassSlots = 4096;          // slots of each hash table; a class id is a slot number
constexpr int kClassProbe = 32;            // linear probes before a row counts as unclassified
constexpr int kClassMaxRow = 64;           // entries per row of A / of B  } classes within these limits get the tables of
constexpr int kClassMaxP = 1024;           // products per row of C        } the register kernels (k_class_patterns)
constexpr int kClassMaxRowBig = 256;       // the same limits of the "big" classes (bhs_class_big.hip.h: block-structured
constexpr int kClassBigMaxP = 8192;        //   grids -- several unknowns per node): their product lists stay in memory
constexpr int kClassBigCap = 1024;         // big classes per multiply
constexpr int kClassMaxNnz = 512;          // entries per row of C
constexpr unsigned long long kClassEmpty = ~0ull;
constexpr int kClassDummy = kClassSlots;    // mixed mode: the "class" of a row without one -- slot kClassSlots of classInfo / classRing / classLane, all zeros: no entries, no products
constexpr int kMixBesideRows = 1024;       // mixed mode: up to this many irregular rows their numeric kernels run beside the ring kernel, not in front of it
constexpr int kClassManyClasses = 256;     // more classes than this in a multiply: are they classes, or single rows? (the mixed flow counts)
constexpr int kClassMixMinRows = 4;        // mixed mode (bhs_class_mix.hip.h): a class with fewer rows than this is not worked out, its rows are irregular

// `stats` block written by k_class_rows / k_class_patterns (ints)
constexpr int kClassSumSlots = 32;
enum { CS_MAXRING = 0 /* most staged B values any class's ring needs (bhs_class_wg.hip.h), 0x7fffffff: some class cannot */,
       CS_MAXLB = 1 /* longest B row behind any class's A entries */, CS_MAXSLAB = 7 /* most values per slab */, CS_FLAGS = 2 /* 1 unclassified row, 2 class beyond the limits */, CS_MAXP = 3, CS_MAXNNZ = 4, CS_CLASSES = 5,
       CS_MAXNA = 6 /* longest A row of any class */,
       CS_SUMS = 8 /* kClassSumSlots x u64: products */, CS_RANGE = 8 + 2 * kClassSumSlots /* 2 ints: columns of A */,
       CS_BIGCOUNT = 8 + 2 * kClassSumSlots + 2 /* big classes */, CS_BIGMAXP = 8 + 2 * kClassSumSlots + 3 /* words of their longest list */,
       CS_SCANTICKET = 8 + 2 * kClassSumSlots + 4 /* tile numbers of k_class_scan */,
       CS_HEADS = 8 + 2 * kClassSumSlots + 5 /* rows of A that went through the class table (k_class_rows on the heads' lists) */,
       CS_RINGFULL = 8 + 2 * kClassSumSlots + 6 /* bhs_class_ring.hip.h: most values of a ring of (longest chain + 1) slabs among the classes whose ring fits kClassRingBudget */,
       CS_RINGONE = 8 + 2 * kClassSumSlots + 7 /* ... most values of (longest chain) slabs among the others (their rows load what they need, row by row); 0x7fffffff: some class cannot */,
       CS_INTS = 8 + 2 * kClassSumSlots + 8 };
constexpr int kClassRingDma = kClassMaxP + 64 + kClassMaxNnz;      // classRing: where the slab-load words of bhs_class_ring.hip.h's kernel begin (kClassMaxLoads x 64, then the slab's size)
constexpr int kClassRingStride = kClassRingDma + 4 * 64 + 4;       // words per class of classRing: 16 steps x 64 lanes, a word per lane, the relative columns in pairs, the slab loads
constexpr int kClassColourUnits = 128;                             // slabs of up to this many 16-byte units get their units coloured over the bank groups
constexpr int kClassRingBudget = 8192;     // bytes of LDS a wave of bhs_class_ring.hip.h's kernel gives its ring: with the slots of a row of C and the row's A values, 16 waves per CU

__device__ __forceinline__ unsigned class_mix(unsigned h, unsigned v)
{
    h = (h ^ v) * 0x9E3779B1u;
    return h ^ (h >> 15);
}

// ---------------------------------------------------------------------------
// Class of every row.  IS_A = false: rows of B, element = relative column.  IS_A = true: rows of A, element =
// (relative column, class of the B row it selects); also counts the rows of every class.
// G lanes share a row (coalesced loads of its entries, one entry per lane and pass); the row's hash is the sum of
// its position-keyed element hashes, the comparison with the class's representative row is one entry per lane too.
// ---------------------------------------------------------------------------
// Smallest and largest column of A: when A is a row block of a larger product (multi-GPU: m rows of A against all k
// rows of B) only the rows of B that A points at need a class.  range[0] = min (preset INT_MAX), range[1] = max (-1).
__global__ __launch_bounds__(256) void k_class_col_range(long long nnzA, const int* __restrict__ Aj, int* __restrict__ range)
{
    __shared__ int smin[4], smax[4];
    int lo = 0x7fffffff, hi = -1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnzA; i += (long long)gridDim.x * 256) {
        const int c = Aj[i];
        lo = min(lo, c);
        hi = max(hi, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&range[0], min(min(smin[0], smin[1]), min(smin[2], smin[3])));
        atomicMax(&range[1], max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}

template <int G>
__device__ __forceinline__ unsigned group_sum_u32(unsigned v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    return v;
}

constexpr int kClassRowsBlock = 1024;      // large blocks: fewer block-local class caches to warm up
template <bool IS_A, int G, int E>         // G lanes per row, E entries per lane: rows of up to G * E entries
__global__ __launch_bounds__(kClassRowsBlock) void k_class_rows(int nrows, const int* __restrict__ Rp, const int* __restrict__ Rj,
                                                    const int* __restrict__ classB,
                                                    unsigned long long* __restrict__ table,
                                                    int* __restrict__ classOut, int* __restrict__ stats,
                                                    const int* __restrict__ range,     // rows [range[0], range[1]] only (nullptr: all)
                                                    const int* __restrict__ rowList,   // nullptr: all rows; else the rows to classify: segment
                                                    const int* __restrict__ rowCount,  //   blockIdx.y of kClassHeadSegs lists (k_class_heads),
                                                    int segCap)                        //   segCap slots apart, its length at rowCount[16 * segment]
{
    static_assert(G * E <= kClassMaxRowBig, "rows of up to kClassMaxRowBig entries");
    constexpr int PW = G * E > kClassMaxRow ? G * E : kClassMaxRow;   // entries per pattern of the block's class cache
    constexpr int RPB = kClassRowsBlock / G;                       // rows per block and pass
    // Block-local cache of the table, indexed by the hash: {slot, 20 bits of the hash} and the class's pattern (the
    // relative columns, for A rows also the B classes) -- a row whose class is here is recognised without touching
    // the representative row in memory.  The blocks are persistent and a stretch of rows has few classes; the
    // device-wide table (coherent loads, compare-and-swap) is for the misses.  An entry is claimed once (LDS
    // compare-and-swap to "busy"), filled, then published; it never changes afterwards.
    constexpr int NC = PW > 2 * kClassMaxRow ? 16 : 32;
    constexpr unsigned kBusy = 0xFFFFFFFEu;
    __shared__ unsigned ctag[NC];
    __shared__ int cpat[NC][PW];
    __shared__ int clen[NC];
    __shared__ int cpatB[IS_A ? NC : 1][PW];
    const int tid = threadIdx.x, lane = tid & 63, g = tid % G;
    if (tid < NC) { ctag[tid] = 0xFFFFFFFFu; clen[tid] = -1; }   // (LDS is not cleared between workgroups)
    __syncthreads();
    const int leaderLane = lane - g;                               // first lane of this lane's group
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << leaderLane;
    // R row sets per pass and lane group: their load chains (rowPtr -> colInd -> B class) are independent, so the
    // three memory round trips of a pass are shared by R rows per group instead of paid per row.  The pass is bound
    // by those round trips (SQ_WAIT_ANY 60-80 % of the wave cycles), so short rows -- few entries per lane -- take
    // more row sets: R * E = 16 entries per lane in flight.
    constexpr int R = E >= 8 ? 2 : (E >= 4 ? 4 : 8);
    bool noClass = false;                                          // a row of this wave's found no class
    long long first = 0;
    if (rowList != nullptr) {                                      // (positions in the segment's list from here on)
        rowList += (size_t)blockIdx.y * segCap;
        nrows = rowCount[16 * blockIdx.y];
        if (IS_A && blockIdx.x == 0 && tid == 0) atomicAdd(&stats[CS_HEADS], nrows);   // (the host's verdict: rows in stretches, or every row for itself?)
    } else if (range != nullptr) {                                 // (wave-uniform values)
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        nrows = lo <= hi ? min(nrows, hi + 1) : 0;
    }
    const long long span = first + ((long long)nrows - first + RPB * R - 1) / (RPB * R) * (RPB * R);   // whole blocks take part in the shuffles
    for (long long row0 = first + (long long)blockIdx.x * RPB * R + tid / G; row0 < span; row0 += (long long)gridDim.x * RPB * R) {
        long long rowv[R];
        bool live[R], ok[R];
        int a0[R], len[R], el[R][E], cb[R][E], cc[R][E];
        unsigned h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            rowv[r] = row0 + (long long)r * RPB;
            live[r] = rowv[r] < nrows;
        }
        if (rowList != nullptr) {
#pragma unroll
            for (int r = 0; r < R; ++r) rowv[r] = rowList[live[r] ? rowv[r] : 0];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {                               // (unpredicated: both row sets' loads in flight)
            const long long rr = live[r] ? rowv[r] : 0;
            a0[r] = Rp[rr];
            len[r] = Rp[rr + 1];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            len[r] = live[r] ? len[r] - a0[r] : 0;
            ok[r] = live[r] && len[r] <= G * E;                   // (longer: no class, the multiply goes to the general pipeline)
        }
        // all loads of the rows first, without predicates (a position past the row's end re-reads its last entry), so
        // that they are in flight together; then the gather of the B classes, likewise; then the hashes
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int lastPos = ok[r] && len[r] > 0 ? a0[r] + len[r] - 1 : 0;
#pragma unroll
            for (int e = 0; e < E; ++e) cc[r][e] = Rj[min(a0[r] + e * G + g, lastPos)];
        }
        if (IS_A) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < E; ++e) cb[r][e] = classB[cc[r][e]];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned hp = 0;
            bool bad = false;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = e * G + g;
                const bool in = ok[r] && pos < len[r];
                el[r][e] = in ? cc[r][e] - (int)rowv[r] : 0;
                if (!IS_A || !in) cb[r][e] = 0;
                unsigned hh = class_mix(0x85EBCA6Bu * (unsigned)(pos + 1), (unsigned)el[r][e]);
                if (IS_A) {
                    bad = bad || cb[r][e] < 0;
                    hh = class_mix(hh, (unsigned)cb[r][e]);
                }
                hp += in ? hh : 0u;
            }
            if (IS_A && (__ballot(bad) & gmask)) ok[r] = false;    // a B row without a class: none for this row either
            h[r] = group_sum_u32<G>(hp) + (unsigned)len[r] * 0x9E3779B1u + 1u;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long row = rowv[r];
            const unsigned hr = h[r];
            const int lenr = len[r];
            int cls = -1;
            // does this row equal row `rep` entry by entry?  (one entry per lane and pass; the group votes)
            auto equals = [&](bool cand, int rep) {
                bool same = true;
                if (__any(cand && rep != (int)row)) {               // (rare: a class this block meets for the first time)
                    const int rp = cand ? rep : 0;
                    const int r0 = Rp[rp], r1 = Rp[rp + 1];
                    const int lastR = r1 > r0 ? r1 - 1 : 0;
                    int cr[E], cbr[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) cr[e] = Rj[min(r0 + e * G + g, lastR)];
                    if (IS_A) {
#pragma unroll
                        for (int e = 0; e < E; ++e) cbr[e] = classB[cr[e]];
                    }
                    same = r1 - r0 == lenr;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool in = e * G + g < lenr;
                        same = same && (!in || (el[r][e] == cr[e] - rp && (!IS_A || cb[r][e] == cbr[e])));
                    }
                    same = same || rep == (int)row;
                }
                return cand && !(__ballot(cand && !same) & gmask);
            };
            bool searching = ok[r];
            const int ci = (int)(hr & (NC - 1));
            {
                unsigned tg = 0xFFFFFFFFu;
                if (searching && g == 0) tg = __hip_atomic_load(&ctag[ci], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                tg = (unsigned)__shfl((int)tg, leaderLane, 64);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the pattern is read after its tag, never before
                const bool cand = searching && tg < kBusy && (tg & 0xFFFFFu) == (hr >> 12);
                bool same = clen[ci] == lenr;
                int pc[E], pb[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { pc[e] = cpat[ci][e * G + g]; pb[e] = IS_A ? cpatB[ci][e * G + g] : 0; }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = e * G + g < lenr;
                    same = same && (!in || (el[r][e] == pc[e] && (!IS_A || cb[r][e] == pb[e])));
                }
                if (cand && !(__ballot(cand && !same) & gmask)) { cls = (int)(tg >> 20); searching = false; }
            }
            const unsigned long long mine = ((unsigned long long)hr << 32) | (unsigned)row;
            int s = (int)(hr & (kClassSlots - 1));
            for (int probe = 0; probe < kClassProbe; ++probe) {
                if (!__any(searching)) break;
                // An entry changes once (empty -> final).  The load is device-coherent: a plain one could keep
                // returning the "empty" line this XCD's L2 cached before another XCD claimed the slot.
                unsigned long long v = kClassEmpty;
                if (searching && g == 0) {
                    v = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == kClassEmpty) {
                        const unsigned long long old = atomicCAS(&table[s], kClassEmpty, mine);
                        v = old == kClassEmpty ? mine : old;
                    }
                }
                v = (unsigned long long)__shfl((long long)v, leaderLane, 64);
                const int rep = (int)(unsigned)v;
                if (equals(searching && (unsigned)(v >> 32) == hr, rep)) {
                    cls = s;
                    searching = false;
                    // publish in the block's cache if its cell is still free
                    unsigned won = 0;
                    if (g == 0) won = atomicCAS(&ctag[ci], 0xFFFFFFFFu, kBusy) == 0xFFFFFFFFu ? 1u : 0u;
                    won = (unsigned)__shfl((int)won, leaderLane, 64);
                    if (won) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int pos = e * G + g;
                            if (pos < lenr) {
                                cpat[ci][pos] = el[r][e];
                                if (IS_A) cpatB[ci][pos] = cb[r][e];
                            }
                        }
                        if (g == 0) clen[ci] = lenr;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (g == 0) ctag[ci] = ((unsigned)s << 20) | (hr >> 12);
                    }
                }
                s = (s + 1) & (kClassSlots - 1);
            }
            if (live[r] && g == 0) classOut[row] = cls;
            noClass = noClass || __any(live[r] && cls < 0);          // (told once, when the wave ends: bhs_class_tile.hip.h)
        }
    }
    if (noClass && lane == 0) atomicOr(&stats[CS_FLAGS], 1);
}

// ---------------------------------------------------------------------------
// Rows that look like the row before them.  On a matrix assembled on a grid a row is, almost always, its predecessor
// shifted by one column -- the same relative pattern (and, for rows of A, the same classes of B rows behind it).
// (With several unknowns per node it is the row `period` rows back that a row repeats -- period = unknowns per node, a
// hint sampled by k_row_period when the data set is handed over: the rows are then walked in `period` interleaved
// sequences, and everything below reads "the row before" as "the row before in its sequence".)
// Such a row needs no hash and no table: it has its predecessor's class.  k_class_heads streams the rows once (every
// lane group takes R CONSECUTIVE rows, so a row's predecessor is in the same lanes' registers, or one group to the
// left; a wave takes one contiguous piece of kClassHeadPiece rows), compares, lists the rows that differ ("heads":
// poisson27pt 128^3 has one in fifty) and leaves in classOut of every other row the head it follows, as -2 - head.
// k_class_rows then classifies the listed rows only, and k_class_propagate hands the classes on.  The comparison is
// entry by entry, like the table's.  The heads go to kClassHeadSegs lists, a block's with one atomic (a single
// counter bumped once per wave queues for most of a millisecond).
// ---------------------------------------------------------------------------
// Everything the class path wants cleared at the start of a multiply, in one launch (six memsets before): the counter
// block, the class statistics, the head counters, both class tables (all ones = empty), the big classes' index (all
// ones = none) and the scan's tile words.
__global__ __launch_bounds__(256) void k_class_reset(int* __restrict__ small, int nSmall, int* __restrict__ cstats, int nStats,
                                                     int* __restrict__ headCnt, int nHead, unsigned long long* __restrict__ tab, int nTab,
                                                     int* __restrict__ bigIdx, int nBig, unsigned long long* __restrict__ scanState, int nScan)
{
    const int i0 = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
    for (int i = i0; i < nSmall; i += step) small[i] = 0;
    for (int i = i0; i < nStats; i += step) cstats[i] = 0;
    for (int i = i0; i < nHead; i += step) headCnt[i] = 0;
    for (int i = i0; i < nTab; i += step) tab[i] = kClassEmpty;
    for (int i = i0; i < nBig; i += step) bigIdx[i] = -1;
    for (int i = i0; i < nScan; i += step) scanState[i] = 0ull;
}

constexpr int kLineWindow = 4096, kLineMax = 512;                  // rows compared for the grid-line hint, the longest line it finds
// The period hint: eight rows spread over the matrix, each compared with the rows 1 .. 8 before it; the smallest
// distance at which a sample repeats, by majority (1 when there is none).  One wave, at hand-over time.
__global__ __launch_bounds__(64) void k_row_period(int nrows, const int* __restrict__ Rp, const int* __restrict__ Rj, int* __restrict__ out,
                                                   int* __restrict__ local = nullptr, int ncols = 0, int* __restrict__ line = nullptr)
{
    __shared__ unsigned changed[(kLineWindow + kLineMax) / 32 + 2];
    const int lane = threadIdx.x, smp = lane >> 3, d = (lane & 7) + 1;
    const long long row = (long long)(smp + 1) * nrows / 9;
    bool same = row - d >= 0 && row < nrows;
    if (same) {
        const int a0 = Rp[row], len = Rp[row + 1] - a0, b0 = Rp[row - d];
        same = Rp[row - d + 1] - b0 == len && len > 0 && len <= kClassMaxRowBig;   // (longer rows never classify: no walk through a hub row)
        for (int e = 0; same && e < len; ++e) same = Rj[a0 + e] - (int)row == Rj[b0 + e] - (int)(row - d);
    }
    const unsigned long long votes = __ballot(same);
    int count[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 8; ++q) {
        const unsigned byte = (unsigned)(votes >> (8 * q)) & 255u;
        if (byte) count[__ffs((int)byte)]++;                       // (the sample's smallest distance)
    }
    int best = 1;
    for (int q = 2; q <= 8; ++q) if (count[q] > count[best]) best = q;
    if (lane == 0) *out = count[best] >= 4 ? best : 1;
    // Second hint (round 4), for the lane-per-row kernels: do the entries of a row stay near its diagonal?  Those kernels
    // give a row to a lane; the 64 lanes of a wave then read the B rows behind 64 consecutive rows of A -- neighbouring
    // memory on a grid or a band, 64 different places for random columns (uniform random, 8 per row, 2^20 rows: 3.8 ms
    // on the lane kernels, 1.4 ms on the wave kernels).  64 sampled rows: mean |column - row| < columns / 16.
    if (local != nullptr) {
        const long long row = (long long)(lane + 1) * nrows / 65;
        long long dist = 0;
        int cnt = 0;
        if (row < nrows) {
            const int a0 = Rp[row], len = min(Rp[row + 1] - a0, 64);
            for (int e = 0; e < len; ++e) { const long long dd = (long long)Rj[a0 + e] - row; dist += dd < 0 ? -dd : dd; }
            cnt = len;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dist += __shfl_xor(dist, o, 64); cnt += __shfl_xor(cnt, o, 64); }
        if (lane == 0) *local = (cnt == 0 || ncols <= 4096 || dist * 16 < (long long)cnt * ncols) ? 1 : 0;   // (a small B sits in the caches anyway)
    }
    // Third hint (round 4), for the ring kernel: the length of a grid line.  Where a grid's line of unknowns ends the row
    // lengths change (the neighbour beyond the end is missing), so the places where a row's length differs from the row
    // before it repeat with the line's length -- also in the grid's first and last lines, whose rows are shorter but change
    // at the same places.  A window of kLineWindow rows from the middle of the matrix.
    if (line != nullptr) {
        const long long first = (long long)nrows / 2;
        int found = 0;
        if (first >= 1 && first + kLineWindow + kLineMax < nrows) {
            // one bit per row of the window: its length differs from the row before it
            int changes = 0;
            for (int i0 = 0; i0 < kLineWindow + kLineMax; i0 += 64 * 8) {     // (eight steps' loads in flight at once)
                int a[8], b[8], c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long long i = first + i0 + u * 64 + lane;
                    a[u] = Rp[i - 1]; b[u] = Rp[i]; c[u] = Rp[i + 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j0 = i0 + u * 64;
                    const unsigned long long m = __ballot(c[u] - b[u] != b[u] - a[u]);
                    if (lane == 0) { changed[j0 >> 5] = (unsigned)m; changed[(j0 >> 5) + 1] = (unsigned)(m >> 32); }
                    if (j0 < kLineWindow) changes += __popcll(m);
                }
            }
            if (lane == 0) changed[(kLineWindow + kLineMax) >> 5] = 0u;
            __syncthreads();
            // lane l tries the lengths 16 + l, 16 + 64 + l, ...: the smallest that fits all but an eighth of the changes (a
            // wrong length misses nearly all of them; the right one a few where the window crosses from plane to plane).
            // 32 rows per step: the window's bits against the same bits P rows on.
            const int allowed = changes / 8;
            found = 0x7fffffff;
            for (int P = 16 + lane; changes > 0 && P <= kLineMax; P += 64) {
                const int ws = P >> 5, sh = P & 31;
                int miss = 0;
                for (int w = 0; w < kLineWindow / 32 && miss <= allowed; ++w) {
                    const unsigned lo = changed[w + ws], hi = changed[w + ws + 1];
                    const unsigned there = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
                    miss += __popc(changed[w] ^ there);
                }
                if (miss <= allowed) { found = P; break; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) found = min(found, __shfl_xor(found, o, 64));
            if (found == 0x7fffffff) found = 0;
        }
        if (lane == 0) *line = found;
    }
}

constexpr int kClassHeadSegs = 8;
constexpr int kClassHeadsBlock = 512;
constexpr int kClassHeadPiece = BHS_HEAD_PIECE;                    // most rows per wave (its first is a head by decree)
// ... fewer where a row takes many lanes: a wave walks its piece pass by pass, three dependent round trips each
constexpr int class_head_piece(int G) { return G >= 64 ? 64 : (G >= 32 ? 128 : kClassHeadPiece); }
template <bool IS_A, int G, int E>
__global__ __launch_bounds__(kClassHeadsBlock) void k_class_heads(int nrows, const int* __restrict__ Rp, const int* __restrict__ Rj,
                                                     const int* __restrict__ classB, int* __restrict__ classOut,
                                                     int* __restrict__ headList, int* __restrict__ headCount, int segCap,
                                                     const int* __restrict__ range,     // rows [range[0], range[1]] only (nullptr: all)
                                                     int period)                        // a row is compared with the row `period` before it
{
    constexpr int GPW = 64 / G;                                    // lane groups per wave
    constexpr int R = E >= 8 ? 2 : (E >= 4 ? 4 : 8);               // consecutive rows per lane group
    constexpr int RPW = GPW * R;                                   // consecutive rows per wave and pass
    constexpr int WPB = kClassHeadsBlock / 64;
    constexpr int PIECE = class_head_piece(G);
    __shared__ int sList[WPB * PIECE], sCount, sBase;              // the block's heads
    const int lane = threadIdx.x & 63, g = lane % G, grp = lane / G;
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << (lane - g);
    long long first = 0;
    if (range != nullptr) {                                        // (wave-uniform values)
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        nrows = lo <= hi ? min(nrows, hi + 1) : 0;
    }
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    const long long wave = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
    const long long pieceBegin = first + wave * PIECE;
    const long long pieceEnd = min((long long)nrows, pieceBegin + PIECE);
    bool okP = false;                                              // the pass before's last row (lane g of every group)
    int lenP = 0, elP[E], cbP[E], followP = -1;
#pragma unroll
    for (int e = 0; e < E; ++e) { elP[e] = 0; cbP[e] = 0; }
    // the wave walks its piece as `period` sequences, one after the other: position q of sequence `seq` = row pieceBegin + q * period + seq
    const int perSeq = (PIECE + period - 1) / period;
    for (int seq = 0; seq < period; ++seq) {
    auto row_at = [&](int q) { return pieceBegin + (long long)q * period + seq; };
    followP = -1;
    for (int qbase = 0; qbase < perSeq && row_at(qbase) < pieceEnd; qbase += RPW) {
        const bool firstPass = qbase == 0;                         // (of this sequence: its first row is a head by decree)
        long long rowv[R];
        int vi[R];
        bool live[R], ok[R];
        int a0[R], len[R], el[R][E], cb[R][E], cc[R][E];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            vi[r] = qbase + grp * R + r;
            rowv[r] = row_at(vi[r]);
            live[r] = vi[r] < perSeq && rowv[r] < pieceEnd;
            const long long rr = live[r] ? rowv[r] : 0;
            a0[r] = Rp[rr];
            len[r] = Rp[rr + 1];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            len[r] = live[r] ? len[r] - a0[r] : 0;
            ok[r] = live[r] && len[r] <= G * E;
            const int lastPos = ok[r] && len[r] > 0 ? a0[r] + len[r] - 1 : 0;
#pragma unroll
            for (int e = 0; e < E; ++e) cc[r][e] = Rj[min(a0[r] + e * G + g, lastPos)];
        }
        if (IS_A) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < E; ++e) cb[r][e] = classB[cc[r][e]];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bool bad = false;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool in = ok[r] && e * G + g < len[r];
                el[r][e] = in ? cc[r][e] - (int)rowv[r] : 0;
                if (!IS_A || !in) cb[r][e] = 0;
                bad = bad || (IS_A && cb[r][e] < 0);
            }
            if (IS_A && (__ballot(bad) & gmask)) ok[r] = false;
        }
        // same as the row before?
        bool head[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bool differs;
            if (r == 0) {                                         // (group 0: the pass before's last row)
                const int src = max(lane - G, 0);
                // (shuffles by ALL lanes, the selection afterwards: bhs_class_tile.hip.h)
                const int okSh = __shfl((int)ok[R - 1], src, 64), lenSh = __shfl(len[R - 1], src, 64);
                const bool okB = grp ? okSh != 0 : okP;
                const int lenB = grp ? lenSh : lenP;
                differs = (grp == 0 && firstPass) || !ok[0] || !okB || len[0] != lenB;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int elB = __shfl(el[R - 1][e], src, 64);
                    differs = differs || el[0][e] != (grp ? elB : elP[e]);
                    if (IS_A) {
                        const int cbB = __shfl(cb[R - 1][e], src, 64);
                        differs = differs || cb[0][e] != (grp ? cbB : cbP[e]);
                    }
                }
            } else {
                differs = !ok[r] || !ok[r - 1] || len[r] != len[r - 1];
#pragma unroll
                for (int e = 0; e < E; ++e) differs = differs || el[r][e] != el[r - 1][e] || (IS_A && cb[r][e] != cb[r - 1][e]);
            }
            head[r] = live[r] && (__ballot(differs) & gmask) != 0;
        }
        // the head every row follows: the last head at or before it in the wave's walk (as positions of the walk)
        int lastIn = -1;
#pragma unroll
        for (int r = 0; r < R; ++r) lastIn = head[r] ? vi[r] : lastIn;
        int incl = lastIn;                                         // inclusive running maximum over the groups
#pragma unroll
        for (int o = G; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            incl = lane >= o ? max(incl, up) : incl;
        }
        int follow = __shfl_up(incl, G, 64);                       // the groups before this one, or the passes before this one
        follow = grp ? max(follow, followP) : followP;
        // the block's list: one LDS atomic per wave and pass that met a head
        int nBefore = 0, nHeads = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned long long hm = __ballot(head[r] && g == 0);
            nBefore += __popcll(hm & ((1ull << lane) - 1ull));
            nHeads += __popcll(hm);
        }
        int slot0 = 0;
        if (lane == 0 && nHeads) slot0 = atomicAdd(&sCount, nHeads);
        slot0 = __builtin_amdgcn_readfirstlane(slot0);
        if (g == 0) {
            int at = slot0 + nBefore;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (head[r]) {
                    sList[at++] = (int)rowv[r];                   // (the list need not be sorted, only complete)
                    follow = vi[r];
                } else if (live[r]) classOut[rowv[r]] = -2 - (int)row_at(follow);
            }
        }
        // hand the pass's last row to the next pass's group 0 (lane g takes lane (GPW - 1) * G + g's)
        const int from = (GPW - 1) * G + g;
        okP = (bool)__shfl((int)ok[R - 1], from, 64);
        lenP = __shfl(len[R - 1], from, 64);
#pragma unroll
        for (int e = 0; e < E; ++e) { elP[e] = __shfl(el[R - 1][e], from, 64); if (IS_A) cbP[e] = __shfl(cb[R - 1][e], from, 64); }
        followP = max(followP, __builtin_amdgcn_readlane(incl, 63));
    }
    }
    // the block's heads go to list blockIdx.x % kClassHeadSegs: one global atomic per block
    __syncthreads();
    const int seg = blockIdx.x % kClassHeadSegs, cnt = sCount;
    if (threadIdx.x == 0) sBase = cnt ? atomicAdd(&headCount[16 * seg], cnt) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kClassHeadsBlock) headList[(size_t)seg * segCap + sBase + i] = sList[i];
}

__global__ __launch_bounds__(256) void k_class_propagate(int nrows, int* __restrict__ classOut, const int* __restrict__ range)
{
    long long first = 0;
    if (range != nullptr) {
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        nrows = lo <= hi ? min(nrows, hi + 1) : 0;
    }
    for (long long i = first + (long long)blockIdx.x * 256 + threadIdx.x; i < nrows; i += (long long)gridDim.x * 256) {
        const int v = classOut[i];
        if (v <= -2) classOut[i] = classOut[-2 - v];               // (a head's entry is its class, or -1: never <= -2)
    }
}

// ---------------------------------------------------------------------------
// Pattern of every class of A rows: one 256-lane workgroup per table slot.
// classInfo[s] = {entries of the A row, products, entries of the C row (-1: beyond the limits), representative row}
// classRel[s * kClassMaxNnz + e] = column of entry e minus the row number, ascending
// classMapA[s * kClassMaxP + p] = A entry | B entry << 6 | position << 16 for product p in A-entry-major order
//   (k_class_numeric_atomic)
// classMap[s * kClassMaxP + (kClassMaxSteps - U + u) * 64 + L] = product descriptor of lane L, step u of the ring kernel.
//   The class's P products are sorted by (position in the row of C, product number) and dealt out in that order,
//   U = ceil(P / 64) consecutive products per lane: the products that sum into one entry of C sit in ONE lane, one
//   after the other (a few entries straddle a lane boundary).  The descriptor is the word the kernel keeps in a
//   register: byte offset of the entry's accumulator slot (bits 0-15), byte offset of the A entry's value in the row's
//   staged values (16-24), B entry (25-30), sign bit: first product of an entry within this lane (the running sum
//   restarts).  An entry that goes on in the NEXT lane has the class's spare slot (number nnz) instead of its own:
//   the lane's sum at the end of its list is a partial sum that it adds to that entry's position, classLane[.. + L]
//   below.  A lane with fewer than U products has its idle steps FIRST (spare slot, restart: the sum it carries at
//   the end of the list is that of real products), and the U steps are the LAST of the kClassMaxSteps stored ones, so
//   that the kernel's loads of its MAXU steps do not wait for the class's U.
// classLane[s * kClassLaneInts + ..]: [L] = tail position of lane L (-1: none).  The rest describes the CHAINS of the A
//   row: maximal stretches of A entries with consecutive columns whose B rows have one length -- consecutive rows of
//   B, also across consecutive rows of the class (row i + 1's entry k selects the B row after row i's).  A "slab" is
//   one B row of every chain side by side, each padded to whole 16-byte lanes: what one more row of the class adds.
//   [64 + L]  as A entry L: its chain's place in a slab (bits 0-15), its place in the chain (16-21), its B row's length (24-30)
//   [128 + L] as chain L:   first A entry (0-5), entries (6-12), B row length (13-19), place in a slab (20-30)
//   [192] chains (0-7), entries of the longest chain (8-15), values per slab (16-31)
//   [256 + j * 64 + L] lane L's 16 bytes of the j-th LDS-direct load of a slab: place in its chain's B row (bits 0-7),
//             that row's length (8-15; 0: a padding lane, no load), the chain's first A entry (16-21)
// ---------------------------------------------------------------------------
constexpr int kClassLaneInts = 512;
constexpr int kClassSpanWords = 4096;                            // k_class_patterns ranks relative columns with a bitmap of this many words where their span fits
constexpr int kClassMaxSteps = kClassMaxP / 64;                  // steps of the stored map
constexpr int kClassMaxLoads = 4;                                // LDS-direct loads per slab (64 lanes x 16 bytes each)

__global__ __launch_bounds__(256) void k_class_patterns(const unsigned long long* __restrict__ tableA,
                                                        const int* __restrict__ Ap, const int* __restrict__ Aj,
                                                        const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                        int4* __restrict__ classInfo, unsigned* __restrict__ classMap,
                                                        unsigned* __restrict__ classMapA,
                                                        int* __restrict__ classRel, int* __restrict__ classLane,
                                                        unsigned* __restrict__ classRing,   // nullptr: not wanted
                                                        int* __restrict__ stats,
                                                        const int* __restrict__ rowCount = nullptr)   // round 6 (bhs_class_mix.hip.h): rows per class; a class of a few rows is not worked out (z = -1: its rows are irregular)
{
    __shared__ int keys[kClassMaxP], srt[kClassMaxP], pk[kClassMaxP];
    __shared__ int sIncl[kClassMaxRow], sB0[kClassMaxRow], scan[256];
    __shared__ int sEnt[kClassMaxRow], sGeo[4], sChain[kClassMaxRow];
    __shared__ unsigned char sUnitPos[4 * 64], sUnitOwner[4 * 64];          // a 16-byte unit of the slab -> its place (in units), and back
    const int tid = threadIdx.x, s = blockIdx.x;
    const unsigned long long v = tableA[s];
    if (v == kClassEmpty) {
        if (tid == 0) classInfo[s] = make_int4(0, 0, 0, -1);
        return;
    }
    const int rep = (int)(unsigned)v;
    if (rowCount != nullptr && rowCount[s] < kClassMixMinRows) {
        if (tid == 0) classInfo[s] = make_int4(0, 0, -1, rep);
        return;
    }
    const int a0 = Ap[rep], nA = Ap[rep + 1] - a0;               // <= kClassMaxRowBig (k_class_rows)
    if (nA > kClassMaxRow) {                                       // a big class: k_class_patterns_big's (z = -2: not done yet)
        if (tid == 0) classInfo[s] = make_int4(nA, 0, -2, rep);
        return;
    }
    if (tid < 64) {
        int b0 = 0, len = 0;
        if (tid < nA) {
            const int j = Aj[a0 + tid];
            b0 = Bp[j];
            len = Bp[j + 1] - b0;
        }
        const int incl = wave_incl_scan_dpp(len);
        sIncl[tid] = incl;
        sB0[tid] = b0 - (incl - len);
        int longest = len;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o, 64));
        if (tid == 0) scan[0] = longest;
    }
    __syncthreads();
    const int P = nA > 0 ? sIncl[nA - 1] : 0;
    if (P > kClassMaxP || scan[0] > kClassMaxRow) {                // (a B entry's number has 6 bits in these tables)
        if (tid == 0) classInfo[s] = make_int4(nA, P, -2, rep);
        return;
    }
    int N2 = 1;
    while (N2 < P) N2 <<= 1;
    auto bitonic = [&]() {                                           // ascending sort of srt[0, N2)
        for (int kk = 2; kk <= N2; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < N2; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const int x = srt[i], y = srt[ixj];
                        const bool up = (i & kk) == 0;
                        if ((x > y) == up) { srt[i] = y; srt[ixj] = x; }
                    }
                }
                __syncthreads();
            }
    };
    for (int p = tid; p < N2; p += 256) {
        int key = 0x7fffffff, code = 0;
        if (p < P) {
            int k = 0;
            while (sIncl[k] <= p) ++k;                           // <= 64 steps, once per class
            key = Bj[sB0[k] + p] - rep;
            code = k | ((p - (k ? sIncl[k - 1] : 0)) << 6);
        }
        keys[p] = key;
        srt[p] = key;
        pk[p] = code;
    }
    __syncthreads();
    // The distinct keys in ascending order -> ulist: ranked with a presence bitmap over [smallest, largest] where that
    // span fits (a grid's neighbours of neighbours: 66 K values on a 128^3 grid), else sorted.
    __shared__ int ulist[kClassMaxNnz];
    __shared__ unsigned bits[kClassSpanWords];
    int lo = 0x7fffffff, hi = -0x7fffffff - 1;
    for (int p = tid; p < P; p += 256) { lo = min(lo, keys[p]); hi = max(hi, keys[p]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((tid & 63) == 0) { scan[tid >> 6] = lo; scan[4 + (tid >> 6)] = hi; }
    __syncthreads();
    lo = min(min(scan[0], scan[1]), min(scan[2], scan[3]));
    hi = max(max(scan[4], scan[5]), max(scan[6], scan[7]));
    __syncthreads();
    const long long span = P > 0 ? (long long)hi - lo + 1 : 0;
    int nnz = 0;
    if (span <= (long long)kClassSpanWords * 32) {
        const int nw = (int)((span + 31) >> 5), wpt = (nw + 255) / 256;
        for (int w = tid; w < nw; w += 256) bits[w] = 0u;
        __syncthreads();
        for (int p = tid; p < P; p += 256) atomicOr(&bits[(keys[p] - lo) >> 5], 1u << ((keys[p] - lo) & 31));
        __syncthreads();
        int mine = 0;
        for (int w = tid * wpt; w < (tid + 1) * wpt && w < nw; ++w) mine += __popc(bits[w]);
        scan[tid] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        nnz = scan[255];
        if (nnz > kClassMaxNnz) {
            if (tid == 0) { classInfo[s] = make_int4(nA, P, -1, rep); atomicOr(&stats[CS_FLAGS], 2); }
            return;
        }
        int at = scan[tid] - mine;
        for (int w = tid * wpt; w < (tid + 1) * wpt && w < nw; ++w)
            for (unsigned mm = bits[w]; mm; mm &= mm - 1) ulist[at++] = lo + (w << 5) + (__ffs((int)mm) - 1);
    } else {
        bitonic();
        // distinct keys: thread t owns srt[t * per .. (t + 1) * per)
        const int per = (N2 + 255) / 256;
        int heads = 0;
        for (int i = tid * per; i < (tid + 1) * per && i < P; ++i) heads += (i == 0 || srt[i] != srt[i - 1]) ? 1 : 0;
        scan[tid] = heads;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        nnz = scan[255];
        if (nnz > kClassMaxNnz) {
            if (tid == 0) { classInfo[s] = make_int4(nA, P, -1, rep); atomicOr(&stats[CS_FLAGS], 2); }
            return;
        }
        int at = scan[tid] - heads;
        for (int i = tid * per; i < (tid + 1) * per && i < P; ++i)
            if (i == 0 || srt[i] != srt[i - 1]) ulist[at++] = srt[i];
    }
    __syncthreads();
    for (int e = tid; e < nnz; e += 256) classRel[(size_t)s * kClassMaxNnz + e] = ulist[e];
    // second sort: the products by (position, product number) -- position << 10 | product
    for (int p = tid; p < N2; p += 256) {
        int sv = 0x7fffffff;
        if (p < P) {
            const int key = keys[p];
            int l = 0, r = nnz - 1;
            while (l < r) { const int mid = (l + r) >> 1; if (ulist[mid] < key) l = mid + 1; else r = mid; }
            sv = (l << 10) | p;
            classMapA[(size_t)s * kClassMaxP + p] = (unsigned)pk[p] | ((unsigned)l << 16);   // k_class_numeric_atomic's form
        }
        srt[p] = sv;
    }
    __syncthreads();
    // The products in the order (entry of C, product number).  Round 5: not a second bitonic sort (55 stages, a barrier
    // each: most of this kernel's 49 us): an entry's products come from DIFFERENT entries of the A row (a row of B holds a
    // column once), so a 64-bit mask per entry of C says which -- the entry's products are as many as its mask has bits,
    // a product's rank inside its entry is the number of bits below its A entry's, and the entries' starts are one scan.
    // (A row of B with a column twice breaks the premise: the counts then do not add up to P, and the sort runs.)
    {
        unsigned long long* emask = reinterpret_cast<unsigned long long*>(bits);          // [kClassMaxNnz] (the bitmap's words are free now)
        int* estart = reinterpret_cast<int*>(bits) + 2 * kClassMaxNnz;                    // [kClassMaxNnz]
        for (int e = tid; e < kClassMaxNnz; e += 256) emask[e] = 0ull;
        __syncthreads();
        for (int p = tid; p < P; p += 256) atomicOr(&emask[srt[p] >> 10], 1ull << (pk[p] & 63));
        __syncthreads();
        const int c0 = 2 * tid < nnz ? __popcll(emask[2 * tid]) : 0, c1 = 2 * tid + 1 < nnz ? __popcll(emask[2 * tid + 1]) : 0;
        scan[tid] = c0 + c1;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        const int excl = scan[tid] - c0 - c1;
        estart[2 * tid] = excl;
        estart[2 * tid + 1] = excl + c0;
        const bool unique = scan[255] == P;
        __syncthreads();
        if (unique) {
            for (int p = tid; p < P; p += 256) {
                const int l = srt[p] >> 10, kA = pk[p] & 63;
                keys[estart[l] + __popcll(emask[l] & ((1ull << kA) - 1ull))] = srt[p];
            }
            __syncthreads();
            for (int p = tid; p < N2; p += 256) srt[p] = p < P ? keys[p] : 0x7fffffff;
            __syncthreads();
        } else {
            bitonic();
        }
    }
    const int U = (P + 63) >> 6;
    const unsigned spare = (unsigned)nnz * (unsigned)sizeof(acc_t);           // the slot behind the row's entries
    for (int idx = tid; idx < kClassMaxSteps * 64; idx += 256) {
        const int L = idx & 63, u = (idx >> 6) - (kClassMaxSteps - U);         // (negative: not a step of this class)
        const int first = L * U, last = min(P, first + U) - 1;      // ranks this lane holds
        const int r = last - (U - 1 - u);                           // (right-aligned: idle steps first)
        unsigned d = spare | 0x80000000u;
        if (u >= 0 && r >= first) {
            const int sv = srt[r], l = sv >> 10;
            const bool start = r == first || (srt[r - 1] >> 10) != l;
            const bool goesOn = l == (srt[last] >> 10) && last + 1 < P && (srt[last + 1] >> 10) == l;
            const unsigned code = (unsigned)pk[sv & 1023];          // A entry | B entry << 6
            d = (goesOn ? spare : (unsigned)l * (unsigned)sizeof(acc_t)) | ((code & 63u) * (unsigned)sizeof(acc_t)) << 16 |
                (code >> 6) << 25 | (start ? 0x80000000u : 0u);
        }
        classMap[(size_t)s * kClassMaxP + idx] = d;
    }
    if (tid < 64) {
        const int first = tid * U, last = min(P, first + U) - 1;
        int tail = -1;
        if (first <= last && last + 1 < P && (srt[last + 1] >> 10) == (srt[last] >> 10)) tail = srt[last] >> 10;
        classLane[(size_t)s * kClassLaneInts + tid] = tail;
        // chains of the A row (stored order): entry k opens one unless its column follows entry k - 1's and its B row is
        // as long as that one's
        const int col = tid < nA ? Aj[a0 + tid] : 0;
        const int myLen = tid < nA ? sIncl[tid] - (tid ? sIncl[tid - 1] : 0) : 0;
        const int prev = __shfl_up(col, 1, 64), prevLen = __shfl_up(myLen, 1, 64);
        const bool opens = tid < nA && (tid == 0 || col != prev + 1 || myLen != prevLen);
        const unsigned long long om = __ballot(opens);
        const unsigned long long upTo = om & ((2ull << tid) - 1ull);
        const int chain = __popcll(upTo) - 1;                                  // chain of entry tid
        const int opened = tid < nA ? 63 - __clzll((long long)upTo) : 0;       // ... and the entry that opened it
        const int nCh = __popcll(om);
        int kf = 0, len = 0, lc = 0;
        if (tid < nCh) {                                                       // as chain tid: its first entry, entries, row length
            unsigned long long rest = om;
            for (int i = 0; i < tid; ++i) rest &= rest - 1;                   // (<= 63 steps, once per class)
            kf = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            len = (rest ? __ffsll((long long)rest) - 1 : nA) - kf;
            lc = sIncl[kf] - (kf ? sIncl[kf - 1] : 0);
        }
        const int lpad = (lc + kClassEpl - 1) & ~(kClassEpl - 1);
        const int place = wave_incl_scan_dpp(lpad) - lpad;                     // of the chain's row in a slab
        const int slab = __builtin_amdgcn_readlane(place + lpad, 63);
        int maxLen = len;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxLen = max(maxLen, __shfl_xor(maxLen, o, 64));
        const int myPlace = __shfl(place, tid < nA ? chain : 0, 64);
        classLane[(size_t)s * kClassLaneInts + 64 + tid] = tid < nA ? (myPlace | ((tid - opened) << 16) | (myLen << 24)) : 0;
        sEnt[tid] = tid < nA ? (myPlace | ((tid - opened) << 16)) : 0;
        classLane[(size_t)s * kClassLaneInts + 128 + tid] = tid < nCh ? (kf | (len << 6) | (lc << 13) | (place << 20)) : 0;
        sChain[tid] = tid < nCh ? (kf | (lc << 8) | (place << 16)) : 0;
        for (int j = 0; j < kClassMaxLoads; ++j) {                             // this lane's pieces of a slab
            const int x = (j * 64 + tid) * kClassEpl;
            int c = 0;
            for (int cc = 1; cc < nCh; ++cc) c += x >= __shfl(place, cc, 64) ? 1 : 0;
            const int o = x - __shfl(place, c, 64), rowLen = __shfl(lc, c, 64), kFirst = __shfl(kf, c, 64);
            const bool piece = x < slab && o < rowLen;
            classLane[(size_t)s * kClassLaneInts + 256 + j * 64 + tid] = piece ? (o | (rowLen << 8) | (kFirst << 16)) : (kFirst << 16);
        }
        if (tid == 0) {
            classLane[(size_t)s * kClassLaneInts + 192] = nCh | (maxLen << 8) | (slab << 16);
            // the ring of the ring kernel: (entries of the longest chain + 1) slabs; 4 x 64 lanes x 16 bytes per slab at most
            atomicMax(&stats[CS_MAXRING], slab <= kClassMaxLoads * 64 * kClassEpl ? (maxLen + 1) * slab : 0x7fffffff);
            atomicMax(&stats[CS_MAXSLAB], slab);
            sGeo[0] = slab;
            sGeo[1] = maxLen;
            sGeo[2] = nCh;
        }
        int mx = myLen;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        if (tid == 0) atomicMax(&stats[CS_MAXLB], mx);
    }
    // classRing (bhs_class_ring.hip.h): per (step, lane) the product's place in the ring at the first row of a
    // stretch (byte offset, bits 0-15), its A entry (16-22; nA: a step without a product -- it multiplies by the zero behind
    // the row's A values -- and reads where the lane's last product reads) and bit 31: an entry of C ends here, in this lane
    // (its last product of the lane, and the entry does not go on in the next lane).  A lane's products come first, its idle
    // steps last; the class's U steps are the last of the kClassMaxSteps stored ones.  Behind them a word per lane: the
    // first entry that ends in it (bits 0-15) and 1 + the entry its last running sum is added to (16-31; 0: none).
    if (classRing != nullptr) {
        __syncthreads();
        // The slab's layout in LDS.  An LDS-direct load puts lane L's 16 bytes at (load's base + 16 L) whatever address the
        // lane reads, so WHICH 16-byte unit of the chains' B rows lands where is free.  The 64 lanes of a step read 64 values
        // scattered over the ring -- 3.3 lanes per bank pair with the chains side by side, 77 LDS cycles per row where 24
        // would do (poisson27pt, simulated and counted) --, and which units are read TOGETHER is known here: two units read in
        // the same step by the same half of the wave, at the same place inside the unit, should not share a bank group
        // (unit position mod 16).  Units are coloured greedily, in their natural order, with the 16 bank groups (at most
        // ceil(units / 16) units each) by the weight of such meetings with the units already placed: 77 -> 50 cycles.
        const int slab0 = sGeo[0], maxLen = sGeo[1], nChains = sGeo[2];
        const int nUnits = slab0 / kClassEpl;
        const int cap = (nUnits + 15) / 16, slabC = 16 * cap * kClassEpl;      // units per bank group, the coloured slab
        const bool fitsBefore = (maxLen + 1) * slab0 * (int)sizeof(value_t) <= kClassRingBudget;
        const bool colour = BHS_CLS_COLOUR && nUnits > 16 && nUnits <= kClassColourUnits && U >= 1 &&
                            (!fitsBefore || (maxLen + 1) * slabC * (int)sizeof(value_t) <= kClassRingBudget);
        auto unit_of = [&](int r, int& sub) {                        // product of rank r: its unit (chains side by side) and its place in it
            const unsigned code = (unsigned)pk[srt[r] & 1023];
            const int v = (sEnt[code & 63u] & 0xFFFF) + (int)(code >> 6);
            sub = v % kClassEpl;
            return v / kClassEpl;
        };
        for (int q = tid; q < 4 * 64; q += 256) { sUnitPos[q] = (unsigned char)q; sUnitOwner[q] = (unsigned char)q; }
        __syncthreads();
        if (colour) {
            unsigned* W = bits;                                       // weights, a byte per pair of units: W[u1 * 32 + u2 / 4], byte u2 % 4 (the bitmap's 16 KB)
            int* cost = scan;                                         // per bank group
            for (int i = tid; i < kClassColourUnits * 32; i += 256) W[i] = 0u;
            __syncthreads();
            // every (step, half of the wave): its <= 32 products, pair by pair
            const int nGroups = 2 * U;
            for (int w = tid; w < nGroups * 32 * 32; w += 256) {
                const int gidx = w >> 10, i = (w >> 5) & 31, j = w & 31;
                if (i >= j) continue;
                const int step = gidx >> 1, half = gidx & 1;
                const int r1 = (half * 32 + i) * U + step, r2 = (half * 32 + j) * U + step;
                if (r1 >= P || r2 >= P) continue;
                int s1, s2;
                const int u1 = unit_of(r1, s1), u2 = unit_of(r2, s2);
                if (s1 != s2 || u1 == u2) continue;
                atomicAdd(&W[u1 * 32 + (u2 >> 2)], 1u << (8 * (u2 & 3)));
                atomicAdd(&W[u2 * 32 + (u1 >> 2)], 1u << (8 * (u1 & 3)));
            }
            __syncthreads();
            __shared__ int sCount[16];
            if (tid < 16) sCount[tid] = 0;
            for (int v = 0; v < nUnits; ++v) {
                if (tid < 16) cost[tid] = 0;
                __syncthreads();
                if (tid < v) {
                    const int wv = (int)((W[v * 32 + (tid >> 2)] >> (8 * (tid & 3))) & 255u);
                    if (wv) atomicAdd(&cost[sUnitPos[tid] & 15], wv);
                }
                __syncthreads();
                if (tid == 0) {
                    int best = -1, bw = 0x7fffffff;
                    for (int c = 0; c < 16; ++c)
                        if (sCount[c] < cap && cost[c] < bw) { bw = cost[c]; best = c; }
                    sUnitPos[v] = (unsigned char)(best + 16 * sCount[best]);
                    sCount[best]++;
                }
                __syncthreads();
            }
            for (int q = tid; q < 4 * 64; q += 256) sUnitOwner[q] = 255;
            __syncthreads();
            if (tid < nUnits) sUnitOwner[sUnitPos[tid]] = (unsigned char)tid;
            __syncthreads();
        }
        const int slabP = colour ? slabC : slab0;                    // values per slot of the ring
        if (tid == 0) {
            // the ring of (longest chain + 1) slabs where it fits that kernel's budget; a class beyond it keeps (longest
            // chain) slabs -- what ONE row needs -- and starts every row as a stretch
            if (slab0 > kClassMaxLoads * 64 * kClassEpl) atomicMax(&stats[CS_RINGONE], 0x7fffffff);
            else if ((maxLen + 1) * slabP * (int)sizeof(value_t) <= kClassRingBudget) atomicMax(&stats[CS_RINGFULL], (maxLen + 1) * slabP);
            else atomicMax(&stats[CS_RINGONE], maxLen * slabP);
            classRing[(size_t)s * kClassRingStride + kClassRingDma + 4 * 64] = (unsigned)slabP;
        }
        // lane L's 16 bytes of the j-th LDS-direct load of a slab (bhs_class_ring.hip.h's copy of classLane[256 ..]): the
        // unit that lives at position 64 j + L -- place in its chain's B row (bits 0-7), that row's length (8-15; 0: no
        // load), the chain's first A entry (16-21)
        for (int q = tid; q < kClassMaxLoads * 64; q += 256) {
            const int un = colour ? (int)sUnitOwner[q] : q;
            unsigned word = 0;
            // (255: a coloured position without a unit.  Uncoloured, position 255 IS unit 255 -- the last 16 bytes of a slab of
            // exactly 512 values, 32 B rows of 15 or 16 entries, say: until round 6's soak found it, that unit was never loaded
            // and the products that read it were products with whatever the ring held there)
            if (!(colour && un == 255) && un * kClassEpl < slab0) {
                const int x = un * kClassEpl;
                int c = 0;
                for (int cc = 1; cc < nChains; ++cc) c += x >= (sChain[cc] >> 16) ? 1 : 0;
                const int o = x - (sChain[c] >> 16), rowLen = (sChain[c] >> 8) & 255, kFirst = sChain[c] & 255;
                word = o < rowLen ? (unsigned)(o | (rowLen << 8) | (kFirst << 16)) : (unsigned)(kFirst << 16);
            }
            classRing[(size_t)s * kClassRingStride + kClassRingDma + q] = word;
        }
        auto place_of = [&](int r) {
            const unsigned code = (unsigned)pk[srt[r] & 1023];
            const int e = sEnt[code & 63u];
            int sub;
            const int un = unit_of(r, sub);
            return (unsigned)(((e >> 16) * slabP + (int)sUnitPos[un] * kClassEpl + sub) * (int)sizeof(value_t));
        };
        for (int idx = tid; idx < kClassMaxSteps * 64; idx += 256) {
            const int L = idx & 63, j = (idx >> 6) - (kClassMaxSteps - U);
            const int first = L * U, last = min(P, first + U) - 1, r = first + j;
            unsigned w;
            if (j >= 0 && r <= last) {
                const int sv = srt[r], l = sv >> 10;
                const bool lastOfEntry = r == last || (srt[r + 1] >> 10) != l;
                const bool goesOn = l == (srt[last] >> 10) && last + 1 < P && (srt[last + 1] >> 10) == l;
                w = place_of(r) | ((unsigned)(pk[sv & 1023] & 63) << 16) | ((lastOfEntry && !goesOn) ? 0x80000000u : 0u);
            } else {
                w = (P > 0 ? place_of(first <= last ? last : 0) : 0u) | ((unsigned)nA << 16);
            }
            classRing[(size_t)s * kClassRingStride + idx] = w;
        }
        if (tid < 64) {
            const int first = tid * U, last = min(P, first + U) - 1;
            int slot0 = 0, tl = -1;
            if (first <= last) {
                slot0 = srt[first] >> 10;
                if (last + 1 < P && (srt[last + 1] >> 10) == (srt[last] >> 10)) tl = srt[last] >> 10;
            }
            classRing[(size_t)s * kClassRingStride + kClassMaxP + tid] = (unsigned)slot0 | ((unsigned)(tl + 1) << 16);
        }
        // ... and the relative columns as the kernel's lanes write them: lane L of the w-th pair of store instructions takes
        // the entries e0, e0 + 1 with e0 = min(2 (64 w + L), nnz - 2) -- an odd row's last lane overlaps its neighbour
        for (int q = tid; q < kClassMaxNnz / 2; q += 256) {
            const int e0 = max(0, min(2 * q, nnz - 2));
            const bool on = 2 * q < nnz;
            classRing[(size_t)s * kClassRingStride + kClassMaxP + 64 + 2 * q] = on ? (unsigned)ulist[e0] : 0u;
            classRing[(size_t)s * kClassRingStride + kClassMaxP + 64 + 2 * q + 1] = on && e0 + 1 < nnz ? (unsigned)ulist[e0 + 1] : 0u;
        }
    }
    if (tid == 0) {
        classInfo[s] = make_int4(nA, P, nnz, rep);
        atomicMax(&stats[CS_MAXP], P);
        atomicMax(&stats[CS_MAXNNZ], nnz);
        atomicMax(&stats[CS_MAXNA], nA);
        atomicAdd(&stats[CS_CLASSES], 1);
    }
}

// ---------------------------------------------------------------------------
// colIndC of rows from their classes alone (bhs_expand_class_columns_device: the values-only all-gatherv of the
// multi-GPU layer rebuilds the columns of the other ranks' blocks instead of receiving them).  One wave per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_class_expand_columns(int n, int row0, const int* __restrict__ classC,
                                                              const int4* __restrict__ classInfo, const int* __restrict__ classRel,
                                                              int relStride, const int* __restrict__ Cp, int* __restrict__ Cj)
{
    const int lane = threadIdx.x & 63;
    for (long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (long long)gridDim.x * 4) {
        const int cls = classC[i];
        if (cls < 0) continue;
        const int nnz = classInfo[cls].z;
        const long long out = Cp[i];
        const int* rel = classRel + (size_t)cls * relStride;
        for (int s = lane; s < nnz; s += 64) Cj[out + s] = rel[s] + row0 + (int)i;
    }
}

// ---------------------------------------------------------------------------
// rowPtrC in one pass: the class's entry count of every row, scanned (replaces k_class_counts + the three scan kernels
// of the general pipeline on the class path; create_C's host scan in the reference, bhsparse_cuda.h:2783-2811).
// Tiles of kClassScanTile (8192) rows in ticket order; a tile publishes its sum, then its first wave looks back over its
// predecessors 64 at a time -- a sum (flag 1) is added, a running total (flag 2) ends the walk -- and publishes its own
// running total.  state[tile] = flag << 62 | value, written and read with relaxed device-scope atomics (one word: no
// ordering between words is needed).  A tile only waits for tiles with smaller tickets, which are running.
// ---------------------------------------------------------------------------
constexpr int kClassScanBlock = 1024, kClassScanPer = 8, kClassScanTile = kClassScanBlock * kClassScanPer;   // (few tiles: short look-backs)
__global__ __launch_bounds__(kClassScanBlock) void k_class_scan(int m, const int* __restrict__ classC, const int4* __restrict__ classInfo,
                                                    int* __restrict__ Cp, unsigned long long* __restrict__ state,
                                                    long long* __restrict__ totalOut, int* __restrict__ stats,
                                                    bool mixed,       // round 6 (bhs_class_mix.hip.h): a row without a class has its count in Cp already (the general pipeline's symbolic kernels)
                                                    const int* __restrict__ Ap, const int* __restrict__ ub, BinSpec numSpec,
                                                    int* __restrict__ binCount)   // ... and is counted into its numeric bin here
{
    constexpr int NW = kClassScanBlock / 64;
    __shared__ int sTile, wsum[NW], hist[kMaxBins];
    if (mixed && threadIdx.x < kMaxBins) hist[threadIdx.x] = 0;       // (before the ticket's barrier)
    __shared__ long long sPrefix;
    __shared__ unsigned long long wprod[NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sTile = atomicAdd(&stats[CS_SCANTICKET], 1);
    __syncthreads();
    const int tile = sTile;
    const long long base = (long long)tile * kClassScanTile + (long long)tid * kClassScanPer;
    int v[kClassScanPer], mine = 0;
    unsigned long long products = 0;
    bool pending = false;
#pragma unroll
    for (int j = 0; j < kClassScanPer; ++j) {
        v[j] = 0;
        const int c = base + j < m ? classC[base + j] : -1;
        if (c >= 0 && c != kClassDummy) {
            const int4 ci = classInfo[c];
            v[j] = ci.z > 0 ? ci.z : 0;
            products += (unsigned long long)ci.y;
            pending = pending || ci.z == -2;
        } else if (mixed && c == kClassDummy) {
            v[j] = Cp[base + j];                                  // (read and overwritten by this thread alone)
            const int b = bin_of(numSpec, v[j], Ap[base + j + 1] - Ap[base + j], v[j], numSpec.hubMin > 0 ? ub[base + j] : 0);
            if (b > 0) atomicAdd(&hist[b], 1);
        }
        mine += v[j];
    }
    if (pending) atomicOr(&stats[CS_FLAGS], 2);                    // (a big class nobody worked out: stale hints)
    const int incl = wave_incl_scan_dpp(mine);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) products += __shfl_xor(products, o, 64);
    if (lane == 63) wsum[wv] = incl;
    if (lane == 0) wprod[wv] = products;
    __syncthreads();
    int before = 0, tileSum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { before += w < wv ? wsum[w] : 0; tileSum += wsum[w]; }
    constexpr unsigned long long kVal = (1ull << 62) - 1ull;
    if (wv == 0) {
        long long run = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&state[tile], (1ull << 62) | (unsigned long long)tileSum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int back = tile - 1; back >= 0;) {               // (wave-uniform)
                const int p = back - lane;
                unsigned long long st = 2ull << 62;               // (before the first tile: a running total of 0)
                if (p >= 0) st = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long unset = __ballot((st >> 62) == 0), total = __ballot((st >> 62) == 2);
                const int firstTotal = total ? __ffsll((long long)total) - 1 : 64;      // nearest predecessor with a running total
                if (unset & ((firstTotal < 64 ? (2ull << firstTotal) : 0ull) - 1ull)) continue;      // one of the nearer ones has not published: again
                long long part = lane <= firstTotal ? (long long)(st & kVal) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                run += part;
                if (firstTotal < 64) break;
                back -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&state[tile], (2ull << 62) | (unsigned long long)(run + tileSum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sPrefix = run;
        }
    }
    __syncthreads();
    long long at = sPrefix + before + incl - mine;
#pragma unroll
    for (int j = 0; j < kClassScanPer; ++j) {
        if (base + j < m) Cp[base + j] = (int)at;
        at += v[j];
    }
    if (base <= m - 1 && m - 1 < base + kClassScanPer) { Cp[m] = (int)at; *totalOut = at; }   // (the thread of the last row)
    if (tid == 0) {
        unsigned long long all = 0;
        for (int w = 0; w < NW; ++w) all += wprod[w];
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + CS_SUMS) + (tile % kClassSumSlots), all);
    }
    if (mixed && tid < kMaxBins && hist[tid]) atomicAdd(&binCount[tid], hist[tid]);     // (behind the barriers above)
}

// ---------------------------------------------------------------------------
// Speculative numeric launch (pipeline_symbolic, bhs_host_pipeline.inc.h).  Between k_class_scan and the numeric kernel the
// host reads the classes' figures back -- which kernel, how much LDS, how large C is -- and the device idles for that round
// trip (30 us of a 0.4 .. 1.6 ms multiply).  On a data set's second and later multiplies the host assumes the figures of the
// multiply before instead and launches at once; this kernel compares what it assumed with what this multiply's kernels found,
// and the numeric kernel returns before its first load unless word == 1 (the host sees the word at the end of the multiply
// and runs it again the slow way: borrowed arrays may change between multiplies, nothing kept from an earlier one is trusted).
// ---------------------------------------------------------------------------
struct ClassSpecKey { int cs[CS_INTS]; long long nnzC; };
__global__ __launch_bounds__(64) void k_class_spec_check(ClassSpecKey key, const int* __restrict__ stats, const long long* __restrict__ total,
                                                         const int* __restrict__ err, int* __restrict__ word)
{
    const int i = threadIdx.x;
    bool ok = *err == 0 && *total == key.nnzC;
    // (the products' partial sums and the scan's ticket are not decisions: CS_SUMS .. CS_RANGE and CS_SCANTICKET are left out)
    for (int j = i; j < CS_INTS; j += 64)
        if (!(j >= CS_SUMS && j < CS_RANGE) && j != CS_SCANTICKET) ok = ok && stats[j] == key.cs[j];
    const bool all = __ballot(!ok) == 0ull;
    if (i == 0) *word = all ? 1 : 2;
}

// ---------------------------------------------------------------------------
// Numeric pass, round 2's form (kept as the default until the workgroup form of bhs_class_wg.hip.h beats it
// everywhere; option class_numeric).  A wave takes runs of kClassRunA consecutive rows (neighbouring rows share their B rows: L1 / L2 hits,
// and the rows of C they write are adjacent); blocks are dealt to the XCDs so that each XCD's L2 sees one contiguous
// band of rows.
//   per run   the A entries of its rows are one contiguous stretch of colIndA / valA: loaded with coalesced loads,
//             their rowPtrB gathered, both parked in LDS -- two memory round trips for the whole run;
//   per row   every lane holds MAXU product triples {A entry, B entry, position} of the row's class in registers:
//             MAXU LDS reads of the B-row starts, MAXU loads of B's values (no predicates: all in flight at once),
//             MAXU multiplies and ds_add_f64 into the row's accumulators, then the row is written out with its
//             columns (class list + row number).  The class data stays in registers until a row of another class
//             comes along.
// What bounds it (ablations, poisson27pt 128^3, 2.3 ms): the LDS pipe -- 12 ds_add_f64 per row at ~33 cycles each
// whatever the number of active lanes, 24 reads of the staged A data.  Variants measured slower: one batch per A entry
// with plain read-fma-write accumulation (3.5 ms: 56 partial-lane LDS instructions per row), the same with four
// accumulator copies (3.9 ms), the same with atomics (4.0 ms).
// ---------------------------------------------------------------------------
constexpr unsigned kClassIdleA = 1u << 12;     // product triple of a lane without a product
// Stores of C that do not stay in the XCD's L2.  The kernel writes 3.3 GB that nobody reads again through a 4 MB L2 per
// XCD: with plain stores those lines push out the rows of B that the next rows of A need again (measured: 5.2 GB read
// per launch where 1.2 GB is compulsory).  Rounds 3-4 wrote through (sc1); round 5 measured the policies side by side on
// one box (poisson27pt 128^3, bhs_class_ring.hip.h's kernel): plain 1.58 ms, sc1 1.63, sc0 sc1 1.63, sc1 nt 1.80,
// nt 1.46 -- non-temporal stores it is (round 4's kernel: 1.61 -> 1.42).
#if BHS_CLS_STORE_SC1 == 2
#define BHS_CLS_STORE_POLICY " nt"
#elif BHS_CLS_STORE_SC1 == 3
#define BHS_CLS_STORE_POLICY " sc1 nt"
#elif BHS_CLS_STORE_SC1 == 4
#define BHS_CLS_STORE_POLICY " sc0 sc1"
#else
#define BHS_CLS_STORE_POLICY " sc1"
#endif
__device__ __forceinline__ void class_store_c(double* p, double v)
{
    if (BHS_CLS_STORE_SC1) asm volatile("global_store_dwordx2 %0, %1, off" BHS_CLS_STORE_POLICY ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
__device__ __forceinline__ void class_store_c(float* p, float v)
{
    if (BHS_CLS_STORE_SC1) asm volatile("global_store_dword %0, %1, off" BHS_CLS_STORE_POLICY ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
__device__ __forceinline__ void class_store_c(int* p, int v)
{
    if (BHS_CLS_STORE_SC1) asm volatile("global_store_dword %0, %1, off" BHS_CLS_STORE_POLICY ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
// ... the same with the address as a wave-uniform base (two SGPRs) and a 32-bit byte offset per lane: half the address
// words per store instruction, and no 64-bit address arithmetic per lane and row
__device__ __forceinline__ void class_store_c_at(double* base, unsigned off, double v)
{
    if (BHS_CLS_STORE_SC1) asm volatile("global_store_dwordx2 %0, %1, %2" BHS_CLS_STORE_POLICY ::"v"(off), "v"(v), "s"(base) : "memory");
    else *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v;
}
__device__ __forceinline__ void class_store_c_at(float* base, unsigned off, float v)
{
    if (BHS_CLS_STORE_SC1) asm volatile("global_store_dword %0, %1, %2" BHS_CLS_STORE_POLICY ::"v"(off), "v"(v), "s"(base) : "memory");
    else *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) = v;
}
__device__ __forceinline__ void class_store_c_at(int* base, unsigned off, int v)
{
    if (BHS_CLS_STORE_SC1) asm volatile("global_store_dword %0, %1, %2" BHS_CLS_STORE_POLICY ::"v"(off), "v"(v), "s"(base) : "memory");
    else *reinterpret_cast<int*>(reinterpret_cast<char*>(base) + off) = v;
}
// ... two neighbouring entries per lane: one instruction where two were
__device__ __forceinline__ void class_store_c2_at(double* base, unsigned off, double a, double b)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 v = {a, b};
    // (s_nop 1: a VMEM store of more than 64 bits reads its data over the next two wait states on gfx940+, and the compiler's
    // hazard recogniser does not see a store inside an asm statement: without it k_class_ring<16, 8, 4> had `v_or_b32 v22, 0x80, v26`
    // -- the next pair's entry number -- one s_or_b64 behind `global_store_dwordx4 v28, v[22:25]`, and lanes 12 .. 15 of every 16
    // stored that in the low word of their first value now and then.  Found by the mixed-mode soak, round 6.)
    asm volatile("global_store_dwordx4 %0, %1, %2" BHS_CLS_STORE_POLICY "\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void class_store_c2_at(float* base, unsigned off, float a, float b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, %2" BHS_CLS_STORE_POLICY ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void class_store_c2_at(int* base, unsigned off, int a, int b)
{
    typedef int i2 __attribute__((ext_vector_type(2)));
    const i2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, %2" BHS_CLS_STORE_POLICY ::"v"(off), "v"(v), "s"(base) : "memory");
}
constexpr int kClassRunA = 8;
constexpr int kClassWavesA = 4;

template <int MAXU, int MAXV, int SE>            // SE: 64-entry passes that stage a run's A entries (<= kClassRunA)
__global__ __launch_bounds__(64 * kClassWavesA) void k_class_numeric_atomic(
    int m, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const value_t* __restrict__ Bx, const int* __restrict__ classC,
    const int4* __restrict__ classInfo, const unsigned* __restrict__ classMap, const int* __restrict__ classRel,
    const int* __restrict__ Cp, int* __restrict__ Cj, value_t* __restrict__ Cx, int accStride, int stageCap,
    int rowBase)                                               // m, Ap, classC, Cp are views of the rows [rowBase, rowBase + m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // per wave: acc[accStride] and sAx[stageCap] doubles; then, after all waves' doubles, sBp[stageCap] ints per wave
    acc_t* acc = reinterpret_cast<acc_t*>(smemRaw) + (size_t)wv * (accStride + stageCap);
    acc_t* sAx = acc + accStride;
    int* sBp = reinterpret_cast<int*>(reinterpret_cast<acc_t*>(smemRaw) + (size_t)kClassWavesA * (accStride + stageCap)) + wv * stageCap;
    for (int i = lane; i < accStride; i += 64) acc[i] = 0.0;

    const int nRuns = (m + kClassRunA - 1) / kClassRunA;
    // XCD-aware: block b runs on XCD b % 8; XCD x takes the runs [x * perX, (x + 1) * perX)
    const int xcd = blockIdx.x & 7, perX = (nRuns + 7) / 8;
    const int wavesPerX = (gridDim.x >> 3) * kClassWavesA;
    const int wIdx = (blockIdx.x >> 3) * kClassWavesA + wv;

    int cur = -2, P = 0, nnz = 0;
    unsigned mp[MAXU];
    int rel[MAXV];
    for (int rr = wIdx; rr < perX; rr += wavesPerX) {
        const int run = xcd * perX + rr;
        if (run >= nRuns) break;
        const int row0 = run * kClassRunA;
        const int nr = min(kClassRunA, m - row0);
        // row pointers and classes of the run: one lane per row (lane nr holds the end of the last row)
        int myAp = 0, myCp = 0, myCls = -1;
        if (lane <= nr) { myAp = Ap[row0 + lane]; myCp = Cp[row0 + lane]; }
        if (lane < nr) myCls = classC[row0 + lane];
        const int base = __builtin_amdgcn_readlane(myAp, 0);
        const int nE = min(__builtin_amdgcn_readlane(myAp, nr) - base, stageCap - 64);
        {
            int aj[SE], bp[SE];
            acc_t ax[SE];
#pragma unroll
            for (int i = 0; i < SE; ++i) {
                aj[i] = -1;
                ax[i] = 0.0;
                if (i * 64 + lane < nE) { aj[i] = Aj[base + i * 64 + lane]; ax[i] = (acc_t)Ax[base + i * 64 + lane]; }
            }
#pragma unroll
            for (int i = 0; i < SE; ++i) { bp[i] = 0; if (aj[i] >= 0) bp[i] = Bp[aj[i]]; }
#pragma unroll
            for (int i = 0; i < SE; ++i)
                if (i * 64 < nE) { sAx[i * 64 + lane] = ax[i]; sBp[i * 64 + lane] = bp[i]; }
        }
        wave_sync();
        for (int t = 0; t < nr; ++t) {
            const int row = row0 + t;
            const int cls = __builtin_amdgcn_readlane(myCls, t);
            const int out = __builtin_amdgcn_readlane(myCp, t);
            const int off = __builtin_amdgcn_readlane(myAp, t) - base;
            if (cls < 0) continue;                                   // (cannot happen: such a multiply was sent back)
            if (cls != cur) {                                        // (wave-uniform)
                cur = cls;
                const int4 ci = classInfo[cls];
                P = __builtin_amdgcn_readfirstlane(ci.y);            // (uniform anyway: tells the compiler so)
                nnz = __builtin_amdgcn_readfirstlane(ci.z);
                // lanes past the class's last product get a harmless triple: entry 0 of B, the spare slot
#pragma unroll
                for (int u = 0; u < MAXU; ++u)
                    mp[u] = u * 64 + lane < P ? classMap[(size_t)cls * kClassMaxP + u * 64 + lane] : (kClassIdleA | ((unsigned)(accStride - 1) << 16));
#pragma unroll
                for (int v = 0; v < MAXV; ++v) rel[v] = v * 64 + lane < nnz ? classRel[(size_t)cls * kClassMaxNnz + v * 64 + lane] : 0;
                __builtin_amdgcn_s_waitcnt(kWaitVm0);                // (so that no later wait has to cover these loads)
            }
            // every lane, every batch: the loads carry no predicate, so the MAXU / BHS_CLS_PARTS of a part are in flight
            // at once (all MAXU at once keep 4 * MAXU registers live and cost two waves per SIMD)
            constexpr int UP = (MAXU + BHS_CLS_PARTS - 1) / BHS_CLS_PARTS;
#pragma unroll
            for (int u0 = 0; u0 < MAXU; u0 += UP) {
                int bpv[UP];
                acc_t bv[UP], axv[UP];
#pragma unroll
                for (int i = 0; i < UP; ++i) if (u0 + i < MAXU) bpv[i] = sBp[off + (int)(mp[u0 + i] & 63u)];
#pragma unroll
                for (int i = 0; i < UP; ++i) {
                    if (u0 + i < MAXU) {
                        const unsigned e = mp[u0 + i];
                        const int valid = (int)((e >> 12) & 1u) - 1;     // idle lane: 0, else all ones
                        const long long idx = (long long)((bpv[i] + (int)((e >> 6) & 63u)) & valid);
                        bv[i] = (acc_t)Bx[idx];
                    }
                }
#pragma unroll
                for (int i = 0; i < UP; ++i) if (u0 + i < MAXU) axv[i] = sAx[off + (int)(mp[u0 + i] & 63u)];
#pragma unroll
                for (int i = 0; i < UP; ++i) {
                    if (u0 + i < MAXU) {
                        unsafeAtomicAdd(&acc[mp[u0 + i] >> 16], axv[i] * bv[i]);
                    }
                }
                if (BHS_CLS_PARTS > 1) __builtin_amdgcn_sched_barrier(0);
            }
            wave_sync();
#pragma unroll
            for (int v = 0; v < MAXV; ++v) {
                const int s = v * 64 + lane;
                if (s < nnz) {
                    const acc_t val = acc[s];
                    acc[s] = 0.0;
                    class_store_c(&Cj[(long long)out + s], rel[v] + row + rowBase);
                    class_store_c(&Cx[(long long)out + s], (value_t)val);
                }
            }
            wave_sync();
        }
    }
}

}  // namespace bhs
