// bhs_rank.hip.h — the "pattern, then rank" pair of wave-per-row kernels (round 2).
//
// The hash kernels of bhs_kernels.hip.h walk every row's products twice and learn the same thing twice: the
// symbolic pass finds the row's column set and throws it away, the numeric pass hashes the same columns again
// (ds_cmpst_rtn per product), compacts the table and sorts the row.  Here the symbolic pass KEEPS what it learns:
//
//   k_sym_sorted   per row: the LDS hash table of the symbolic pass (CAS on the column), plus a list of the keys in
//                  the order they were claimed (one ballot per batch).  The list -- the row's distinct columns -- is
//                  sorted in registers (wave_flip_sort_u32) and written to a scratch array in HBM, kPatStride ints
//                  per row.  nnz of the row = length of the list.
//   k_num_rank     per row: the sorted columns are read back (coalesced), dropped into a small LDS lookup table
//                  {column, rank} -- one CAS per ENTRY of the row, not per product -- and every product finds its
//                  final position with one plain ds_read_b64 and adds its value into a dense fp64 image of the row
//                  (ds_add_f64).  No compare-and-swap per product, no compaction, no sort: the image IS the sorted
//                  row and leaves with coalesced stores, the columns straight from the registers they arrived in.
//                  (Replaces what ESC_bitonic_scan :1400-1518 and EM_mergepath :1902-2157 do for these rows.)
//
// Both kernels are latency-bound at any occupancy the register file allows (measured: time ~ 1 / resident waves), so
// k_num_rank is software-pipelined ACROSS rows: while the products of one window are looked up and added, the loads
// of the next window -- of the same row or of the next row, whose scan / A-entry table were prepared meanwhile --
// are already in flight in a second register set.
//
// Rows the fixed-size structures cannot hold (more than 64 entries in the A row, more than kRankRowMax entries in the
// row of C) are appended to an overflow queue by k_sym_sorted and take the workgroup-per-row hash kernels
// (k_row_block) in both passes.
#pragma once
#include "bhs_kernels.hip.h"
// measurement-only ablation mask of k_num_rank (tools/build_variants.sh): 1 no value adds, 2 no valB loads,
// 16 no stores of C
#ifndef BHS_RABL
#define BHS_RABL 0
#endif
#ifndef BHS_RANK_WAVES
#define BHS_RANK_WAVES 4
#endif
#ifndef BHS_RANK_W
#define BHS_RANK_W 6
#endif

namespace bhs {

constexpr int kPatStride = 256;               // ints per row in the pattern scratch array
constexpr int kRankRowMax = 256;              // entries per row of C the numeric image holds
constexpr int kSymTabLog2 = 10, kSymTab = 1 << kSymTabLog2;   // symbolic hash table (keys only)
constexpr int kPatOverflow = -2;              // pat[row * kPatStride]: the row went to the overflow queue

// XCD-aware persistent schedule of the wave kernels (see k_row_wave): queue position of the it-th row of this wave
struct WaveSched {
    int chunkLog2, chunk, xcd, lb, perX, nIt;
    __device__ __forceinline__ void init(int qn, int chunkLog2_, int wave, int WPB)
    {
        chunkLog2 = chunkLog2_;
        chunk = 1 << chunkLog2;
        xcd = blockIdx.x & 7;
        lb = (blockIdx.x >> 3) * WPB + wave;
        perX = (gridDim.x >> 3) * WPB;
        const int nChunks = (qn + chunk - 1) >> chunkLog2;
        int positions = 0;
        if (nChunks > xcd) {
            positions = ((nChunks - xcd + 7) >> 3) << chunkLog2;
            if (((nChunks - 1) & 7) == xcd) positions -= (nChunks << chunkLog2) - qn;
        }
        nIt = lb < positions ? (positions - lb + perX - 1) / perX : 0;
    }
    __device__ __forceinline__ int q_of(int it) const
    {
        const int t = lb + it * perX;
        return ((((t >> chunkLog2) << 3) + xcd) << chunkLog2) + (t & (chunk - 1));
    }
};

struct SymSortSmem {
    int keys[kSymTab];
    unsigned lst[kRankRowMax];                // distinct columns in the order they were claimed
    int sBase[64];
    alignas(8) unsigned marks[2 * kMaxBSym];
};

// ---------------------------------------------------------------------------
// Symbolic pass with pattern hand-off.  Rows come straight from rowPtrA (queue entry q = row q), as in the
// wave-first launches of k_row_wave: ub[row] and the product total are delivered on the side.
// ---------------------------------------------------------------------------
template <bool SMALLB>
__global__ __launch_bounds__(64 * kWavesPerBlock, 6) void k_sym_sorted(
    int qn, int chunkLog2, const int* __restrict__ Ap, const int* __restrict__ Aj,
    const int* __restrict__ Bp, const int* __restrict__ Bj,
    int* __restrict__ cntOut, int* __restrict__ ubOut, unsigned long long* __restrict__ ctSlots,
    int* __restrict__ pat, int4* __restrict__ ovfQueue, int* __restrict__ ovfCount)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = kWavesPerBlock;
    constexpr int MAXB = kMaxBSym, GRP = 4;
    constexpr int TS = kSymTab;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SymSortSmem& sm = reinterpret_cast<SymSortSmem*>(smem_raw)[wave];
    WaveSched sch;
    sch.init(qn, chunkLog2, wave, WPB);
    const int nIt = sch.nIt;
    if (nIt == 0) return;
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    auto load_desc = [&](int it_) {                      // (row, a0, a1): always a global load (see k_row_wave)
        const bool has = it_ < nIt;
        const int q = sch.q_of(has ? it_ : 0) + vzero;
        int2 aa;
        __builtin_memcpy(&aa, Ap + q, 8);
        int4 r = make_int4(q, aa.x, aa.y, 0);
        if (!has) r = make_int4(-1, 0, 0, 0);
        return r;
    };
    unsigned long long prodSum = 0;
    int4 dC = load_desc(0);
    int4 d1 = load_desc(1);
    int4 d2 = load_desc(2);
    int cC = 0, c1 = 0;
    if (lane < dC.z - dC.y) cC = Aj[dC.y + lane];
    if (lane < d1.z - d1.y) c1 = Aj[d1.y + lane];
    int2 beC = make_int2(0, 0);
    if (lane < dC.z - dC.y) __builtin_memcpy(&beC, Bp + cC, 8);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    for (int it = 0; it < nIt; ++it) {
        const int4 d3 = load_desc(it + 3);
        int c2 = 0;
        if (lane < d2.z - d2.y) c2 = Aj[d2.y + lane];
        int2 be1 = make_int2(0, 0);
        if (lane < d1.z - d1.y) __builtin_memcpy(&be1, Bp + c1, 8);
        const int row = dC.x, a0 = dC.y, a1 = dC.z;
        // ---- clear the table: 16 keys per lane
#pragma unroll
        for (int k = 0; k < TS / 256; ++k)
            *reinterpret_cast<int4*>(&sm.keys[k * 256 + lane * 4]) = make_int4(kEmpty, kEmpty, kEmpty, kEmpty);
        int nNew = 0;                                        // wave-uniform: distinct columns so far
        bool ovf = a1 - a0 > 64;                             // rows with more than 64 A entries: overflow queue
        const int b0 = beC.x;
        const int len = beC.y - beC.x;
        const int incl = wave_incl_scan_dpp(len);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        long long rowProducts = total;
        if (ovf) {                                           // (rare) the rest of the row still counts as products
            for (int ea = a0 + 64 + lane; __any(ea < a1); ea += 64) {
                int l2 = 0;
                if (ea < a1) {
                    int2 be;
                    __builtin_memcpy(&be, Bp + Aj[ea], 8);
                    l2 = be.y - be.x;
                }
                rowProducts += wave_sum_dpp(l2);
            }
            __builtin_amdgcn_s_waitcnt(kWaitVm0);
        }
        const int last = incl - 1;
        const unsigned long long nz = __ballot(len > 0);
        const int jc = mbcnt64(nz);
        wave_sync();
        if (len > 0) sm.sBase[jc] = b0 - (incl - len);
        int done = 0;
        for (int w0 = 0; w0 < total && !ovf; w0 += 64 * MAXB) {
            const int nb = (total - w0 + 63) >> 6;
            if (lane < 2 * MAXB) sm.marks[lane] = 0;
            wave_sync();
            const int rel = last - w0;
            if (len > 0 && rel >= 0 && rel < 64 * MAXB) atomicOr(&sm.marks[rel >> 5], 1u << (rel & 31));
            wave_sync();
            int col[MAXB];
            int cum = done;
#pragma unroll
            for (int u = 0; u < MAXB; ++u) {
                col[u] = kEmpty;
                if (u < nb) {
                    const unsigned long long mk = *reinterpret_cast<const unsigned long long*>(&sm.marks[2 * u]);
                    const int p = w0 + u * 64 + lane;
                    const int j = cum + mbcnt64(mk);
                    cum += __popcll(mk);
                    if (p < total) {
                        if constexpr (SMALLB) {
                            const unsigned idx32 = (unsigned)(sm.sBase[j] + p);
                            col[u] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(Bj) + (idx32 << 2));
                        } else {
                            col[u] = Bj[(long long)sm.sBase[j] + p];
                        }
                    }
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
                        hh[v] = hash_col(col[g + v], kSymTabLog2);
                        cur[v] = kEmpty;
                        if (col[g + v] != kEmpty) cur[v] = atomicCAS(&sm.keys[hh[v]], kEmpty, col[g + v]);
                    }
#pragma unroll
                    for (int v = 0; v < GRP; ++v) {
                        const int cv = col[g + v];
                        bool isNew = false;
                        if (cv != kEmpty) {
                            isNew = cur[v] == kEmpty;
                            if (!isNew && cur[v] != cv) {                   // collision: bounded linear probing
                                unsigned h = hh[v];
                                int left = TS;
                                for (;;) {
                                    h = (h + 1) & (TS - 1);
                                    const int c2 = atomicCAS(&sm.keys[h], kEmpty, cv);
                                    if (c2 == kEmpty) { isNew = true; break; }
                                    if (c2 == cv) break;
                                    if (--left == 0) { ovf = true; break; }
                                }
                            }
                        }
                        // claimed keys join the list in claim order (any order will do: the list is sorted below)
                        const unsigned long long nbal = __ballot(isNew);
                        if (isNew) {
                            const int pos = nNew + mbcnt64(nbal);
                            if (pos < kRankRowMax) sm.lst[pos] = (unsigned)cv;
                        }
                        nNew += __popcll(nbal);
                    }
                }
            }
            ovf = __any(ovf) || nNew > kRankRowMax;
            __builtin_amdgcn_s_waitcnt(kWaitVm0);
        }
        wave_sync();
        // ---- rotate the row pipeline (ahead of the stores, see k_row_wave)
        dC = d1; d1 = d2; d2 = d3;
        c1 = c2;
        beC = be1;
        asm volatile("" : "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(c1), "+v"(beC.x), "+v"(beC.y));
        int* prow = pat + (size_t)row * kPatStride;
        if (ovf) {
            // the fixed-size structures cannot hold this row: the workgroup-per-row kernels take it in both passes
            if (lane == 0) {
                const int slot = atomicAdd(ovfCount, 1);
                ovfQueue[slot] = make_int4(row, a0, a1, 0);
                prow[0] = kPatOverflow;
                cntOut[row] = 0;                             // (the overflow kernel writes the row's count)
                ubOut[row] = rowProducts > 0x7fffffffLL ? 0x7fffffff : (int)rowProducts;
            }
        } else {
            const int cnt = nNew;
            if (cnt <= 64) {
                unsigned x[1];
                x[0] = lane < cnt ? sm.lst[lane] : 0xffffffffu;
                wave_flip_sort_u32<1>(x, lane);
                if (lane < cnt) prow[lane] = (int)x[0];
            } else if (cnt <= 128) {
                unsigned x[2];
                const uint2 v = *reinterpret_cast<const uint2*>(&sm.lst[lane * 2]);
                x[0] = lane * 2 < cnt ? v.x : 0xffffffffu;
                x[1] = lane * 2 + 1 < cnt ? v.y : 0xffffffffu;
                wave_flip_sort_u32<2>(x, lane);
                if (lane * 2 + 1 < cnt) *reinterpret_cast<int2*>(&prow[lane * 2]) = make_int2((int)x[0], (int)x[1]);
                else if (lane * 2 < cnt) prow[lane * 2] = (int)x[0];
            } else {
                unsigned x[4];
                const uint4 v = *reinterpret_cast<const uint4*>(&sm.lst[lane * 4]);
                x[0] = lane * 4 < cnt ? v.x : 0xffffffffu;
                x[1] = lane * 4 + 1 < cnt ? v.y : 0xffffffffu;
                x[2] = lane * 4 + 2 < cnt ? v.z : 0xffffffffu;
                x[3] = lane * 4 + 3 < cnt ? v.w : 0xffffffffu;
                wave_flip_sort_u32<4>(x, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (lane * 4 + e < cnt) prow[lane * 4 + e] = (int)x[e];
            }
            if (lane == 0) {
                cntOut[row] = cnt;
                ubOut[row] = (int)rowProducts;
            }
        }
        prodSum += (unsigned long long)rowProducts;
        wave_sync();
    }
    if (lane == 0 && prodSum) atomicAdd(&ctSlots[blockIdx.x & 63], prodSum);
}

// ---------------------------------------------------------------------------
// Numeric pass by rank, software-pipelined across rows.
// ---------------------------------------------------------------------------
struct NumRankEnt {                           // per A entry of the row: where its B row sits, its value
    int base;
    int pad;
    acc_t av;
};

template <int RMAX>
struct NumRankSmem {
    uint2 tbl[2 * RMAX];                      // {column, rank}
    acc_t vals[RMAX];                         // the row of C, in column order
    NumRankEnt ent[2][64];                    // double-buffered: the next row is prepared while this one is accumulated
    alignas(8) unsigned marks[2][2 * BHS_RANK_W];   // per register set: "last product of an entry" marks of its window
    unsigned magic[64];                       // ceil(2^32 / L): product index -> A entry for uniform rows
};

template <int RMAX, bool SMALLB>
__global__ __launch_bounds__(64 * kWavesPerBlock, BHS_RANK_WAVES) void k_num_rank(
    int qn, int chunkLog2, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    const int* __restrict__ Cp, int* __restrict__ Cj, value_t* __restrict__ Cx,
    const int* __restrict__ pat, int* __restrict__ errFlag)
{
    static_assert(RMAX == 128 || RMAX == 256, "image sizes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = kWavesPerBlock;
    constexpr int W = BHS_RANK_W;                 // batches of 64 products per window
    constexpr int TS = 2 * RMAX, LOG2TS = RMAX == 128 ? 8 : 9;
    constexpr int NP = RMAX / 64;                 // pattern columns per lane
    constexpr int rabl = BHS_RABL;
    using Smem = NumRankSmem<RMAX>;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Smem& sm = reinterpret_cast<Smem*>(smem_raw)[wave];
    WaveSched sch;
    sch.init(qn, chunkLog2, wave, WPB);
    const int nIt = sch.nIt;
    if (nIt == 0) return;
    sm.magic[lane] = 0xffffffffu / (unsigned)(lane + 1) + 1u;

    // ---- row pipeline.  Everything wave-uniform about a row (its extent in A, its place in C) is read with SCALAR
    // loads from the queue position -- no VGPRs, and nothing that sits on vmcnt -- and the per-lane stages run deep
    // enough that whatever prepare() needs is OLDER than every load of the window being accumulated:
    //   iteration `it` loads: the column of this lane's A entry of row it+3; the extent of its B row (a gather through
    //   that column) and its value for row it+2; the sorted columns (pattern) of row it+1
    struct RowS { int row, a0, a1, c0, c1; };            // uniform: row, extent in A, extent in C
    auto row_s = [&](int it_) {
        RowS r;
        const bool has = it_ < nIt;
        const int q = sch.q_of(has ? it_ : 0);
        r.row = has ? q : -1;
        r.a0 = Ap[q]; r.a1 = has ? Ap[q + 1] : r.a0;
        r.c0 = Cp[q]; r.c1 = has ? Cp[q + 1] : r.c0;
        return r;
    };
    struct RowIn {                                       // what prepare() needs of a row
        RowS s;
        int2 be;                                         // per lane: extent of this lane's B row
        value_t av;                                      // per lane: value of this lane's A entry
    };
    auto load_c = [&](const RowS& s) {
        int c = 0;
        if (lane < s.a1 - s.a0) c = Aj[s.a0 + lane];
        return c;
    };
    auto load_be = [&](RowIn& r, int c) {                // r.s is set
        r.be = make_int2(0, 0);
        r.av = 0.0;
        if (lane < r.s.a1 - r.s.a0) { __builtin_memcpy(&r.be, Bp + c, 8); r.av = Ax[r.s.a0 + lane]; }
    };
    auto load_pat = [&](const RowS& s, int (&pc)[NP]) {
        const int row = s.row < 0 ? 0 : s.row;
#pragma unroll
        for (int k = 0; k < NP; ++k) pc[k] = pat[(size_t)row * kPatStride + k * 64 + lane];
    };
    // A prepared row: everything issue() / process() / finish_row() need; wave-uniform unless noted
    struct Prep {
        int total;                                       // products
        int cnt;                                         // entries of the row of C (0: nothing to do / overflow row)
        int outBase;
        unsigned magic;                                  // != 0: all B rows have the same length L, magic = ceil(2^32 / L)
        int last;                                        // per lane: flat index of this A entry's last product
        int len;                                         // per lane
        int pc[NP];                                      // per lane: sorted columns
    };
    // scan of the B row lengths, A-entry table into ent[eb]
    auto prepare = [&](const RowIn& r, int eb, Prep& P) {
        const int nA = r.s.a1 - r.s.a0;
        const int len = lane < nA ? r.be.y - r.be.x : 0;
        const int incl = wave_incl_scan_dpp(len);
        P.total = __builtin_amdgcn_readlane(incl, 63);
        P.last = incl - 1;
        P.len = len;
        const unsigned long long nz = __ballot(len > 0);
        const int jc = mbcnt64(nz);
        if (len > 0) {
            NumRankEnt e;
            e.base = r.be.x - (incl - len);
            e.pad = 0;
            e.av = (acc_t)r.av;
            sm.ent[eb][jc] = e;
        }
        const int L0 = __builtin_amdgcn_readfirstlane(len);
        const bool uni = L0 >= 2 && L0 <= 64 && __ballot(lane < nA && len != L0) == 0ull;
        P.magic = 0;
        if (uni) P.magic = sm.magic[L0 - 1];
        P.outBase = r.s.c0;
        P.cnt = r.s.c1 - r.s.c0;
        // rows k_sym_sorted sent to the overflow queue (exactly these: more than 64 A entries or more than
        // kRankRowMax entries) belong to k_row_block; rows past the end of the queue have no work either
        if (r.s.row < 0 || P.cnt > kRankRowMax || P.cnt > RMAX || nA > 64) { P.cnt = 0; P.total = 0; }
        wave_sync();
    };
    // A entry of product p of a window: uniform rows divide (one v_mul_hi), general rows count the "last product of
    // an entry" marks below p.  Computed when the loads are issued AND again when they are consumed (two to four VALU
    // instructions) rather than carried in W registers per set: the register file is what limits the pipeline depth.
    auto entry_of = [&](const Prep& P, int set, int u, int p, int& cum) {
        if (P.magic) return (int)__umulhi((unsigned)p, P.magic);
        const unsigned long long mk = *reinterpret_cast<const unsigned long long*>(&sm.marks[set][2 * u]);
        const int j = cum + mbcnt64(mk);
        cum += __popcll(mk);
        return j;
    };
    // loads of window [w0, w0 + 64 W) of a prepared row into register set `set`; always W pairs of loads (lanes and
    // batches beyond the row read element 0 of B: the load count per window is fixed, so the wait counts are exact)
    auto issue = [&](const Prep& P, int eb, int w0, int wEnd, int set, int (&col)[W], value_t (&bx)[W]) {
        int cum = 0;
        if (P.magic == 0 && wEnd > w0) {               // general rows: marks of "last product of an entry"
            if (lane < 2 * W) sm.marks[set][lane] = 0;
            wave_sync();
            const int rel = P.last - w0;
            if (P.len > 0 && rel >= 0 && rel < 64 * W) atomicOr(&sm.marks[set][rel >> 5], 1u << (rel & 31));
            wave_sync();
            cum = __popcll(__ballot(P.len > 0 && P.last < w0));   // entries completed before the window
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int p = w0 + u * 64 + lane;
            int j = entry_of(P, set, u, p, cum);
            const bool valid = p < wEnd;
            j = valid ? j : 0;
            const int base = sm.ent[eb][j].base;
            if constexpr (SMALLB) {
                const unsigned idx32 = valid ? (unsigned)(base + p) : 0u;
                col[u] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(Bj) + (idx32 << 2));
                if (rabl & 2) bx[u] = (value_t)1;
                else bx[u] = *reinterpret_cast<const value_t*>(reinterpret_cast<const char*>(Bx) +
                                                               idx32 * (unsigned)sizeof(value_t));
            } else {
                const long long idx = valid ? (long long)base + p : 0ll;
                col[u] = Bj[idx];
                bx[u] = (rabl & 2) ? (value_t)1 : Bx[idx];
            }
        }
    };
    // lookups and adds of one window
    auto process = [&](const Prep& P, int eb, int w0, int wEnd, int set, const int (&col)[W], const value_t (&bx)[W]) {
        int cum = 0;
        if (P.magic == 0 && wEnd > w0) cum = __popcll(__ballot(P.len > 0 && P.last < w0));
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (w0 + u * 64 < wEnd) {                     // (wave-uniform)
                const int p = w0 + u * 64 + lane;
                const int j = entry_of(P, set, u, p, cum);
                if (p < wEnd) {
                    const int cv = col[u];
                    unsigned h = hash_col(cv, LOG2TS);
                    uint2 e = sm.tbl[h];
                    if ((int)e.x != cv) {                 // collision chain (load <= 50 %)
                        int left = TS;
#pragma unroll 1
                        do {
                            h = (h + 1) & (TS - 1);
                            e = sm.tbl[h];
                        } while ((int)e.x != cv && --left > 0);
                        if (left == 0) { atomicOr(errFlag, 1); continue; }
                    }
                    const acc_t av = sm.ent[eb][j].av;
                    if (!(rabl & 1)) unsafeAtomicAdd(&sm.vals[e.y], av * (acc_t)bx[u]);
                    else asm volatile("" ::"v"(e.y), "v"(av), "v"(bx[u]));
                }
            }
        }
    };
    // table and image of a row: cleared, then one CAS per entry of the row
    auto begin_row = [&](const Prep& P) {
        if (P.cnt == 0) return;
#pragma unroll
        for (int k = 0; k < TS / 128; ++k)
            *reinterpret_cast<uint4*>(&sm.tbl[k * 128 + lane * 2]) = make_uint4(0xffffffffu, 0u, 0xffffffffu, 0u);
#pragma unroll
        for (int k = 0; k < RMAX / 128; ++k)
            *reinterpret_cast<double2*>(&sm.vals[k * 128 + lane * 2]) = make_double2(0.0, 0.0);
        wave_sync();
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int r = k * 64 + lane;
            if (r < P.cnt) {
                const int cv = P.pc[k];
                unsigned h = hash_col(cv, LOG2TS);
                int left = TS;
#pragma unroll 1
                for (;;) {
                    const unsigned old = atomicCAS(&sm.tbl[h].x, 0xffffffffu, (unsigned)cv);
                    if (old == 0xffffffffu) break;
                    h = (h + 1) & (TS - 1);
                    if (--left == 0) { atomicOr(errFlag, 1); break; }
                }
                sm.tbl[h].y = (unsigned)r;
            }
        }
        wave_sync();
    };
    auto finish_row = [&](const Prep& P) {
        if (P.cnt == 0) return;
        wave_sync();
        if (rabl & 16) return;
        const long long outBase = P.outBase;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int r = k * 64 + lane;
            if (r < P.cnt) {
                Cj[outBase + r] = P.pc[k];
                Cx[outBase + r] = (value_t)sm.vals[r];
            }
        }
        wave_sync();
    };

    // ---- prologue: fill the row pipeline
    RowIn rC, r1, r2;
    rC.s = row_s(0);
    r1.s = row_s(1);
    r2.s = row_s(2);
    const int cC = load_c(rC.s), c1 = load_c(r1.s);
    int c2 = load_c(r2.s);
    load_be(rC, cC);
    load_be(r1, c1);
    // Two register sets with FIXED roles: every trip of a row accumulates two windows, A (set 0) then B (set 1), of up
    // to W batches each -- a trip's batches are split evenly between them, so even a row with two batches has a window
    // in flight while the other is accumulated.  (Letting the sets swap roles by parity of a row's window count made
    // the compiler rotate the sets with register copies, each of which waits for the loads in flight.)
    int colA[W], colB[W];
    value_t bxA[W], bxB[W];
    struct Trip { int a0, a1, b0, b1; bool last; };      // product ranges [a0,a1) and [b0,b1) of windows A and B
    auto trip_of = [&](int total, int batch0) {
        const int nbatch = (total + 63) >> 6;
        const int k = nbatch - batch0 < 2 * W ? nbatch - batch0 : 2 * W;
        const int kA = (k + 1) >> 1;
        Trip t;
        t.a0 = batch0 * 64;
        t.a1 = (batch0 + kA) * 64 < total ? (batch0 + kA) * 64 : total;
        t.b0 = (batch0 + kA) * 64;
        t.b1 = (batch0 + k) * 64 < total ? (batch0 + k) * 64 : total;
        if (t.a1 < t.a0) t.a1 = t.a0;
        if (t.b1 < t.b0) t.b1 = t.b0;
        t.last = batch0 + 2 * W >= nbatch;
        return t;
    };
    Prep P;
    load_pat(rC.s, P.pc);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    int eb = 0;
    prepare(rC, eb, P);
    {
        const Trip t0 = trip_of(P.total, 0);
        issue(P, eb, t0.a0, t0.a1, 0, colA, bxA);
    }

    for (int it = 0; it < nIt; ++it) {
        // ---- stages of the rows behind this one (see above); all of them older than the next window's loads
        const RowS s3 = row_s(it + 3);
        const int c3 = load_c(s3);
        load_be(r2, c2);
        Prep Pn;
        load_pat(r1.s, Pn.pc);
        begin_row(P);
        for (int batch0 = 0;; batch0 += 2 * W) {
            const Trip t = trip_of(P.total, batch0);
            // window B's loads go out, window A (in flight since the previous half-step) is accumulated
            issue(P, eb, t.b0, t.b1, 1, colB, bxB);
            process(P, eb, t.a0, t.a1, 0, colA, bxA);
            // next window A -- of this row or, after its scan and entry table are prepared, of the next row -- goes
            // out, window B is accumulated
            if (!t.last) {
                const Trip tn = trip_of(P.total, batch0 + 2 * W);
                issue(P, eb, tn.a0, tn.a1, 0, colA, bxA);
            } else {
                prepare(r1, eb ^ 1, Pn);
                const Trip tn = trip_of(Pn.total, 0);
                issue(Pn, eb ^ 1, tn.a0, tn.a1, 0, colA, bxA);
            }
            process(P, eb, t.b0, t.b1, 1, colB, bxB);
            if (t.last) break;
        }
        finish_row(P);
        // ---- rotate
        P = Pn;
        eb ^= 1;
        r1 = r2;
        r2.s = s3;
        c2 = c3;
    }
}

}  // namespace bhs
