// bhs_row_wg.hip.h -- workgroup-per-row accumulators of the general pipeline: LDS hash tables (k_row_block), bitmap accumulators in HBM slots (k_row_spa) and in LDS (k_row_bitmap_lds) for long rows.  (Split from bhs_kernels.hip.h in round 4; bhsparse_hip.hip includes the parts in the old order.)
#pragma once

namespace bhs {

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
    int reverse,                                               // 1: queue taken from its end (longest rows there)
    const int* __restrict__ qnDev = nullptr)                  // the queue's length is this word on the device (k_row_wave_window's spill list)
{
    constexpr int BLOCK = kLdsBitmapBlock, CH = kLdsBitmapChunk, U = 4, NW = BLOCK / 64;
    if (qnDev != nullptr) qn = *qnDev;
    if (qn <= 0) return;
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
    // (round 6: the NEXT chunk's A entries and B row pointers -- two dependent loads -- are requested before this chunk's
    // products are walked: a row of a web graph's portal pages is 4 .. 10 chunks, two passes each, and every chunk used to
    // start with those two round trips in front of its first product)
    auto chunk_rows = [&](int ca, int a1, int& b0, int& len) {
        b0 = 0; len = 0;
        const int e = ca + tid;
        if (tid < CH && e < a1) {
            const int c = Aj[e];
            int2 be;
            __builtin_memcpy(&be, Bp + c, sizeof(be));
            b0 = be.x;
            len = be.y - be.x;
        }
    };
    auto expand = [&](int a0, int a1, auto&& f) {
        int b0n = 0, lenn = 0;
        chunk_rows(a0, a1, b0n, lenn);
        for (int ca = a0; ca < a1; ca += CH) {
            const int b0 = b0n, len = lenn;
            if (ca + CH < a1) chunk_rows(ca + CH, a1, b0n, lenn);   // (wave-uniform)
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

}  // namespace bhs
