// bhs_row_wave.hip.h -- the wavefront-per-row accumulator k_row_wave (symbolic and numeric), the workhorse of the general pipeline.  (Split from bhs_kernels.hip.h in round 4.)
#pragma once

namespace bhs {

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
            gen_store_c(&Cj[outBase + r], (int)(x[e] >> 32));
            gen_store_c(&Cx[outBase + r], (value_t)vals[(unsigned)x[e]]);
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
            gen_store_c(&Cj[outBase + r], col);
            gen_store_c(&Cx[outBase + r], (value_t)vals[slot]);
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
                    gen_store_c(&Cj[outBase + r], PACK32 ? (int)(e >> LOG2TS) : (int)((unsigned long long)e >> 32));
                    gen_store_c(&Cx[outBase + r], (value_t)sm.vals[PACK32 ? (unsigned)(e & ((1u << LOG2TS) - 1)) : (unsigned)e]);
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

}  // namespace bhs
