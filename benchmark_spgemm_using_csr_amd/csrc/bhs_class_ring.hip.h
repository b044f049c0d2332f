// bhs_class_ring.hip.h -- numeric pass by row classes, third form (round 5): the ring kernel with its per-product
// bookkeeping taken out of the row loop.  (Included after bhs_class_wg.hip.h; the classification and the per-class
// tables are bhs_class.hip.h's, the ring of slabs is bhs_class_wg.hip.h's -- read that header first.)
//
// What round 4's kernel (k_class_numeric) spends per product and row, counted in its ISA: 11 VALU instructions -- A
// address from the descriptor (2), slot address (1), restart of the running sum (compare + two selects: 3), the fma (1),
// the product's place in the ring moved on (add, compare, select, subtract: 4) -- and three LDS instructions; 200 VALU
// per row in all, at 156 VGPRs and 12.7 KB of LDS: 12 waves per CU.  Here:
//   the ring     is a power of two of bytes at LDS address 0 (slots and slab rounded up to powers of two): a product's
//                place moves on with an add and an and-or, and the bank of every read is the same in every row;
//                the places are CLASS constants (k_class_patterns works them out; a stretch starts at whatever phase
//                the ring is in), so no descriptor, chain table or B entry number is kept in registers;
//   A's values   of the row are copied to a fixed place in LDS (two stores per row, from the registers that hold the
//                chunk of valA at hand -- no staging area for a whole run): a product's A address is a class constant;
//   the sums     a lane's running sum goes to LDS only where an entry of C ENDS in it: per step a 64-bit mask in SGPRs
//                (one ballot per step and class) is moved into exec around {ds_write_b64 of the sum to the lane's slot
//                pointer, pointer += 8, sum = 0}; the restart needs no compare and no select, the slot no address
//                arithmetic, and the stores of a step touch ~16 bank pairs instead of 64 lanes' worth;
//   metadata     row pointers / classes of 64 rows per load (a lane per row), A's values in chunks of <= 128 entries
//                (whole rows), requested a chunk ahead right behind a row's vmcnt(0); colIndA and rowPtrB are read
//                where a stretch starts and nowhere else.
// Per row and product: fma, masked {mov, add}, add + and-or = 5 VALU; ~9.5 KB of LDS (8 KB ring, 1 KB slots, the row's
// A values) and <= 128 VGPRs: 16 waves per CU.
// (The ablation switches this kernel was measured with -- no slab loads, no stores, fixed operands ...: wrong results by
// design -- live in its lab copy, tools/lab/bhs_class_ring_lab.hip.h, outside the product's sources.)
#pragma once

namespace bhs {

#ifndef BHS_RING_LOAD_AUX      // cache policy of the slab loads (gfx940 cpol bits: 1 sc0, 2 nt, 16 sc1)
#define BHS_RING_LOAD_AUX 0
#endif
#ifndef BHS_RING_WAVES
#define BHS_RING_WAVES 4
#endif
// ... of the instance for up to 1024 products and 512 entries a row (MAXU 16, MAXV 8): it wants 175 VGPRs.  At 4 waves (128
// VGPRs, as until round 6) it kept 62 of them in scratch, reloaded row by row behind an s_waitcnt vmcnt(0) each: rows of 32 x 15
// entries (400 k rows) numeric_class 1.42 ms, 9 x 31 entries 0.93; at 2 waves 0.49 - 0.58 / 0.37; at 3 (168 VGPRs, 7 spilled;
// 13 KB of LDS a wave give 12 waves a CU anyway) 0.48 - 0.58 / 0.35.
#ifndef BHS_RING_WAVES_WIDE
#define BHS_RING_WAVES_WIDE 3
#endif

// {store the running sum to the lane's slot pointer, move the pointer on, restart the sum} in the lanes of `mask`
__device__ __forceinline__ void ring_end_step(unsigned long long mask, unsigned& slotPtr, acc_t& sum)
{
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "ds_write_b64 %[p], %[s]\n\t"
                 "v_add_u32 %[p], 8, %[p]\n\t"
                 "v_mov_b64 %[s], 0\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [p] "+v"(slotPtr), [s] "+v"(sum), [sv] "=&s"(saved)
                 : [m] "s"(mask)
                 : "memory");
}

template <int MAXU, int MAXV, int MAXJ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MAXU >= 16 ? BHS_RING_WAVES_WIDE : BHS_RING_WAVES, 8))) void k_class_ring(
    int m, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax, long long nnzA,
    const int* __restrict__ Bp, const value_t* __restrict__ Bx, long long nnzB, const int* __restrict__ classC,
    const int4* __restrict__ classInfo, const unsigned* __restrict__ classRing, const int* __restrict__ classRel,
    const int* __restrict__ classLane, const int* __restrict__ Cp, int* __restrict__ Cj, value_t* __restrict__ Cx,
    int ringBytes, int accStride, int rowBase, int superRows, int chunkRows,     // m, Ap, classC, Cp are views of the rows [rowBase, rowBase + m)
    const int* __restrict__ specWord,                                             // launched before the host saw this multiply's classes (k_class_spec_check): go on only if 1
    int* __restrict__ tickets)                                                    // 8 counters, zero at launch: the next super-run of every XCD's band (round 6)
{
    if (specWord != nullptr && *specWord != 1) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    // (the LDS addresses below are plain integers counted from 0: the dynamic area must be all the LDS this kernel has --
    // a compile-time constant, the test costs nothing)
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    const int lane = threadIdx.x;
    // the ring (ringBytes: a power of two, at LDS address 0 -- the kernel has no static LDS), acc[accStride], the row's A values
    value_t* ring = reinterpret_cast<value_t*>(smemRaw);
    acc_t* acc = reinterpret_cast<acc_t*>(smemRaw + ringBytes);
    acc_t* afix = acc + accStride;
    const unsigned accBase = (unsigned)ringBytes, afixBase = accBase + (unsigned)accStride * (unsigned)sizeof(acc_t);   // (LDS byte addresses: the dynamic area starts at 0)

    // super-runs of superRows consecutive rows (a grid line of A where it has such lines), XCD x takes the super-runs
    // [x * perX, (x + 1) * perX), block b runs on XCD b % 8 (bhs_class_wg.hip.h); a super-run is walked in blocks of <= 64
    // rows: a lane per row holds its row pointers and class
    const int nSuper = (m + superRows - 1) / superRows;
    const int xcd = blockIdx.x & 7, perX = (nSuper + 7) / 8;
    const int wavesPerX = gridDim.x >> 3, wIdx = blockIdx.x >> 3;
    const int BPS = (superRows + 63) >> 6;                       // blocks per super-run
    // A wave's first super-run is its own (wave w of its XCD: the w-th of the band); the ones after it are handed out by the
    // XCD's counter as the waves come for them (round 6).  Until round 5 wave w took the super-runs w, w + waves, w + 2 waves
    // ...: 25 600 grid lines over 4096 waves (poisson27pt 160^3) are 6.25 each -- a quarter of the waves walked seven lines
    // while the others idled behind their six, an eighth of the kernel's time; and a workgroup that starts late (mixed mode:
    // the irregular rows' kernels still hold its CU) made the whole launch late by as much.
    int srCur = wIdx;
    auto block_of = [&](int i, int& nr) -> int {                 // the i-th block of this wave: first row, rows (0: no such block).  Called for i = 0, 1, 2, ...
        nr = 0;
        if (i > 0 && i % BPS == 0) {
            if (tickets != nullptr) {
                int tk = 0;
                if (lane == 0) tk = atomicAdd(&tickets[xcd], 1);
                srCur = wavesPerX + __builtin_amdgcn_readfirstlane(tk);
            } else {
                srCur += wavesPerX;                              // (option "ring_dynamic" 0: wave w takes the super-runs w, w + waves, ... as until round 5)
            }
        }
        const long long sr = srCur;
        if (sr >= perX) return 0;
        const long long sup = (long long)xcd * perX + sr;
        if (sup >= nSuper) return 0;
        const long long r0 = sup * superRows + (long long)(i % BPS) * 64;
        const long long end = min((long long)m, (sup + 1) * superRows);
        nr = (int)max(0ll, min(64ll, end - r0));
        return (int)r0;
    };
    struct Ptrs { int ap, ap1, cp, cls; };                       // of row (first row of the block + lane): entries [ap, ap1) of A, first entry of C, class
    auto load_ptrs = [&](int row0, int nr) {
        Ptrs r{0, 0, 0, -1};
        if (lane < nr) { r.ap = Ap[row0 + lane]; r.ap1 = Ap[row0 + lane + 1]; r.cp = Cp[row0 + lane]; r.cls = classC[row0 + lane]; }
        return r;
    };
    // Where the chunk of A's values that starts at row s of a block ends (exclusive): G rows on, at the block's end -- or,
    // mixed mode (bhs_class_mix.hip.h), in front of and behind a row without a class (kClassDummy): such a row may be longer
    // than the 128 entries a chunk holds, so it is a chunk of its own (loaded, never looked at) and the rows of a class behind
    // it start a new one.  Scalar arithmetic on a ballot, where a chunk is taken up: nothing here lives across rows.
    auto chunk_end = [&](const Ptrs& p, int s, int nrB) -> int {
        const unsigned long long dummy = __ballot(p.cls == kClassDummy);      // (lanes beyond the block's rows carry -1)
        unsigned long long ends = dummy | (dummy >> 1) | (1ull << (nrB - 1));
        if (s + chunkRows - 1 < 64) ends |= 1ull << (s + chunkRows - 1);
        return s + __builtin_ctzll(ends >> s) + 1;
    };
    // a chunk of valA: the entries [base, base + nE), nE <= 128, two per lane (the last value of valA is not read as the
    // first half of a pair)
    auto load_chunk = [&](int base, int nE) {
        bhs_val2 ax = bhs_val2{(value_t)0, (value_t)0};
        const int e = 2 * lane;
        if (e < nE) {
            if ((long long)base + e + 1 < nnzA) ax = *reinterpret_cast<const bhs_val2*>(Ax + base + e);
            else ax.x = Ax[base + e];
        }
        return ax;
    };

    // ---- the class at hand
    int cur = -2, nA = 0, nnz = 0, need = 1, slab = 0;
    unsigned at[MAXU], aAddr[MAXU];                              // per product: its place in the ring (moves on by a slab per row), its A value's
    unsigned long long endMask[MAXU];                            // per step: the lanes in which an entry of C ends with this product
    unsigned slot0 = 0;                                          // LDS address of the slot of the first entry that ends in this lane
    int tail = -1;                                               // the entry this lane's last running sum is added to (-1: none)
    constexpr int MAXW = (MAXV + 1) / 2;                         // pairs of store instructions per row: a lane writes two neighbouring entries
    int relA[MAXW], relB[MAXW];                                  // their columns, relative to the row
    int dma[MAXJ];                                               // this lane's 16 bytes of each load instruction of a slab (bhs_class.hip.h: classLane[256 ..])
    unsigned src[MAXJ];                                          // ... where its next piece comes from (values of B are counted in int32: nnzB < 2^31)
    unsigned stepB = 0, wrapB = 0;                               // bytes per slab, bytes of this class's ring
    int slots = 1;                                               // slabs of the ring
    bool oneRow = false;                                         // the class's ring is beyond the budget: every row loads its own slabs
    int phase = 0, loadSlot = 0, lastRow = -2;
    acc_t bv[MAXU], av[MAXU];                                    // the operands of the row at hand's products (read from LDS a row ahead where the rows carry on)
    bool preIssued = false;                                      // ... they are the row at hand's already
    bool ringOK = false;
#if BHS_PHASES_CLS
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tPh = __builtin_readcyclecounter();
#endif

    // one slab: this lane's 16 bytes of each of its load instructions, if they are a piece of a B row
    auto request_slab = [&]() {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (j * 64 * kClassEpl < slab) {
                const bool piece = (dma[j] & 0xFF00) != 0;
                if (piece && (long long)src[j] + kClassEpl <= nnzB)
                    __builtin_amdgcn_global_load_lds((bhs_glb_void*)(Bx + src[j]), (bhs_lds_void*)(ring + loadSlot * slab + j * 64 * kClassEpl), 16, 0, BHS_RING_LOAD_AUX);
                else if (piece)                                      // (the last few values of valB: no 16-byte load past its end)
                    for (int e2 = 0; e2 < kClassEpl; ++e2)
                        if ((long long)src[j] + e2 < nnzB) ring[loadSlot * slab + (j * 64 + lane) * kClassEpl + e2] = Bx[src[j] + e2];
            }
        }
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) src[j] += (unsigned)((dma[j] >> 8) & 255);
        loadSlot = loadSlot + 1 == slots ? 0 : loadSlot + 1;
    };

    // ---- prologue: the first block's pointers, the first chunk of A's values, the next chunk's request
    int nr = 0, nrN = 0;
    int row0 = block_of(0, nr), row0N = 0;
    Ptrs pc = load_ptrs(row0, nr), pn{0, 0, 0, -1};
    bhs_val2 axCur = bhs_val2{(value_t)0, (value_t)0}, axNxt = axCur;
    int curBase = 0;                                             // first entry of the chunk at hand
    int cEnd = 0;                                                // the chunk at hand: rows [.., cEnd) of the block at hand
    bool nxtValid = false;
    int nEnd = 0;                                                // the requested chunk: rows [.., nEnd) of its block
    if (nr > 0) {
        __builtin_amdgcn_s_waitcnt(kWaitVm0);                    // (the first block's classes: chunk_end looks at them)
        cEnd = chunk_end(pc, 0, nr);
        curBase = __builtin_amdgcn_readlane(pc.ap, 0);
        axCur = load_chunk(curBase, __builtin_amdgcn_readlane(pc.ap1, cEnd - 1) - curBase);
        if (cEnd < nr) {
            nEnd = chunk_end(pc, cEnd, nr);
            const int b = __builtin_amdgcn_readlane(pc.ap, cEnd);
            axNxt = load_chunk(b, __builtin_amdgcn_readlane(pc.ap1, nEnd - 1) - b);
            nxtValid = true;
        }
    }
    // (Everything the prologue asked for is waited for HERE: left pending into the loops, the compiler's wait for the first
    // block's pointers lands in the header of the row loop -- an s_waitcnt vmcnt(0) at the top of every row, in front of
    // which the row before's slab request and stores have just been issued.)
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    for (int ib = 0; nr > 0; ++ib) {
        row0N = block_of(ib + 1, nrN);
        pn = load_ptrs(row0N, nrN);                              // (consumed behind this block's first vmcnt(0) at the earliest)
        BHS_TICK_CLS(0);
        for (int t = 0; t < nr; ++t) {
            // (>= 0.  Mixed mode, bhs_class_mix.hip.h: a row without a class carries kClassDummy, the class of no entries and no
            // products whose tables are all zeros -- the row loads nothing, multiplies nothing and stores nothing, it ends a
            // stretch like any change of class, and this loop has no branch for it: at <= 128 VGPRs and 33 spilled SGPRs one more
            // edge to the loop's latch cost 150 spilled VGPRs.)
            const int cls = __builtin_amdgcn_readlane(pc.cls, t);
            const int row = row0 + t;
            const int apT = __builtin_amdgcn_readlane(pc.ap, t);
            if (cls != cur) {                                        // (wave-uniform)
                cur = cls;
                ringOK = false;
                const int4 ci = classInfo[cls];
                unsigned w[MAXU];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) w[u] = classRing[(size_t)cls * kClassRingStride + (kClassMaxSteps - MAXU + u) * 64 + lane];
                const unsigned lw = classRing[(size_t)cls * kClassRingStride + kClassMaxP + lane];
                const int geoV = classLane[(size_t)cls * kClassLaneInts + 192];
                const unsigned slabV = classRing[(size_t)cls * kClassRingStride + kClassRingDma + 4 * 64];
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) dma[j] = (int)classRing[(size_t)cls * kClassRingStride + kClassRingDma + j * 64 + lane];
#pragma unroll
                for (int w2 = 0; w2 < MAXW; ++w2) {                  // (beyond the row: never stored)
                    const uint2 rp = *reinterpret_cast<const uint2*>(classRing + (size_t)cls * kClassRingStride + kClassMaxP + 64 + 2 * (w2 * 64 + lane));
                    relA[w2] = (int)rp.x;
                    relB[w2] = (int)rp.y;
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);                // (so that no later wait has to cover these loads)
                nA = __builtin_amdgcn_readfirstlane(ci.x);
                nnz = __builtin_amdgcn_readfirstlane(ci.z);
                const int geo = __builtin_amdgcn_readfirstlane(geoV);
                const int chainMax = (geo >> 8) & 255;               // entries of the longest chain
                slab = __builtin_amdgcn_readfirstlane((int)slabV);   // (k_class_patterns' layout of the slab: its units coloured over the bank groups)
                // the ring: (longest chain + 1) slabs -- a row's request replaces the slab only that row still needed -- where
                // that fits the budget; else what one row needs, loaded row by row (every row starts a stretch)
                oneRow = (chainMax + 1) * slab * (int)sizeof(value_t) > kClassRingBudget;
                need = oneRow ? chainMax : chainMax + 1;
                slots = max(need, 1);
                stepB = oneRow ? 0u : (unsigned)slab * (unsigned)sizeof(value_t);
                wrapB = (unsigned)slots * (unsigned)slab * (unsigned)sizeof(value_t);
                phase = 0;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    at[u] = w[u] & 0xFFFFu;
                    aAddr[u] = afixBase + ((w[u] >> 16) & 0x7Fu) * (unsigned)sizeof(acc_t);
                    endMask[u] = __ballot((int)w[u] < 0);
                }
                slot0 = accBase + (lw & 0xFFFFu) * (unsigned)sizeof(acc_t);
                tail = (int)(lw >> 16) - 1;
                if (lane == 0) afix[nA] = (acc_t)0;                  // what a step without a product multiplies by
                BHS_TICK_CLS(1);
            }
            if (!ringOK || oneRow || row != lastRow + 1) {           // a stretch begins: its first slabs, all at once, from the slot the ring is at
                const int ajv = lane < nA ? Aj[apT + lane] : -1;
                const int bo = ajv >= 0 ? Bp[ajv] : 0;
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) src[j] = (unsigned)(__shfl(bo, (dma[j] >> 16) & 63, 64) + (dma[j] & 255));
                loadSlot = phase;
                for (int s2 = 0; s2 < need; ++s2) request_slab();
                ringOK = true;
                BHS_TICK_CLS(2);
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
                wave_sync();
                BHS_TICK_CLS(3);
            }
            // The row's operands: A's values to their fixed place, then the LDS reads of both operands of every product, and
            // the products' places moved on by one slab around the ring.  Where the next row carries on (same class, same
            // chunk of A's values, same block) this was done a row ago, BEHIND that row's arithmetic and in front of its
            // write-out: the reads' latency -- 16 waves queue at the LDS pipe -- runs beside the slab request and the stores.
            // (The stores of A's values go through an address made of an integer, like every LDS access of the arithmetic:
            // for a store through a pointer derived from the shared array the compiler cannot rule out that a slab's
            // LDS-direct load writes the same bytes and puts an s_waitcnt vmcnt(0) in front of it.)
            auto issue_reads = [&](int apRow) {
                typedef __attribute__((address_space(3))) acc_t* lds_acc;
                typedef __attribute__((address_space(3))) const value_t* lds_val;
                typedef __attribute__((address_space(3))) const acc_t* lds_acc_c;
                const int k0 = 2 * lane - (apRow - curBase);
                const unsigned a0 = afixBase + (unsigned)k0 * (unsigned)sizeof(acc_t);
                if ((unsigned)k0 < (unsigned)nA) *(lds_acc)(size_t)a0 = (acc_t)axCur.x;
                if ((unsigned)(k0 + 1) < (unsigned)nA) *(lds_acc)(size_t)(a0 + (unsigned)sizeof(acc_t)) = (acc_t)axCur.y;
                wave_sync();
#pragma unroll
                for (int u = 0; u < MAXU; ++u) bv[u] = (acc_t)*(lds_val)(size_t)at[u];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) av[u] = *(lds_acc_c)(size_t)aAddr[u];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {                     // (unsigned: below the ring's end the difference wraps to something huge)
                    const unsigned nx = at[u] + stepB;
                    at[u] = min(nx, nx - wrapB);
                }
                if (!oneRow) phase = phase + 1 == slots ? 0 : phase + 1;
            };
            if (!preIssued) issue_reads(apT);
            BHS_TICK_CLS(8);
            // the row's arithmetic
            {
                acc_t sum = 0.0;
                unsigned slotPtr = slot0;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    sum = __builtin_fma(av[u], bv[u], sum);
                    ring_end_step(endMask[u], slotPtr, sum);
                }
                if (tail >= 0) unsafeAtomicAdd(&acc[tail], sum);     // (after every plain store of the row: in order)
            }
            wave_sync();
            BHS_TICK_CLS(4);
            // whatever is in flight was requested a row ago: the slab the next row needs first -- and the row before's stores,
            // which nobody here waits for (their acknowledgement takes longer than a row's arithmetic)
            // (A counted wait that leaves the row before's stores in flight -- tools/lab/bhs_class_ring_lab.hip.h, bit 8 -- measured the same, and
            // keeps the compiler from knowing that the chunk of A's values has arrived: it then waits in front of their use.)
            __builtin_amdgcn_s_waitcnt(kWaitVm0);
            BHS_TICK_CLS(6);
            // this row's sums, out of the slots before the next row's arithmetic writes there
            const int out = __builtin_amdgcn_readlane(pc.cp, t);
            acc_t x0[MAXW], x1[MAXW];
            {
                typedef __attribute__((address_space(3))) const acc_t* lds_acc_c;
#pragma unroll
                for (int w2 = 0; w2 < MAXW; ++w2) {
                    const int e0 = max(0, min(2 * (w2 * 64 + lane), nnz - 2));
                    const unsigned a0 = accBase + (unsigned)e0 * (unsigned)sizeof(acc_t);
                    x0[w2] = *(lds_acc_c)(size_t)a0;
                    x1[w2] = *(lds_acc_c)(size_t)(a0 + (unsigned)sizeof(acc_t));
                }
            }
            // the chunk of A's values: the next one becomes the one at hand behind the last row of this one, and the one
            // after it is requested (whatever was requested a chunk ago has arrived: the wait above)
            if (t + 1 == cEnd) {
                const bool more = t + 1 < nr;                        // the next chunk is in this block
                if (more || nrN > 0) {
                    const int s0 = more ? t + 1 : 0, nrB = more ? nr : nrN;
                    const int e0 = more ? chunk_end(pc, s0, nrB) : chunk_end(pn, s0, nrB);
                    const int b0 = more ? __builtin_amdgcn_readlane(pc.ap, s0) : __builtin_amdgcn_readlane(pn.ap, 0);
                    if (nxtValid) axCur = axNxt;
                    else {                                           // (a block of one chunk: on demand)
                        axCur = load_chunk(b0, (more ? __builtin_amdgcn_readlane(pc.ap1, e0 - 1) : __builtin_amdgcn_readlane(pn.ap1, e0 - 1)) - b0);
                        __builtin_amdgcn_s_waitcnt(kWaitVm0);
                    }
                    curBase = b0;
                    cEnd = e0;                                       // (rows of the block the chunk is in)
                    // the chunk after it: in the same block as the new one, or -- that block ending with it -- in the block
                    // behind it, if that is the next block (its pointers are here); else on demand
                    nxtValid = false;
                    if (e0 < nrB) {
                        const int e1 = more ? chunk_end(pc, e0, nrB) : chunk_end(pn, e0, nrB);
                        const int b1 = more ? __builtin_amdgcn_readlane(pc.ap, e0) : __builtin_amdgcn_readlane(pn.ap, e0);
                        axNxt = load_chunk(b1, (more ? __builtin_amdgcn_readlane(pc.ap1, e1 - 1) : __builtin_amdgcn_readlane(pn.ap1, e1 - 1)) - b1);
                        nxtValid = true;
                    } else if (more && nrN > 0) {
                        const int e1 = chunk_end(pn, 0, nrN);
                        const int b1 = __builtin_amdgcn_readlane(pn.ap, 0);
                        axNxt = load_chunk(b1, __builtin_amdgcn_readlane(pn.ap1, e1 - 1) - b1);
                        nxtValid = true;
                    }
                }
            }
            BHS_TICK_CLS(10);
            // the next row's operands, if it carries on from this one (its slabs are here: the wait above) -- behind this row's
            // sums: no store below has to wait for an LDS read
            __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
            preIssued = false;
            if (!oneRow && t + 1 < nr && __builtin_amdgcn_readlane(pc.cls, min(t + 1, 63)) == cur) {
                issue_reads(__builtin_amdgcn_readlane(pc.ap, min(t + 1, 63)));
                preIssued = true;
            }
            BHS_TICK_CLS(9);
            if (!oneRow) request_slab();
            BHS_TICK_CLS(11);
            {
                int* const cjRow = Cj + (long long)out;
                value_t* const cxRow = Cx + (long long)out;
                if (nnz >= 2) {
#pragma unroll
                    for (int w2 = 0; w2 < MAXW; ++w2) {
                        const int q = w2 * 64 + lane;
                        if (2 * q < nnz) {                           // (an odd row's last lane: its neighbour's second entry once more, and the last)
                            const int e0 = min(2 * q, nnz - 2);
                            class_store_c2_at(cjRow, (unsigned)e0 * (unsigned)sizeof(int), relA[w2] + row + rowBase, relB[w2] + row + rowBase);
                            class_store_c2_at(cxRow, (unsigned)e0 * (unsigned)sizeof(value_t), (value_t)x0[w2], (value_t)x1[w2]);
                        }
                    }
                } else if (nnz == 1 && lane == 0) {
                    class_store_c_at(cjRow, 0u, relA[0] + row + rowBase);
                    class_store_c_at(cxRow, 0u, (value_t)x0[0]);
                }
            }
            wave_sync();
            lastRow = row;
            BHS_TICK_CLS(5);
#if BHS_PHASES_CLS
            ph[7] += 1;
#endif
        }
        pc = pn;
        row0 = row0N;
        nr = nrN;
    }
#if BHS_PHASES_CLS
    if (lane == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_phase_cycles[i], ph[i]);
#endif
}

}  // namespace bhs
