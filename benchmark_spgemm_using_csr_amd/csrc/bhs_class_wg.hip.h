// bhs_class_wg.hip.h -- numeric pass by row classes, second form: products summed in registers, B values in a ring in LDS.
// (Included after bhs_class.hip.h; see there for the classification and the per-class tables.)
#pragma once

namespace bhs {

// ---------------------------------------------------------------------------
// One wave per workgroup.  A wave takes super-runs of consecutive rows -- a grid line of A where it has such lines (the
// host's superRows; kClassSuper rows otherwise) -- kClassRun at a time; workgroups are dealt to the XCDs so that each
// XCD's L2 sees one contiguous band of rows.
//   per run      the A entries of its rows are one contiguous stretch of colIndA / valA: loaded with coalesced loads,
//                their rowPtrB gathered, both parked in LDS.  These loads run three runs ahead of the arithmetic in a
//                register pipeline (row pointers -> A entries -> rowPtrB words);
//   the ring     rows i, i + 1, .. of ONE class: entry k of row i + 1 selects the B row after the one entry k of row i
//                selects.  So for a CHAIN of A entries with consecutive columns (and B rows of one length) the B rows
//                that row i + 1 needs are those of row i shifted by one: one NEW B row per chain and row.  A "slab" is
//                that new row of every chain, side by side; the wave keeps the last (longest chain + 1) slabs in a ring
//                in LDS and brings in one slab per row with global_load_lds_dwordx4 (two instructions for poisson27pt's
//                9 x 27 values), two rows ahead of the row that needs it first.  Every value of B enters the CU once
//                per wave and stretch (1.9 KB per row of C; round 2's kernel gathers 5.8 KB per row through an L1 that
//                20 waves overrun, and its CU is bound by the L2 requests in flight), at 8 KB of LDS per wave;
//   per row      every lane holds MAXU product descriptors of the class in registers (k_class_patterns: consecutive
//                products of the position-sorted list): per product one LDS read of B's value in the ring (its place
//                moves on by one slab per row), one of A's value and one fused multiply-add into ONE register -- the
//                running sum restarts where the next entry of C begins and is stored to the entry's LDS slot after
//                every product (plain ds_write_b64; the last store of an entry leaves its sum there; the LDS pipe keeps
//                a wave's stores in order).  An entry that straddles a lane boundary gets its earlier lanes' partial
//                sums by one ds_add_f64 after the loop.  Then the row leaves (write-through stores);
//   the waits    a row's order is: arithmetic (LDS only) -> s_waitcnt vmcnt(0) -> request the slab two rows ahead ->
//                store the row.  Whatever the vmcnt(0) waits for -- the slab requested during the row before, that
//                row's stores, the next run's metadata -- was issued a whole row's arithmetic earlier.
// Round 4, fewer vector-memory instructions per row (the CU's memory pipe is what the waves queue for: a VMEM instruction
// took ~230 cycles to ISSUE in round 3's phase timers):
//   * colIndA / rowPtrB of a run are only requested when a stretch STARTS in that run (first run of a super-run, or a
//     class change): inside a stretch the kernel needs A's values and nothing else of A -- the columns of C come from
//     the class's relative list, the B rows from the ring.  A's values are loaded two per lane;
//   (Rows leaving in PAIRS -- the first row's sums kept in registers, both rows written with 8- / 16-byte stores per
//   lane, half the store instructions -- measured 15 % SLOWER: 1.78 against 1.51 ms on one box; the 16-byte stores start
//   at 8-byte-aligned addresses.  Removed.)
// Round 2's form (k_class_numeric_atomic: 64-lane batches in A-entry-major order, one ds_add_f64 per product) stands at
// two walls of equal height: 30 LDS cycles per atomic (3.6 lanes per bank pair) and the CU's L1-miss parallelism
// (profiles/r03_class_numeric_forms.md, which also has the forms of this kernel that staged whole stretches of rows
// and were left with 4 to 6 waves per CU).
// ---------------------------------------------------------------------------
constexpr int kClassRun = BHS_CLS_RUN;                           // rows per run (metadata granularity; <= 63)
constexpr int kClassSuper = BHS_CLS_SUPER;                       // consecutive rows a wave takes before it moves on
constexpr int kClassMaxJ = kClassMaxLoads;                       // most LDS-direct load instructions per slab (64 lanes x 16 bytes each)
static_assert(kClassSuper % kClassRun == 0, "whole runs");

typedef __attribute__((address_space(3))) void bhs_lds_void;
typedef __attribute__((address_space(1))) const void bhs_glb_void;
__device__ __forceinline__ unsigned ringBaseOf(const value_t* ring) { return (unsigned)(size_t)ring; }

typedef value_t bhs_val2 __attribute__((ext_vector_type(2), aligned(sizeof(value_t))));   // two consecutive values of A: one load

template <int MAXU, int MAXV, int SE, int MAXJ>   // SE: 64-entry passes over a run's A entries (even); MAXJ: load instructions per slab
__global__ __launch_bounds__(64) void k_class_numeric(
    int m, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    long long nnzA, const int* __restrict__ Bp, const value_t* __restrict__ Bx, long long nnzB, const int* __restrict__ classC,
    const int4* __restrict__ classInfo, const unsigned* __restrict__ classMap, const int* __restrict__ classRel,
    const int* __restrict__ classLane, const int* __restrict__ Cp, int* __restrict__ Cj, value_t* __restrict__ Cx,
    int accStride, int stageCap, int ringCap, int rowBase, int superRows)     // m, Ap, classC, Cp are views of the rows [rowBase, rowBase + m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    const int lane = threadIdx.x;
    // acc[accStride] + sAx[stageCap] doubles, ring[ringCap] values (all 16-byte aligned: the three counts are even)
    acc_t* acc = reinterpret_cast<acc_t*>(smemRaw);
    acc_t* sAx = acc + accStride;                                // A values of the run at hand
    value_t* ring = reinterpret_cast<value_t*>(sAx + stageCap);

    // super-runs of superRows consecutive rows (a grid line of A where it has such lines -- the waves of an XCD then walk
    // neighbouring lines side by side and meet in the same rows of B -- else kClassSuper), cut into runs of kClassRun rows,
    // the last one of a super-run shorter if it must be: XCD x takes the super-runs [x * perX, (x + 1) * perX); block b
    // runs on XCD b % 8
    const int RPS = (superRows + kClassRun - 1) / kClassRun;     // runs per super-run
    const int nSuper = (m + superRows - 1) / superRows;
    const int nRuns = nSuper * RPS;
    const int xcd = blockIdx.x & 7, perX = (nSuper + 7) / 8;
    const int wavesPerX = gridDim.x >> 3, wIdx = blockIdx.x >> 3;
    // the i-th run this wave works on: run (i % RPS) of its (i / RPS)-th super-run
    auto run_of = [&](int i) {
        const long long sr = (long long)wIdx + (long long)(i / RPS) * wavesPerX;
        if (sr >= perX) return nRuns;
        const long long run = ((long long)xcd * perX + sr) * RPS + i % RPS;
        return (int)min((long long)nRuns, run);
    };
    auto first_row = [&](int run) { return (run / RPS) * superRows + (run % RPS) * kClassRun; };
    auto rows_of = [&](int run) {
        if (run >= nRuns) return 0;
        return max(0, min(min(kClassRun, superRows - (run % RPS) * kClassRun), m - first_row(run)));
    };

    // A run's metadata travels through a three-deep register pipeline so that no load is waited for where it is
    // issued: at the top of the work on run i the row pointers / classes of run i + 3, the A entries of run i + 2 and
    // the rowPtrB words of run i + 1 are requested; the vmcnt(0) of the run's last row covers them (every row has one),
    // and right behind it -- the last row's arithmetic is done -- run i + 1's A values and B row starts replace run
    // i's in the LDS staging area.
    struct RunPtrs { int ap, cp, cls; };
    auto load_ptrs = [&](int run) {
        RunPtrs r{0, 0, -1};
        const int nr = rows_of(run);
        if (nr > 0) {
            const int row0 = first_row(run);
            if (lane <= nr) { r.ap = Ap[row0 + lane]; r.cp = Cp[row0 + lane]; }
            if (lane < nr) r.cls = classC[row0 + lane];
        }
        return r;
    };
    auto entries_of = [&](const RunPtrs& r, int nr) { return min(__builtin_amdgcn_readlane(r.ap, nr) - __builtin_amdgcn_readlane(r.ap, 0), stageCap); };

    // class at hand (registers) ...
    int cur = -2, nnz = 0, tail = -1, slab = 0, slots = 1;
    // per product (lane, step) ONE word, LDS byte addresses / offsets so that the row loop computes none: the slot of the entry
    // of C (bits 0-15), the A entry's place in a row's staged values (16-24), the B entry (25-30; read where a stretch
    // begins), and the sign bit: the running sum restarts here.  A step without a product reads A's and the ring's first
    // value and stores to the dump slot; the product behind it restarts the sum.
    int desc[MAXU];
    int ent = 0;                                                 // the class's chain table: this lane as A entry (k_class_patterns)
    // ... its slab as seen by this lane's share of the MAXJ load instructions: the chain's first A entry, the
    // lane's place in that chain's row (-1: a padding lane), the row's length
    int dma[MAXJ];                                               // place in the chain's row (bits 0-7), row length (8-15, 0: padding lane), the chain's first A entry (16-21)
    // the ring: where this lane's next piece of a slab comes from, the slot it goes to, every product's place, the last row done
    unsigned src[MAXJ];                                          // (values of B are counted in int32: nnzB < 2^31)
    int loadSlot = 0, lastRow = -2, wrapB = 0;
    unsigned at[MAXU];
    bool ringOK = false;
    int rel[MAXV];                                               // the columns of this lane's entries of a row, relative to the row
#if BHS_PHASES_CLS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tPh = __builtin_readcyclecounter();
#endif
    // Does a stretch start in the i-th run of this wave (rows of classes cls, nrI of them; the run before ended in class
    // prevLast)?  Only then are its column indices and their rowPtrB words needed.
    constexpr bool LEAN = !(BHS_CLS_LAB & 8);
    auto starts_in = [&](int i, int cls, int nrI, int prevLast) -> bool {
        if (!LEAN || i % RPS == 0) return true;
        int before = __shfl_up(cls, 1, 64);
        if (lane == 0) before = prevLast;
        return __ballot(lane < nrI && cls != before) != 0ull;
    };
    auto last_class = [&](const int cls, int nrI) { return nrI > 0 ? __builtin_amdgcn_readlane(cls, nrI - 1) : -3; };
    // two consecutive values of A per lane (the last value of valA is not read as the first half of a pair)
    constexpr int SP = SE / 2;
    auto load_ax = [&](int base, int nE, bhs_val2 (&ax)[SP]) {
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int e = i * 128 + 2 * lane;
            ax[i] = bhs_val2{(value_t)0, (value_t)0};
            if (e < nE) {
                if ((long long)base + e + 1 < nnzA) ax[i] = *reinterpret_cast<const bhs_val2*>(Ax + base + e);
                else ax[i].x = Ax[base + e];
            }
        }
    };
    auto stage_ax = [&](int nE, const bhs_val2 (&ax)[SP]) {
        typedef acc_t a2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int e = i * 128 + 2 * lane;
            if (e < nE) *reinterpret_cast<a2*>(sAx + e) = a2{(acc_t)ax[i].x, (acc_t)ax[i].y};   // (stageCap is even: e + 1 is inside)
        }
    };
    // prologue: run 0 staged, run 1's A entries and run 2's pointers in registers
    RunPtrs p0 = load_ptrs(run_of(0)), p1 = load_ptrs(run_of(1)), p2 = load_ptrs(run_of(2));
    int aj1[SE], bp0[SE];                                        // (bp0: B row start of every A entry of the run at hand -- if a stretch starts in it)
    bhs_val2 ax1[SP];
    bool ns1 = true;                                             // a stretch starts in the next run: aj1 holds its columns
    {
        const int nr0 = rows_of(run_of(0)), nr1 = rows_of(run_of(1));
        const int b0 = __builtin_amdgcn_readlane(p0.ap, 0), nE0 = nr0 ? entries_of(p0, nr0) : 0;
        const int b1 = __builtin_amdgcn_readlane(p1.ap, 0), nE1 = nr1 ? entries_of(p1, nr1) : 0;
        int aj0[SE];
        bhs_val2 ax0[SP];
#pragma unroll
        for (int i = 0; i < SE; ++i) {
            aj0[i] = aj1[i] = -1;
            if (i * 64 + lane < nE0) aj0[i] = Aj[b0 + i * 64 + lane];
            if (i * 64 + lane < nE1) aj1[i] = Aj[b1 + i * 64 + lane];
        }
        load_ax(b0, nE0, ax0);
        load_ax(b1, nE1, ax1);
#pragma unroll
        for (int i = 0; i < SE; ++i) bp0[i] = aj0[i] >= 0 ? Bp[aj0[i]] : 0;
        stage_ax(nE0, ax0);
    }
    wave_sync();
    for (int it = 0;; ++it) {
        const int run = run_of(it);
        if (run >= nRuns) break;
        const int row0 = first_row(run), nr = rows_of(run);
        const int base = __builtin_amdgcn_readlane(p0.ap, 0);
        // requests for the runs behind this one (consumed behind this run's first vmcnt(0))
        const RunPtrs p3 = load_ptrs(run_of(it + 3));
        const int nr1 = rows_of(run_of(it + 1)), nr2 = rows_of(run_of(it + 2));
        const int nE1 = nr1 ? entries_of(p1, nr1) : 0;
        const bool ns2 = nr2 > 0 && starts_in(it + 2, p2.cls, nr2, last_class(p1.cls, nr1));
        int aj2[SE], bp1[SE];
        bhs_val2 ax2[SP];
        {
            const int b2 = __builtin_amdgcn_readlane(p2.ap, 0), nE2 = nr2 ? entries_of(p2, nr2) : 0;
#pragma unroll
            for (int i = 0; i < SE; ++i) aj2[i] = -1;
            if (ns2) {
#pragma unroll
                for (int i = 0; i < SE; ++i)
                    if (i * 64 + lane < nE2) aj2[i] = Aj[b2 + i * 64 + lane];
            }
            load_ax(b2, nE2, ax2);
#pragma unroll
            for (int i = 0; i < SE; ++i) bp1[i] = 0;
            if (ns1) {
#pragma unroll
                for (int i = 0; i < SE; ++i) bp1[i] = aj1[i] >= 0 ? Bp[aj1[i]] : 0;
            }
        }
        auto stage_next = [&]() {               // (behind the last row's vmcnt(0): the requests above have arrived, the row is done)
            stage_ax(nE1, ax1);
#pragma unroll
            for (int i = 0; i < SE; ++i) {
                bp0[i] = bp1[i];
                aj1[i] = aj2[i];
            }
#pragma unroll
            for (int i = 0; i < SP; ++i) ax1[i] = ax2[i];
        };
        // one slab: this lane's 16 bytes of each of its load instructions, if they are a piece of a B row
        auto request_slab = [&]() {
            if (!(BHS_CLS_LAB & 1)) {
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {
                    if (j * 64 * kClassEpl < slab) {
                        const bool piece = (dma[j] & 0xFF00) != 0;
                        if (piece && (long long)src[j] + kClassEpl <= nnzB)
                            __builtin_amdgcn_global_load_lds((bhs_glb_void*)(Bx + src[j]), (bhs_lds_void*)(ring + loadSlot * slab + j * 64 * kClassEpl), 16, 0, 0);
                        else if (piece)                              // (the last few values of valB: no 16-byte load past its end)
                            for (int e2 = 0; e2 < kClassEpl; ++e2)
                                if ((long long)src[j] + e2 < nnzB) ring[loadSlot * slab + (j * 64 + lane) * kClassEpl + e2] = Bx[src[j] + e2];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) src[j] += (unsigned)((dma[j] >> 8) & 255);
            loadSlot = loadSlot + 1 == slots ? 0 : loadSlot + 1;
        };
        BHS_TICK_CLS(0);
        for (int t = 0; t < nr; ++t) {
            const int cls = __builtin_amdgcn_readlane(p0.cls, t);
            if (cls < 0) continue;                                   // (cannot happen: such a multiply was sent back)
            const int row = row0 + t;
            const int offT = __builtin_amdgcn_readlane(p0.ap, t) - base;
            if (cls != cur) {                                        // (wave-uniform)
                cur = cls;
                ringOK = false;
                // every word of the class's tables is where the kernel can ask for it without knowing the class (right-aligned
                // map, fixed places): one round trip
                const int4 ci = classInfo[cls];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) desc[u] = (int)classMap[(size_t)cls * kClassMaxP + (kClassMaxSteps - MAXU + u) * 64 + lane];
                tail = classLane[(size_t)cls * kClassLaneInts + lane];
                ent = classLane[(size_t)cls * kClassLaneInts + 64 + lane];     // as A entry
                const int geoV = classLane[(size_t)cls * kClassLaneInts + 192];
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) dma[j] = classLane[(size_t)cls * kClassLaneInts + 256 + j * 64 + lane];
#pragma unroll
                for (int v = 0; v < MAXV; ++v) rel[v] = classRel[(size_t)cls * kClassMaxNnz + v * 64 + lane];   // (beyond the row: never stored)
                __builtin_amdgcn_s_waitcnt(kWaitVm0);                // (so that no later wait has to cover these loads)
                nnz = __builtin_amdgcn_readfirstlane(ci.z);
                const int geo = __builtin_amdgcn_readfirstlane(geoV);
                slots = ((geo >> 8) & 255) + 1;                      // slabs a stretch starts with (rows 0 and 1 find theirs) = slots of
                                                                     // the ring: a row's request replaces the slab only that row still needed
                slab = geo >> 16;
                wrapB = slots * slab * (int)sizeof(value_t);
                const unsigned accBase = (unsigned)(size_t)acc;      // (low half of a flat LDS address = the LDS byte address)
#pragma unroll
                for (int u = 0; u < MAXU; ++u) desc[u] += (int)accBase;
                BHS_TICK_CLS(1);
            }
            if (!ringOK || row != lastRow + 1) {                     // a stretch begins: its first slabs, all at once
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {                     // the B row of the chain's first A entry, from its holder's register
                    const int idx = offT + ((dma[j] >> 16) & 63);
                    int bo = 0;
#pragma unroll
                    for (int i = 0; i < SE; ++i) {
                        const int got = __shfl(bp0[i], idx & 63, 64);
                        bo = (idx >> 6) == i ? got : bo;
                    }
                    src[j] = (unsigned)(bo + (dma[j] & 255));
                }
                loadSlot = 0;
                for (int s2 = 0; s2 < slots; ++s2) request_slab();
                // every product's place in the ring at the stretch's first row: the slot = its A entry's place in its chain
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    const int ek = __shfl(ent, ((desc[u] >> 16) & 0x1FF) / (int)sizeof(acc_t), 64);
                    at[u] = ringBaseOf(ring) + (unsigned)((((ek >> 16) & 63) * slab + (ek & 0xFFFF) + ((desc[u] >> 25) & 63)) * (int)sizeof(value_t));
                }
                ringOK = true;
                BHS_TICK_CLS(2);
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
                wave_sync();
                BHS_TICK_CLS(3);
            }
            // the row's arithmetic: LDS only
            {
                typedef __attribute__((address_space(3))) const value_t* lds_val;
                typedef __attribute__((address_space(3))) const acc_t* lds_acc_c;
                typedef __attribute__((address_space(3))) acc_t* lds_acc;
                const unsigned aBase = (unsigned)(size_t)sAx + (unsigned)offT * (unsigned)sizeof(acc_t);
                acc_t bv[MAXU], axv[MAXU];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) bv[u] = (acc_t)*(lds_val)(size_t)at[u];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) axv[u] = *(lds_acc_c)(size_t)(aBase + (((unsigned)desc[u] >> 16) & 0x1FFu));
                acc_t sum = 0.0;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    sum = desc[u] < 0 ? 0.0 : sum;
                    sum = __builtin_fma(axv[u], bv[u], sum);
                    *(lds_acc)(size_t)((unsigned)desc[u] & 0xFFFFu) = sum;
                }
                if (tail >= 0) unsafeAtomicAdd(&acc[tail], sum);     // (after every plain store of the row: in order)
                // every product moves on by one slab, around the ring
                const unsigned ringEnd = ringBaseOf(ring) + (unsigned)wrapB, stepB = (unsigned)slab * (unsigned)sizeof(value_t);
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    const unsigned nx = at[u] + stepB;
                    at[u] = nx >= ringEnd ? nx - (unsigned)wrapB : nx;
                }
            }
            wave_sync();
            BHS_TICK_CLS(4);
            // whatever is in flight was requested a row ago: the slab the next row needs first, the row before's stores
            __builtin_amdgcn_s_waitcnt(kWaitVm0);
            BHS_TICK_CLS(6);
            request_slab();
            const int out = __builtin_amdgcn_readlane(p0.cp, t);
            if (!(BHS_CLS_LAB & 2)) {
#pragma unroll
                for (int v = 0; v < MAXV; ++v) {
                    const int s = v * 64 + lane;
                    if (s < nnz) {
                        class_store_c(&Cj[(long long)out + s], rel[v] + row + rowBase);
                        class_store_c(&Cx[(long long)out + s], (value_t)acc[s]);
                    }
                }
            }
            wave_sync();
            lastRow = row;
            BHS_TICK_CLS(5);
#if BHS_PHASES_CLS
            ph[7] += 1;
#endif
        }
        // (the last row's vmcnt(0) came after this run's requests were issued, or there was no row: wait here)
        __builtin_amdgcn_s_waitcnt(kWaitVm0);
        stage_next();
        wave_sync();
        p0 = p1; p1 = p2; p2 = p3;
        ns1 = ns2;
    }
#if BHS_PHASES_CLS
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_cycles[i], ph[i]);
#endif
}

}  // namespace bhs
