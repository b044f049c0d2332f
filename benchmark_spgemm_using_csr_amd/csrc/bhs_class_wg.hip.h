// bhs_class_wg.hip.h -- numeric pass by row classes, second form: products summed in registers, B values staged in LDS.
// (Included after bhs_class.hip.h; see there for the classification and the per-class tables.)
#pragma once

namespace bhs {

// ---------------------------------------------------------------------------
// Numeric pass.  One wave per workgroup; a wave takes runs of kClassRun consecutive rows; blocks are dealt to the XCDs
// so that each XCD's L2 sees one contiguous band of rows.
//   per run      the A entries of its rows are one contiguous stretch of colIndA / valA: loaded with coalesced loads,
//                their rowPtrB gathered, both parked in LDS (the A values one row per LP slots).  These loads run
//                three runs ahead of the arithmetic in a register pipeline (row pointers -> A entries -> rowPtrB
//                words), so the only memory wait a run sees is the one for its B values;
//   per stretch  of consecutive rows of ONE class (rows i, i + 1, ..: entry k of row i + 1 selects the B row after
//                the one entry k of row i selects, so a CHAIN of A entries with consecutive columns, over the rows of
//                the stretch, selects consecutive rows of B): every B row any row of the stretch needs is copied
//                into LDS ONCE, one row per LP slots, by global_load_lds_dwordx4 (no registers, all of the wave's
//                loads in flight together; a load instruction carries 64 / (LP / 2) B rows).  On poisson27pt 4 rows
//                x 27 B rows become 54 staged rows, 14 load instructions: 3.5 per row against the 12 gathers of
//                round 2 (16 CU cycles of texture addresser each -- and 64 different cache lines per instruction in
//                the product order used here);
//   kClassRV rows of a stretch are worked on TOGETHER: every lane holds MAXU product descriptors of the class in
//                registers (k_class_patterns: consecutive products of the position-sorted list); for each of them
//                and each row one LDS read of B's value, one of A's value and one fused multiply-add into that row's
//                running sum.  From one row to the next a product's operands and slot move by the constants LP, LP
//                and LPO: the rows' reads and writes pair up into ds_read2 / ds_write2 with immediate offsets.  A
//                running sum restarts where the next entry of C begins and is stored to the entry's LDS slot after
//                every product (the last store of an entry leaves its sum there; the LDS pipe keeps a wave's stores
//                in order).  An entry that straddles a lane boundary gets its earlier lanes' partial sums by one
//                ds_add_f64 per row after the loop.  Then the rows leave: values with 16-byte stores, columns =
//                class list + row number.  The class data stays in registers until a row of another class comes.
// Round 2's form (64-lane batches in A-entry-major order, one ds_add_f64 per product) spent 30 CU cycles per atomic:
// the lanes of a batch hit a few dozen scattered slots, several lanes per slot (tools/lds_probe.hip: 8 cycles for a
// conflict-free ds_add_f64, 6 for a ds_write_b64 however many lanes are active, 2.3 / 7.8 for a ds_read_b64 with
// consecutive / scattered addresses).  What bounds this form is the instruction stream of a wave (in-order issue:
// about 12 cycles per instruction at the few waves per CU that the staged B rows leave room for), hence the paired
// LDS instructions and the shared work per stretch.
// ---------------------------------------------------------------------------
#ifndef BHS_CLS_RUN
#define BHS_CLS_RUN 8
#endif
#ifndef BHS_CLS_RV
#define BHS_CLS_RV 4
#endif
#ifndef BHS_CLS_SB_BYTES
#define BHS_CLS_SB_BYTES 14336
#endif
constexpr int kClassRun = BHS_CLS_RUN;                           // rows per run (<= 63)
constexpr int kClassRV = BHS_CLS_RV;                             // rows of a stretch worked on together
constexpr int kClassWaves = 1;
constexpr int kClassEpl = 16 / (int)sizeof(value_t);            // elements of B per lane of a 16-byte LDS-direct load

#ifndef BHS_CLS_LAB      // measurement builds only (tools/build_variants.sh): 1 no LDS-direct loads, 2 no stores of C (wrong results)
#define BHS_CLS_LAB 0
#endif
#if BHS_PHASES_CLS      // measurement builds only (tools/phase_profile_cls.py): wave cycles per phase, summed by lane 0
#define BHS_TICK_CLS(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); ph[i] += t__ - tPh; tPh = t__; } while (0)
#else
#define BHS_TICK_CLS(i) do { } while (0)
#endif
typedef __attribute__((address_space(3))) void bhs_lds_void;
typedef __attribute__((address_space(1))) const void bhs_glb_void;
struct __attribute__((aligned(8))) ClassPair { value_t a, b; };  // two values of C: one store, 8-byte aligned

// LDS of one wave, in bytes (host and kernel agree through this)
template <int LP, int LPO>
struct ClassLds {
    static constexpr int kStage = kClassRun * LP;                // A entries of a run
    // staged B rows per wave: BHS_CLS_SB_BYTES worth, and never fewer than one row of C can need (LP) plus a few
    static constexpr int kRows = BHS_CLS_SB_BYTES / (LP * (int)sizeof(value_t)) > LP + 8 ? BHS_CLS_SB_BYTES / (LP * (int)sizeof(value_t)) : LP + 8;
    static constexpr int kSbElems = kRows * LP;
    static constexpr int kOut = 0;                               // acc_t out[kClassRV][LPO]
    static constexpr int kA = kOut + kClassRV * LPO * (int)sizeof(acc_t);          // acc_t sA[2][kClassRun][LP]
    static constexpr int kB = kA + 2 * kClassRun * LP * (int)sizeof(acc_t);        // value_t sB[kClassSbElems]
    static constexpr int kBo = kB + kSbElems * (int)sizeof(value_t);               // int sBo[2][kStage]: B row start of every A entry
    static constexpr int kRow = kBo + 2 * kStage * (int)sizeof(int);               // int sRow[kRows]: start of every staged B row
    static constexpr int kBase = kRow + kRows * (int)sizeof(int);                  // int sBase[64]: per A entry of the stretch's first row, its B row's place in sB
    static constexpr int kBytes = (kBase + 64 * (int)sizeof(int) + 15) & ~15;
};

// The arithmetic of kClassRV rows of a stretch (rows past the stretch's end compute on whatever the LDS holds and are
// never written out).  bAt / aAt / oAt: element indices of the first row's operands and slots.
// (__restrict__: the three LDS arrays do not overlap, so a step's reads need not wait for the step before's stores.)
template <int MAXU, int LP, int LPO>
__device__ __forceinline__ void class_rows(const unsigned (&mp)[MAXU], const int (&bAt)[MAXU], const int (&aAt)[MAXU],
                                           int tail, const value_t* __restrict__ sB, const acc_t* __restrict__ sA,
                                           acc_t* __restrict__ out)
{
    acc_t sum[kClassRV];
#pragma unroll
    for (int r = 0; r < kClassRV; ++r) sum[r] = 0.0;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const bool restart = (mp[u] & kClassStart) != 0;
        const int o = (int)(mp[u] >> 16);
        acc_t bv[kClassRV], av[kClassRV];
#pragma unroll
        for (int r = 0; r < kClassRV; ++r) { bv[r] = (acc_t)sB[bAt[u] + r * LP]; av[r] = sA[aAt[u] + r * LP]; }
#pragma unroll
        for (int r = 0; r < kClassRV; ++r) {
            sum[r] = restart ? 0.0 : sum[r];
            sum[r] = __builtin_fma(av[r], bv[r], sum[r]);
            out[o + r * LPO] = sum[r];
        }
    }
    if (tail >= 0) {                                                 // (after every plain store of the rows: in order)
#pragma unroll
        for (int r = 0; r < kClassRV; ++r) unsafeAtomicAdd(&out[tail + r * LPO], sum[r]);
    }
}

template <int MAXU, int MAXV, int LP>            // LP: slots per staged row of A / of B (power of two >= the longest)
__global__ __launch_bounds__(64) void k_class_numeric(
    int m, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const value_t* __restrict__ Bx, long long nnzB, const int* __restrict__ classC,
    const int4* __restrict__ classInfo, const unsigned* __restrict__ classMap, const int* __restrict__ classRel,
    const int* __restrict__ classLane, const int* __restrict__ Cp, int* __restrict__ Cj, value_t* __restrict__ Cx,
    int rowBase)                                               // m, Ap, classC, Cp are views of the rows [rowBase, rowBase + m)
{
    constexpr int LPO = 64 * MAXV;                               // slots per row of C (one of them, the last, takes the strays)
    constexpr int SE = kClassRun * LP / 64 > 0 ? kClassRun * LP / 64 : 1;          // 64-entry passes over a run's A entries
    constexpr int LPR = LP / kClassEpl;                          // lanes per staged B row
    constexpr int RPJ = 64 / LPR;                                // B rows per load instruction
    using L = ClassLds<LP, LPO>;
    constexpr int kMaxJ = (L::kRows + RPJ - 1) / RPJ;            // load instructions that fill the staging area
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    const int lane = threadIdx.x;
    acc_t* out = reinterpret_cast<acc_t*>(smemRaw + L::kOut);
    acc_t* sA2 = reinterpret_cast<acc_t*>(smemRaw + L::kA);
    value_t* sB = reinterpret_cast<value_t*>(smemRaw + L::kB);
    int* sBo2 = reinterpret_cast<int*>(smemRaw + L::kBo);
    int* sRow = reinterpret_cast<int*>(smemRaw + L::kRow);
    int* sBase = reinterpret_cast<int*>(smemRaw + L::kBase);

    const int nRuns = (m + kClassRun - 1) / kClassRun;
    // XCD-aware: block b runs on XCD b % 8; XCD x takes the runs [x * perX, (x + 1) * perX)
    const int xcd = blockIdx.x & 7, perX = (nRuns + 7) / 8;
    const int wavesPerX = gridDim.x >> 3;
    const int wIdx = blockIdx.x >> 3;
    auto run_of = [&](int i) { const long long rr = (long long)wIdx + (long long)i * wavesPerX; return rr < perX ? (int)min((long long)nRuns, (long long)xcd * perX + rr) : nRuns; };

    // A run's metadata travels through a three-deep register pipeline so that no load is waited for where it is
    // issued: at the top of the work on run i the row pointers / classes of run i + 3, the A entries of run i + 2 and
    // the rowPtrB words of run i + 1 are requested; the one vmcnt(0) a run needs anyway (its staged B values) covers
    // them, and right behind it run i + 1's A values and B row starts go to the other half of the LDS staging area.
    struct RunPtrs { int ap, cp, cls; };
    auto load_ptrs = [&](int run) {
        RunPtrs r{0, 0, -1};
        if (run < nRuns) {
            const int row0 = run * kClassRun, nr = min(kClassRun, m - row0);
            if (lane <= nr) { r.ap = Ap[row0 + lane]; r.cp = Cp[row0 + lane]; }
            if (lane < nr) r.cls = classC[row0 + lane];
        }
        return r;
    };
    auto rows_of = [&](int run) { return run < nRuns ? min(kClassRun, m - run * kClassRun) : 0; };
    auto entries_of = [&](const RunPtrs& r, int nr) { return min(__builtin_amdgcn_readlane(r.ap, nr) - __builtin_amdgcn_readlane(r.ap, 0), L::kStage); };
    // A values of a run: flat entry i -> row tt (the rows' offsets are in r.ap), slot tt * LP + its place in the row
    auto stage_run = [&](const RunPtrs& r, int nr, int nE, const acc_t (&ax)[SE], const int (&bp)[SE], int half) {
        acc_t* dA = sA2 + half * kClassRun * LP;
        int* dBo = sBo2 + half * L::kStage;
        const int b = __builtin_amdgcn_readlane(r.ap, 0);
#pragma unroll
        for (int i = 0; i < SE; ++i) {
            if (i * 64 < nE) {
                const int e = i * 64 + lane;
                int tt = 0, start = 0;
#pragma unroll
                for (int s2 = 1; s2 < kClassRun; ++s2) {
                    const int o = __builtin_amdgcn_readlane(r.ap, s2) - b;
                    const bool past = s2 < nr && e >= o;
                    tt += past ? 1 : 0;
                    start = past ? o : start;
                }
                if (e < nE) {
                    dBo[e] = bp[i];
                    if (e - start < LP) dA[tt * LP + (e - start)] = ax[i];
                }
            }
        }
    };

    int cur = -2, nnz = 0, tail = -1, aux = 0, nA = 0, nCh = 0, stretchCap = 1;
    unsigned mp[MAXU];
    int rel[MAXV];
#if BHS_PHASES_CLS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tPh = __builtin_readcyclecounter();
#endif
    // prologue: run 0 staged in half 0, run 1's A entries and run 2's pointers in registers
    RunPtrs p0 = load_ptrs(run_of(0)), p1 = load_ptrs(run_of(1)), p2 = load_ptrs(run_of(2));
    int aj1[SE];
    acc_t ax1[SE];
    {
        const int nr0 = rows_of(run_of(0)), nr1 = rows_of(run_of(1));
        const int b0 = __builtin_amdgcn_readlane(p0.ap, 0), nE0 = nr0 ? entries_of(p0, nr0) : 0;
        const int b1 = __builtin_amdgcn_readlane(p1.ap, 0), nE1 = nr1 ? entries_of(p1, nr1) : 0;
        int aj0[SE], bp0[SE];
        acc_t ax0[SE];
#pragma unroll
        for (int i = 0; i < SE; ++i) {
            aj0[i] = aj1[i] = -1;
            ax0[i] = ax1[i] = 0.0;
            if (i * 64 + lane < nE0) { aj0[i] = Aj[b0 + i * 64 + lane]; ax0[i] = (acc_t)Ax[b0 + i * 64 + lane]; }
            if (i * 64 + lane < nE1) { aj1[i] = Aj[b1 + i * 64 + lane]; ax1[i] = (acc_t)Ax[b1 + i * 64 + lane]; }
        }
#pragma unroll
        for (int i = 0; i < SE; ++i) bp0[i] = aj0[i] >= 0 ? Bp[aj0[i]] : 0;
        stage_run(p0, nr0, nE0, ax0, bp0, 0);
    }
    wave_sync();
    for (int it = 0;; ++it) {
        const int run = run_of(it);
        if (run >= nRuns) break;
        const int row0 = run * kClassRun, nr = rows_of(run);
        const int half = it & 1;
        const acc_t* sA = sA2 + half * kClassRun * LP;
        const int* sBo = sBo2 + half * L::kStage;
        const int base = __builtin_amdgcn_readlane(p0.ap, 0);
        // requests for the runs behind this one (consumed behind this run's first vmcnt(0))
        const RunPtrs p3 = load_ptrs(run_of(it + 3));
        const int nr1 = rows_of(run_of(it + 1)), nr2 = rows_of(run_of(it + 2));
        const int nE1 = nr1 ? entries_of(p1, nr1) : 0;
        int aj2[SE], bp1[SE];
        acc_t ax2[SE];
        {
            const int b2 = __builtin_amdgcn_readlane(p2.ap, 0), nE2 = nr2 ? entries_of(p2, nr2) : 0;
#pragma unroll
            for (int i = 0; i < SE; ++i) {
                aj2[i] = -1;
                ax2[i] = 0.0;
                if (i * 64 + lane < nE2) { aj2[i] = Aj[b2 + i * 64 + lane]; ax2[i] = (acc_t)Ax[b2 + i * 64 + lane]; }
            }
#pragma unroll
            for (int i = 0; i < SE; ++i) bp1[i] = aj1[i] >= 0 ? Bp[aj1[i]] : 0;
        }
        bool nextStaged = false;
        auto stage_next = [&]() {                                   // (behind a vmcnt(0): the requests above have arrived)
            stage_run(p1, nr1, nE1, ax1, bp1, half ^ 1);
#pragma unroll
            for (int i = 0; i < SE; ++i) { aj1[i] = aj2[i]; ax1[i] = ax2[i]; }
            nextStaged = true;
        };
        BHS_TICK_CLS(0);
        int t = 0;
        while (t < nr) {
            const int cls = __builtin_amdgcn_readlane(p0.cls, t);
            if (cls < 0) { ++t; continue; }                          // (cannot happen: such a multiply was sent back)
            if (cls != cur) {                                        // (wave-uniform)
                cur = cls;
                const int4 ci = classInfo[cls];
                nA = __builtin_amdgcn_readfirstlane(ci.x);           // (uniform anyway: tells the compiler so)
                const int P = __builtin_amdgcn_readfirstlane(ci.y);
                const int U = (P + 63) >> 6;
                nnz = __builtin_amdgcn_readfirstlane(ci.z);
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {                     // (the class's U steps are the LAST U of the MAXU)
                    unsigned d = u >= MAXU - U ? classMap[(size_t)cls * kClassMaxP + (u - (MAXU - U)) * 64 + lane] : kClassIdle;
                    if ((d >> 16) == kClassDump) d = (d & 0xFFFFu) | ((unsigned)(LPO - 1) << 16);   // strays: the row's last slot
                    mp[u] = d;
                }
                tail = classLane[(size_t)cls * kClassLaneInts + lane];
                aux = classLane[(size_t)cls * kClassLaneInts + 64 + lane];
#pragma unroll
                for (int v = 0; v < MAXV; ++v) rel[v] = v * 64 + lane < nnz ? classRel[(size_t)cls * kClassMaxNnz + v * 64 + lane] : 0;
                __builtin_amdgcn_s_waitcnt(kWaitVm0);                // (so that no later wait has to cover these loads)
                if (!nextStaged) stage_next();
                nCh = __popcll(__ballot(aux < 0));
                // rows of this class whose B rows fit the staging area together: nA + (rows - 1) * chains of them
                stretchCap = max(1, min(kClassRun, nCh > 0 ? 1 + (L::kRows - nA) / nCh : kClassRun));
                BHS_TICK_CLS(1);
            }
            // the stretch: rows t .. t + R - 1 of this class
            const unsigned long long same = __ballot(p0.cls == cls) >> t;
            const int R = min(__builtin_ctzll(~same), stretchCap);
            // chains (lane c < nCh): their B rows are the staged rows [rowBase, rowBase + entries + R - 1)
            const int rowsC = aux < 0 ? ((aux >> 18) & 63) - ((aux >> 12) & 63) + R : 0;
            const int rowBase0 = wave_incl_scan_dpp(rowsC) - rowsC;
            const int totalRows = __builtin_amdgcn_readlane(rowBase0 + rowsC, 63);
            // as A entry k: row (t, k) of B is staged row q0, row (t + tt, k) is q0 + tt
            const int q0 = __shfl(rowBase0, aux & 63, 64) + ((aux >> 6) & 63);
            {
                int st[kClassRun];
#pragma unroll
                for (int tt = 0; tt < kClassRun; ++tt)
                    st[tt] = sBo[max(0, min(__builtin_amdgcn_readlane(p0.ap, min(t + tt, kClassRun)) - base + lane, L::kStage - 1))];
                if (lane < nA) {
                    sBase[lane] = q0 * LP;
#pragma unroll
                    for (int tt = 0; tt < kClassRun; ++tt)
                        if (tt < R) sRow[q0 + tt] = st[tt];
                }
            }
            wave_sync();
            // the staged rows' values: load instruction j carries rows [j * RPJ, (j + 1) * RPJ), LPR lanes each
            const int nJ = (totalRows + RPJ - 1) / RPJ;
            {
                const int e0 = (lane % LPR) * kClassEpl;
                long long from[kMaxJ];
#pragma unroll
                for (int j = 0; j < kMaxJ; ++j) {
                    const int q = j * RPJ + lane / LPR;
                    from[j] = q < totalRows ? (long long)sRow[q] + e0 : -1;
                }
#pragma unroll
                for (int j = 0; j < kMaxJ; ++j) {
                    if (j < nJ && !(BHS_CLS_LAB & 1)) {
                        if (from[j] >= 0) {
                            if (from[j] + kClassEpl <= nnzB)
                                __builtin_amdgcn_global_load_lds((bhs_glb_void*)(Bx + from[j]), (bhs_lds_void*)(sB + j * 64 * kClassEpl), 16, 0, 0);
                            else                                     // (the last few values of valB: no 16-byte load past its end)
                                for (int e2 = 0; e2 < kClassEpl; ++e2)
                                    if (from[j] + e2 < nnzB) sB[j * 64 * kClassEpl + lane * kClassEpl + e2] = Bx[from[j] + e2];
                        }
                    }
                }
            }
            // every product's operands for the stretch's first row
            int bAt[MAXU], aAt[MAXU];
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                const unsigned e = mp[u];
                const int k = (int)(e & 63u);
                aAt[u] = t * LP + k;
                bAt[u] = (e & kClassIdleBit) ? 0 : sBase[k] + (int)((e >> 6) & 63u);
            }
            BHS_TICK_CLS(2);
            __builtin_amdgcn_s_waitcnt(kWaitVm0);
            if (!nextStaged) stage_next();
            wave_sync();
            BHS_TICK_CLS(3);
            for (int g0 = 0; g0 < R; g0 += kClassRV) {
                class_rows<MAXU, LP, LPO>(mp, bAt, aAt, tail, sB, sA, out);
                wave_sync();
                BHS_TICK_CLS(4);
                if (!(BHS_CLS_LAB & 2)) {
#pragma unroll
                    for (int r = 0; r < kClassRV; ++r) {
                        if (g0 + r < R) {
                            const int o = __builtin_amdgcn_readlane(p0.cp, t + g0 + r);
                            for (int s2 = lane * 2; s2 < nnz; s2 += 128) {
                                if (s2 + 1 < nnz) {
                                    ClassPair pr;
                                    pr.a = (value_t)out[r * LPO + s2];
                                    pr.b = (value_t)out[r * LPO + s2 + 1];
                                    *reinterpret_cast<ClassPair*>(Cx + (long long)o + s2) = pr;
                                } else Cx[(long long)o + s2] = (value_t)out[r * LPO + s2];
                            }
#pragma unroll
                            for (int v = 0; v < MAXV; ++v) {
                                const int s2 = v * 64 + lane;
                                if (s2 < nnz) Cj[(long long)o + s2] = rel[v] + row0 + t + g0 + r + rowBase;
                            }
                        }
                    }
                }
                wave_sync();
#pragma unroll
                for (int u = 0; u < MAXU; ++u) { bAt[u] += (mp[u] & kClassIdleBit) ? 0 : kClassRV * LP; aAt[u] += kClassRV * LP; }
                BHS_TICK_CLS(5);
            }
            t += R;
#if BHS_PHASES_CLS
            ph[7] += R;
#endif
        }
        if (!nextStaged) { __builtin_amdgcn_s_waitcnt(kWaitVm0); stage_next(); }
        wave_sync();
        p0 = p1; p1 = p2; p2 = p3;
    }
#if BHS_PHASES_CLS
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_cycles[i], ph[i]);
#endif
}

}  // namespace bhs
