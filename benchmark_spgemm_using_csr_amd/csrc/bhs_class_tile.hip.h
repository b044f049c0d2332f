// bhs_class_tile.hip.h -- the classes of a matrix's rows in one pass, a LANE per row (round 5).  (Included after bhs_class_fused.hip.h.)
//
// k_class_fused gives a row to G lanes, E entries each: a wave's load instruction then fetches 64 / G pieces of G consecutive
// column indices, 4 G bytes each and a row apart -- 48 vector-memory instructions per 64 rows of poisson27pt, every one of
// them eight partly used cache lines, and for A as many scattered reads of B's classes again.  The pass is bound by those
// instructions (classify_rows 0.25 ms on 128^3: 1.9 - 2.3 TB/s), not by its bytes.
//
// Here a wave takes 63 consecutive rows per step (lane 0 holds the row before them once more: every lane finds the row before
// its own in the lane below, and nothing is carried from step to step):
//   * their column indices are ONE contiguous piece of colInd: it comes in as 16-byte loads from the 16-byte boundary at or
//     below its first entry, 1 KB per instruction (7 instructions for 63 rows of 27), into registers and from there into the
//     wave's tile in LDS when the step begins -- the tile of step s + 1 and the row pointers of step s + 2 are on their way
//     while step s is compared;
//   * lane i owns row i: it reads its entries from the tile (rows 27 words apart: no two lanes on a bank), gets the row
//     before's from lane i - 1 by DPP (wave_shr:1), and compares.  For A the classes of the B rows behind entry e are
//     classB[c + i] for 64 consecutive i on a grid -- one coalesced load per entry, eight entries in flight;
//   * a row that differs from the row before it (a head: one in fifty on a grid) goes through the class table exactly as in
//     k_class_fused -- G lanes, E entries each, read back from the tile -- eight heads at a time; the classes are handed on
//     along the wave's walk by a running maximum of (position, class);
//   * 512 lanes per workgroup share the block's class cache; 16 waves per CU (128 VGPRs, 9 KB of LDS per wave); the host cuts
//     the rows into one piece per wave slot of the device.
// Everything that crosses lanes (DPP moves, shuffles, ballots) is computed by ALL lanes, unconditionally: behind `a || b` the
// lanes that have their answer are masked out, and a masked-out lane hands its neighbour nothing (the first form of this
// kernel made every row behind a head a head that way).
// Rows compared `period` apart (block-structured matrices) stay with k_class_fused.
#pragma once

#ifndef BHS_TILE_EH       // entries of a row compared at a time (for A: so many gathers of B's classes in flight per lane): poisson27pt 160^3 classify_rows 0.327 / 0.303 / 0.293 / 0.385 ms with 4 / 8 / 16 / 32
#define BHS_TILE_EH 16
#endif

namespace bhs {

constexpr int kClassTileBlock = 512;
#ifdef BHS_TILE_DEBUG
__device__ int g_tileDbg[64 * 8];
#endif

// lane i gets lane i - 1's value, lane 0 gets `first` (DPP wave_shr:1)
__device__ __forceinline__ int wave_prev(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false);
}

template <bool IS_A, int G, int E>
__global__ __launch_bounds__(kClassTileBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_class_tile(int nrows, const int* __restrict__ Rp, const int* __restrict__ Rj,
                                                  const int* __restrict__ classB, int* __restrict__ classOut,
                                                  unsigned long long* __restrict__ table, int* __restrict__ stats, long long nnzR, int pieceRows,
                                                  const int* __restrict__ range)     // rows [range[0], range[1]] only (nullptr: all)
{
    constexpr int LMAX = G * E;                                    // longest row that can have a class
    constexpr int GPW = 64 / G;                                    // heads through the table at a time
    constexpr int WPB = kClassTileBlock / 64;
    constexpr int TCAP = 64 * LMAX;                                // column indices of a tile
    constexpr int PW = LMAX, NC = 32;                              // the block's class cache: patterns of at most LMAX entries
    constexpr unsigned kBusy = 0xFFFFFFFEu;
    __shared__ unsigned ctag[NC];                                  // the block's cache of the class table (as in k_class_fused)
    __shared__ int cpat[NC][PW];
    __shared__ int clen[NC];
    __shared__ int cpatB[IS_A ? NC : 1][PW];
    __shared__ int sCount;
    __shared__ __attribute__((aligned(16))) int tileAll[WPB][TCAP + 8 + LMAX];
    __shared__ int sClsAll[WPB][64];
    // (75 KB at G x E = 32: more than 64 KB of LDS per workgroup is gfx950's -- the Makefile's ARCH is not meant to be anything else)
    static_assert(sizeof(int) * ((size_t)WPB * (TCAP + 8 + LMAX) + (size_t)NC * PW * 2 + WPB * 64 + 3 * NC) <= 160 * 1024, "k_class_tile: the tiles of a workgroup's waves must fit gfx950's 160 KB of LDS");
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
    static_assert(G * E < 0, "bhs_class_tile.hip.h is written for gfx950 (160 KB of LDS per workgroup)");
#endif
    const int lane = threadIdx.x & 63, g = lane % G, grp = lane / G, wv = threadIdx.x >> 6;
    const int leaderLane = lane - g;
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << (lane - g);
    int* tile = tileAll[wv];
    int* sCls = sClsAll[wv];
    long long first = 0;
    if (range != nullptr) {                                        // (wave-uniform values)
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        nrows = lo <= hi ? min(nrows, hi + 1) : 0;
    }
    if (threadIdx.x == 0) sCount = 0;
    if (threadIdx.x < NC) { ctag[threadIdx.x] = 0xFFFFFFFFu; clen[threadIdx.x] = -1; }
    __syncthreads();
    const long long wave = (long long)blockIdx.x * WPB + wv;
    if (pieceRows <= 0) {                                          // one piece per wave of the launch, whole steps: the rows in question are known here, not on the host
        const long long rows = max(0ll, (long long)nrows - first), waves = (long long)gridDim.x * WPB;
        pieceRows = 63 * (int)max(1ll, (rows + waves * 63 - 1) / (waves * 63));
    }
    const long long pieceBegin = first + wave * pieceRows;
    const long long pieceEnd = min((long long)nrows, pieceBegin + pieceRows);
    // a step takes 63 new rows: lane 0 holds the row BEFORE them (classified by the step before, or by the wave before) once
    // more, so that every lane finds the row before its own in the lane below -- nothing is carried from step to step.
    // Loads run two steps ahead: while step s is compared, the tile of step s + 1 is on its way (in registers until the tile
    // in LDS is free) and the row pointers of step s + 2 are.
    int followP = 0;
    int nHeadsMine = 0;
    bool noClass = false;                                          // a row of this wave's found no class
    constexpr int NV = (TCAP + 3 + 255) / 256;                     // 16-byte loads per lane and tile
    struct Geom { int t0, tlen, off; bool inLds; };
    auto rp_issue = [&](long long base, int& a0, int& a1) {        // the lanes' row pointers of the step at `base`
        a0 = a1 = 0;
        if (base < pieceEnd) {                                     // (wave-uniform)
            const long long row = base - 1 + lane;
            const bool there = lane == 0 ? row >= first : row < pieceEnd;
            const long long rr = there ? row : base;               // (a lane without a row reads row `base`: lane 0's tile starts there)
            a0 = Rp[rr];
            a1 = Rp[rr + 1];
        }
    };
    auto geom_of = [&](long long base, int a0, int a1) {
        Geom gm{0, 0, 0, false};
        if (base < pieceEnd) {
            const int nthere = (int)min(64ll, pieceEnd - (base - 1));          // lanes [0 or 1, nthere) hold rows
            gm.t0 = __builtin_amdgcn_readlane(a0, 0);
            gm.tlen = __builtin_amdgcn_readlane(a1, nthere - 1) - gm.t0;
            gm.inLds = gm.tlen >= 0 && gm.tlen <= TCAP;            // (a row beyond LMAX entries among them: straight from memory)
            gm.off = gm.inLds ? (int)((reinterpret_cast<uintptr_t>(Rj + gm.t0) & 15u) >> 2) : 0;   // the tile starts on a 16-byte boundary of colInd
        }
        return gm;
    };
    // (16-byte loads on 16-byte boundaries: a chunk that holds one of the tile's entries lies in that entry's page whatever
    // else it holds -- the words before the tile's first entry and behind its last are loaded and never looked at)
#define BHS_TILE_ISSUE(gm_)                                                                                        \
    do {                                                                                                           \
        const int4* src_ = reinterpret_cast<const int4*>(Rj + ((long long)(gm_).t0 - (gm_).off));                  \
        _Pragma("unroll") for (int j = 0; j < NV; ++j)                                                             \
            if ((gm_).inLds && lane * 4 + j * 256 < (gm_).off + (gm_).tlen) tv[j] = src_[lane + j * 64];           \
    } while (0)
    int4 tv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) tv[j] = make_int4(0, 0, 0, 0);
    int a0, a1, a0n, a1n, a0nn = 0, a1nn = 0;
    rp_issue(pieceBegin, a0, a1);
    rp_issue(pieceBegin + 63, a0n, a1n);
    Geom gm = geom_of(pieceBegin, a0, a1);
    BHS_TILE_ISSUE(gm);
    for (long long base = pieceBegin; base < pieceEnd; base += 63) {
        const long long row = base - 1 + lane;
        const bool ghost = lane == 0;
        const bool live = !ghost && row < pieceEnd;
        const bool there = ghost ? row >= first : live;             // a row of the matrix (lane 0 of the first piece: none)
        const int len = there ? a1 - a0 : 0;
        const int t0 = gm.t0, off = gm.off;
        const bool inLds = gm.inLds;
        if (inLds) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int k = lane * 4 + j * 256;
                if (k < off + gm.tlen) *reinterpret_cast<int4*>(tile + k) = tv[j];
            }
        }
        // the loads of the steps to come
        const Geom gmN = geom_of(base + 63, a0n, a1n);
        BHS_TILE_ISSUE(gmN);
        rp_issue(base + 126, a0nn, a1nn);
        // (a step with a row so long that its rows do not fit the tile: none of them gets a class -- the data set is for the
        // general pipeline anyway)
        bool ok = there && len <= LMAX && inLds;
        const int lenOk = ok ? len : 0;
        const int my = ok ? a0 - t0 + off : 0;                      // the row's first entry in the tile (the tile has LMAX words to spare behind it)
        const int cv0 = tile[my];
        // (no short-circuit and no branch anywhere near a DPP move or a shuffle: a lane that has stopped evaluating is masked out
        // and hands its neighbour nothing)
        int diff = 0, neg = 0;
        constexpr int EH = LMAX > BHS_TILE_EH ? BHS_TILE_EH : LMAX;    // entries at a time: EH loads in flight per lane
#pragma unroll
        for (int e0 = 0; e0 < LMAX; e0 += EH) {
            int cv[EH], cbv[IS_A ? EH : 1];
#pragma unroll
            for (int e = 0; e < EH; ++e) cv[e] = tile[my + e0 + e];
            if (IS_A) {
                const int cSafe = lenOk > 0 ? cv0 : 0;             // a column that exists, for the lanes past their row's end
#pragma unroll
                for (int e = 0; e < EH; ++e) cbv[e] = classB[e0 + e < lenOk ? cv[e] : cSafe];
            }
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                const bool in = e0 + e < lenOk;
                const int elv = in ? cv[e] - (int)row : 0;         // (entries beyond a row's end count as 0 on both sides)
                diff |= elv ^ wave_prev(elv, 0);
                if (IS_A) {
                    const int cbe = in ? cbv[e] : 0;
                    neg |= cbe;
                    diff |= cbe ^ wave_prev(cbe, 0);
                }
            }
            asm volatile("" : "+v"(diff), "+v"(neg));               // (the chunk's verdict here and now: left to itself the compiler keeps every entry and every
            __builtin_amdgcn_sched_barrier(0);                      // neighbour's entry of the LMAX in registers until the end -- 100 registers, half the waves)
        }
        const bool bad = neg < 0;
        ok = ok && !bad;
        const int okBi = wave_prev((int)ok, 0);
        const int lenB = wave_prev(len, 0);
        const bool differs = diff != 0 || !ok || okBi == 0 || len != lenB || row == pieceBegin;     // (the piece's first row is a head by decree)
        const bool head = live && differs;
        // the heads through the class table, GPW at a time: group j takes the batch's j-th head
        unsigned long long hm = __ballot(head);
        nHeadsMine += __popcll(hm);
        while (hm) {                                               // (wave-uniform)
            int src = -1;
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                const int b = hm ? __ffsll((long long)hm) - 1 : -1;
                if (grp == j) src = b;
                hm = hm ? hm & (hm - 1ull) : 0ull;
            }
            const bool has = src >= 0;
            const int sl = has ? src : 0;
            const long long hrow = base - 1 + sl;
            const int ha0 = __shfl(a0, sl, 64), lenr = __shfl(len, sl, 64), okS = __shfl((int)ok, sl, 64);
            const bool hok = has && okS != 0;
            int el[E], cb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = e * G + g;
                const bool in = hok && pos < lenr;
                int c = 0;
                if (in) c = tile[ha0 - t0 + off + pos];
                el[e] = in ? c - (int)hrow : 0;
                cb[e] = IS_A && in ? classB[c] : 0;
            }
            int cls = -1;
            unsigned hp = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = e * G + g;
                const bool in = hok && pos < lenr;
                unsigned hh = class_mix(0x85EBCA6Bu * (unsigned)(pos + 1), (unsigned)el[e]);
                if (IS_A) hh = class_mix(hh, (unsigned)cb[e]);
                hp += in ? hh : 0u;
            }
            const unsigned hr = group_sum_u32<G>(hp) + (unsigned)lenr * 0x9E3779B1u + 1u;
            auto equals = [&](bool cand, int rep) {                 // does this row equal row `rep` entry by entry?
                bool same = true;
                if (__any(cand && rep != (int)hrow)) {              // (rare: a class this block meets for the first time)
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
                        same = same && (!in || (el[e] == cr[e] - rp && (!IS_A || cb[e] == cbr[e])));
                    }
                    same = same || rep == (int)hrow;
                }
                return cand && !(__ballot(cand && !same) & gmask);
            };
            bool searching = hok;
            const int ci = (int)(hr & (NC - 1));
            {
                unsigned tg = 0xFFFFFFFFu;
                if (searching && g == 0) tg = __hip_atomic_load(&ctag[ci], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                tg = (unsigned)__shfl((int)tg, leaderLane, 64);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the pattern is read after its tag, never before
                const bool cand = searching && tg < kBusy && (tg & 0xFFFFFu) == (hr >> 12);
                bool same = clen[ci] == lenr;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int pos = e * G + g;
                    const bool in = pos < lenr;
                    const int pc = cpat[ci][pos], pb = IS_A ? cpatB[ci][pos] : 0;
                    same = same && (!in || (el[e] == pc && (!IS_A || cb[e] == pb)));
                }
                if (cand && !(__ballot(cand && !same) & gmask)) { cls = (int)(tg >> 20); searching = false; }
            }
            const unsigned long long mine = ((unsigned long long)hr << 32) | (unsigned)hrow;
            int s = (int)(hr & (kClassSlots - 1));
            for (int probe = 0; probe < kClassProbe; ++probe) {
                if (!__any(searching)) break;
                unsigned long long v = kClassEmpty;                 // (device-coherent: an entry changes once, empty -> final)
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
                    unsigned won = 0;                               // publish in the block's cache if its cell is still free
                    if (g == 0) won = atomicCAS(&ctag[ci], 0xFFFFFFFFu, kBusy) == 0xFFFFFFFFu ? 1u : 0u;
                    won = (unsigned)__shfl((int)won, leaderLane, 64);
                    if (won) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int pos = e * G + g;
                            if (pos < lenr) {
                                cpat[ci][pos] = el[e];
                                if (IS_A) cpatB[ci][pos] = cb[e];
                            }
                        }
                        if (g == 0) clen[ci] = lenr;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (g == 0) ctag[ci] = ((unsigned)s << 20) | (hr >> 12);
                    }
                }
                s = (s + 1) & (kClassSlots - 1);
            }
            noClass = noClass || __any(has && cls < 0);              // (told once, when the wave ends: one same-address atomic per batch of heads -- 150 k of them on poisson27pt 128^3 with 1 % of its rows perturbed -- was 1 ms of queueing at one L2 word)
            if (has && g == 0) sCls[sl] = cls;
        }
        // the head every row follows: the last head at or before it in the wave's walk, as (position << 13) | (class + 1)
        int incl = head ? ((((int)(row - pieceBegin)) << 13) | (sCls[lane] + 1)) : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            incl = lane >= o ? max(incl, up) : incl;
        }
        const int follow = max(incl, followP);
#ifdef BHS_TILE_DEBUG
        if (blockIdx.x == 0 && wv == 0 && base == pieceBegin) {
            g_tileDbg[lane] = head; g_tileDbg[64 + lane] = sCls[lane]; g_tileDbg[128 + lane] = incl; g_tileDbg[192 + lane] = len;
            g_tileDbg[256 + lane] = cv0; g_tileDbg[320 + lane] = wave_prev(lane, -7); g_tileDbg[384 + lane] = (int)ok; g_tileDbg[448 + lane] = my;
        }
#endif
        if (live) classOut[row] = (follow & 0x1FFF) - 1;
        followP = max(followP, __builtin_amdgcn_readlane(incl, 63));
        a0 = a0n; a1 = a1n; a0n = a0nn; a1n = a1nn; gm = gmN;
    }
    if (noClass && lane == 0) atomicOr(&stats[CS_FLAGS], 1);
    // (statistics: the rows of A that went through the class table)
    if (IS_A && lane == 0 && nHeadsMine) atomicAdd(&sCount, nHeadsMine);
    __syncthreads();
    if (IS_A && threadIdx.x == 0 && sCount) atomicAdd(&stats[CS_HEADS], sCount);
}

#undef BHS_TILE_ISSUE

}  // namespace bhs
