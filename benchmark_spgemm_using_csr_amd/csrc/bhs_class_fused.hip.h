// bhs_class_fused.hip.h -- the classes of a matrix's rows in ONE pass (round 5).  (Included after bhs_class.hip.h.)
//
// Rounds 3-4 classified a matrix with three launches: k_class_heads streams the rows and lists those that differ from the
// row before them (one in fifty on a grid), k_class_rows takes the listed rows through the class table, k_class_propagate
// hands the classes on -- poisson27pt 128^3, in microseconds: 90 + 46 + 10 for B, 135 + 90 + 10 for A, and five gaps
// between dependent launches.  The table passes are latency: 42 000 rows, each a hash, a probe and a comparison, on a
// sixteenth of the device.  Here the wave that finds a head takes it through the table ITSELF, on the spot: the row is
// in its lanes' registers in exactly the layout k_class_rows wants (G lanes per row, E entries per lane), its block keeps
// the classes it has met in LDS, and the rows behind a head get its class as the wave walks on -- no lists, no second
// kernel, no third.  A wave that waits for a table probe is one of sixteen on its CU; the others stream.
#pragma once

#ifndef BHS_FUSED_VEC       // a lane's E entries of a row: 1 consecutive ones, one vector load (classify_rows 0.30 -> 0.48 ms: 4-byte-aligned 8-byte loads); 0 G apart, E loads
#define BHS_FUSED_VEC 0
#endif
#ifndef BHS_FUSED_RPSHARE   // a group's R + 1 row pointers: 1 one load and shuffles (0.30 -> 0.41 ms: the shuffles sit between two dependent loads), 0 two loads per row and lane
#define BHS_FUSED_RPSHARE 0
#endif
#define BHS_FUSED_POS(e) (BHS_FUSED_VEC ? g * E + (e) : (e) * G + g)

namespace bhs {

template <bool IS_A, int G, int E>
__global__ __launch_bounds__(kClassHeadsBlock) void k_class_fused(int nrows, const int* __restrict__ Rp, const int* __restrict__ Rj,
                                                     const int* __restrict__ classB, int* __restrict__ classOut,
                                                     unsigned long long* __restrict__ table, int* __restrict__ stats, long long nnzR, int pieceRows,
                                                     const int* __restrict__ range,     // rows [range[0], range[1]] only (nullptr: all)
                                                     int period)                        // a row is compared with the row `period` before it
{
    constexpr int GPW = 64 / G;                                    // lane groups per wave
    constexpr int R = E >= 8 ? 2 : (E >= 4 ? 4 : 8);               // consecutive rows per lane group
    constexpr int RPW = GPW * R;                                   // consecutive rows per wave and pass
    constexpr int WPB = kClassHeadsBlock / 64;
    const int PIECE = pieceRows;                                   // consecutive rows a wave walks: its first goes through the class table whatever it looks like
    // the block's cache of the class table (k_class_rows: tag = slot << 20 | 20 bits of the hash, then the pattern)
    constexpr int PW = G * E > kClassMaxRow ? G * E : kClassMaxRow;
    constexpr int NC = PW > 2 * kClassMaxRow ? 16 : 32;
    constexpr unsigned kBusy = 0xFFFFFFFEu;
    __shared__ unsigned ctag[NC];
    __shared__ int cpat[NC][PW];
    __shared__ int clen[NC];
    __shared__ int cpatB[IS_A ? NC : 1][PW];
    __shared__ int sCount;                                         // heads of the block (statistics)
    const int leaderLane = (threadIdx.x & 63) - (threadIdx.x & 63) % G;
    const int lane = threadIdx.x & 63, g = lane % G, grp = lane / G;
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << (lane - g);
    long long first = 0;
    if (range != nullptr) {                                        // (wave-uniform values)
        const int lo = range[0], hi = range[1];
        first = lo <= hi ? lo : 0;
        nrows = lo <= hi ? min(nrows, hi + 1) : 0;
    }
    if (threadIdx.x == 0) sCount = 0;
    if (threadIdx.x < NC) { ctag[threadIdx.x] = 0xFFFFFFFFu; clen[threadIdx.x] = -1; }   // (LDS is not cleared between workgroups)
    __syncthreads();
    const long long wave = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
    const long long pieceBegin = first + wave * PIECE;
    const long long pieceEnd = min((long long)nrows, pieceBegin + PIECE);
    bool noClass = false;                                          // a row of this wave's found no class
    bool okP = false;                                              // the pass before's last row (lane g of every group)
    int lenP = 0, elP[E], cbP[E], followP = 0;                       // followP: (position of the last head << 13) | (its class + 1)
#pragma unroll
    for (int e = 0; e < E; ++e) { elP[e] = 0; cbP[e] = 0; }
    // the wave walks its piece as `period` sequences, one after the other: position q of sequence `seq` = row pieceBegin + q * period + seq
    const int perSeq = (PIECE + period - 1) / period;
    for (int seq = 0; seq < period; ++seq) {
    auto row_at = [&](int q) { return pieceBegin + (long long)q * period + seq; };
    followP = 0;
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
        }
        // the row pointers: a group's R rows are consecutive (period 1), so their R + 1 pointers are ONE load by the group's
        // first lanes, handed round by shuffles (2 R loads per lane otherwise: the pass is bound by its vector-memory
        // instructions, not by its bytes)
        if (BHS_FUSED_RPSHARE && G > R && period == 1) {
            const int rpv = Rp[min(rowv[0] + min(g, R), (long long)nrows)];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a0[r] = __shfl(rpv, leaderLane + r, 64);
                len[r] = __shfl(rpv, leaderLane + r + 1, 64);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long rr = live[r] ? rowv[r] : 0;
                a0[r] = Rp[rr];
                len[r] = Rp[rr + 1];
            }
        }
        // the columns: E consecutive entries per lane, one load (positions past the row's end read the next rows' entries --
        // real columns, never compared; only the last lanes of the array's last row must not read past its end)
        typedef int intE __attribute__((ext_vector_type(E == 1 ? 2 : E), aligned(4)));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            len[r] = live[r] ? len[r] - a0[r] : 0;
            ok[r] = live[r] && len[r] <= G * E;
            if (BHS_FUSED_VEC) {
                const long long base = (long long)(ok[r] ? a0[r] : 0) + g * E;
                if (base + E <= nnzR) {
                    if (E == 1) cc[r][0] = Rj[base];
                    else {
                        const intE v = *reinterpret_cast<const intE*>(Rj + base);
#pragma unroll
                        for (int e = 0; e < E; ++e) cc[r][e] = v[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) cc[r][e] = Rj[min(base + e, max(nnzR - 1, 0ll))];
                }
            } else {
                const int lastPos = ok[r] && len[r] > 0 ? a0[r] + len[r] - 1 : 0;
#pragma unroll
                for (int e = 0; e < E; ++e) cc[r][e] = Rj[min(a0[r] + e * G + g, lastPos)];
            }
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
                const bool in = ok[r] && BHS_FUSED_POS(e) < len[r];
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
                // (shuffles by ALL lanes, the selection afterwards: inside a lane-divergent ternary a masked-out lane hands its
                // neighbour nothing -- bhs_class_tile.hip.h)
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
        // the heads through the class table, here and now (the body of k_class_rows for the row sets that have a head: rare)
        int cls[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cls[r] = -1;
            if (!__any(head[r])) continue;                          // (wave-uniform)
            unsigned hp = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = BHS_FUSED_POS(e);
                const bool in = ok[r] && pos < len[r];
                unsigned hh = class_mix(0x85EBCA6Bu * (unsigned)(pos + 1), (unsigned)el[r][e]);
                if (IS_A) hh = class_mix(hh, (unsigned)cb[r][e]);
                hp += in ? hh : 0u;
            }
            const unsigned hr = group_sum_u32<G>(hp) + (unsigned)len[r] * 0x9E3779B1u + 1u;
            const long long row = rowv[r];
            const int lenr = len[r];
            auto equals = [&](bool cand, int rep) {                 // does this row equal row `rep` entry by entry?
                bool same = true;
                if (__any(cand && rep != (int)row)) {               // (rare: a class this block meets for the first time)
                    const int rp = cand ? rep : 0;
                    const int r0 = Rp[rp], r1 = Rp[rp + 1];
                    const int lastR = r1 > r0 ? r1 - 1 : 0;
                    int cr[E], cbr[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) cr[e] = Rj[min(r0 + BHS_FUSED_POS(e), lastR)];
                    if (IS_A) {
#pragma unroll
                        for (int e = 0; e < E; ++e) cbr[e] = classB[cr[e]];
                    }
                    same = r1 - r0 == lenr;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool in = BHS_FUSED_POS(e) < lenr;
                        same = same && (!in || (el[r][e] == cr[e] - rp && (!IS_A || cb[r][e] == cbr[e])));
                    }
                    same = same || rep == (int)row;
                }
                return cand && !(__ballot(cand && !same) & gmask);
            };
            bool searching = ok[r] && head[r];
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
                for (int e = 0; e < E; ++e) { pc[e] = cpat[ci][BHS_FUSED_POS(e)]; pb[e] = IS_A ? cpatB[ci][BHS_FUSED_POS(e)] : 0; }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = BHS_FUSED_POS(e) < lenr;
                    same = same && (!in || (el[r][e] == pc[e] && (!IS_A || cb[r][e] == pb[e])));
                }
                if (cand && !(__ballot(cand && !same) & gmask)) { cls[r] = (int)(tg >> 20); searching = false; }
            }
            const unsigned long long mine = ((unsigned long long)hr << 32) | (unsigned)row;
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
                    cls[r] = s;
                    searching = false;
                    unsigned won = 0;                               // publish in the block's cache if its cell is still free
                    if (g == 0) won = atomicCAS(&ctag[ci], 0xFFFFFFFFu, kBusy) == 0xFFFFFFFFu ? 1u : 0u;
                    won = (unsigned)__shfl((int)won, leaderLane, 64);
                    if (won) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int pos = BHS_FUSED_POS(e);
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
            noClass = noClass || __any(head[r] && cls[r] < 0);       // (told once, when the wave ends: bhs_class_tile.hip.h)
        }
        // the head every row follows: the last head at or before it in the wave's walk, as (position << 13) | (class + 1)
        int lastIn = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) lastIn = head[r] ? ((vi[r] << 13) | (cls[r] + 1)) : lastIn;
        int incl = lastIn;                                         // inclusive running maximum over the groups
#pragma unroll
        for (int o = G; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            incl = lane >= o ? max(incl, up) : incl;
        }
        int follow = __shfl_up(incl, G, 64);                       // the groups before this one, or the passes before this one
        follow = grp ? max(follow, followP) : followP;
        int nHeads = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) nHeads += __popcll(__ballot(head[r] && g == 0));
        if (IS_A && lane == 0 && nHeads) atomicAdd(&sCount, nHeads);
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (head[r]) follow = (vi[r] << 13) | (cls[r] + 1);
                if (live[r]) classOut[rowv[r]] = (follow & 0x1FFF) - 1;
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
    if (noClass && lane == 0) atomicOr(&stats[CS_FLAGS], 1);
    // (statistics: the rows of A that went through the class table -- the host's verdict "rows in stretches, or every row
    // for itself?")
    __syncthreads();
    if (IS_A && threadIdx.x == 0 && sCount) atomicAdd(&stats[CS_HEADS], sCount);
}

}  // namespace bhs
