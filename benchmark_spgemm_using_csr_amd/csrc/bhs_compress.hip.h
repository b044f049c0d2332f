// bhs_compress.hip.h -- the compressed symbolic pass: B's pattern as (column block, mask) pairs (k_compress_b) and the wave kernel that ORs masks (k_row_wave_csym).  (Split from bhs_kernels.hip.h in round 4.)
#pragma once

namespace bhs {

// ===========================================================================
// Compressed pattern of B for the symbolic pass.  The symbolic pass only needs
// the SET of columns of every C row, so B's pattern is first rewritten as
// (column >> 5, 32-bit occupancy mask) pairs: a run of adjacent columns (stencils,
// FEM blocks, bands) collapses into one pair -- a poisson27pt row of 27 entries
// becomes 9.6 pairs -- and the hash table sees 2.8x fewer inserts, OR-ing masks
// instead of counting keys; a row's nnz is the popcount of its table's masks.
// (Same idea as the compression step of KokkosKernels' KKMEM symbolic phase; the
// reference has no counterpart: it over-allocates by the upper bound instead,
// bhsparse.h:365-481.)  Matrices whose rows have no adjacent columns gain
// nothing; the host checks the pair count and falls back to the plain pass.
//
// k_compress_b<G>: G <= 16 lanes per row of B.  cLen[j] = (entries, pairs) of row j;
// ext[j] = (first pair, one past the last pair) in `pair`, which reuses rowPtrB's offsets (a row never has more pairs
// than entries), so no scan is needed.  Requires strictly ascending columns inside
// a row (the host only enables it for sorted B); a block that straddles a chunk of
// G entries simply appears twice, which the OR-accumulation absorbs.
// ===========================================================================
template <int G>
__global__ __launch_bounds__(256) void k_compress_b(int k, const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                    int2* __restrict__ ext, int2* __restrict__ cLen,
                                                    int2* __restrict__ pair,
                                                    unsigned long long* __restrict__ totalPairs)
{
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "a lane group lives inside one 16-lane DPP row");
    constexpr int R = 4;                                      // rows in flight per lane group: the loads are a dependent chain
    constexpr int rows_per_block = 256 / G;
    const int tid = threadIdx.x, lane = tid & 63, g = tid & (G - 1);
    unsigned long long mine = 0;
    // one chunk of <= G entries of a row: run heads, segmented OR of the bits towards the head (DPP row shifts,
    // no LDS), compaction of the heads by ballot
    auto chunk = [&](int c, int s, int& cnt) {
        const bool v = c >= 0;
        const int blk = c >> 5;                               // idle lanes: -1, never equal to a real block
        unsigned acc = v ? 1u << (c & 31) : 0u;
        const int prev = (int)dpp_u32<0x111, 0xf, 0xf, false>((unsigned)-2, (unsigned)blk);     // lane - 1
        const bool head = v && (g == 0 || blk != prev);
#define BHS_SEG_OR(D)                                                                                           \
        if (D < G) {                                                                                            \
            const unsigned o = dpp_u32<0x100 + D, 0xf, 0xf, true>(0u, acc);                    /* lane + D */   \
            const int ob = (int)dpp_u32<0x100 + D, 0xf, 0xf, false>((unsigned)-2, (unsigned)blk);               \
            if (g + D < G && ob == blk) acc |= o;                                                               \
        }
        BHS_SEG_OR(1) BHS_SEG_OR(2) BHS_SEG_OR(4) BHS_SEG_OR(8)
#undef BHS_SEG_OR
        const unsigned long long hb = __ballot(head);
        const unsigned gm = (unsigned)(hb >> (lane - g)) & ((1u << G) - 1u);
        if (head) pair[(long long)s + cnt + __popc(gm & ((1u << g) - 1u))] = make_int2(blk, (int)acc);
        cnt += __popc(gm);
    };
    const long long stride = (long long)gridDim.x * rows_per_block * R;
    for (long long rb = (long long)blockIdx.x * rows_per_block * R; rb < k; rb += stride) {
        int s[R], e[R], c[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long long r = rb + q * rows_per_block + tid / G;
            s[q] = e[q] = 0;
            if (r < k) { s[q] = Bp[r]; e[q] = Bp[r + 1]; }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) { c[q] = -1; if (s[q] + g < e[q]) c[q] = Bj[s[q] + g]; }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            int cnt = 0;
            chunk(c[q], s[q], cnt);
            for (int base = s[q] + G; __any(base < e[q]); base += G) {       // rows longer than G entries
                int cc = -1;
                if (base + g < e[q]) cc = Bj[base + g];
                chunk(cc, s[q], cnt);
            }
            const long long r = rb + q * rows_per_block + tid / G;
            if (r < k && g == 0) {
                ext[r] = make_int2(s[q], s[q] + cnt);
                cLen[r] = make_int2(e[q] - s[q], cnt);           // (entries, pairs): what k_upper_bound<.., CMP> gathers
                mine += (unsigned long long)cnt;
            }
        }
    }
    // one same-address global atomic per block (they serialise in L2: one per wave cost 0.3 ms on 8192 blocks)
    __shared__ unsigned long long bsum;
    if (tid == 0) bsum = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(&bsum, mine);
    __syncthreads();
    if (tid == 0 && bsum) atomicAdd(totalPairs, bsum);
}

// ---------------------------------------------------------------------------
// Symbolic wave-per-row kernel on the compressed pattern: the structure of
// k_row_wave<.., NUM = false> (XCD-aware persistent schedule, row-pipelined
// metadata, flat product mapping) with (block, mask) pairs as the products:
// CAS on the block key, ds_or on the slot's mask, nnz = sum of popcounts.
// ---------------------------------------------------------------------------
constexpr int kMaxBCsym = 6;
template <int TS>
struct CsymSmem {
    int keys[TS];
    unsigned masks[TS];
    int sBase[64];
    alignas(8) unsigned marks[2 * kMaxBCsym];
};

template <int TS, int LOG2TS>
__global__ __launch_bounds__(64 * kWavesPerBlock) BHS_WAVE_ATTR void k_row_wave_csym(
    const int4* __restrict__ desc, int qn, int chunkLog2, const int* __restrict__ Aj,
    const int2* __restrict__ cExt, const int2* __restrict__ cPair, int* __restrict__ cntOut,
    int* __restrict__ errFlag)
{
    using Smem = CsymSmem<TS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = kWavesPerBlock;
    constexpr int MAXB = kMaxBCsym, GRP = 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Smem& sm = reinterpret_cast<Smem*>(smem_raw)[wave];
    const unsigned long long* __restrict__ cPair64 = reinterpret_cast<const unsigned long long*>(cPair);

    // schedule and row pipeline: see k_row_wave
    const int chunk = 1 << chunkLog2;
    const int xcd = blockIdx.x & 7, lb = (blockIdx.x >> 3) * WPB + wave, perX = (gridDim.x >> 3) * WPB;
    const int nChunks = (qn + chunk - 1) >> chunkLog2;
    int positions = 0;
    if (nChunks > xcd) {
        positions = ((nChunks - xcd + 7) >> 3) << chunkLog2;
        if (((nChunks - 1) & 7) == xcd) positions -= (nChunks << chunkLog2) - qn;
    }
    const int nIt = lb < positions ? (positions - lb + perX - 1) / perX : 0;
    auto q_of = [&](int it) {
        const int t = lb + it * perX;
        return ((((t >> chunkLog2) << 3) + xcd) << chunkLog2) + (t & (chunk - 1));
    };
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    auto load_desc = [&](int it_) {
        const bool has = it_ < nIt;
        int4 r = desc[q_of(has ? it_ : 0) + vzero];
        if (!has) r = make_int4(-1, 0, 0, 0);
        return r;
    };
    if (nIt == 0) return;
    int4 dC = load_desc(0);
    int4 d1 = load_desc(1);
    int4 d2 = load_desc(2);
    int cC = 0, c1 = 0;
    if (lane < dC.z - dC.y) cC = Aj[dC.y + lane];
    if (lane < d1.z - d1.y) c1 = Aj[d1.y + lane];
    int2 beC = make_int2(0, 0);
    if (lane < dC.z - dC.y) beC = cExt[cC];
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    for (int it = 0; it < nIt; ++it) {
        const int4 d3 = load_desc(it + 3);
        int c2 = 0;
        if (lane < d2.z - d2.y) c2 = Aj[d2.y + lane];
        int2 be1 = make_int2(0, 0);
        if (lane < d1.z - d1.y) be1 = cExt[c1];
        const int row = dC.x, a0 = dC.y, a1 = dC.z;
#pragma unroll
        for (int k = 0; k < (TS + 255) / 256; ++k) {
            const int s = k * 256 + lane * 4;
            if (TS >= 256 || s < TS) {
                *reinterpret_cast<int4*>(&sm.keys[s]) = make_int4(kEmpty, kEmpty, kEmpty, kEmpty);
                *reinterpret_cast<int4*>(&sm.masks[s]) = make_int4(0, 0, 0, 0);
            }
        }
        for (int ca = a0; ca < a1; ca += 64) {
            int b0 = beC.x, len = beC.y - beC.x;
            if (ca != a0) {                                   // rows with more than 64 A entries: later chunks, unpipelined
                const int ea = ca + lane;
                b0 = 0; len = 0;
                if (ea < a1) {
                    const int2 be = cExt[Aj[ea]];
                    b0 = be.x;
                    len = be.y - be.x;
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
            }
            const int incl = wave_incl_scan_dpp(len);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int last = incl - 1;
            const unsigned long long nz = __ballot(len > 0);
            const int jc = mbcnt64(nz);
            wave_sync();
            if (len > 0) sm.sBase[jc] = b0 - (incl - len);
            int done = 0;
            for (int w0 = 0; w0 < total; w0 += 64 * MAXB) {
                const int nb = (total - w0 + 63) >> 6;
                if (lane < 2 * MAXB) sm.marks[lane] = 0;
                wave_sync();
                const int rel = last - w0;
                if (len > 0 && rel >= 0 && rel < 64 * MAXB) atomicOr(&sm.marks[rel >> 5], 1u << (rel & 31));
                wave_sync();
                // a pair stays ONE 64-bit register tuple until it is inserted: splitting it where it is loaded
                // puts a v_mov -- and with it an s_waitcnt vmcnt(0) -- right behind every load
                unsigned long long pr[MAXB];
                int cum = done;
#pragma unroll
                for (int u = 0; u < MAXB; ++u) {
                    pr[u] = 0x00000000ffffffffull;                // (kEmpty, no bits)
                    if (u < nb) {
                        const unsigned long long mk = *reinterpret_cast<const unsigned long long*>(&sm.marks[2 * u]);
                        const int p = w0 + u * 64 + lane;
                        const int j = cum + mbcnt64(mk);
                        cum += __popcll(mk);
                        if (p < total) pr[u] = cPair64[(long long)sm.sBase[j] + p];
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
                            const int bk = (int)(unsigned)pr[g + v];
                            hh[v] = hash_col(bk, LOG2TS);
                            cur[v] = kEmpty;
                            if (bk != kEmpty) cur[v] = atomicCAS(&sm.keys[hh[v]], kEmpty, bk);
                        }
#pragma unroll
                        for (int v = 0; v < GRP; ++v) {
                            const int cv = (int)(unsigned)pr[g + v];
                            if (cv != kEmpty) {
                                if (cur[v] != kEmpty && cur[v] != cv) {        // collision: linear probing
                                    unsigned h = hh[v];
                                    int left = TS;             // bounded probing (see k_row_wave)
                                    for (;;) {
                                        h = (h + 1) & (TS - 1);
                                        const int c2 = atomicCAS(&sm.keys[h], kEmpty, cv);
                                        if (c2 == kEmpty || c2 == cv) break;
                                        if (--left == 0) { atomicOr(errFlag, 1); break; }
                                    }
                                    hh[v] = h;
                                }
                                atomicOr(&sm.masks[hh[v]], (unsigned)(pr[g + v] >> 32));
                            }
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
            }
        }
        wave_sync();
        dC = d1; d1 = d2; d2 = d3;
        c1 = c2;
        beC = be1;
        asm volatile("" : "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(c1), "+v"(beC.x), "+v"(beC.y));
        // ---- nnz of the row: popcount of every mask
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < (TS + 255) / 256; ++k) {
            const int s = k * 256 + lane * 4;
            if (TS >= 256 || s < TS) {
                const int4 mk = *reinterpret_cast<const int4*>(&sm.masks[s]);
                cnt += __popc((unsigned)mk.x) + __popc((unsigned)mk.y) + __popc((unsigned)mk.z) + __popc((unsigned)mk.w);
            }
        }
        cnt = wave_sum_dpp(cnt);
        if (lane == 0) cntOut[row] = cnt;
        wave_sync();
    }
}

}  // namespace bhs
