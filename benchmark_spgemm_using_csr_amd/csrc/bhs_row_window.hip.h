// bhs_row_window.hip.h -- rows of thousands of entries of C, one WAVE (k_row_wave_window) or 256 lanes (k_row_wg_window) per
// row, column window by column window: the bitmap accumulator of bhs_row_wg.hip.h with a bitmap of one window, the window's
// piece of the row put together in LDS.
#pragma once

namespace bhs {

// ===========================================================================
// The reference sends rows of more than 512 entries through EM_mergepath / EM_mergepath_global (bhsparse_cuda.h:1043-
// 1489): a merge of the row's B rows in shared memory, out to HBM when the row outgrows it, one workgroup per row.
// k_row_bitmap_lds (bhs_row_wg.hip.h) keeps one occupancy bit per COLUMN OF B in LDS instead -- 128 KB for 2^20 columns,
// one row per CU at a time -- and every phase of a row is a handful of dependent round trips to L2 with nothing to hide
// them behind.  That is fine for a row of 10^5 products and poor for the rows a graph has most of: an R-MAT graph of
// 2^20 rows has 38 k rows of 2 k .. 8 k entries of C (12 entries of A times B rows of 250, typically), and that kernel
// spent 54 us on each.  What such rows cost is (a) the round trips and (b) requests: one 4-byte store per column from the
// ordered sweep, one 8-byte store or fp64 atomic per product, each its own 64-byte request to L2 (measured on this
// kernel's first form, 125 M products: 3.0 ms as a whole, 2.1 with plain stores for the atomics, 1.4 without the
// products' stores, 0.66 without any store).
//
// Here:
//   * the columns are cut into WINDOWS of at most 2^16 columns (an 8 KB bitmap) and a wave walks its row window by
//     window: no barrier anywhere, nine rows in flight per CU;
//   * the windows hold about equal shares of B's entries, not equal shares of the columns (k_window_hist /
//     k_window_pick: a graph's columns are anything but uniform -- with equal widths the first window of an R-MAT row
//     held a third of it);
//   * an index of B built at the start of the multiply (k_b_windows16) says where each window begins in every row of B
//     (rows of B ascend, so a window's entries are one contiguous piece of the row): a window's products are a flat space
//     over the pieces, every product is loaded once per pass, and a lane reads its B row's window starts one window ahead;
//   * pass 1 sets bits; ONE wave scan of the lanes' totals ranks everything (a lane owns 32 consecutive bitmap words,
//     stored group-swizzled so that the lanes' 16-byte reads spread over the banks); pass 2 adds every product to its
//     entry's value IN LDS (ds_add_f64) and writes the entry's column beside it -- no sweep over the bitmap -- and both
//     arrays leave 64 lanes wide.  A window with more entries than the staging arrays hold is done in rounds by rank;
//   * a window of at most 256 products (most of them) is worked on in two halves, one window apart: its products are
//     found and requested while the window before it is finished, and stay in registers -- pass 2 neither searches nor
//     loads.
// ===========================================================================
constexpr int kWwLog2 = 16;                                        // most columns per window
constexpr int kWwWords = 1 << (kWwLog2 - 5);                       // 2048 bitmap words
constexpr int kWwBucketLog2 = 12;                                  // window boundaries are multiples of 4096 columns
constexpr int kWwBuckets = 256;                                    // ... so at most 2^20 columns
constexpr int kWwMax = 32;                                         // most windows: 15 cuts by share of B's entries + 16 by width
constexpr int kWwU = 4;                                            // products per lane and batch (one wave per row)
#ifndef BHS_WG_U
#define BHS_WG_U 8
#endif
constexpr int kWgU = BHS_WG_U;                                     // ... (256 lanes per row)
#ifndef BHS_WW_CAP
#define BHS_WW_CAP 512                                             // (R-MAT, rows of 2 k .. 8 k entries: 256 1.61 ms, 384 1.58, 512 1.54, 640 1.68)
#endif
constexpr int kWwCap = BHS_WW_CAP;                                 // entries of C staged per round
constexpr int kWwSpillA = 128;                                     // longest row of A this kernel keeps
constexpr int kWwSpillWin = 2048;                                  // most products of one window it keeps
constexpr int kWwStride = kWwMax + 2;                              // 16-bit offsets per row of B in the index (even: a row starts on a 4-byte boundary)
constexpr int kWwTabInts = kWwMax + 2;                             // device table: the number of windows, their first columns, n
#ifndef BHS_WW_LAB
#define BHS_WW_LAB 0                                               // measurement switches: 2 no pass 2, 4 no write-out
#endif

#if BHS_PHASES_SPA
#define BHS_TICK_WW(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); phw[i] += t__ - tWw; tWw = t__; } while (0)
#else
#define BHS_TICK_WW(i) do { } while (0)
#endif

// entries of B per bucket of 4096 columns
__global__ __launch_bounds__(256) void k_window_hist(long long nnzB, const int* __restrict__ Bj, unsigned* __restrict__ hist)
{
    __shared__ unsigned h[kWwBuckets];
    h[threadIdx.x] = 0u;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnzB; i += (long long)gridDim.x * 256)
        atomicAdd(&h[min(Bj[i] >> kWwBucketLog2, kWwBuckets - 1)], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// the windows: a new one begins when the one at hand holds a fifteenth of B's entries or is 2^16 columns wide (at most
// 15 + 16 such places in 256 buckets: 32 windows)
// tab[0] = number of windows, tab[1 + w] = first column of window w, tab[1 + nWin] = n
__global__ __launch_bounds__(64) void k_window_pick(int n, long long nnzB, const unsigned* __restrict__ hist, int* __restrict__ tab)
{
    if (threadIdx.x != 0) return;
    const int nb = (n + (1 << kWwBucketLog2) - 1) >> kWwBucketLog2;
    const long long target = (nnzB + 14) / 15;
    int nWin = 0, width = 0;
    long long acc = 0;
    tab[1] = 0;
    for (int b = 0; b < nb; ++b) {
        acc += hist[b];
        ++width;
        if (b + 1 < nb && (acc >= target || width == (1 << (kWwLog2 - kWwBucketLog2))) && nWin + 1 < kWwMax) {
            ++nWin;
            tab[1 + nWin] = (b + 1) << kWwBucketLog2;
            acc = 0;
            width = 0;
        }
    }
    ++nWin;
    tab[1 + nWin] = n;
    tab[0] = nWin;
}

// where the windows begin in every row of B, as 16-bit offsets from the row's start (rows of B are shorter than 2^16):
// win[r * kWwStride + w] = entries of row r with a column before window w (w = 0 .. kWwMax, the tail repeats the row's length)
__global__ __launch_bounds__(256) void k_b_windows16(int k, const int* __restrict__ tab, const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                     unsigned short* __restrict__ win)
{
    // (a row's offsets are staged in LDS and leave as one contiguous piece per block)
    __shared__ unsigned short stage[256 * kWwStride + 2];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int nWin = tab[0];
    if (r < k) {
        const int s = Bp[r], e = Bp[r + 1];
        unsigned short* out = stage + threadIdx.x * kWwStride;
        int pos = s;
        out[0] = 0;
        for (int w = 1; w < nWin; ++w) {
            const int lim = tab[1 + w];
            if (e - pos <= 8) {
                while (pos < e && Bj[pos] < lim) ++pos;
            } else {
                int lo = pos, hi = e;                              // first entry in [pos, e) with column >= lim
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (Bj[mid] < lim) lo = mid + 1; else hi = mid; }
                pos = lo;
            }
            out[w] = (unsigned short)(pos - s);
        }
        for (int w = nWin; w < kWwStride; ++w) out[w] = (unsigned short)(e - s);
    }
    __syncthreads();
    const int rows = min(256, k - blockIdx.x * 256);
    const unsigned* src = reinterpret_cast<const unsigned*>(stage);
    unsigned* dst = reinterpret_cast<unsigned*>(win + (size_t)blockIdx.x * 256 * kWwStride);
    const int n32 = rows * kWwStride / 2;
    for (int i = threadIdx.x; i < n32; i += 256) dst[i] = src[i];
}

// bitmap, rank of every 8-word group + the eight words' offsets from it (one byte each), chunk arrays, staged columns and values
constexpr size_t wave_window_smem() { return (size_t)kWwWords * 4 + (size_t)(kWwWords / 8) * 12 + 2 * 64 * sizeof(int) + (size_t)kWwCap * (sizeof(int) + sizeof(acc_t)); }

// word x of a wave's bitmap lives at: its lane's block (x >> 5), the 8-word group rotated by the block number, the word
__device__ __forceinline__ int ww_phys(int x) { return (x & ~31) | ((((x >> 3) ^ (x >> 5)) & 3) << 3) | (x & 7); }

__global__ __launch_bounds__(64) void k_row_wave_window(
    const int4* __restrict__ desc, int qn, const int* __restrict__ tab, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const unsigned short* __restrict__ Bwin, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    int* __restrict__ Cj, value_t* __restrict__ Cx, int* __restrict__ ticket, int reverse,
    int4* __restrict__ spill)                                  // spill[0].x: rows handed on, spill[1 ..]: their descriptors
{
    constexpr int U = kWwU, CAP = kWwCap;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    acc_t* vals = reinterpret_cast<acc_t*>(smemRaw);               // (first: 8-byte aligned whatever follows)
    unsigned* bm = reinterpret_cast<unsigned*>(vals + CAP);
    int* rank8 = reinterpret_cast<int*>(bm + kWwWords);
    uint2* sub8 = reinterpret_cast<uint2*>(rank8 + kWwWords / 8);      // entries before each word of a group, from the group's first (<= 224: a byte)
    int* sIncl = reinterpret_cast<int*>(sub8 + kWwWords / 8);
    int* sBase = sIncl + 64;
    int* cols = sBase + 64;
    const int lane = threadIdx.x;
    const int nWin = __builtin_amdgcn_readfirstlane(tab[0]);
    for (int i = lane; i < kWwWords / 4; i += 64) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    wave_sync();
#if BHS_PHASES_SPA
    unsigned long long phw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tWw = __builtin_readcyclecounter();
#endif

    for (;;) {
        int q = 0;
        if (lane == 0) q = atomicAdd(ticket, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= qn) break;
        const int4 d = desc[reverse ? qn - 1 - q : q];
        const int a0 = d.y, a1 = d.z;
        // A row of A with hundreds of entries and short B rows (a hub of a web graph) is the wrong shape for windows --
        // entries x windows pieces to look up, nearly all of them empty: it goes on to k_row_bitmap_lds, which walks
        // whole B rows, through the spill list.
        if (a1 - a0 > kWwSpillA) {
            if (lane == 0) spill[1 + atomicAdd(&spill[0].x, 1)] = d;
            wave_sync();                                           // (lane-0 work never next to the loop's back edge: see the end of the loop)
            continue;
        }
        // ... and so does a row whose products crowd into one window (a web graph's rows: neighbouring pages): thousands
        // of products of one window want the 1024 lanes of that kernel, not rounds of 384 entries by one wave.  The
        // windows' product counts come from the index alone: 17 loads per entry of A, 32 wave sums.
        {
            unsigned pk[kWwStride / 2];
            int len[kWwMax];
#pragma unroll
            for (int w = 0; w < kWwMax; ++w) len[w] = 0;
            for (int ca = a0; ca < a1; ca += 64) {
                if (ca + lane < a1) {
                    const unsigned* src = reinterpret_cast<const unsigned*>(Bwin + (size_t)Aj[ca + lane] * kWwStride);
#pragma unroll
                    for (int i = 0; i < kWwStride / 2; ++i) pk[i] = src[i];
#pragma unroll
                    for (int w = 0; w < kWwMax; ++w) {
                        const unsigned lo = w & 1 ? pk[w / 2] >> 16 : pk[w / 2] & 0xFFFFu;
                        const unsigned hi = (w + 1) & 1 ? pk[(w + 1) / 2] >> 16 : pk[(w + 1) / 2] & 0xFFFFu;
                        len[w] += (int)(hi - lo);
                    }
                }
            }
            int most = 0;
#pragma unroll
            for (int w = 0; w < kWwMax; ++w) most = max(most, wave_sum_dpp(len[w]));
            if (most > kWwSpillWin) {
                if (lane == 0) spill[1 + atomicAdd(&spill[0].x, 1)] = d;
                wave_sync();
                continue;
            }
        }
        long long base = (long long)d.w;                           // where the window at hand begins in the row of C
        // a row of A of at most 64 entries (nearly all of them) is read once: a lane keeps its entry's value and walks the
        // window starts of its B row one load ahead of the window at hand
        const bool single = a1 - a0 <= 64;
        acc_t av = 0.0;                                            // this lane's A value of the chunk at hand
        const unsigned short* wptr = Bwin;
        int rowB = 0, cur = 0, nxt = 0;
        if (single && a0 + lane < a1) {
            const int c = Aj[a0 + lane];
            wptr = Bwin + (size_t)c * kWwStride;
            rowB = Bp[c];
            av = (acc_t)Ax[a0 + lane];
            nxt = wptr[1];
        }
        BHS_TICK_WW(0);
        // the window's pieces of 64 B rows -> a flat product space (sIncl / sBase); returns its size
        auto stage_chunk = [&](int w, int ca, int sb0, int slen) {
            int b0 = sb0, len = slen;
            if (!single) {
                b0 = len = 0;
                av = 0.0;
                if (ca + lane < a1) {
                    const int c = Aj[ca + lane];
                    const unsigned short* src = Bwin + (size_t)c * kWwStride + w;
                    b0 = Bp[c] + src[0];
                    len = (int)src[1] - (int)src[0];
                    av = (acc_t)Ax[ca + lane];
                }
            }
            const int incl = wave_incl_scan_dpp(len);
            sIncl[lane] = incl;
            sBase[lane] = b0 - (incl - len);
            wave_sync();
            return __builtin_amdgcn_readlane(incl, 63);
        };
        auto find = [&](int p) {                                   // first entry j with sIncl[j] > p
            int l = 0, r = 63;
#pragma unroll
            for (int t = 0; t < 6; ++t) { const int mid = (l + r) >> 1; if (sIncl[mid] > p) r = mid; else l = mid + 1; }
            return l;
        };
        // A window of a single-chunk row with at most 64 * UP products is worked on in two halves, and the second half of
        // window w runs AFTER the first half of window w + 1: `issue` finds the products and requests their columns and B
        // values -- no branch around the loads, so that the wait for this set's data is a count, not a drain -- and `finish`,
        // one window later, finds them arrived.  (Phase timers of the form without this: 65 % of the kernel between the
        // request and the data.)
        constexpr int UP = 4;
        struct WinSet { int w, sb0, slen, total; int kc[UP], ke[UP]; acc_t kv[UP]; };
        auto issue = [&](WinSet& S, int w) {                       // (w >= nWin: an empty window -- the loop below never branches around an issue)
            S.w = w;
            S.sb0 = rowB + cur;
            S.slen = w < nWin ? nxt - cur : 0;
            cur = nxt;
            {
                const int t = wptr[min(w + 2, kWwMax + 1)];        // (the index repeats the row's length to its end)
                if (a0 + lane < a1) nxt = t;
            }
            S.total = stage_chunk(w, a0, S.sb0, S.slen);
            long long idx[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int p = u * 64 + lane;
                S.ke[u] = p < S.total ? find(p) : 0;
                idx[u] = p < S.total ? (long long)sBase[S.ke[u]] + p : 0;
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int c = Bj[idx[u]];
                S.kv[u] = (acc_t)Bx[idx[u]];
                S.kc[u] = u * 64 + lane < S.total ? c : -1;
            }
            wave_sync();
        };
        auto finish = [&](WinSet& S) {
            const int w = S.w;
            if (w >= nWin) return;
            const int colBase = tab[1 + w], words = (tab[2 + w] - colBase + 31) >> 5;   // (wave-uniform: scalar loads)
            const bool kept = single && S.total <= 64 * UP;
            int kc[U], ke[U];
            acc_t kv[U];
            int total = 0, rowTotal = 0;
            // one batch of a chunk's products (the windows that are not kept): column within the window (-1: none), A value x B value
            auto load_batch = [&](int p0, bool values) {
                long long idx[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * 64 + lane;
                    ke[u] = p < total ? find(p) : 0;
                    idx[u] = (long long)sBase[ke[u]] + p;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool on = p0 + u * 64 + lane < total;
                    kc[u] = on ? Bj[idx[u]] - colBase : -1;
                    kv[u] = on && values ? (acc_t)Bx[idx[u]] : (acc_t)0;
                }
                if (values) {
#pragma unroll
                    for (int u = 0; u < U; ++u) kv[u] *= __shfl(av, ke[u], 64);   // (product formed in acc_t)
                }
            };
            // ---- pass 1: occupancy bits
            if (kept) {
                rowTotal = S.total;
                if (rowTotal == 0) return;
#pragma unroll
                for (int u = 0; u < UP; ++u) {
                    S.kv[u] *= __shfl(av, S.ke[u], 64);
                    if (S.kc[u] >= 0) {
                        S.kc[u] -= colBase;
                        atomicOr(&bm[ww_phys(S.kc[u] >> 5)], 1u << (S.kc[u] & 31));
                    }
                }
            } else {
                for (int ca = a0; ca < a1; ca += 64) {
                    total = stage_chunk(w, ca, S.sb0, S.slen);
                    rowTotal += total;
                    for (int p0 = 0; p0 < total; p0 += 64 * U) {
                        load_batch(p0, false);
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (kc[u] >= 0) atomicOr(&bm[ww_phys(kc[u] >> 5)], 1u << (kc[u] & 31));
                    }
                    wave_sync();                                   // (the next chunk overwrites sIncl / sBase)
                }
                if (rowTotal == 0) return;
            }
            BHS_TICK_WW(1);
            // ---- the lanes' totals, their ranks, the rank of every 8-word group and of every word within it
            const int blk = lane * 32, rot = lane & 3;
            int cnt4[4] = {0, 0, 0, 0}, mine = 0;
            if (blk < words) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint4 lo = *reinterpret_cast<const uint4*>(&bm[blk + ((j ^ rot) << 3)]);
                    const uint4 hi = *reinterpret_cast<const uint4*>(&bm[blk + ((j ^ rot) << 3) + 4]);
                    const int c0 = __popc(lo.x), c1 = c0 + __popc(lo.y), c2 = c1 + __popc(lo.z), c3 = c2 + __popc(lo.w);
                    const int c4 = c3 + __popc(hi.x), c5 = c4 + __popc(hi.y), c6 = c5 + __popc(hi.z);
                    cnt4[j] = c6 + __popc(hi.w);
                    sub8[lane * 4 + j] = make_uint2((unsigned)(c0 << 8 | c1 << 16 | c2 << 24), (unsigned)(c3 | c4 << 8 | c5 << 16 | c6 << 24));
                    mine += cnt4[j];
                }
            }
            const int incl = wave_incl_scan_dpp(mine);
            const int winCount = __builtin_amdgcn_readlane(incl, 63);
            const int first = incl - mine;                         // rank of this lane's first column
            *reinterpret_cast<int4*>(&rank8[lane * 4]) = make_int4(first, first + cnt4[0], first + cnt4[0] + cnt4[1], first + cnt4[0] + cnt4[1] + cnt4[2]);
            BHS_TICK_WW(2);
            // rank of column c of the window (its bit is set): its group's, its word's offset in the group, the bits below it
            auto rank_of = [&](int c) {
                const int wd = c >> 5, grp = wd >> 3, k = wd & 7;
                const uint2 sb = sub8[grp];
                const unsigned word = bm[ww_phys(wd)];
                const unsigned off = ((k & 4 ? sb.y : sb.x) >> ((k & 3) * 8)) & 255u;
                return rank8[grp] + (int)off + __popc(word & ((1u << (c & 31)) - 1u));
            };
            wave_sync();
            // ---- rounds of CAP entries: zeroed values in LDS, every product added to its entry, out in full-width stores
            for (int r0 = 0; r0 < winCount; r0 += CAP) {
                const int nr = min(CAP, winCount - r0);
                for (int i = lane; i < nr; i += 64) vals[i] = 0.0;
                wave_sync();
                BHS_TICK_WW(3);
                auto add = [&](int c, acc_t v) {
                    const int pos = rank_of(c) - r0;
                    if ((unsigned)pos < (unsigned)CAP) {
                        cols[pos] = colBase + c;                   // (every product of the entry writes the same column: no sweep over the bitmap)
                        unsafeAtomicAdd(&vals[pos], v);
                    }
                };
                if (BHS_WW_LAB & 2) {
                } else if (kept) {
#pragma unroll
                    for (int u = 0; u < UP; ++u)
                        if (S.kc[u] >= 0) add(S.kc[u], S.kv[u]);
                } else {
                    for (int ca = a0; ca < a1; ca += 64) {
                        total = stage_chunk(w, ca, S.sb0, S.slen);
                        for (int p0 = 0; p0 < total; p0 += 64 * U) {
                            load_batch(p0, true);
#pragma unroll
                            for (int u = 0; u < U; ++u)
                                if (kc[u] >= 0) add(kc[u], kv[u]);
                        }
                        wave_sync();
                    }
                }
                wave_sync();
                BHS_TICK_WW(4);
                if (!(BHS_WW_LAB & 4)) {
                    for (int i = lane; i < nr; i += 64) {
                        Cj[base + r0 + i] = cols[i];
                        Cx[base + r0 + i] = (value_t)vals[i];
                    }
                }
                wave_sync();
                BHS_TICK_WW(5);
            }
            base += winCount;
            // ---- leave the bitmap clean
            for (int i = lane; i * 4 < ((words + 31) & ~31); i += 64) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);   // (whole blocks: the words are swizzled within them)
            wave_sync();
            BHS_TICK_WW(6);
        };
        WinSet A, B;
        if (single) {
            issue(A, 0);
            for (int w = 0; w < nWin; w += 2) {
                issue(B, w + 1);
                finish(A);
                issue(A, w + 2);
                finish(B);
            }
        } else {
            for (int w = 0; w < nWin; ++w) {
                A.w = w; A.sb0 = 0; A.slen = 0; A.total = 0x7fffffff;
                finish(A);
            }
        }
        // (A convergent operation between whatever lane 0 does alone at the end of a row -- the hand-over above -- and its
        // ticket draw at the top of the next: without one hipcc folds the two into ONE divergent region around the loop's
        // back edge, lane 0 in an outer loop and the other 63 in an inner one where readfirstlane finds the ticket of a
        // lane that never drew: the wave takes row 0 forever.  Seen twice in this file; the ISA shows the two loop nests.)
        wave_sync();
    }
#if BHS_PHASES_SPA
    if (lane == 0) for (int i = 0; i < 7; ++i) atomicAdd(&g_phase_cycles[i], phw[i]);
#endif
}

// ===========================================================================
// The long rows (more than 8 k entries of C) the same way, 256 lanes per row: where a multiply has thousands of them (the
// R-MAT graph: 14 k rows, 232 M products) four rows per CU in flight beat k_row_bitmap_lds's one.  A lane owns ONE
// 8-word group of the window's bitmap (no swizzle needed: the lanes' 32-byte pieces are consecutive), the ranks come
// from a block scan, the staging arrays hold 2048 entries, a window of up to 2048 products stays in registers.  Rows of A
// beyond 512 entries and rows with more than 16 k products in one window go on to k_row_bitmap_lds.
// ===========================================================================
#ifndef BHS_WG_LANES
#define BHS_WG_LANES 256
#define BHS_WG_CAP 3072                                            // (2048: R-MAT's long rows 3.79 ms, 3072: 3.56, 1024: 4.33; 128 lanes: 4.8-5.3; 3 workgroups per CU either way)
#endif
constexpr int kWgLanes = BHS_WG_LANES, kWgCap = BHS_WG_CAP, kWgSpillA = 2 * BHS_WG_LANES, kWgSpillWin = 64 * BHS_WG_LANES;

static_assert((kWwWords / 8) % kWgLanes == 0, "whole groups per lane");
constexpr size_t wg_window_smem()
{
    return (size_t)kWwWords * 4 + (size_t)(kWwWords / 8) * 12 + 2 * kWgLanes * sizeof(int) + kWgLanes * sizeof(acc_t) + (size_t)kWgCap * (sizeof(int) + sizeof(acc_t)) +
           (8 + (kWgLanes / 64) * kWwMax) * sizeof(int);
}

__global__ __launch_bounds__(kWgLanes) void k_row_wg_window(   // (133 VGPRs: three workgroups per CU; held to 128 it spills and gains nothing)

    const int4* __restrict__ desc, int qn, const int* __restrict__ tab, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const unsigned short* __restrict__ Bwin, const int* __restrict__ Bj, const value_t* __restrict__ Bx,
    int* __restrict__ Cj, value_t* __restrict__ Cx, int* __restrict__ ticket, int reverse, int4* __restrict__ spill)
{
    constexpr int L = kWgLanes, NW = L / 64, U = kWgU, CAP = kWgCap;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    acc_t* vals = reinterpret_cast<acc_t*>(smemRaw);
    acc_t* sAv = vals + CAP;                                       // A values of the chunk at hand
    unsigned* bm = reinterpret_cast<unsigned*>(sAv + L);
    int* rank8 = reinterpret_cast<int*>(bm + kWwWords);
    uint2* sub8 = reinterpret_cast<uint2*>(rank8 + kWwWords / 8);
    int* sIncl = reinterpret_cast<int*>(sub8 + kWwWords / 8);
    int* sBase = sIncl + L;
    int* cols = sBase + L;
    int* misc = cols + CAP;                                        // [0] ticket, [1 .. NW] wave totals, then NW x kWwMax window sums
    int* winSum = misc + 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nWin = tab[0];
    for (int i = tid; i < kWwWords / 4; i += L) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // inclusive scan over the workgroup; total = the sum
    auto block_scan = [&](int v, int& total) {
        int incl = wave_incl_scan_dpp(v);
        if (lane == 63) misc[1 + wv] = incl;
        __syncthreads();
        int off = 0;
        total = 0;
#pragma unroll
        for (int t = 0; t < NW; ++t) {
            const int x = misc[1 + t];
            if (t < wv) off += x;
            total += x;
        }
        __syncthreads();
        return incl + off;
    };

    for (;;) {
        if (tid == 0) misc[0] = atomicAdd(ticket, 1);
        __syncthreads();
        const int q = misc[0];
        __syncthreads();
        if (q >= qn) break;
        const int4 d = desc[reverse ? qn - 1 - q : q];
        const int a0 = d.y, a1 = d.z;
        // products per window from the index (as in k_row_wave_window); rows of the wrong shape go on
        int most = 0x7fffffff;
        if (a1 - a0 <= kWgSpillA) {
            unsigned pk[kWwStride / 2];
            int len[kWwMax];
#pragma unroll
            for (int w = 0; w < kWwMax; ++w) len[w] = 0;
            for (int ca = a0; ca < a1; ca += L) {
                if (ca + tid < a1) {
                    const unsigned* src = reinterpret_cast<const unsigned*>(Bwin + (size_t)Aj[ca + tid] * kWwStride);
#pragma unroll
                    for (int i = 0; i < kWwStride / 2; ++i) pk[i] = src[i];
#pragma unroll
                    for (int w = 0; w < kWwMax; ++w) {
                        const unsigned lo = w & 1 ? pk[w / 2] >> 16 : pk[w / 2] & 0xFFFFu;
                        const unsigned hi = (w + 1) & 1 ? pk[(w + 1) / 2] >> 16 : pk[(w + 1) / 2] & 0xFFFFu;
                        len[w] += (int)(hi - lo);
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < kWwMax; ++w) {
                const int sum = wave_sum_dpp(len[w]);
                if (lane == 0) winSum[wv * kWwMax + w] = sum;
            }
            __syncthreads();
            most = 0;
            for (int w = 0; w < kWwMax; ++w) {
                int sum = 0;
#pragma unroll
                for (int t = 0; t < NW; ++t) sum += winSum[t * kWwMax + w];
                most = max(most, sum);
            }
        }
        if (most > kWgSpillWin) {
            if (tid == 0) spill[1 + atomicAdd(&spill[0].x, 1)] = d;
            __syncthreads();                                       // (lane-0 work never next to the loop's back edge: see the end of the loop)
            continue;
        }
        long long base = (long long)d.w;
        const bool single = a1 - a0 <= L;
        const unsigned short* wptr = Bwin;
        int rowB = 0, cur = 0, nxt = 0;
        if (single) {
            sAv[tid] = 0.0;
            if (a0 + tid < a1) {
                const int c = Aj[a0 + tid];
                wptr = Bwin + (size_t)c * kWwStride;
                rowB = Bp[c];
                sAv[tid] = (acc_t)Ax[a0 + tid];
                nxt = wptr[1];
            }
        }
        for (int w = 0; w < nWin; ++w) {
            const int colBase = tab[1 + w];
            int total = 0, rowTotal = 0;
            int kc[U], ke[U];
            acc_t kv[U];
            const int sb0 = rowB + cur, slen = nxt - cur;
            cur = nxt;
            if (single && a0 + tid < a1 && w + 2 <= nWin) nxt = wptr[w + 2];
            auto stage_chunk = [&](int ca) {
                int b0 = sb0, len = slen;
                if (!single) {
                    b0 = len = 0;
                    acc_t a = 0.0;
                    if (ca + tid < a1) {
                        const int c = Aj[ca + tid];
                        const unsigned short* src = Bwin + (size_t)c * kWwStride + w;
                        b0 = Bp[c] + src[0];
                        len = (int)src[1] - (int)src[0];
                        a = (acc_t)Ax[ca + tid];
                    }
                    sAv[tid] = a;
                }
                int tot;
                const int incl = block_scan(len, tot);
                sIncl[tid] = incl;
                sBase[tid] = b0 - (incl - len);
                __syncthreads();
                return tot;
            };
            auto find = [&](int p) {                               // first entry j with sIncl[j] > p
                int l = 0, r = L - 1;
#pragma unroll
                for (int t = 0; t < 8; ++t) { const int mid = (l + r) >> 1; if (sIncl[mid] > p) r = mid; else l = mid + 1; }
                return l;
            };
            auto load_batch = [&](int p0, bool values) {
                long long idx[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * L + tid;
                    ke[u] = p < total ? find(p) : 0;
                    idx[u] = (long long)sBase[ke[u]] + p;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool on = p0 + u * L + tid < total;
                    kc[u] = on ? Bj[idx[u]] - colBase : -1;
                    kv[u] = on && values ? (acc_t)Bx[idx[u]] : (acc_t)0;
                }
                if (values) {
#pragma unroll
                    for (int u = 0; u < U; ++u) kv[u] *= sAv[ke[u]];
                }
            };
            // ---- pass 1: occupancy bits
            for (int ca = a0; ca < a1; ca += L) {
                total = stage_chunk(ca);
                rowTotal += total;
                for (int p0 = 0; p0 < total; p0 += L * U) {
                    load_batch(p0, single && total <= L * U);
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (kc[u] >= 0) atomicOr(&bm[kc[u] >> 5], 1u << (kc[u] & 31));
                }
                __syncthreads();
            }
            if (rowTotal == 0) continue;                           // (uniform: every lane has the same totals)
            const bool kept = single && rowTotal <= L * U;
            // ---- a lane's groups (256 / L of them, consecutive): their words' offsets, the lane's total; ranks by a block scan
            constexpr int GP = (kWwWords / 8) / L;
            int mine = 0, cntG[GP];
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const int grp = tid * GP + g;
                const uint4 lo = *reinterpret_cast<const uint4*>(&bm[grp * 8]);
                const uint4 hi = *reinterpret_cast<const uint4*>(&bm[grp * 8 + 4]);
                const int c0 = __popc(lo.x), c1 = c0 + __popc(lo.y), c2 = c1 + __popc(lo.z), c3 = c2 + __popc(lo.w);
                const int c4 = c3 + __popc(hi.x), c5 = c4 + __popc(hi.y), c6 = c5 + __popc(hi.z);
                cntG[g] = c6 + __popc(hi.w);
                sub8[grp] = make_uint2((unsigned)(c0 << 8 | c1 << 16 | c2 << 24), (unsigned)(c3 | c4 << 8 | c5 << 16 | c6 << 24));
                mine += cntG[g];
            }
            int winCount;
            const int incl = block_scan(mine, winCount);
            {
                int r = incl - mine;
#pragma unroll
                for (int g = 0; g < GP; ++g) { rank8[tid * GP + g] = r; r += cntG[g]; }
            }
            __syncthreads();
            auto rank_of = [&](int c) {
                const int wd = c >> 5, grp = wd >> 3, k = wd & 7;
                const uint2 sb = sub8[grp];
                const unsigned word = bm[wd];
                const unsigned off = ((k & 4 ? sb.y : sb.x) >> ((k & 3) * 8)) & 255u;
                return rank8[grp] + (int)off + __popc(word & ((1u << (c & 31)) - 1u));
            };
            for (int r0 = 0; r0 < winCount; r0 += CAP) {
                const int nr = min(CAP, winCount - r0);
                for (int i = tid; i < nr; i += L) vals[i] = 0.0;
                __syncthreads();
                auto add = [&](int c, acc_t v) {
                    const int pos = rank_of(c) - r0;
                    if ((unsigned)pos < (unsigned)CAP) {
                        cols[pos] = colBase + c;
                        unsafeAtomicAdd(&vals[pos], v);
                    }
                };
                if (kept) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (kc[u] >= 0) add(kc[u], kv[u]);
                } else {
                    for (int ca = a0; ca < a1; ca += L) {
                        total = stage_chunk(ca);
                        for (int p0 = 0; p0 < total; p0 += L * U) {
                            load_batch(p0, true);
#pragma unroll
                            for (int u = 0; u < U; ++u)
                                if (kc[u] >= 0) add(kc[u], kv[u]);
                        }
                        __syncthreads();
                    }
                }
                __syncthreads();
                for (int i = tid; i < nr; i += L) {
                    Cj[base + r0 + i] = cols[i];
                    Cx[base + r0 + i] = (value_t)vals[i];
                }
                __syncthreads();
            }
            base += winCount;
            for (int i = tid; i < kWwWords / 4; i += L) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        // (A barrier between whatever lane 0 does alone at the end of a row and its ticket draw at the top of the next:
        // without one hipcc folds the two into ONE divergent region around the loop's back edge -- lane 0 in an outer
        // loop, everyone else in an inner one that meets the ticket's barriers again before lane 0 has drawn, and the
        // workgroup takes its row forever.  Seen in this kernel's first form; `llvm-objdump` shows the two loop nests.)
        __syncthreads();
    }
}

}  // namespace bhs
