// bhs_hub.hip.h -- hub rows: ONE row of C assembled by MANY workgroups.
//
// Every other kernel of the pipeline gives a row to one lane, wave or workgroup.  A row with millions of
// intermediate products (the hub pages of a web graph, a dense row of an otherwise sparse matrix) then runs on one
// CU while 255 wait.  The rows of the hub bin (BinSpec::hubMin products or more) are instead cut into ITEMS of
// about kHubItemProducts products, the items are handed out to the whole device, and the row's accumulator is a
// bitmap slot in HBM that all of its workgroups share (the algorithm of k_row_spa, bhs_kernels.hip.h, across
// workgroups; kernel boundaries are the device-wide barriers between its phases):
//   k_hub_plan    per hub row: chunks of kHubChunk A entries, each cut into ceil(products / item) items
//   k_hub_mark    per item: every product sets its column's bit in the row's bitmap (device-scope atomic OR); a bit
//                 that was already set raises the duplicate flag of its 16-column group (numeric stage only)
//   k_hub_count   per (row, bitmap segment): population count -> the row's entry count (symbolic stage) and the
//                 per-segment counts (numeric stage)
//   k_hub_emit    per (row, segment): ordered sweep -- colIndC in ascending order, one rank word per 32 columns,
//                 zeros into the entries of flagged groups
//   k_hub_place   per item: every product goes straight to valC[rowBase + rank[c / 32] + popc(bits below c)] -- a
//                 plain store for columns hit once, an fp64 atomic add for flagged groups
// This replaces, for those rows, the reference's multi-round global merge (EM_mergepath_global,
// SpGEMM_cuda/bhsparse_cuda.h:2270-2525, and its progressive re-allocation loop :2527-2780), which is also the one
// place where the reference spreads a row over more than one thread block.
#pragma once

namespace bhs {

constexpr int kHubChunk = 512;            // A entries per chunk: their B-row lengths are prefix-summed in LDS
constexpr int kHubBlock = 1024;           // lanes of the item kernels and of the bitmap sweeps
constexpr int kHubMaxSeg = 64;            // bitmap segments per row (one workgroup each in count / emit)

// geometry of a slot: nW bitmap words (a multiple of 1024 * seg), then nW / 16 duplicate-flag words
struct HubGeom {
    int nW, seg, segW;
    long long slotWords;                  // bitmap + flags: cleared together
};
inline HubGeom hub_geom(long long ncols)
{
    HubGeom g;
    const long long words = (ncols + 31) / 32;
    int seg = 1;
    while (seg < kHubMaxSeg && (long long)seg * 2 * 4096 <= words) seg *= 2;
    const long long unit = 1024LL * seg;
    g.nW = (int)(((words > 0 ? words : 1) + unit - 1) / unit * unit);
    g.seg = seg;
    g.segW = g.nW / seg;
    g.slotWords = (long long)g.nW + g.nW / 16;
    return g;
}

// ---------------------------------------------------------------------------
// plan: kHubPlanWG 256-lane workgroups per hub row, chunks dealt round robin (items may land in any order)
// ---------------------------------------------------------------------------
constexpr int kHubPlanWG = 32;
__global__ __launch_bounds__(256) void k_hub_plan(const int4* __restrict__ hubQ, const int* __restrict__ Aj,
                                                  const int* __restrict__ Bp, int4* __restrict__ items,
                                                  int* __restrict__ itemCount, int cap, int itemProducts,
                                                  int* __restrict__ cntOut, int* __restrict__ errFlag)
{
    __shared__ int wsum[4];
    __shared__ int sBase;
    const int tid = threadIdx.x;
    const int hub = blockIdx.x / kHubPlanWG, pw = blockIdx.x % kHubPlanWG;
    const int4 d = hubQ[hub];
    if (cntOut && tid == 0 && pw == 0) cntOut[d.x] = 0;      // symbolic stage: the segments add their counts
    for (long long ca64 = (long long)d.y + (long long)pw * kHubChunk; ca64 < d.z; ca64 += (long long)kHubPlanWG * kHubChunk) {
        const int ca = (int)ca64;
        int s = 0;
#pragma unroll
        for (int t = 0; t < kHubChunk / 256; ++t) {
            const int e = ca + t * 256 + tid;
            if (e < d.z && t * 256 + tid < kHubChunk) {
                int2 be;
                __builtin_memcpy(&be, Bp + Aj[e], sizeof(be));
                s += be.y - be.x;
            }
        }
        s = wave_sum_dpp(s);
        if ((tid & 63) == 0) wsum[tid >> 6] = s;
        __syncthreads();
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int parts = (int)(((long long)total + itemProducts - 1) / itemProducts);
        if (tid == 0 && parts) sBase = atomicAdd(itemCount, parts);
        __syncthreads();
        if (parts) {
            const int base = sBase;
            if (base + parts > cap) { if (tid == 0) atomicOr(errFlag, 1); }
            else for (int k = tid; k < parts; k += 256) items[base + k] = make_int4(hub, ca, k, parts);
        }
        __syncthreads();
    }
}

// The products [pBeg, pEnd) of one item: the chunk's B-row lengths are prefix-summed in LDS, product p of the
// chunk belongs to the first A entry l with incl[l] > p (binary search) and is entry sBase[l] + p of B.
// f(valid, column, index into B, index into A), called by every lane of a wave that holds at least one product
template <typename F>
__device__ __forceinline__ void hub_item_products(const int4 item, const int4 d, const int* __restrict__ Aj,
                                                  const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                  int* sIncl, int* sBase, int* wtot, F&& f)
{
    constexpr int BLOCK = kHubBlock, CH = kHubChunk, U = 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ca = item.y;
    const int e = ca + tid;
    int b0 = 0, len = 0;
    if (tid < CH && e < d.z) {
        int2 be;
        __builtin_memcpy(&be, Bp + Aj[e], sizeof(be));
        b0 = be.x;
        len = be.y - be.x;
    }
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
    const int pBeg = (int)((long long)total * item.z / item.w);
    const int pEnd = (int)((long long)total * (item.z + 1) / item.w);
    for (int p0 = pBeg; p0 < pEnd; p0 += BLOCK * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * BLOCK + tid;
            const bool valid = p < pEnd;
            int l = 0, c = -1;
            long long idx = 0;
            if (valid) {
                int r = CH - 1;
                while (l < r) { const int mid = (l + r) >> 1; if (sIncl[mid] > p) r = mid; else l = mid + 1; }
                idx = (long long)sBase[l] + p;
                c = Bj[idx];
            }
            if (p0 + u * BLOCK + (tid & ~63) < pEnd) f(valid, c, idx, ca + l);    // whole waves: f may use cross-lane operations
        }
    }
    __syncthreads();
}

// persistent grid over the item list (its length is read on the device)
template <bool NUM>
__global__ __launch_bounds__(kHubBlock) void k_hub_mark(const int4* __restrict__ items,
                                                        const int* __restrict__ itemCount,
                                                        const int4* __restrict__ hubQ, const int* __restrict__ Aj,
                                                        const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                        unsigned* __restrict__ slots, long long slotWords, int nW,
                                                        int* __restrict__ ticket, int aggregate)
{
    __shared__ int sIncl[kHubChunk], sBase[kHubChunk], wtot[kHubBlock / 64 + 1];
    const int tid = threadIdx.x;
    const int n = *itemCount;
    for (;;) {
        if (tid == 0) wtot[kHubBlock / 64] = atomicAdd(ticket, 1);
        __syncthreads();
        const int t = wtot[kHubBlock / 64];
        __syncthreads();
        if (t >= n) break;
        const int4 item = items[t];
        unsigned* bits = slots + (size_t)item.x * (size_t)slotWords;
        unsigned* dup = bits + nW;
        hub_item_products(item, hubQ[item.x], Aj, Bp, Bj, sIncl, sBase, wtot, [&](bool valid, int c, long long, int e) {
            const unsigned bit = valid ? 1u << (c & 31) : 0u;
            // Bits only ever go from 0 to 1 in this phase, so a plain (possibly stale) load that already shows
            // the bits is the truth and saves the atomic; one that does not is followed by the atomic, whose
            // return value decides.  Rows with many duplicate columns would otherwise queue on their words.
            if (aggregate) {
                // Neighbouring lanes walk one ascending B row: those that fall into the same bitmap word merge their
                // bits (segmented OR over runs of equal (word, A entry)) and the run's first lane issues one atomic.
                // Inside one strictly ascending B row a run cannot hold a column twice.
                const int lane = threadIdx.x & 63;
                const int key = valid ? (((c >> 5) << 9) | (e & 511)) : -1 - lane;
                unsigned m = bit;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const int k2 = __shfl_down(key, dd, 64);
                    const unsigned m2 = __shfl_down(m, dd, 64);
                    if (lane + dd < 64 && k2 == key) m |= m2;
                }
                const int kp = __shfl_up(key, 1, 64);
                if (valid && (lane == 0 || kp != key)) {
                    const int w = c >> 5;
                    unsigned old = bits[w];
                    if ((old & m) != m) old = atomicOr(&bits[w], m);
                    const unsigned ov = old & m;
                    if (NUM && ov) {
                        const unsigned flags = ((ov & 0xffffu) ? 1u : 0u) | ((ov >> 16) ? 2u : 0u);
                        const unsigned fl = flags << ((w & 15) * 2);
                        if ((dup[w >> 4] & fl) != fl) atomicOr(&dup[w >> 4], fl);
                    }
                }
            } else if (valid) {
#if BHS_HUB_PRECHECK & 1
                unsigned old = bits[c >> 5];
                if (!(old & bit)) old = atomicOr(&bits[c >> 5], bit);
#else
                const unsigned old = atomicOr(&bits[c >> 5], bit);
#endif
                if (NUM && (old & bit)) {
                    const unsigned flag = 1u << ((c >> 4) & 31);
#if BHS_HUB_PRECHECK & 2
                    if (!(dup[c >> 9] & flag))
#endif
                    atomicOr(&dup[c >> 9], flag);
                }
            }
        });
    }
}

// one workgroup per (row, segment); wave wv owns the words [w0 + wv * wpw, w0 + (wv + 1) * wpw) of the segment
template <bool NUM>
__global__ __launch_bounds__(kHubBlock) void k_hub_count(const int4* __restrict__ hubQ,
                                                         const unsigned* __restrict__ slots, long long slotWords,
                                                         int seg, int segW, int* __restrict__ segCnt,
                                                         int* __restrict__ cntOut)
{
    __shared__ int wtot[kHubBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hub = blockIdx.x / seg, sg = blockIdx.x % seg;
    const unsigned* bits = slots + (size_t)hub * (size_t)slotWords + (size_t)sg * segW;
    const int wpw = segW / (kHubBlock / 64);
    int mine = 0;
    for (int i = lane; i < wpw; i += 64) mine += __popc(bits[wv * wpw + i]);
    mine = wave_sum_dpp(mine);
    if (lane == 0) wtot[wv] = mine;
    __syncthreads();
    if (tid == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < kHubBlock / 64; ++w) total += wtot[w];
        if (NUM) segCnt[blockIdx.x] = total;
        else if (total) atomicAdd(&cntOut[hubQ[hub].x], total);
    }
}

__global__ __launch_bounds__(kHubBlock) void k_hub_emit(const int4* __restrict__ hubQ,
                                                        const unsigned* __restrict__ slots, long long slotWords,
                                                        int nW, int seg, int segW, const int* __restrict__ segCnt,
                                                        int* __restrict__ rankBase, int* __restrict__ Cj,
                                                        value_t* __restrict__ Cx)
{
    __shared__ int wtot[kHubBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hub = blockIdx.x / seg, sg = blockIdx.x % seg;
    const unsigned* bits = slots + (size_t)hub * (size_t)slotWords;
    const unsigned* dup = bits + nW;
    int* rank = rankBase + (size_t)hub * (size_t)nW;
    const int wpw = segW / (kHubBlock / 64);
    const int w0 = sg * segW + wv * wpw;
    int mine = 0;
    for (int i = lane; i < wpw; i += 64) mine += __popc(bits[w0 + i]);
    mine = wave_sum_dpp(mine);
    if (lane == 0) wtot[wv] = mine;
    int before = 0;                                           // entries of the row in earlier segments
    for (int s = lane; s < sg; s += 64) before += segCnt[hub * seg + s];
    before = wave_sum_dpp(before);
    __syncthreads();
    int run = before;
#pragma unroll
    for (int w = 0; w < kHubBlock / 64; ++w) if (w < wv) run += wtot[w];
    const long long base = hubQ[hub].w;
    for (int i = 0; i < wpw; i += 64) {
        const int w = w0 + i + lane;
        unsigned mm = bits[w];
        const unsigned dd = (dup[w >> 4] >> ((w & 15) * 2)) & 3u;    // flags of this word's two 16-column halves
        const int cnt = __popc(mm);
        const int incl = wave_incl_scan_dpp(cnt);
        int r = run + incl - cnt;
        if (mm) rank[w] = r;                                   // only occupied words are ever looked up
        while (mm) {
            const int b = __ffs((int)mm) - 1;
            mm &= mm - 1;
            Cj[base + r] = (w << 5) + b;
            if ((dd >> (b >> 4)) & 1u) Cx[base + r] = (value_t)0;
            ++r;
        }
        run += __builtin_amdgcn_readlane(incl, 63);
    }
}

__global__ __launch_bounds__(kHubBlock) void k_hub_place(const int4* __restrict__ items,
                                                         const int* __restrict__ itemCount,
                                                         const int4* __restrict__ hubQ, const int* __restrict__ Aj,
                                                         const value_t* __restrict__ Ax, const int* __restrict__ Bp,
                                                         const int* __restrict__ Bj, const value_t* __restrict__ Bx,
                                                         const unsigned* __restrict__ slots, long long slotWords,
                                                         int nW, const int* __restrict__ rankBase,
                                                         value_t* __restrict__ Cx, int* __restrict__ ticket)
{
    __shared__ int sIncl[kHubChunk], sBase[kHubChunk], wtot[kHubBlock / 64 + 1];
    const int tid = threadIdx.x;
    const int n = *itemCount;
    for (;;) {
        if (tid == 0) wtot[kHubBlock / 64] = atomicAdd(ticket, 1);
        __syncthreads();
        const int t = wtot[kHubBlock / 64];
        __syncthreads();
        if (t >= n) break;
        const int4 item = items[t];
        const int4 d = hubQ[item.x];
        const unsigned* bits = slots + (size_t)item.x * (size_t)slotWords;
        const unsigned* dup = bits + nW;
        const int* rank = rankBase + (size_t)item.x * (size_t)nW;
        const long long base = d.w;
        hub_item_products(item, d, Aj, Bp, Bj, sIncl, sBase, wtot, [&](bool valid, int c, long long idx, int e) {
            if (!valid) return;
            const int w = c >> 5;
            const int pos = rank[w] + __popc(bits[w] & ((1u << (c & 31)) - 1u));
            const value_t v = (value_t)((acc_t)Ax[e] * (acc_t)Bx[idx]);   // product formed in acc_t, narrowed once
            if ((dup[c >> 9] >> ((c >> 4) & 31)) & 1u) unsafeAtomicAdd(&Cx[base + pos], v);
            else Cx[base + pos] = v;                           // the only product of this column
        });
    }
}

}  // namespace bhs
