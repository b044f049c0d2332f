// bhs_class_big.hip.h -- row classes whose product lists do not fit the registers of a wave.
// (Included after bhs_class.hip.h; see there for the classification and the tables of the small classes.)
//
// A grid with several unknowns per node (3-dof elasticity on a 27-point stencil: 81 entries per row, 6561 products and
// 375 entries per row of C) still has one relative pattern per kind of row, so its rows classify -- but a class's
// product list is 26 KB, not 12 words per lane.  For such classes
//   k_class_patterns_big   works the list out like k_class_patterns does (one workgroup per class: the representative
//                          row's products, sorted and made unique -> the relative column list; then every product's
//                          {A entry, B entry, position} word, in A-entry-major order) and leaves it in memory;
//   k_class_numeric_big    a workgroup takes a range of ~200 consecutive rows and goes through them CLASS BY CLASS: the
//                          class's list is copied to LDS once and every wave multiplies rows of that class with it
//                          (per 64 products: one LDS read of the list, the A value and B row start of the entry from
//                          LDS, one gather of B's values -- consecutive lanes read consecutive entries of one B row --
//                          and one ds_add_f64 into the row's accumulators), then the next class of the range.
// Replaces, for the matrices that qualify, the hash kernels (SpGEMM_cuda/bhsparse_cuda.h:210-2780, as
// bhs_class.hip.h does): a class's list is read from memory once per range and class, not once per row.
#pragma once

namespace bhs {

constexpr int kClassBigRangeMax = 224;     // consecutive rows a workgroup takes at a time: the largest multiple of (waves x rows per group) up to this
constexpr int kClassBigPatThreads = 1024;
constexpr int kClassBigSpan = 1 << 18;      // widest span of a class's relative columns that is ranked with a bitmap (32 KB of bits)

// ---------------------------------------------------------------------------
// classInfo[s].z == -2 (k_class_patterns found the class beyond its tables): {A entry (8 bits), B entry (8), position (16)}
// of every product -- of every group of T entries' product, see below -- to bigMap[i * kClassBigMaxP + ..], with
// classBigIdx[s] = i | T << 16; the relative columns to classRel like a small class's.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kClassBigPatThreads) void k_class_patterns_big(const unsigned long long* __restrict__ tableA,
                                                            const int* __restrict__ Ap, const int* __restrict__ Aj,
                                                            const int* __restrict__ Bp, const int* __restrict__ Bj,
                                                            int4* __restrict__ classInfo, int* __restrict__ classBigIdx,
                                                            unsigned* __restrict__ bigMap, int* __restrict__ classRel,
                                                            int* __restrict__ stats)
{
    extern __shared__ int smemBig[];                               // keys[kClassBigMaxP], srt[kClassBigMaxP]
    int* keys = smemBig;
    int* srt = smemBig + kClassBigMaxP;
    __shared__ int sIncl[kClassMaxRowBig], sB0[kClassMaxRowBig], scan[kClassBigPatThreads], ulist[kClassMaxNnz], sIdx, sGroup[5];
    constexpr int NT = kClassBigPatThreads;
    const int tid = threadIdx.x, s = blockIdx.x;
    if (tableA[s] == kClassEmpty) return;
    const int4 ci = classInfo[s];
    if (ci.z != -2) return;
    const int rep = ci.w, nA = ci.x;
    const int a0 = Ap[rep];
    auto fail = [&]() { if (tid == 0) { classInfo[s] = make_int4(nA, 0, -1, rep); atomicOr(&stats[CS_FLAGS], 2); } };
    // B row starts and lengths of the representative row's entries (one per thread), inclusive scan of the lengths
    int b0 = 0, len = 0;
    if (tid < nA) {
        const int j = Aj[a0 + tid];
        b0 = Bp[j];
        len = Bp[j + 1] - b0;
    }
    scan[tid] = len;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {
        const int add = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    if (tid < kClassMaxRowBig) {
        sIncl[tid] = scan[tid];
        sB0[tid] = b0 - (scan[tid] - len);
    }
    __syncthreads();
    const int P = nA > 0 ? sIncl[nA - 1] : 0;
    int longest = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) scan[tid >> 6] = longest;
    __syncthreads();
    longest = 0;
    for (int w = 0; w < kClassMaxRowBig / 64; ++w) longest = max(longest, scan[w]);   // (entries sit in the first threads)
    __syncthreads();
    if (P > kClassBigMaxP || longest > kClassMaxRowBig) { fail(); return; }
    int N2 = 64;
    while (N2 < P) N2 <<= 1;
    auto entry_of = [&](int p) {                                   // the A entry product p belongs to: first k with sIncl[k] > p
        int l = 0, r = nA - 1;
        while (l < r) { const int mid = (l + r) >> 1; if (sIncl[mid] <= p) l = mid + 1; else r = mid; }
        return l;
    };
    for (int p = tid; p < N2; p += NT) {
        int key = 0x7fffffff;
        if (p < P) key = Bj[sB0[entry_of(p)] + p] - rep;
        keys[p] = key;
        srt[p] = key;
    }
    __syncthreads();
    // Groups of T consecutive A entries whose B rows have the same columns (the unknowns of one node: the B rows of a
    // node's unknowns are that node's neighbours, all of them): entry k's product e and entry k + 1's product e fall on
    // the same entry of C, so the list carries one word per GROUP and B entry -- the numeric kernel sums the group's T
    // products in a register and adds once.  T = 4, 3 or 2 if every group of the row qualifies, else 1.
    if (tid < 5) sGroup[tid] = nA % max(tid, 1) == 0 ? 1 : 0;
    __syncthreads();
    for (int k = 1 + tid; k < nA; k += NT) {                       // lengths, entry by entry (an empty B row has no product to speak for it)
        const int lenK = sIncl[k] - sIncl[k - 1], lenPrev = sIncl[k - 1] - (k > 1 ? sIncl[k - 2] : 0);
        if (lenK != lenPrev)
            for (int T = 2; T <= 4; ++T)
                if (k % T) sGroup[T] = 0;                          // (a benign race: every writer writes 0)
    }
    for (int p = tid; p < P; p += NT) {                            // columns, product by product
        const int k = entry_of(p);
        if (k == 0) continue;
        const int lenK = sIncl[k] - sIncl[k - 1], lenPrev = sIncl[k - 1] - (k > 1 ? sIncl[k - 2] : 0);
        if (lenK == lenPrev && keys[p] != keys[p - lenPrev])       // (p - lenPrev: product e of entry k - 1)
            for (int T = 2; T <= 4; ++T)
                if (k % T) sGroup[T] = 0;
    }
    __syncthreads();
    const int T = sGroup[4] ? 4 : (sGroup[3] ? 3 : (sGroup[2] ? 2 : 1));
    // The distinct keys in ascending order -> ulist.  A class's relative columns span a few thousand values (a grid's
    // neighbours of neighbours): a presence bitmap over [smallest, largest] ranks them with a dozen barriers; a wider
    // span takes the bitonic sort (91 barriers for 8192 keys).
    int lo = 0x7fffffff, hi = -0x7fffffff - 1;
    for (int p = tid; p < P; p += NT) { lo = min(lo, keys[p]); hi = max(hi, keys[p]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((tid & 63) == 0) { scan[tid >> 6] = lo; scan[NT / 64 + (tid >> 6)] = hi; }
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) { lo = min(lo, scan[w]); hi = max(hi, scan[NT / 64 + w]); }
    __syncthreads();
    const long long span = P > 0 ? (long long)hi - lo + 1 : 0;
    int nnz = 0;
    if (span <= kClassBigSpan) {
        unsigned* bits = reinterpret_cast<unsigned*>(srt);          // (srt is free on this path: kClassBigSpan / 32 words)
        const int nw = (int)((span + 31) >> 5), wpt = (nw + NT - 1) / NT;
        for (int w = tid; w < nw; w += NT) bits[w] = 0u;
        __syncthreads();
        for (int p = tid; p < P; p += NT) atomicOr(&bits[(keys[p] - lo) >> 5], 1u << ((keys[p] - lo) & 31));
        __syncthreads();
        int mine = 0;
        for (int w = tid * wpt; w < (tid + 1) * wpt && w < nw; ++w) mine += __popc(bits[w]);
        scan[tid] = mine;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {
            const int add = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        nnz = scan[NT - 1];
        if (nnz > kClassMaxNnz) { fail(); return; }
        int at = scan[tid] - mine;
        for (int w = tid * wpt; w < (tid + 1) * wpt && w < nw; ++w)
            for (unsigned mm = bits[w]; mm; mm &= mm - 1) ulist[at++] = lo + (w << 5) + (__ffs((int)mm) - 1);
    } else {
    for (int kk = 2; kk <= N2; kk <<= 1)                           // ascending bitonic sort of srt[0, N2)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < N2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int x = srt[i], y = srt[ixj];
                    const bool up = (i & kk) == 0;
                    if ((x > y) == up) { srt[i] = y; srt[ixj] = x; }
                }
            }
            __syncthreads();
        }
    // distinct keys: thread t owns srt[t * per .. (t + 1) * per)
    const int per = (N2 + NT - 1) / NT;
    int heads = 0;
    for (int i = tid * per; i < (tid + 1) * per && i < P; ++i) heads += (i == 0 || srt[i] != srt[i - 1]) ? 1 : 0;
    scan[tid] = heads;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {
        const int add = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    nnz = scan[NT - 1];
    if (nnz > kClassMaxNnz) { fail(); return; }
    int at = scan[tid] - heads;
    for (int i = tid * per; i < (tid + 1) * per && i < P; ++i)
        if (i == 0 || srt[i] != srt[i - 1]) ulist[at++] = srt[i];
    }
    if (tid == 0) sIdx = atomicAdd(&stats[CS_BIGCOUNT], 1);
    __syncthreads();
    const int idx = sIdx;
    if (idx >= kClassBigCap) { fail(); return; }
    for (int e = tid; e < nnz; e += NT) classRel[(size_t)s * kClassMaxNnz + e] = ulist[e];
    for (int p = tid; p < P; p += NT) {
        const int key = keys[p], k = entry_of(p);
        if (k % T) continue;                                       // (its group's first entry speaks for it)
        int l = 0, r = nnz - 1;
        while (l < r) { const int mid = (l + r) >> 1; if (ulist[mid] < key) l = mid + 1; else r = mid; }
        const int start = k ? sIncl[k - 1] : 0;                    // (the groups before this one: T equal rows each)
        bigMap[(size_t)idx * kClassBigMaxP + start / T + (p - start)] = (unsigned)k | (unsigned)(p - start) << 8 | (unsigned)l << 16;
    }
    if (tid == 0) {
        classInfo[s] = make_int4(nA, P, nnz, rep);
        classBigIdx[s] = idx | (T << 16);
        atomicMax(&stats[CS_BIGMAXP], P / T);                      // (the longest list, in words)
        atomicMax(&stats[CS_MAXNNZ], nnz);
        atomicMax(&stats[CS_MAXNA], nA);
        atomicAdd(&stats[CS_CLASSES], 1);
    }
}

// ---------------------------------------------------------------------------
// Numeric pass of a multiply that has big classes (all of its rows: a small class's list is read from classMapA in the
// same way).  Workgroups are dealt to the XCDs so that each XCD's L2 sees one contiguous band of ranges.
//
// Rows in groups: the rmax unknowns of a node are rmax consecutive rows with the SAME columns of A -- the same B rows --
// and word for word the same list (their classes differ only in the relative columns).  Where the pass finds such a
// tuple of classes (lists compared word by word once per pass; the rows' A columns entry by entry for every group), a
// wave takes the group's rows together: one gather of B's values serves all of them (rmax accumulator sets per wave).
// A group whose rows do not share their columns after all is multiplied row by row.
//
// LDS: per wave acc[rmax][accStride] + sAx[rmax][stageCap] (acc_t) and sBp[stageCap] (int); per workgroup the class's
// list sDesc[descCap], the relative columns sRel[rmax][accStride], the classes of the range's rows and the rows (group
// leaders) of the pass at hand.
// ---------------------------------------------------------------------------
constexpr int kClassBigMaxGroup = 4;

// the products of R rows with the same A columns: list word = first A entry of the group | B entry << eShift | position << 16.
// Two batches of 64 x UN words alternate: the gathers of one are in flight while the other is multiplied and added.
template <int T, int R>
__device__ __forceinline__ void class_big_rows(const unsigned* sDesc, int words, int eShift, unsigned kMask, const acc_t* sAx,
                                               int stageCap, const int* sBp, const value_t* __restrict__ Bx, acc_t* acc,
                                               int accStride, int lane)
{
    constexpr int UN = T >= 3 ? 2 : 4;                             // (6 to 8 loads per lane and batch)
    if (words <= 0) return;                                        // (an empty row of A, or only empty B rows behind it: no list word
                                                                   //  to clamp to -- the row leaves with its 0 entries)
    auto load =[&](int base, unsigned (&d)[UN], acc_t (&b)[UN][T]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) d[u] = sDesc[min(base + u * 64 + lane, words - 1)];   // (beyond the list: its last word again)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (int)(d[u] & kMask), e = (int)((d[u] >> eShift) & kMask);
#pragma unroll
            for (int t = 0; t < T; ++t) b[u][t] = (acc_t)Bx[sBp[k + t] + e];
        }
    };
    auto add = [&](int base, const unsigned (&d)[UN], const acc_t (&b)[UN][T]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (int)(d[u] & kMask);
            const bool valid = base + u * 64 + lane < words;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc_t v = sAx[r * stageCap + k] * b[u][0];
#pragma unroll
                for (int t = 1; t < T; ++t) v = __builtin_fma(sAx[r * stageCap + k + t], b[u][t], v);
                if (valid) unsafeAtomicAdd(&acc[r * accStride + (d[u] >> 16)], v);
            }
        }
    };
    unsigned d0[UN], d1[UN];
    acc_t b0[UN][T], b1[UN][T];
    constexpr int STEP = 64 * UN;
    load(0, d0, b0);
    for (int base = 0; base < words; base += 2 * STEP) {
        load(base + STEP, d1, b1);
        add(base, d0, b0);
        if (base + STEP >= words) break;
        load(base + 2 * STEP, d0, b0);
        add(base + STEP, d1, b1);
    }
}

template <int R>
__device__ __forceinline__ void class_big_rows_t(int T, const unsigned* sDesc, int words, int eShift, unsigned kMask, const acc_t* sAx,
                                                 int stageCap, const int* sBp, const value_t* __restrict__ Bx, acc_t* acc,
                                                 int accStride, int lane)
{
    if (T == 1) class_big_rows<1, R>(sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
    else if (T == 2) class_big_rows<2, R>(sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
    else if (T == 3) class_big_rows<3, R>(sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
    else class_big_rows<4, R>(sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
}

__global__ __launch_bounds__(1024) void k_class_numeric_big(
    int m, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const value_t* __restrict__ Bx, const int* __restrict__ classC,
    const int4* __restrict__ classInfo, const unsigned* __restrict__ classMapA, const int* __restrict__ classBigIdx,
    const unsigned* __restrict__ bigMap, const int* __restrict__ classRel, const int* __restrict__ Cp,
    int* __restrict__ Cj, value_t* __restrict__ Cx, int accStride, int stageCap, int descCap,
    int rmax, int range,                                           // rows per group (1: none), rows per range (<= blockDim.x)
    int rowBase)                                                   // m, Ap, classC, Cp are views of the rows [rowBase, rowBase + m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    const int NT = blockDim.x, NW = NT >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int perWave = rmax * (accStride + stageCap);
    acc_t* acc = reinterpret_cast<acc_t*>(smemRaw) + (size_t)wv * perWave;
    acc_t* sAx = acc + rmax * accStride;
    int* ints = reinterpret_cast<int*>(reinterpret_cast<acc_t*>(smemRaw) + (size_t)NW * perWave);
    int* sBp = ints + wv * stageCap;
    unsigned* sDesc = reinterpret_cast<unsigned*>(ints + NW * stageCap);
    int* sRel = reinterpret_cast<int*>(sDesc + descCap);
    int* sCls = sRel + rmax * accStride;
    int* sList = sCls + range;
    int* sMisc = sList + range;                                    // [0] first row without a result, [1..] leaders per wave
    for (int i = lane; i < rmax * accStride; i += 64) acc[i] = 0.0;

    const int nRanges = (m + range - 1) / range;
    const int xcd = blockIdx.x & 7, perX = (nRanges + 7) / 8, wgPerX = gridDim.x >> 3;
    for (int i = blockIdx.x >> 3; i < perX; i += wgPerX) {
        const int rg = xcd * perX + i;
        if (rg >= nRanges) break;
        const int row0 = rg * range, nr = min(range, m - row0);
        __syncthreads();                                           // (the range before is done with sCls)
        // (mixed mode, bhs_class_mix.hip.h: a row without a class carries kClassDummy -- the general pipeline's kernels write it)
        for (int t = tid; t < range; t += NT) {
            const int c = t < nr ? classC[row0 + t] : -1;
            sCls[t] = c == kClassDummy ? -1 : c;
        }
        for (;;) {
            if (tid == 0) sMisc[0] = 0x7fffffff;
            __syncthreads();
            for (int t = tid; t < range; t += NT)
                if (sCls[t] >= 0) atomicMin(&sMisc[0], t);
            __syncthreads();
            const int first = sMisc[0];
            if (first == 0x7fffffff) break;                        // (workgroup-uniform)
            const int cls = sCls[first];
            const int4 ci = classInfo[cls];
            const int nA = ci.x, P = ci.y, nnz = ci.z;
            const int bi = classBigIdx[cls];
            const bool big = bi >= 0;
            const int T = big ? bi >> 16 : 1, words = P / T;      // (entries per group: one list word per group and B entry)
            const unsigned* list = big ? bigMap + (size_t)(bi & 0xFFFF) * kClassBigMaxP : classMapA + (size_t)cls * kClassMaxP;
            const int eShift = big ? 8 : 6;
            const unsigned kMask = big ? 255u : 63u;
            // a tuple of classes for rows in groups: the rmax rows from `first` on, if their tables are this class's
            int R = 1, cj[kClassBigMaxGroup] = {cls, -1, -1, -1};
            if (rmax > 1 && big && first + rmax <= nr) {
                bool ok = true;
                const unsigned* lj[kClassBigMaxGroup] = {list, list, list, list};
                for (int j = 1; j < rmax; ++j) {
                    cj[j] = sCls[first + j];
                    ok = ok && cj[j] >= 0;
                    for (int jj = 0; jj < j; ++jj) ok = ok && cj[j] != cj[jj];      // (distinct: no row leads two groups)
                    if (ok) {
                        const int4 c2 = classInfo[cj[j]];
                        const int b2 = classBigIdx[cj[j]];
                        ok = b2 >= 0 && (b2 >> 16) == T && c2.x == nA && c2.y == P && c2.z == nnz;
                        lj[j] = bigMap + (size_t)(b2 & 0xFFFF) * kClassBigMaxP;
                    }
                }
                int same = 1;
                if (ok)
                    for (int p = tid; p < words; p += NT) {
                        const unsigned w0 = list[p];
                        for (int j = 1; j < rmax; ++j) same &= lj[j][p] == w0 ? 1 : 0;
                    }
                if (__syncthreads_and(ok && same)) R = rmax;       // (ok is workgroup-uniform)
            }
            for (int p = tid; p < words; p += NT) sDesc[p] = list[p];
            for (int j = 0; j < R; ++j)
                for (int e = tid; e < nnz; e += NT) sRel[j * accStride + e] = classRel[(size_t)cj[j] * kClassMaxNnz + e];
            // the (leading) rows of this pass, in order (range <= blockDim.x: one row per thread)
            bool lead = tid < range && tid + R <= nr && sCls[tid] == cls;
            for (int j = 1; j < R; ++j) lead = lead && sCls[min(tid + j, range - 1)] == cj[j];
            const unsigned long long mm = __ballot(lead);
            if (lane == 0) sMisc[1 + wv] = __popcll(mm);
            __syncthreads();
            int before = 0, cnt = 0;
            for (int w = 0; w < NW; ++w) {
                const int c = sMisc[1 + w];
                before += w < wv ? c : 0;
                cnt += c;
            }
            if (lead) {
                sList[before + __popcll(mm & ((1ull << lane) - 1ull))] = tid;
                for (int j = 0; j < R; ++j) sCls[tid + j] = -1;
            }
            __syncthreads();
            for (int q = wv; q < cnt; q += NW) {
                const int row = row0 + sList[q];
                int a0[kClassBigMaxGroup];
                for (int r = 0; r < R; ++r) a0[r] = Ap[row + r];
                bool twins = true;
                for (int e = lane; e < nA; e += 64) {
                    const int aj = Aj[a0[0] + e];
                    sAx[e] = (acc_t)Ax[a0[0] + e];
                    sBp[e] = Bp[aj];
                    for (int r = 1; r < R; ++r) {
                        twins = twins && Aj[a0[r] + e] == aj;
                        sAx[r * stageCap + e] = (acc_t)Ax[a0[r] + e];
                    }
                }
                twins = __all(twins);
                wave_sync();
                if (R == 1) class_big_rows_t<1>(T, sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
                else if (twins) {
                    if (R == 2) class_big_rows_t<2>(T, sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
                    else if (R == 3) class_big_rows_t<3>(T, sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
                    else class_big_rows_t<4>(T, sDesc, words, eShift, kMask, sAx, stageCap, sBp, Bx, acc, accStride, lane);
                } else {
                    for (int r = 0; r < R; ++r) {                  // (same list, other B rows: one row at a time)
                        if (r) {
                            wave_sync();
                            for (int e = lane; e < nA; e += 64) sBp[e] = Bp[Aj[a0[r] + e]];
                            wave_sync();
                        }
                        class_big_rows_t<1>(T, sDesc, words, eShift, kMask, sAx + r * stageCap, stageCap, sBp, Bx, acc + r * accStride, accStride, lane);
                    }
                }
                wave_sync();
                for (int r = 0; r < R; ++r) {
                    const long long out = Cp[row + r];
                    for (int e = lane; e < nnz; e += 64) {
                        const acc_t v = acc[r * accStride + e];
                        acc[r * accStride + e] = 0.0;
                        class_store_c(&Cj[out + e], sRel[r * accStride + e] + row + r + rowBase);
                        class_store_c(&Cx[out + e], (value_t)v);
                    }
                }
                wave_sync();
            }
        }
    }
}

}  // namespace bhs
