// bhs_class_big.hip.h -- row classes whose product lists do not fit the registers of a wave.
// (Included after bhs_class.hip.h; see there for the classification and the tables of the small classes.)
//
// A grid with several unknowns per node (3-dof elasticity on a 27-point stencil: 81 entries per row, 6561 products and
// 375 entries per row of C) still has one relative pattern per kind of row, so its rows classify -- but a class's
// product list is 26 KB, not 12 words per lane.  For such classes
//   k_class_patterns_big   works the list out like k_class_patterns does (one workgroup per class: the representative
//                          row's products, sorted and made unique -> the relative column list; then every product's
//                          {A entry, B entry, position} word, in A-entry-major order) and leaves it in memory;
//   k_class_numeric_big    a workgroup takes kClassBigRange consecutive rows and goes through them CLASS BY CLASS: the
//                          class's list is copied to LDS once and every wave multiplies rows of that class with it
//                          (per 64 products: one LDS read of the list, the A value and B row start of the entry from
//                          LDS, one gather of B's values -- consecutive lanes read consecutive entries of one B row --
//                          and one ds_add_f64 into the row's accumulators), then the next class of the range.
// Replaces, for the matrices that qualify, the hash kernels (SpGEMM_cuda/bhsparse_cuda.h:210-2780, as
// bhs_class.hip.h does): a class's list is read from memory once per range and class, not once per row.
#pragma once

namespace bhs {

constexpr int kClassBigRange = 192;        // consecutive rows a workgroup takes at a time (a multiple of 3 and 4 unknowns per node)
constexpr int kClassBigWaves = 8;
constexpr int kClassBigPatThreads = 1024;

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
    const int nnz = scan[NT - 1];
    if (nnz > kClassMaxNnz) { fail(); return; }
    if (tid == 0) sIdx = atomicAdd(&stats[CS_BIGCOUNT], 1);
    int at = scan[tid] - heads;
    for (int i = tid * per; i < (tid + 1) * per && i < P; ++i)
        if (i == 0 || srt[i] != srt[i - 1]) ulist[at++] = srt[i];
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
        atomicMax(&stats[CS_BIGMAXP], P);
        atomicMax(&stats[CS_MAXNNZ], nnz);
        atomicMax(&stats[CS_MAXNA], nA);
        atomicAdd(&stats[CS_CLASSES], 1);
    }
}

// ---------------------------------------------------------------------------
// Numeric pass of a multiply that has big classes (all of its rows: a small class's list is read from classMapA in the
// same way).  Workgroups are dealt to the XCDs so that each XCD's L2 sees one contiguous band of ranges.
// LDS: per wave acc[accStride] + sAx[stageCap] (acc_t) and sBp[stageCap] (int); per workgroup the class's list
// sDesc[descCap], its relative columns sRel[accStride], the classes of the range's rows and the rows of the class at hand.
// ---------------------------------------------------------------------------
// the products of one row: list word = first A entry of the group | B entry << eShift | position << 16
template <int T>
__device__ __forceinline__ void class_big_row(const unsigned* sDesc, int words, int eShift, unsigned kMask, const acc_t* sAx,
                                              const int* sBp, const value_t* __restrict__ Bx, acc_t* acc, int lane)
{
    constexpr int UN = T >= 3 ? 2 : 4;                             // (8 to 4 loads in flight per lane)
    for (int base = 0; base < words; base += 64 * UN) {
        unsigned d[UN];
        acc_t a[UN][T], b[UN][T];
#pragma unroll
        for (int u = 0; u < UN; ++u) d[u] = sDesc[min(base + u * 64 + lane, words - 1)];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (int)(d[u] & kMask), e = (int)((d[u] >> eShift) & kMask);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                b[u][t] = (acc_t)Bx[sBp[k + t] + e];
                a[u][t] = sAx[k + t];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc_t v = a[u][0] * b[u][0];
#pragma unroll
            for (int t = 1; t < T; ++t) v = __builtin_fma(a[u][t], b[u][t], v);
            if (base + u * 64 + lane < words) unsafeAtomicAdd(&acc[d[u] >> 16], v);
        }
    }
}

__global__ __launch_bounds__(64 * kClassBigWaves) void k_class_numeric_big(
    int m, const int* __restrict__ Ap, const int* __restrict__ Aj, const value_t* __restrict__ Ax,
    const int* __restrict__ Bp, const value_t* __restrict__ Bx, const int* __restrict__ classC,
    const int4* __restrict__ classInfo, const unsigned* __restrict__ classMapA, const int* __restrict__ classBigIdx,
    const unsigned* __restrict__ bigMap, const int* __restrict__ classRel, const int* __restrict__ Cp,
    int* __restrict__ Cj, value_t* __restrict__ Cx, int accStride, int stageCap, int descCap,
    int rowBase)                                                   // m, Ap, classC, Cp are views of the rows [rowBase, rowBase + m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    constexpr int NT = 64 * kClassBigWaves;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    acc_t* acc = reinterpret_cast<acc_t*>(smemRaw) + (size_t)wv * (accStride + stageCap);
    acc_t* sAx = acc + accStride;
    int* ints = reinterpret_cast<int*>(reinterpret_cast<acc_t*>(smemRaw) + (size_t)kClassBigWaves * (accStride + stageCap));
    int* sBp = ints + wv * stageCap;
    unsigned* sDesc = reinterpret_cast<unsigned*>(ints + kClassBigWaves * stageCap);
    int* sRel = reinterpret_cast<int*>(sDesc + descCap);
    int* sCls = sRel + accStride;
    int* sList = sCls + kClassBigRange;
    int* sMisc = sList + kClassBigRange;                           // [0] first row without a result, [1..] rows of the class per wave
    for (int i = lane; i < accStride; i += 64) acc[i] = 0.0;

    const int nRanges = (m + kClassBigRange - 1) / kClassBigRange;
    const int xcd = blockIdx.x & 7, perX = (nRanges + 7) / 8, wgPerX = gridDim.x >> 3;
    for (int i = blockIdx.x >> 3; i < perX; i += wgPerX) {
        const int rg = xcd * perX + i;
        if (rg >= nRanges) break;
        const int row0 = rg * kClassBigRange, nr = min(kClassBigRange, m - row0);
        __syncthreads();                                           // (the range before is done with sCls)
        for (int t = tid; t < kClassBigRange; t += NT) sCls[t] = t < nr ? classC[row0 + t] : -1;
        for (;;) {
            if (tid == 0) sMisc[0] = 0x7fffffff;
            __syncthreads();
            for (int t = tid; t < kClassBigRange; t += NT)
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
            for (int p = tid; p < words; p += NT) sDesc[p] = list[p];
            for (int e = tid; e < nnz; e += NT) sRel[e] = classRel[(size_t)cls * kClassMaxNnz + e];
            // the rows of this class, in order (kClassBigRange <= 64 * kClassBigWaves: one row per thread)
            const bool match = tid < kClassBigRange && sCls[tid] == cls;
            const unsigned long long mm = __ballot(match);
            if (lane == 0) sMisc[1 + wv] = __popcll(mm);
            __syncthreads();
            int before = 0, cnt = 0;
#pragma unroll
            for (int w = 0; w < kClassBigWaves; ++w) {
                const int c = sMisc[1 + w];
                before += w < wv ? c : 0;
                cnt += c;
            }
            if (match) {
                sList[before + __popcll(mm & ((1ull << lane) - 1ull))] = tid;
                sCls[tid] = -1;
            }
            __syncthreads();
            for (int q = wv; q < cnt; q += kClassBigWaves) {
                const int row = row0 + sList[q];
                const int a0 = Ap[row];
                const long long out = Cp[row];
                for (int e = lane; e < nA; e += 64) {
                    const int aj = Aj[a0 + e];
                    sAx[e] = (acc_t)Ax[a0 + e];
                    sBp[e] = Bp[aj];
                }
                wave_sync();
                if (T == 1) class_big_row<1>(sDesc, words, eShift, kMask, sAx, sBp, Bx, acc, lane);
                else if (T == 2) class_big_row<2>(sDesc, words, eShift, kMask, sAx, sBp, Bx, acc, lane);
                else if (T == 3) class_big_row<3>(sDesc, words, eShift, kMask, sAx, sBp, Bx, acc, lane);
                else class_big_row<4>(sDesc, words, eShift, kMask, sAx, sBp, Bx, acc, lane);
                wave_sync();
                for (int e = lane; e < nnz; e += 64) {
                    const acc_t v = acc[e];
                    acc[e] = 0.0;
                    class_store_c(&Cj[out + e], sRel[e] + row + rowBase);
                    class_store_c(&Cx[out + e], (value_t)v);
                }
                wave_sync();
            }
        }
    }
}

}  // namespace bhs
