// bhs_sort.hip.h -- per-row sort of a CSR matrix on the device (bhs_csr_sort_indices_device).  (Split from bhs_kernels.hip.h in round 4.)
#pragma once

namespace bhs {

// ===========================================================================
// Per-row sort of a CSR matrix by column, in place and stable: the device
// counterpart of ref_spgemm::csr_sort_indices (SpGEMM_cuda/ref_spgemm.h:37-62),
// which the reference's driver runs on the host over every Matrix Market input
// (main.cu:62-64) because the long-row kernels want ascending B rows.
//   k_sort_rows_wave : one wavefront per row; rows of <= 1024 entries are sorted
//                      in registers as (column << 32 | position) keys -- the
//                      position makes the order stable and tells where the value
//                      comes from; rows already in order are left alone; longer
//                      rows are appended to a list
//   k_sort_rows_block: one workgroup per listed row, "flip" bitonic network for
//                      any length (partners past the end are +inf and never move
//                      down), keys in LDS up to 4096 entries, in a scratch array
//                      in HBM beyond
// ===========================================================================
constexpr int kSortLdsMax = 4096;

template <int E>
__device__ __forceinline__ void sort_row_wave(long long start, int len, int lane, int* __restrict__ Aj,
                                              value_t* __restrict__ Ax)
{
    using T = unsigned long long;
    T x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        x[e] = i < len ? (((T)(unsigned)Aj[start + i] << 32) | (unsigned)i) : ~0ull;
    }
    wave_bitonic_sort<T, E>(x, lane);
    value_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        v[e] = 0;
        if (lane * E + e < len) v[e] = Ax[start + (unsigned)x[e]];
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                      // every value of the row is in registers before one is overwritten
    wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        if (i < len) { Aj[start + i] = (int)(x[e] >> 32); Ax[start + i] = v[e]; }
    }
}

__global__ __launch_bounds__(256) void k_sort_rows_wave(int m, const int* __restrict__ Ap, int* __restrict__ Aj,
                                                        value_t* __restrict__ Ax, int* __restrict__ longList,
                                                        int* __restrict__ longCount)
{
    const int lane = threadIdx.x & 63;
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < m; row += (long long)gridDim.x * 4) {
        const long long start = Ap[row];
        const int len = Ap[row + 1] - Ap[row];
        bool bad = false;
        for (int i = lane; i + 1 < len; i += 64) bad |= Aj[start + i] > Aj[start + i + 1];
        if (!__any(bad)) continue;                             // (non-decreasing: a stable sort would not move anything)
        if (len <= 64) sort_row_wave<1>(start, len, lane, Aj, Ax);
        else if (len <= 128) sort_row_wave<2>(start, len, lane, Aj, Ax);
        else if (len <= 256) sort_row_wave<4>(start, len, lane, Aj, Ax);
        else if (len <= 512) sort_row_wave<8>(start, len, lane, Aj, Ax);
        else if (len <= 1024) sort_row_wave<16>(start, len, lane, Aj, Ax);
        else if (lane == 0) longList[atomicAdd(longCount, 1)] = (int)row;
    }
}

__global__ __launch_bounds__(256) void k_sort_rows_block(const int* __restrict__ Ap, int* __restrict__ Aj,
                                                         value_t* __restrict__ Ax, const int* __restrict__ longList,
                                                         const int* __restrict__ longCount,
                                                         unsigned long long* __restrict__ scratchK,
                                                         value_t* __restrict__ scratchV)
{
    using T = unsigned long long;
    __shared__ T ldsK[kSortLdsMax];
    __shared__ value_t ldsV[kSortLdsMax];
    const int tid = threadIdx.x;
    const int nLong = *longCount;
    for (int li = blockIdx.x; li < nLong; li += gridDim.x) {
        const int row = longList[li];
        const long long start = Ap[row];
        const int len = Ap[row + 1] - Ap[row];
        int P = 1;
        while (P < len) P <<= 1;
        const bool inLds = len <= kSortLdsMax;
        T* buf = inLds ? ldsK : scratchK + start;
        value_t* vb = inLds ? ldsV : scratchV + start;
        for (int i = tid; i < len; i += 256) buf[i] = ((T)(unsigned)Aj[start + i] << 32) | (unsigned)i;
        __syncthreads();
        auto cmpx = [&](int a, int b) {
            const T x = buf[a], y = buf[b];
            if (x > y) { buf[a] = y; buf[b] = x; }
        };
        for (int k = 2; k <= P; k <<= 1) {
            const int hk = k >> 1;
            for (int i = tid; i < (P >> 1); i += 256) {          // flip: o-th element of a block with its mirror image
                const int blk = i / hk, o = i - blk * hk;
                const int a = blk * k + o, b = blk * k + k - 1 - o;
                if (b < len) cmpx(a, b);
            }
            __syncthreads();
            for (int j = k >> 2; j > 0; j >>= 1) {
                for (int i = tid; i < (P >> 1); i += 256) {
                    const int a = (i / j) * 2 * j + (i % j), b = a + j;
                    if (b < len) cmpx(a, b);
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < len; i += 256) vb[i] = Ax[start + (unsigned)buf[i]];
        __syncthreads();
        for (int i = tid; i < len; i += 256) { Aj[start + i] = (int)(buf[i] >> 32); Ax[start + i] = vb[i]; }
        __syncthreads();
    }
}

}  // namespace bhs
