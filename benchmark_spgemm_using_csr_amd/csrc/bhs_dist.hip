// bhs_dist.hip — libbhsparse_dist.so: row-block sharding of the SpGEMM hot path over the GPUs of one node and the
// RCCL all-gatherv that assembles C (see include/bhsparse_dist.h).  Host C++ plus one trivial kernel; everything that
// multiplies lives in libbhsparse_hip.so and is reached through its C-ABI only.
#include "../../include/bhsparse_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

static_assert(sizeof(ncclUniqueId) == BHS_DIST_ID_BYTES, "id size");

struct bhs_dist {
    bhs_handle* h = nullptr;
    int world = 1, rank = 0, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t hstream = nullptr;      // the handle's stream (borrowed)
    hipStream_t cstream = nullptr;      // transfers: a stream of their own so that they overlap the numeric kernels
    hipEvent_t evRange = nullptr, evDone = nullptr;
    long long* dSizes = nullptr;        // device: world x kSizeSlots int64 (sizes exchange)
    long long* hSizes = nullptr;        // pinned mirror
    long long* dStatus = nullptr;       // device: 2 x int64 (agreements)
    bool dead = false;                  // the communicator was aborted after a failed send / receive group
    double linkFloorMs = 0.0;
    // assembled C owned by this object (the host-pointer entry points): grow-only
    int *ownRp = nullptr, *ownCj = nullptr;
    bhs_value_t* ownCx = nullptr;
    long long ownRows = 0, ownCap = 0, lastTotal = 0;
    int lastRows = 0;
    // values-only mode (option "values_only"): when every rank's multiply went by row classes, colIndC of the other
    // ranks' blocks is REBUILT from their classes (4 bytes per row + the class tables) instead of received (4 bytes per
    // entry): two thirds of the bytes on every link
    int valuesOnly = 0, lastValuesOnly = 0;
    int* classAll = nullptr;            // class of every row of the job (this rank's block copied in, the others received)
    long long classAllRows = 0;
    int* tablesAll = nullptr;           // world x (classInfo | classRel) of every rank
    long long tablesAllInts = 0;
};

namespace {

constexpr int kMaxSub = 16;
constexpr int kSizeSlots = kMaxSub + 5;   // m_local (or an error code < 0), nnzCt, capacity (-1: own), cut[0..S], [kMaxSub + 4]: this rank can serve the values-only mode
constexpr int kFlagSlot = kMaxSub + 4;

#define DIST_HIP(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return (int)BHS_ERR_LAUNCH; } } while (0)
#define DIST_NCCL(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { fprintf(stderr, "[bhsparse_dist] %s: %s\n", #call, ncclGetErrorString(r__)); return (int)BHS_ERR_LAUNCH; } } while (0)
#define DIST_TRY(call) do { const int rc__ = (call); if (rc__ != BHS_SUCCESS) return rc__; } while (0)

// rowPtrC of the rank's block, rebased by the entries of the ranks before it, into the block's place in the full array
__global__ void k_rebase_rowptr(int n, const int* __restrict__ local, long long off, int* __restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (int)(local[i] + off);
}

__global__ void k_set_int(int* __restrict__ p, int v) { *p = v; }

__global__ void k_gather_cuts(int S, int m_local, const int* __restrict__ rowPtr, long long* __restrict__ out)
{
    const int s = threadIdx.x;
    if (s <= S) out[s] = rowPtr[(long long)m_local * s / S];
}

// ---- transfer plan -------------------------------------------------------------------------------------------------
// The point-to-point operations of one rank, in issue order, as plain numbers: the executor below walks it, and the CPU
// tests replay the plans of all ranks of a job against each other (every send must meet a receive of the same size, in
// the same order, between the same two ranks; the received pieces must tile the assembled arrays exactly once) --
// the N > 1 transfers cannot run on the one-GPU development boxes.
struct PlanOp { long long kind, peer, array, offset, count, group; };   // kind 0 send / 1 recv; array 0 col, 1 val, 2 rowPtr, 3 class of every row, 4 class tables

// valuesOnly: no colInd transfers; instead, with the first range, the classes of the block's rows (array 3, in rows)
// and the rank's class tables (array 4: tableInts ints per rank, rank r's at r * tableInts)
static int build_plan(int W, int me, int S, const long long* sizes /* W x kSizeSlots */, PlanOp* out, int cap,
                      bool valuesOnly = false, long long tableInts = 0)
{
    std::vector<long long> rowOff(W + 1, 0), nnzOff(W + 1, 0);
    for (int r = 0; r < W; ++r) {
        rowOff[r + 1] = rowOff[r] + sizes[(size_t)r * kSizeSlots];
        nnzOff[r + 1] = nnzOff[r] + sizes[(size_t)r * kSizeSlots + 3 + S];
    }
    auto cut_nnz = [&](int r, int s) { return sizes[(size_t)r * kSizeSlots + 3 + s]; };
    int n = 0;
    auto put = [&](long long kind, long long peer, long long array, long long off, long long cnt, long long grp) {
        if (n < cap) out[n] = PlanOp{kind, peer, array, off, cnt, grp};
        ++n;
    };
    for (int s = 0; s < S; ++s)
        for (int step = 1; step < W; ++step) {                  // staggered peers: rank r sends to r + step, receives from r - step
            const int dst = (me + step) % W, src = (me - step + W) % W;
            const long long a = cut_nnz(me, s), b = cut_nnz(me, s + 1);
            if (b > a) {
                if (!valuesOnly) put(0, dst, 0, nnzOff[me] + a, b - a, s);
                put(0, dst, 1, nnzOff[me] + a, b - a, s);
            }
            const long long ra = cut_nnz(src, s), rb = cut_nnz(src, s + 1);
            if (rb > ra) {
                if (!valuesOnly) put(1, src, 0, nnzOff[src] + ra, rb - ra, s);
                put(1, src, 1, nnzOff[src] + ra, rb - ra, s);
            }
            if (s == 0) {                                       // row pointers travel with the first range
                const long long mm = rowOff[me + 1] - rowOff[me], rm = rowOff[src + 1] - rowOff[src];
                if (mm > 0) put(0, dst, 2, rowOff[me], mm, s);
                if (rm > 0) put(1, src, 2, rowOff[src], rm, s);
                if (valuesOnly) {                               // ... and what the columns are rebuilt from
                    if (mm > 0) { put(0, dst, 3, rowOff[me], mm, s); put(0, dst, 4, (long long)me * tableInts, tableInts, s); }
                    if (rm > 0) { put(1, src, 3, rowOff[src], rm, s); put(1, src, 4, (long long)src * tableInts, tableInts, s); }
                }
            }
        }
    return n;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int bhs_dist_unique_id(char id_out[BHS_DIST_ID_BYTES])
{
    if (!id_out) return BHS_ERR_INVALID_ARG;
    ncclUniqueId id;
    DIST_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return BHS_SUCCESS;
}

int bhs_dist_create(bhs_dist** out, bhs_handle* h, int world, int rank, const char id[BHS_DIST_ID_BYTES])
{
    if (!out || !h || !id || world < 1 || rank < 0 || rank >= world) return BHS_ERR_INVALID_ARG;
    *out = nullptr;
    bhs_dist* d = new (std::nothrow) bhs_dist();
    if (!d) return BHS_ERR_ALLOC;
    d->h = h;
    d->world = world;
    d->rank = rank;
    void* s = nullptr;
    if (bhs_get_stream(h, &s) != BHS_SUCCESS) { delete d; return BHS_ERR_INVALID_ARG; }
    d->hstream = (hipStream_t)s;
    if (hipGetDevice(&d->device) != hipSuccess) { delete d; return BHS_ERR_NO_DEVICE; }
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof(nid));
    if (ncclCommInitRank(&d->comm, world, nid, rank) != ncclSuccess) { delete d; return BHS_ERR_LAUNCH; }
    if (hipStreamCreateWithFlags(&d->cstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->evRange, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->evDone, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&d->dSizes, sizeof(long long) * kSizeSlots * (size_t)world) != hipSuccess ||
        hipMalloc((void**)&d->dStatus, sizeof(long long) * 2) != hipSuccess ||
        hipHostMalloc((void**)&d->hSizes, sizeof(long long) * kSizeSlots * (size_t)world, hipHostMallocDefault) != hipSuccess) {
        bhs_dist_destroy(d);
        return BHS_ERR_ALLOC;
    }
    *out = d;
    return BHS_SUCCESS;
}

int bhs_dist_destroy(bhs_dist* d)
{
    if (!d) return BHS_ERR_INVALID_ARG;
    if (d->cstream) (void)hipStreamSynchronize(d->cstream);
    if (d->comm && !d->dead) (void)ncclCommDestroy(d->comm);
    if (d->cstream) (void)hipStreamDestroy(d->cstream);
    if (d->evRange) (void)hipEventDestroy(d->evRange);
    if (d->evDone) (void)hipEventDestroy(d->evDone);
    if (d->dSizes) (void)hipFree(d->dSizes);
    if (d->dStatus) (void)hipFree(d->dStatus);
    if (d->hSizes) (void)hipHostFree(d->hSizes);
    if (d->ownRp) (void)hipFree(d->ownRp);
    if (d->ownCj) (void)hipFree(d->ownCj);
    if (d->ownCx) (void)hipFree(d->ownCx);
    if (d->classAll) (void)hipFree(d->classAll);
    if (d->tablesAll) (void)hipFree(d->tablesAll);
    delete d;
    return BHS_SUCCESS;
}

int bhs_dist_partition_rows(int m, const int* rowPtrA, const int* colIndA, const int* rowPtrB, int world, int* starts_out)
{
    if (m < 0 || world < 1 || !starts_out || (m > 0 && (!rowPtrA || !rowPtrB))) return BHS_ERR_INVALID_ARG;
    // work of a row = its products + a constant for the row itself (rows without products still cost a visit)
    std::vector<long long> pre((size_t)m + 1, 0);
    for (int i = 0; i < m; ++i) {
        long long w = 8;
        for (int e = rowPtrA[i]; e < rowPtrA[i + 1]; ++e) {
            const int c = colIndA[e];
            w += rowPtrB[c + 1] - rowPtrB[c];
        }
        pre[(size_t)i + 1] = pre[i] + w;
    }
    const long long total = pre[m];
    starts_out[0] = 0;
    for (int r = 1; r < world; ++r) {
        const long long target = total / world * r + (total % world) * r / world;
        int lo = (int)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin());
        lo = std::max(lo, starts_out[r - 1]);
        starts_out[r] = std::min(lo, m);
    }
    starts_out[world] = m;
    return BHS_SUCCESS;
}

// Error handling across ranks.  A rank that fails alone must not leave its peers waiting in a collective, so the call
// is built from three agreements that EVERY rank reaches whatever happened to it locally:
//   (1) the sizes all-gather carries a status word (slot 0 < 0: this rank's symbolic half failed) and each rank's
//       capacity, so that everyone derives the same verdict on sizes, row totals and capacities from the same numbers;
//   (2) a one-word all-reduce (max) after the output arrays are in place (own-mode allocation, binding);
//   (3) the same after the numeric half -- a rank whose numeric kernels failed still posts every send / receive
//       group (its peers get garbage they will not use) and everyone leaves with the error.
// Inside a send / receive group nothing returns: the first RCCL error is remembered, the group is closed, and the
// communicator is aborted (it cannot be trusted after a half-posted group; the object then refuses further calls).
// Every exit finishes the open multiply and unbinds the caller's output arrays from the handle.
int bhs_dist_spgemm_allgatherv(bhs_dist* d, int m_local, int m_total, int sub_blocks, int* d_rowPtrC, int* d_colIndC,
                               bhs_value_t* d_valC, int64_t capacity, int64_t* nnzCt_total_out,
                               int64_t* nnzC_total_out, double ms_out[3])
{
    if (!d || m_local < 0 || m_total < m_local || capacity < 0) return BHS_ERR_INVALID_ARG;
    if (d->dead) return BHS_ERR_LAUNCH;
    const bool own = d_rowPtrC == nullptr;            // host-pointer callers: the assembled C lives in this object
    const int S = std::max(1, std::min(sub_blocks, kMaxSub));
    const int W = d->world, me = d->rank;
    DIST_HIP(hipSetDevice(d->device));
    const double t0 = now_ms();
    bool open = false;                                // a multiply is open on the handle (between its halves)
    auto leave = [&](int rc) {                        // every exit: no open multiply, no borrowed output arrays left behind
        if (open) (void)bhs_spgemm_finish(d->h, nullptr);
        (void)bhs_set_output_device(d->h, nullptr, nullptr, 0);
        return rc;
    };
    auto fatal = [&](const char* what, ncclResult_t r) {          // the communicator is unusable from here on
        fprintf(stderr, "[bhsparse_dist] rank %d: %s: %s -- communicator aborted\n", me, what, ncclGetErrorString(r));
        (void)ncclCommAbort(d->comm);
        d->comm = nullptr;
        d->dead = true;
        return leave((int)BHS_ERR_LAUNCH);
    };
    auto agree = [&](int mine, int* all) -> ncclResult_t {        // max of one status word over the ranks (0 = fine everywhere)
        *all = mine;
        if (W == 1) return ncclSuccess;
        d->hSizes[0] = mine;
        if (hipMemcpyAsync(d->dStatus, d->hSizes, sizeof(long long), hipMemcpyHostToDevice, d->hstream) != hipSuccess) return ncclUnhandledCudaError;
        const ncclResult_t r = ncclAllReduce(d->dStatus, d->dStatus + 1, 1, ncclInt64, ncclMax, d->comm, d->hstream);
        if (r != ncclSuccess) return r;
        if (hipMemcpyAsync(d->hSizes, d->dStatus + 1, sizeof(long long), hipMemcpyDeviceToHost, d->hstream) != hipSuccess ||
            hipStreamSynchronize(d->hstream) != hipSuccess) return ncclUnhandledCudaError;
        *all = (int)d->hSizes[0];
        return ncclSuccess;
    };

    // ---- local work before the first agreement; errors are carried, not returned
    int err = BHS_SUCCESS;
    if (hipStreamWaitEvent(d->hstream, d->evDone, 0) != hipSuccess) err = BHS_ERR_LAUNCH;   // the previous call's transfers read what this multiply overwrites
    if (!err && own && (long long)m_total + 1 > d->ownRows) {
        if (d->ownRp) (void)hipFree(d->ownRp);
        d->ownRp = nullptr; d->ownRows = 0;
        if (hipMalloc((void**)&d->ownRp, sizeof(int) * ((size_t)m_total + 1)) != hipSuccess) { (void)hipGetLastError(); err = BHS_ERR_ALLOC; }
        else d->ownRows = (long long)m_total + 1;
    }
    if (own) d_rowPtrC = d->ownRp;
    // symbolic half: nnz(C) of the block and its rowPtrC are final afterwards (the library allocates no colIndC / valC
    // of its own for it: the numeric half writes into the assembled arrays)
    int64_t nnzCt = 0;
    int nnzC = 0;
    const int* lp = nullptr;
    if (!err) err = bhs_set_output_device(d->h, nullptr, nullptr, 0);
    if (!err) {
        err = bhs_spgemm_symbolic(d->h, &nnzCt, &nnzC);
        open = err == BHS_SUCCESS;
    }
    if (!err) err = bhs_get_C_device(d->h, &lp, nullptr, nullptr);
    // values-only mode: can this rank's block be rebuilt from its row classes?
    const int* cC = nullptr;
    const void* cInfo = nullptr;
    const int* cRel = nullptr;
    int cSlots = 0, cStride = 0, cUsable = 0;
    if (bhs_get_class_tables_device(d->h, &cC, &cInfo, &cRel, &cSlots, &cStride, &cUsable) != BHS_SUCCESS) cUsable = 0;   // (the table geometry comes back either way)
    if (err || !d->valuesOnly) cUsable = 0;
    const long long tableInts = (long long)cSlots * 4 + (long long)cSlots * cStride;   // (classInfo | classRel, the same on every rank of one build)

    // ---- (1) sizes of every rank: status | rows, products, capacity, rowPtrC at the sub-block boundaries
    long long* mine = d->dSizes + (size_t)me * kSizeSlots;
    {
        // (error codes are negative: slot 0 < 0 tells the others that this rank's symbolic half failed, and how)
        long long head[4] = {err ? (long long)(err < 0 ? err : -1) : (long long)m_local, (long long)nnzCt, own ? -1 : (long long)capacity,
                             cUsable ? tableInts : 0};                                     // [3] -> the flag slot: 0, or the table size
        memcpy(d->hSizes, head, sizeof(head));
        bool okc = hipMemcpyAsync(mine, d->hSizes, 3 * sizeof(long long), hipMemcpyHostToDevice, d->hstream) == hipSuccess &&
                   hipMemcpyAsync(mine + kFlagSlot, d->hSizes + 3, sizeof(long long), hipMemcpyHostToDevice, d->hstream) == hipSuccess;
        if (okc && !err) {
            hipLaunchKernelGGL(k_gather_cuts, dim3(1), dim3(64), 0, d->hstream, S, m_local, lp, mine + 3);
            okc = hipGetLastError() == hipSuccess;
        }
        if (!okc && W == 1) return leave(BHS_ERR_LAUNCH);
        if (!okc) return fatal("staging the sizes", ncclUnhandledCudaError);   // (cannot even tell the others)
    }
    if (W > 1) {
        const ncclResult_t r = ncclAllGather(mine, d->dSizes, kSizeSlots, ncclInt64, d->comm, d->hstream);
        if (r != ncclSuccess) return fatal("ncclAllGather(sizes)", r);
    }
    if (hipMemcpyAsync(d->hSizes, d->dSizes, sizeof(long long) * kSizeSlots * (size_t)W, hipMemcpyDeviceToHost, d->hstream) != hipSuccess ||
        hipStreamSynchronize(d->hstream) != hipSuccess) {
        if (W == 1) return leave(BHS_ERR_LAUNCH);
        return fatal("reading the sizes", ncclUnhandledCudaError);
    }
    // the same verdict on every rank, from the same numbers
    std::vector<long long> rowOff(W + 1, 0), nnzOff(W + 1, 0), caps(W, 0);
    std::vector<long long> sizes(d->hSizes, d->hSizes + (size_t)kSizeSlots * W);    // (hSizes is reused by agree())
    long long ctTotal = 0;
    int verdict = BHS_SUCCESS;
    for (int r = 0; r < W; ++r) {
        const long long* sz = sizes.data() + (size_t)r * kSizeSlots;
        if (sz[0] < 0) { verdict = verdict ? verdict : (int)sz[0]; continue; }
        rowOff[r + 1] = rowOff[r] + sz[0];
        nnzOff[r + 1] = nnzOff[r] + sz[3 + S];          // cut[S] = nnz(C) of the block
        ctTotal += sz[1];
        caps[r] = sz[2];
    }
    const long long total = nnzOff[W];
    // values-only: every rank asked for it and can serve it (same table size everywhere); a rank without rows does not matter
    // (judged from the exchanged numbers alone, so that every rank -- also one without rows or without the option -- agrees)
    bool vo = tableInts > 0, anyRows = false;
    for (int r = 0; r < W && vo; ++r) {
        const long long* sz = sizes.data() + (size_t)r * kSizeSlots;
        if (sz[0] > 0) { anyRows = true; if (sz[kFlagSlot] != tableInts) vo = false; }
    }
    vo = vo && anyRows;
    d->lastValuesOnly = vo ? 1 : 0;
    if (!verdict && rowOff[W] != m_total) verdict = BHS_ERR_INVALID_ARG;
    if (!verdict && total > 0x7fffffffLL) verdict = BHS_ERR_NNZ_OVERFLOW;
    for (int r = 0; r < W && !verdict; ++r)
        if (caps[r] >= 0 && total > caps[r]) verdict = BHS_ERR_ALLOC;          // some rank's arrays are too small
    if (verdict) return leave(err ? err : verdict);

    // ---- output arrays in place, then (2): everyone is ready, or nobody transfers
    int ready = BHS_SUCCESS;
    if (own) {
        if (total > d->ownCap) {
            if (d->ownCj) (void)hipFree(d->ownCj);
            if (d->ownCx) (void)hipFree(d->ownCx);
            d->ownCj = nullptr; d->ownCx = nullptr; d->ownCap = 0;
            const size_t cap = (size_t)std::max<long long>(total, 1);
            if (hipMalloc((void**)&d->ownCj, sizeof(int) * cap) != hipSuccess ||
                hipMalloc((void**)&d->ownCx, sizeof(bhs_value_t) * cap) != hipSuccess) { (void)hipGetLastError(); ready = BHS_ERR_ALLOC; }
            else d->ownCap = (long long)cap;
        }
        d_colIndC = d->ownCj;
        d_valC = d->ownCx;
        capacity = d->ownCap;
    }
    d->lastTotal = total;
    d->lastRows = m_total;
    if (!ready && total > 0 && (!d_colIndC || !d_valC)) ready = BHS_ERR_INVALID_ARG;
    if (!ready && vo) {                                             // where the others' classes and tables land
        if ((long long)m_total > d->classAllRows) {
            if (d->classAll) (void)hipFree(d->classAll);
            d->classAll = nullptr; d->classAllRows = 0;
            if (hipMalloc((void**)&d->classAll, sizeof(int) * (size_t)std::max(m_total, 1)) != hipSuccess) { (void)hipGetLastError(); ready = BHS_ERR_ALLOC; }
            else d->classAllRows = m_total;
        }
        if (!ready && tableInts * W > d->tablesAllInts) {
            if (d->tablesAll) (void)hipFree(d->tablesAll);
            d->tablesAll = nullptr; d->tablesAllInts = 0;
            if (hipMalloc((void**)&d->tablesAll, sizeof(int) * (size_t)(tableInts * W)) != hipSuccess) { (void)hipGetLastError(); ready = BHS_ERR_ALLOC; }
            else d->tablesAllInts = tableInts * W;
        }
        if (!ready && m_local > 0) {                                // this rank's share of both, in place (the sends read them there)
            int* tb = d->tablesAll + (size_t)me * tableInts;
            if (hipMemcpyAsync(d->classAll + rowOff[me], cC, sizeof(int) * (size_t)m_local, hipMemcpyDeviceToDevice, d->hstream) != hipSuccess ||
                hipMemcpyAsync(tb, cInfo, sizeof(int) * 4 * (size_t)cSlots, hipMemcpyDeviceToDevice, d->hstream) != hipSuccess ||
                hipMemcpyAsync(tb + (size_t)cSlots * 4, cRel, sizeof(int) * (size_t)cSlots * cStride, hipMemcpyDeviceToDevice, d->hstream) != hipSuccess)
                ready = BHS_ERR_LAUNCH;
        }
    }
    // per-link floor: the largest block this rank receives over one link
    long long worst = 0;
    for (int r = 0; r < W; ++r)
        if (r != me) worst = std::max(worst, (nnzOff[r + 1] - nnzOff[r]) * (long long)((vo ? 0 : sizeof(int)) + sizeof(bhs_value_t)) +
                                                 (rowOff[r + 1] - rowOff[r]) * (long long)sizeof(int) * (vo ? 2 : 1) +
                                                 (vo && rowOff[r + 1] > rowOff[r] ? tableInts * (long long)sizeof(int) : 0));
    d->linkFloorMs = (double)worst / 153.0e9 * 1e3;
    // this rank's block is produced in place: the numeric kernels write into the assembled arrays
    if (!ready) ready = bhs_set_output_device(d->h, d_colIndC ? d_colIndC + nnzOff[me] : nullptr,
                                              d_valC ? d_valC + nnzOff[me] : nullptr, nnzOff[me + 1] - nnzOff[me]);
    if (!ready && m_local > 0) {
        const int grid = (int)std::min<long long>(((long long)m_local + 255) / 256, 1024);
        hipLaunchKernelGGL(k_rebase_rowptr, dim3(grid), dim3(256), 0, d->hstream, m_local, lp, nnzOff[me],
                           d_rowPtrC + rowOff[me]);
        if (hipGetLastError() != hipSuccess) ready = BHS_ERR_LAUNCH;
    }
    if (!ready) {
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, d->hstream, d_rowPtrC + m_total, (int)total);
        if (hipGetLastError() != hipSuccess) ready = BHS_ERR_LAUNCH;
    }
    {
        int all = 0;
        const ncclResult_t r = agree(ready ? 1 : 0, &all);
        if (r != ncclSuccess) return fatal("agreement before the transfers", r);
        if (all) return leave(ready ? ready : BHS_ERR_PEER);
    }
    const double t1 = now_ms();

    // ---- numeric half in row ranges; the transfers of range s ride the second stream while range s + 1 computes
    auto cut_row = [&](int r, int s) { return (int)((long long)(rowOff[r + 1] - rowOff[r]) * s / S); };
    std::vector<PlanOp> plan;
    if (W > 1) {
        plan.resize((size_t)S * (W - 1) * 10);
        const int np = build_plan(W, me, S, sizes.data(), plan.data(), (int)plan.size(), vo, tableInts);
        plan.resize((size_t)np);
    }
    size_t next = 0;
    int numErr = BHS_SUCCESS;
    for (int s = 0; s < S; ++s) {
        if (!numErr) {
            numErr = bhs_spgemm_numeric(d->h, cut_row(me, s), cut_row(me, s + 1));
            if (numErr) open = false;                              // (the library closed the multiply)
        }
        if (W == 1) continue;
        // (after a local failure the groups are still posted, so that no peer waits for this rank)
        if (hipEventRecord(d->evRange, d->hstream) != hipSuccess || hipStreamWaitEvent(d->cstream, d->evRange, 0) != hipSuccess)
            numErr = numErr ? numErr : BHS_ERR_LAUNCH;
        ncclResult_t first = ncclGroupStart();
        if (first != ncclSuccess) return fatal("ncclGroupStart", first);
        for (; next < plan.size() && plan[next].group == s; ++next) {
            const PlanOp& op = plan[next];
            void* ptr;
            size_t count;
            ncclDataType_t type = ncclInt32;
            if (op.array == 0) { ptr = d_colIndC + op.offset; count = (size_t)op.count; }
            else if (op.array == 1) { ptr = d_valC + op.offset; count = (size_t)op.count * sizeof(bhs_value_t); type = ncclInt8; }
            else if (op.array == 2) { ptr = d_rowPtrC + op.offset; count = (size_t)op.count; }
            else if (op.array == 3) { ptr = d->classAll + op.offset; count = (size_t)op.count; }
            else { ptr = d->tablesAll + op.offset; count = (size_t)op.count; }
            const ncclResult_t r = op.kind == 0 ? ncclSend(ptr, count, type, (int)op.peer, d->comm, d->cstream)
                                                : ncclRecv(ptr, count, type, (int)op.peer, d->comm, d->cstream);
            if (r != ncclSuccess && first == ncclSuccess) first = r;   // remembered; the group is closed before anything else
        }
        const ncclResult_t end = ncclGroupEnd();
        if (first != ncclSuccess) return fatal("ncclSend / ncclRecv", first);
        if (end != ncclSuccess) return fatal("ncclGroupEnd", end);
    }
    if (open) {
        const int rcFin = bhs_spgemm_finish(d->h, nullptr);
        open = false;
        if (rcFin != BHS_SUCCESS && !numErr) numErr = rcFin;
    }
    if (vo && W > 1) {
        // the other ranks' column indices, rebuilt behind their receives: entry s of a row = its class's relative column
        // s + the row's number in ITS rank's block (the classes were made on that block's local row numbers)
        for (int r = 0; r < W; ++r) {
            const long long rows = rowOff[r + 1] - rowOff[r];
            if (r == me || rows == 0) continue;
            const int* tb = d->tablesAll + (size_t)r * tableInts;
            const int rcE = bhs_expand_class_columns_device((void*)d->cstream, (int)rows, 0, d->classAll + rowOff[r], (const void*)tb,
                                                            tb + (size_t)cSlots * 4, cStride, d_rowPtrC + rowOff[r], d_colIndC);
            if (rcE != BHS_SUCCESS && !numErr) numErr = rcE;
        }
    }
    const double t2 = now_ms();
    if (hipEventRecord(d->evDone, d->cstream) != hipSuccess || hipStreamSynchronize(d->cstream) != hipSuccess)
        numErr = numErr ? numErr : BHS_ERR_LAUNCH;
    // ---- (3) everyone's numeric half went through, or everyone knows it did not
    {
        int all = 0;
        const ncclResult_t r = agree(numErr ? 1 : 0, &all);
        if (r != ncclSuccess) return fatal("agreement after the transfers", r);
        if (all) return leave(numErr ? numErr : BHS_ERR_PEER);
    }
    const double t3 = now_ms();
    if (nnzCt_total_out) *nnzCt_total_out = ctTotal;
    if (nnzC_total_out) *nnzC_total_out = total;
    if (ms_out) { ms_out[0] = t1 - t0; ms_out[1] = t2 - t1; ms_out[2] = t3 - t2; }
    return leave(BHS_SUCCESS);
}

int bhs_dist_nranks(bhs_dist* d, int* nranks_out)
{
    if (!d || !nranks_out) return BHS_ERR_INVALID_ARG;
    if (d->dead || !d->comm) return BHS_ERR_LAUNCH;
    int n = 0;
    DIST_NCCL(ncclCommCount(d->comm, &n));
    *nranks_out = n;
    return BHS_SUCCESS;
}

double bhs_dist_last_link_floor_ms(bhs_dist* d) { return d ? d->linkFloorMs : 0.0; }

int bhs_dist_set_option(bhs_dist* d, const char* key, int64_t value)
{
    if (!d || !key) return BHS_ERR_INVALID_ARG;
    if (!strcmp(key, "values_only")) { d->valuesOnly = value ? 1 : 0; return BHS_SUCCESS; }
    return BHS_ERR_INVALID_ARG;
}

int bhs_dist_last_values_only(bhs_dist* d) { return d ? d->lastValuesOnly : 0; }

static int plan_impl(int world, int rank, int sub_blocks, const int64_t* rows, const int64_t* cuts, int64_t* ops_out, int cap_ops,
                     bool valuesOnly, long long tableInts);

int bhs_dist_plan(int world, int rank, int sub_blocks, const int64_t* rows, const int64_t* cuts, int64_t* ops_out, int cap_ops)
{
    return plan_impl(world, rank, sub_blocks, rows, cuts, ops_out, cap_ops, false, 0);
}

int bhs_dist_plan_values_only(int world, int rank, int sub_blocks, const int64_t* rows, const int64_t* cuts, int64_t table_ints,
                              int64_t* ops_out, int cap_ops)
{
    if (table_ints <= 0) return BHS_ERR_INVALID_ARG;
    return plan_impl(world, rank, sub_blocks, rows, cuts, ops_out, cap_ops, true, table_ints);
}

static int plan_impl(int world, int rank, int sub_blocks, const int64_t* rows, const int64_t* cuts, int64_t* ops_out, int cap_ops,
                     bool valuesOnly, long long tableInts)
{
    if (world < 1 || rank < 0 || rank >= world || sub_blocks < 1 || sub_blocks > kMaxSub || !rows || !cuts) return BHS_ERR_INVALID_ARG;
    std::vector<long long> sizes((size_t)world * kSizeSlots, 0);
    for (int r = 0; r < world; ++r) {
        sizes[(size_t)r * kSizeSlots] = rows[r];
        for (int s = 0; s <= sub_blocks; ++s) sizes[(size_t)r * kSizeSlots + 3 + s] = cuts[(size_t)r * (sub_blocks + 1) + s];
    }
    std::vector<PlanOp> plan((size_t)std::max(cap_ops, 0));
    const int n = build_plan(world, rank, sub_blocks, sizes.data(), plan.data(), (int)plan.size(), valuesOnly, tableInts);
    for (int i = 0; i < n && i < cap_ops; ++i) memcpy(ops_out + (size_t)i * 6, &plan[i], sizeof(PlanOp));
    return n;
}

int bhs_dist_spgemm_allgatherv_host(bhs_dist* d, int m_local, int m_total, int sub_blocks, int* rowPtrC_out,
                                    int64_t* nnzCt_total_out, int64_t* nnzC_total_out, double ms_out[3])
{
    if (!d) return BHS_ERR_INVALID_ARG;
    DIST_TRY(bhs_dist_spgemm_allgatherv(d, m_local, m_total, sub_blocks, nullptr, nullptr, nullptr, 0, nnzCt_total_out,
                                        nnzC_total_out, ms_out));
    if (rowPtrC_out) {
        DIST_HIP(hipMemcpy(rowPtrC_out, d->ownRp, sizeof(int) * ((size_t)m_total + 1), hipMemcpyDeviceToHost));
    }
    return BHS_SUCCESS;
}

int bhs_dist_get_C_host(bhs_dist* d, int* csrColIndC, bhs_value_t* csrValC)
{
    if (!d || !d->ownRp) return BHS_ERR_NOT_READY;
    if (d->lastTotal && (!csrColIndC || !csrValC)) return BHS_ERR_INVALID_ARG;
    if (d->lastTotal) {
        DIST_HIP(hipMemcpy(csrColIndC, d->ownCj, sizeof(int) * (size_t)d->lastTotal, hipMemcpyDeviceToHost));
        DIST_HIP(hipMemcpy(csrValC, d->ownCx, sizeof(bhs_value_t) * (size_t)d->lastTotal, hipMemcpyDeviceToHost));
    }
    return BHS_SUCCESS;
}

}  // extern "C"
