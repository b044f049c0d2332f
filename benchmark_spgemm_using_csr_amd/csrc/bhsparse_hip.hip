// bhsparse_hip.hip — pipeline + C-ABI of libbhsparse_hip.so (see include/bhsparse_hip.h).
//
// Pipeline of one bhs_spgemm() (replaces bhsparse::spgemm_cuda, bhsparse.h:297-339):
//   stage 1  k_upper_bound (ub per row, nnzCt, symbolic-bin histogram)          <- compute_nnzCt + statistics()
//            k_fill_queues  (16-byte row descriptors grouped by symbolic bin, on device)
//   stage 2  symbolic pass per non-empty bin: exact nnz of every C row          <- replaces upper-bound Ct + copy stage
//            (k_row_quad / k_row_wave / k_row_block / k_row_spa with NUM = false)
//   stage 3  k_scan_* : rowPtrC = exclusive scan, nnz(C), numeric-bin histogram <- create_C
//            (grow-only pool) make room for C; k_fill_queues by nnz per row
//   stage 4  numeric pass per non-empty bin: C written once, sorted             <- ESC_*/EM_* + copyCt2C_*
//            (the same four kernel families with NUM = true)
// Two host<->device round trips of a few hundred bytes (bin counts, nnzCt, nnzC)
// instead of the reference's whole-array D2H/H2D of rowPtrCt, the 6*m queue and
// rowPtrC (bhsparse_cuda.h:280, 289, 2787-2808).
#include "../../include/bhsparse_hip.h"
#include "bhs_kernels.hip.h"
#include "bhs_row_wg.hip.h"
#include "bhs_row_wave.hip.h"
#include "bhs_row_window.hip.h"
#include "bhs_row_quad.hip.h"
#include "bhs_compress.hip.h"
#include "bhs_row_lane.hip.h"
#include "bhs_sort.hip.h"
#include "bhs_hub.hip.h"
#include "bhs_class.hip.h"
#include "bhs_class_wg.hip.h"
#include "bhs_class_ring.hip.h"
#include "bhs_class_fused.hip.h"
#include "bhs_class_big.hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <tuple>
#include <unordered_map>
#include <vector>

using namespace bhs;

namespace {

#define BHS_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            if (h && h->verbose)                                                            \
                fprintf(stderr, "[bhsparse_hip] %s failed: %s (%s:%d)\n", #call,            \
                        hipGetErrorString(e__), __FILE__, __LINE__);                        \
            (void)hipGetLastError();                                                        \
            return e__ == hipErrorOutOfMemory ? (int)BHS_ERR_ALLOC : (int)BHS_ERR_LAUNCH;   \
        }                                                                                   \
    } while (0)

#define BHS_TRY(call)                     \
    do {                                  \
        const int rc__ = (call);          \
        if (rc__ != BHS_SUCCESS) return rc__; \
    } while (0)

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
};

// ---- bin tables ------------------------------------------------------------
// Symbolic bins are chosen by the per-row upper bound ub (the table must hold
// every product's column in the worst case: ub <= 3/4 * TS); numeric bins by the
// exact per-row nnz found by the symbolic pass (nnz <= 3/4 * TS).
struct KernelCfg {
    int log2ts;     // table size
    int block;      // lanes per row
    bool win;       // column-window variant
};
// bin 0 = empty rows (no kernel); bin 1 = quad bin (block 16: four rows per wavefront, k_row_quad)
constexpr int kNumSymBins = 12;
const KernelCfg kSymCfg[kNumSymBins] = {
    {0, 0, false},   {6, 16, false},  {6, 64, false},  {7, 64, false},   {8, 64, false},   {9, 64, false},
    {10, 64, false}, {11, 64, false}, {12, 64, false}, {13, 256, false}, {15, 1024, false}, {15, 1024, true}};
constexpr int kNumNumBins = 11;
const KernelCfg kNumCfg[kNumNumBins] = {
    {0, 0, false},  {6, 16, false},  {6, 64, false},  {7, 64, false},   {8, 64, false},  {9, 64, false},
    {10, 64, false}, {11, 256, false}, {12, 256, false}, {13, 512, false}, {13, 512, true}};
constexpr int kQuadMax = 48;        // products (symbolic) / entries (numeric) a 64-slot quarter table admits

constexpr int kLaneMaxK = 12;       // heads a lane keeps in registers (k_row_lane<K>: K = 4, 6, .., 12)
constexpr int kLaneMax = kLaneMaxK * kLaneMaxK;   // products (symbolic) / entries (numeric) of a lane-bin row

BinSpec make_spec(const KernelCfg* cfg, int nbins, int maxLog2, int loadPct, bool quad, int laneK, int hubMin)
{
    BinSpec s;
    memset(&s, 0, sizeof(s));
    s.nbins = nbins;
    s.hubMin = hubMin;
    s.laneMax = laneK > 0 ? kLaneMax : 0;
    s.laneMaxA = laneK;
    s.quadMax = (quad && maxLog2 >= 6) ? kQuadMax : 0;
    s.upper[0] = 0;
    s.upper[1] = 0;
    for (int b = 2; b < nbins; ++b) {
        int lg = std::min(cfg[b].log2ts, maxLog2);
        int ts = 1 << lg;
        s.upper[b] = cfg[b].win ? 0x7fffffff : (int)((long long)ts * loadPct / 100);
        if (b > 2 && s.upper[b] < s.upper[b - 1]) s.upper[b] = s.upper[b - 1];
    }
    s.upper[nbins - 1] = 0x7fffffff;
    return s;
}

struct StatRec {
    const char* name;
    int launches = 0;
    double ms = 0;
    int64_t rows = 0, products = 0, nnz_out = 0, nnzA_rows = 0;
};

struct EventPair {
    hipEvent_t a, b;
    int stat;   // index into stats
};

}  // namespace

struct bhs_handle {
    int device = 0;
    int numCU = 256;
    int verbose = 0;
    bool bannerDone = false;
    hipStream_t stream = nullptr;
    bool hasData = false, ownAB = false, hasC = false;
    int m = 0, k = 0, n = 0, nnzA = 0, nnzB = 0;
    const int *dAp = nullptr, *dAj = nullptr, *dBp = nullptr, *dBj = nullptr;
    const value_t *dAx = nullptr, *dBx = nullptr;
    DevBuf ownA[3], ownB[3];
    int bSorted = 1;
    int logL = 5, ubG = 8, ubLong = kUbLongA;   // k_upper_bound: lanes per row of A, rows beyond ubLong entries go to its long list
    // C
    DevBuf Cp, Cj, Cx;
    long long nnzC = 0;
    long long nnzCt = 0;
    // workspace
    DevBuf ub, queue, blockSum, small;   // small: counters (see layout below)
    DevBuf spaRank, spaBits;             // bitmap-accumulator slots for rows beyond the LDS tables (bitmaps kept all-zero)
    int spaSlots = 0, spaCols = -1, useSpa = 1, spaMaxSlots = 0, useLdsBitmap = 1, ldsBitmapMinLog2 = 12;
    bool spaDirty = false;
    // hub rows (bhs_hub.hip.h): rows with at least hubMin products are cut into items of hubItemProducts products
    // that the whole device works on; one bitmap slot (+ rank words in the numeric stage) per row of a batch
    DevBuf hubBits, hubRank, hubItems, hubSeg, hubCtl;
    DevBuf bWinSpill;                    // k_row_wave_window's spill lists
    DevBuf bWin, bWinTab;                // where the column windows begin in every row of B, the windows themselves (k_b_windows16, k_window_pick): rebuilt by the multiplies that need them
#ifndef BHS_WINDOW_DEFAULT
#define BHS_WINDOW_DEFAULT 1                  // (a measurement build may force the window kernels on every multiply: 2)
#endif
    int useWindowBitmap = BHS_WINDOW_DEFAULT;             // rows of 2 k .. 8 k entries one wave each, window by window (k_row_wave_window): 0 never, 1 if there are many, 2 always
    // row classes (bhs_class.hip.h): the structure of a row of C worked out once per class of rows
    int classGridMul = 4, classPerLane = 2, classMinProducts = 64;    // tuning hooks of k_class_rows
    int scanOnePass = 1;                 // stage 3 of the general pipeline: k_scan_onepass (0: the three scan kernels of rounds 1-3)
    unsigned scanEpoch = 0;              // tag of this multiply's tile words
    int classHeadsOn = 2;                // 2: one pass per matrix (k_class_fused: the wave that finds a row differing from the row before it takes it through the class table itself); 1: rounds 3-4's three launches (k_class_heads, k_class_rows on its lists, k_class_propagate); 0: every row through the table
    int classNumeric = 2;                // numeric kernel of the class path: 2 round 5's ring kernel (bhs_class_ring.hip.h) where its LDS fits, 1 round 4's (bhs_class_wg.hip.h), 0 k_class_numeric_atomic (round 2) always
    int classPath = 1;                   // 0 never; 1 for data sets whose rows of A and B have <= 64 entries and >= classMinProducts products on average (2: any), until one multiply finds
    int classState = 0;                  //   rows it cannot classify (classState -1: the data set stays on the general pipeline)
    DevBuf classB, classC, classTab, classInfo, classMap, classMapA, classRing, classRel, classLane, classHeads, classHeadCnt, classBigIdx, classBigMap;
    DevBuf longList, longPart;           // rows k_upper_bound / k_check_sorted leave to their *_long kernels; partial sums
    int mergeBitmapBins = 1;
    int hubMin = 1 << 17, hubItemProducts = 8192, hubMaxSlots = 0, hubAggregate = 1;
    int* hostSmall = nullptr;            // pinned mirror of `small`
    int* hostRowPtr = nullptr;           // pinned staging of rowPtrC for the host-pointer API
    size_t hostRowPtrCap = 0;
    hipStream_t copyStream = nullptr;    // D2H of rowPtrC overlaps the numeric stage
    // The bins of a stage touch disjoint rows, so their kernels are independent: they are launched on a few
    // side streams (forked from / joined into `stream` with events) and the small bins fill the tail of the
    // large ones instead of each paying its own ramp-up and drain.
    static constexpr int kBinStreams = 4;
    hipStream_t binStream[kBinStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evFork = nullptr, evJoin[kBinStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t ls = nullptr;            // stream the launch helpers use (== stream unless a bin is being launched)
    int ticketSlot = 101;                // ticket word (index into `small`) of the bin being launched; S_TICKET
    int concurrentBins = 2;              // 0 never, 1 always, 2 when a stage has >= 8 non-empty bins
    bool binsForked = false;
    int allowSmallB = 1;
    int waveFirst = 1;                   // same idea for rows bounded by maxRow(A) x maxRow(B) <= a wave table
    double avgRowA = 1.0, avgRowB = 1.0;
    int localA = 1;                      // hint from bhs_set_data: A's entries stay near the diagonal (k_row_period) -- the lane kernels want that
    int laneFirst = 1;                   // matrices of tiny rows: no upper-bound pass, the lane symbolic kernel counts products too
    int maxRowB = 0;
    int kernelStats = 0;                 // per-kernel-family hipEvent pairs (bhs_get_kernel_stats): off unless asked for -- they cost
                                         // 34 us of a 0.24 ms poisson5pt 1024^2 multiply; the four stage timers are always read
    bool specFailed = false;             // a lane-first / wave-first launch met a row beyond the bounds seen at set_data time
    int directBins = 1;                  // skip the queue of a stage whose rows all sit in the lane or quad bin
    int sortB = 1;                       // unsorted rows of B are sorted (on a private copy) at set_data time
    int laneRows = 1;                    // lane-per-row kernel for tiny rows: 0 never, 1 when every A row has <= 12 entries, 2 always
    int laneNumeric = 2;                 // numeric stage of lane-bin rows through k_row_lane too: 0 never, 1 always, 2 when K <= 8 (where it wins)
    int maxRowA = 0;
    int periodA = 1, periodB = 1;        // rows repeat the row this many rows back (k_row_period: a hint for k_class_heads)
    // compressed pattern of B for the symbolic pass (k_compress_b): 0 never, 1 (default) when it moves rows out of
    // the workgroup-per-row symbolic kernels -- the average row has more than 1536 products (beyond the 2048-slot
    // wave table) and the data has <= 60 % as many (block, mask) pairs as entries --, 2 always (needs sorted B rows
    // either way).  Where the plain pass already runs the wave kernels it does not pay: that kernel is bound by its
    // per-row work, and 2.8x fewer inserts buy back less than the compression costs (poisson27pt 160^3: 10.46 ms
    // with, 10.34 ms without).  A 3-dof FEM-like matrix (81 entries per row, 6561 products) goes from 12.5 to 5.9 ms.
    int compressB = 1;
    int cmpState = 0;                    // per data set: 0 undecided, 1 pays, -1 does not
    bool cmpActive = false;              // this multiply's symbolic wave bins run on the compressed pattern
    DevBuf sortList, sortCnt, sortK, sortV;   // bhs_csr_sort_indices_device: long-row list, its counter, scratch keys / values
    DevBuf cExt, cLen, cPair, symKey;    // per B row: pair extents, (entries, pairs); pairs; per A row: symbolic bin key
    hipEvent_t evScanDone = nullptr, evCopyDone = nullptr;
    bool wantHostRowPtr = false, rowPtrStaged = false;
    bool lazyOut = false;                // bhs_spgemm_symbolic: the library's own colIndC / valC are allocated by the first numeric range that needs them
    // options
    int forcePath = 0;
    int noPack32 = 0;                    // test hook: force 64-bit sort keys
    int wgPerCU = 0;                     // tuning hook: persistent workgroups per CU (0 = occupancy API)
    int classSuperRows = 0;              // tuning hook: consecutive rows a wave of the ring kernel takes (0 = a grid line of A, or kClassSuper)
    int lineA = 0;                       // rows per grid line of A if it has such lines, starting at row 0 (k_row_period), else 0
    int symLoadPct = 75, numLoadPct = 75; // max table load factor (percent) that decides a row's bin
    int maxTableLog2 = 15;
    // timing
    hipEvent_t ev[5] = {};
    std::vector<EventPair> evPool;
    size_t evUsed = 0;
    std::vector<StatRec> stats;
    double stageMs[4] = {0, 0, 0, 0};
    // per-handle (hence per-device) launch cache: resident workgroups per CU of every kernel instantiation, filled
    // by kernel_occupancy(), which also raises the dynamic-LDS limit of kernels that need more than 48 KB.  Both
    // are properties of (kernel, device): a process-wide static would hand a second device the first one's answers.
    std::map<std::tuple<const void*, int, size_t>, int> occ;
    std::unordered_map<const void*, size_t> occLds;   // largest dynamic-LDS size a kernel was granted so far
    // state handed from the symbolic half of a multiply (stages 1-3) to the numeric half (stage 4), which may be
    // run in row ranges (bhs_spgemm_symbolic / bhs_spgemm_numeric / bhs_spgemm_finish)
    struct PipeState {
        bool open = false;                // symbolic done, finish pending
        bool empty = false;               // empty product: nothing to launch
        bool noUpperBound = false, symDirect = false;
        bool useClass = false;            // numeric half: k_class_numeric
        int classMaxP = 0, classMaxNnz = 0, classMaxNA = 0, classMaxLB = 0, classMaxRing = 0, classMaxRing2 = 0, classMaxSlab = 0;
        int classBig = 0, classBigMaxP = 0;   // classes beyond the register kernels' tables (bhs_class_big.hip.h), their longest product list
        int laneK = 0, maxCnt = 0, hubRows = 0;
        BinSpec numSpec;
        int symStat[kMaxBins], numStat[kMaxBins];
        int fullCount[kMaxBins];          // numeric-bin histogram of all rows (from the scan)
        unsigned long long symSums[kMaxBins * 3];
        bool numDirectFull = false;
        int rangesRun = 0;
        bool bWinBuilt = false;          // bWin / bWinTab belong to this multiply
        long long midRows = 0, longRows = 0;   // rows of the numeric bins between the hash tables and the long rows; the long rows
    } ps;
    // external output arrays for the numeric half (bhs_set_output_device): C lands in the caller's buffers
    int* extCj = nullptr;
    const int* resCj = nullptr;          // the colIndC array the finished multiply wrote (own or bound): get_C serves no other
    value_t* extCx = nullptr;
    long long extCap = 0;
};

namespace {

// layout of the `small` device buffer (ints): see the enum below
enum { S_SYM_COUNT = 0, S_SYM_START = 16, S_SYM_CURSOR = 32, S_NUM_COUNT = 48, S_NUM_START = 64,
       S_NUM_CURSOR = 80, S_TOTAL_CT = 96 /* 2 ints = u64 */, S_TOTAL_C = 98 /* 2 ints = i64 */,
       S_ERR = 100, S_TICKET = 101 /* dynamic row scheduler of the workgroup-per-row kernels */,
       S_PAIRS = 102 /* 2 ints = u64: (block, mask) pairs of the compressed B */,
       S_SYM_SUMS = 104 /* kMaxBins x 3 u64: products, nnz(C rows), nnz(A rows) */,
       S_NUM_SUMS = 104 + 96,
       S_MAXCNT = 104 + 192 /* longest row of C */, S_UB_LONG = 104 + 193 /* rows on k_upper_bound's long list */,
       S_SCAN_TICKET = 104 + 194 /* tile numbers of k_scan_onepass */,
       S_ZERO_END = 104 + 195,   /* everything below is zeroed at the start of every spgemm */
       S_SORTED = 300, S_MAXROW = 301, S_OVF = 302 /* (free) */,
       S_LONG_B = 303 /* rows on k_check_sorted's long list */,
       S_TICKETS = 304 /* kMaxBins: one scheduler ticket per bin, bins run concurrently */,
       S_CT_SLOTS = 320 /* 64 x u64: product count of a lane-first multiply, spread over 64 counters */,
       S_SCAN = 448 /* bhs_set_data's scans: longest row of A, its period hint, the same for B, A's entries near the diagonal,
                       the length of A's grid lines */,
       S_SMALL_INTS = 454 };

template <int V> struct template_int { static constexpr int value = V; };

// zeroed: a NEW allocation is cleared on the handle's stream (the scans' tile words: k_scan_onepass takes a word whose epoch
// matches for published, and what hipMalloc hands out may hold the words another handle's scan left there at that epoch)
int ensure(bhs_handle* h, DevBuf& b, size_t bytes, bool zeroed = false)
{
    if (bytes <= b.cap && b.p) return BHS_SUCCESS;
    if (b.p) { BHS_HIP(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    if (bytes == 0) bytes = 16;
    BHS_HIP(hipMalloc(&b.p, bytes));
    b.cap = bytes;
    if (zeroed) BHS_HIP(hipMemsetAsync(b.p, 0, bytes, h->stream));
    return BHS_SUCCESS;
}

int ensure_host_rowptr(bhs_handle* h, size_t bytes)
{
    if (h->hostRowPtrCap >= bytes) return BHS_SUCCESS;
    if (h->hostRowPtr) BHS_HIP(hipHostFree(h->hostRowPtr));
    h->hostRowPtr = nullptr;
    h->hostRowPtrCap = 0;
    BHS_HIP(hipHostMalloc((void**)&h->hostRowPtr, bytes, hipHostMallocDefault));
    h->hostRowPtrCap = bytes;
    return BHS_SUCCESS;
}

void release(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// resident workgroups per CU of `kern` on this handle's device (>= 1), cached per handle; kernels with more than
// 48 KB of dynamic LDS get their limit raised here, once per handle
int kernel_occupancy(bhs_handle* h, const void* kern, int block, size_t smem, int* out)
{
    const auto key = std::make_tuple(kern, block, smem);              // (a kernel's dynamic LDS can depend on the data set)
    auto it = h->occ.find(key);
    if (it != h->occ.end()) { *out = it->second; return BHS_SUCCESS; }
    if (smem > 48 * 1024 && smem > h->occLds[kern]) {
        BHS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        h->occLds[kern] = smem;
    }
    int nb = 0;
    BHS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, block, smem));
    nb = std::max(1, nb);
    h->occ.emplace(key, nb);
    *out = nb;
    return BHS_SUCCESS;
}

// Output arrays of the numeric half: the library's own pool, or the caller's (bhs_set_output_device)
int* out_cj(bhs_handle* h) { return h->extCj ? h->extCj : (int*)h->Cj.p; }
value_t* out_cx(bhs_handle* h) { return h->extCx ? h->extCx : (value_t*)h->Cx.p; }

int stat_index(bhs_handle* h, const char* name)
{
    for (size_t i = 0; i < h->stats.size(); ++i)
        if (h->stats[i].name == name || strcmp(h->stats[i].name, name) == 0) return (int)i;
    StatRec r;
    r.name = name;
    h->stats.push_back(r);
    return (int)h->stats.size() - 1;
}

int timed_begin(bhs_handle* h, const char* name, EventPair** out)
{
    if (!h->kernelStats) {                                   // no events: the record still counts launches / rows
        static thread_local EventPair dummy;
        dummy.a = dummy.b = nullptr;
        dummy.stat = stat_index(h, name);
        *out = &dummy;
        return BHS_SUCCESS;
    }
    if (h->evUsed == h->evPool.size()) {
        EventPair p;
        BHS_HIP(hipEventCreate(&p.a));
        BHS_HIP(hipEventCreate(&p.b));
        p.stat = 0;
        h->evPool.push_back(p);
    }
    EventPair* p = &h->evPool[h->evUsed++];
    p->stat = stat_index(h, name);
    BHS_HIP(hipEventRecord(p->a, h->ls));
    *out = p;
    return BHS_SUCCESS;
}

int timed_end(bhs_handle* h, EventPair* p)
{
    if (!p->b) return BHS_SUCCESS;
    BHS_HIP(hipEventRecord(p->b, h->ls));
    return BHS_SUCCESS;
}

template <int LOG2TS, int BLOCK, bool NUM>
int launch_row_block(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt, const int* qnPtr = nullptr)
{
    constexpr int TS = 1 << LOG2TS;
    auto kern = k_row_block<TS, LOG2TS, BLOCK, NUM>;
    const size_t smem = sizeof(BlockSmem<TS, BLOCK, NUM>);
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), BLOCK, smem, &perCU));
    long long grid = std::max<long long>(1, std::min<long long>((long long)qn, (long long)h->numCU * perCU));
    BHS_HIP(hipMemsetAsync((int*)h->small.p + h->ticketSlot, 0, sizeof(int), h->ls));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), smem, h->ls, queue, qn, h->n, h->bSorted, h->dAj,
                       h->dAx, h->dBp, h->dBj, h->dBx, (const int*)h->ub.p, CpOrCnt, out_cj(h), out_cx(h),
                       (int*)h->small.p + S_ERR, (int*)h->small.p + h->ticketSlot, qnPtr);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

// Bitmap-accumulator slots: one n-bit bitmap + one rank word per 32 columns per resident workgroup, sized
// against 1/16 of the device memory, allocated once per column count; the kernel leaves the bitmaps all-zero.
int ensure_spa(bhs_handle* h)
{
    const size_t n = (size_t)std::max(h->n, 1), nWords = ((n + 31) / 32 + 3) & ~(size_t)3;   // 16-byte groups
    if (h->spaCols == h->n && h->spaSlots > 0 && !h->spaDirty) return BHS_SUCCESS;
    size_t freeB = 0, totalB = 0;
    BHS_HIP(hipMemGetInfo(&freeB, &totalB));
    const size_t perSlot = nWords * (sizeof(int) + sizeof(unsigned));
    long long slots = (long long)(std::min(totalB / 16, freeB / 2) / perSlot);
    slots = std::min<long long>(slots, h->spaMaxSlots > 0 ? (long long)h->spaMaxSlots : (long long)h->numCU);   // 1 per CU measured best
    // every row scans the whole bitmap: beyond 2^25 columns (4 MB of bits) the column-window path stays in charge
    if (slots < 8 || n > ((size_t)1 << 25)) { h->spaSlots = 0; return BHS_SUCCESS; }
    if (h->spaCols != h->n || h->spaSlots != (int)slots) {
        BHS_TRY(ensure(h, h->spaRank, (size_t)slots * nWords * sizeof(int)));      // rank words
        BHS_TRY(ensure(h, h->spaBits, (size_t)slots * nWords * sizeof(unsigned)));
    }
    BHS_HIP(hipMemsetAsync(h->spaBits.p, 0, (size_t)slots * nWords * sizeof(unsigned), h->stream));
    h->spaSlots = (int)slots;
    h->spaCols = h->n;
    h->spaDirty = false;
    return BHS_SUCCESS;
}

template <bool NUM>
int launch_row_spa(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt)
{
    constexpr int BLOCK = BHS_SPA_BLOCK;
    const long long grid = std::max<long long>(1, std::min<long long>(qn, h->spaSlots));
    int* small = (int*)h->small.p;
    BHS_HIP(hipMemsetAsync(small + h->ticketSlot, 0, sizeof(int), h->ls));
    hipLaunchKernelGGL((k_row_spa<BLOCK, NUM>), dim3((unsigned)grid), dim3(BLOCK), 0, h->ls, queue, qn, h->n,
                       h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h), out_cx(h),
                       small + h->ticketSlot, (int*)h->spaRank.p, (unsigned*)h->spaBits.p);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

// Long rows of matrices with <= 2^20 columns: bitmap accumulator in LDS, one 1024-lane workgroup per CU.
template <bool NUM>
int launch_row_bitmap_lds(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt, int reverse = 0, const int* qnDev = nullptr)
{
    auto kern = k_row_bitmap_lds<NUM>;
    int perCUunused = 1;     // (one workgroup per CU by design; the call raises the dynamic-LDS limit for this device)
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), kLdsBitmapBlock,
                             lds_bitmap_smem<NUM>(kLdsBitmapCols / 32), &perCUunused));
    const int nWords = (int)((((long long)std::max(h->n, 1) + 31) / 32 + 1023) / 1024 * 1024);
    const long long grid = std::max<long long>(1, std::min<long long>(qn, h->numCU));
    int* small = (int*)h->small.p;
    BHS_HIP(hipMemsetAsync(small + h->ticketSlot, 0, sizeof(int), h->ls));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kLdsBitmapBlock), lds_bitmap_smem<NUM>(nWords), h->ls, queue,
                       qn, nWords, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h), out_cx(h),
                       small + h->ticketSlot, reverse, qnDev);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

// Numeric pass by row classes on the rows [r0, r1): round 2's kernel (one LDS atomic per product)
template <int MAXU, int MAXV, int SE>
int launch_class_numeric_atomic_impl(bhs_handle* h, int r0, int r1)
{
    auto kern = k_class_numeric_atomic<MAXU, MAXV, SE>;
    const int accStride = (h->ps.classMaxNnz + 1 + 63) & ~63;      // (one spare slot for idle lanes)
    // staging area of a run: its rows' A entries (rounded up to whole 64-entry passes) and 64 entries of slack
    const int stageCap = ((kClassRunA * h->ps.classMaxNA + 63) & ~63) + 64;
    const size_t smem = (size_t)kClassWavesA * ((size_t)(accStride + stageCap) * sizeof(acc_t) + (size_t)stageCap * sizeof(int));
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64 * kClassWavesA, smem, &perCU));
    perCU = std::max(1, std::min(perCU, 32 / kClassWavesA));
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    const int mR = r1 - r0;
    const long long nRuns = ((long long)mR + kClassRunA - 1) / kClassRunA;
    long long grid = std::min<long long>((nRuns + kClassWavesA - 1) / kClassWavesA, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kClassWavesA), smem, h->ls, mR, h->dAp + r0, h->dAj, h->dAx,
                       h->dBp, h->dBx, (const int*)h->classC.p + r0, (const int4*)h->classInfo.p,
                       (const unsigned*)h->classMapA.p, (const int*)h->classRel.p, (const int*)h->Cp.p + r0, out_cj(h),
                       out_cx(h), accStride, stageCap, r0);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

template <int MAXU, int MAXV>
int launch_class_numeric_atomic_uv(bhs_handle* h, int r0, int r1)
{
    const int passes = (kClassRunA * h->ps.classMaxNA + 63) / 64;   // 64-entry passes that stage the A entries of a run
    if (passes <= 2) return launch_class_numeric_atomic_impl<MAXU, MAXV, 2>(h, r0, r1);
    if (passes <= 4) return launch_class_numeric_atomic_impl<MAXU, MAXV, 4>(h, r0, r1);
    return launch_class_numeric_atomic_impl<MAXU, MAXV, kClassRunA>(h, r0, r1);
}

int launch_class_numeric_atomic(bhs_handle* h, int r0, int r1)
{
    const int U = (h->ps.classMaxP + 63) / 64, V = (h->ps.classMaxNnz + 63) / 64;
    if (U <= 1 && V <= 1) return launch_class_numeric_atomic_uv<1, 1>(h, r0, r1);
    if (U <= 2 && V <= 1) return launch_class_numeric_atomic_uv<2, 1>(h, r0, r1);
    if (U <= 4 && V <= 2) return launch_class_numeric_atomic_uv<4, 2>(h, r0, r1);
    if (U <= 8 && V <= 4) return launch_class_numeric_atomic_uv<8, 4>(h, r0, r1);
    if (U <= 12 && V <= 2) return launch_class_numeric_atomic_uv<12, 2>(h, r0, r1);
    return launch_class_numeric_atomic_uv<16, 8>(h, r0, r1);
}


// Numeric pass of a multiply with big classes on the rows [r0, r1) (bhs_class_big.hip.h)
int launch_class_numeric_big(bhs_handle* h, int r0, int r1)
{
    auto kern = k_class_numeric_big;
    const int accStride = (h->ps.classMaxNnz + 3) & ~3, stageCap = (h->ps.classMaxNA + 3) & ~3;
    const int descCap = (std::max(h->ps.classMaxP, h->ps.classBigMaxP) + 3) & ~3;
    // rows per group: the period sampled at hand-over time (the unknowns of a node share their columns of A); waves per
    // workgroup: twelve when the groups' accumulator sets still fit the LDS, else eight, else no groups
    int rmax = h->periodA >= 2 && h->periodA <= kClassBigMaxGroup ? h->periodA : 1, waves = 8;
    auto lds = [&](int rm, int wv, int range) {
        return (size_t)wv * ((size_t)rm * (accStride + stageCap) * sizeof(acc_t) + (size_t)stageCap * sizeof(int)) +
               sizeof(int) * ((size_t)descCap + (size_t)rm * accStride + 2 * (size_t)range + 32);
    };
    auto range_of = [&](int rm, int wv) { return kClassBigRangeMax / (wv * rm) * (wv * rm); };
    const size_t ldsMax = 160 * 1024;
    if (rmax > 1) {
        if (lds(rmax, 12, range_of(rmax, 12)) <= ldsMax) waves = 12;
        else if (lds(rmax, 8, range_of(rmax, 8)) > ldsMax) rmax = 1;
    }
    const int range = range_of(rmax, waves);
    const size_t smem = lds(rmax, waves, range);
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64 * waves, smem, &perCU));
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    const int mR = r1 - r0;
    const long long nRanges = ((long long)mR + range - 1) / range;
    long long grid = std::min<long long>(nRanges, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    if (h->verbose > 1) printf("  [class numeric (big): rows in groups of %d, %d waves per workgroup, ranges of %d rows, %d workgroups per CU, %zu bytes of LDS each, grid %lld]\n", rmax, waves, range, perCU, smem, grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * waves), smem, h->ls, mR, h->dAp + r0, h->dAj, h->dAx,
                       h->dBp, h->dBx, (const int*)h->classC.p + r0, (const int4*)h->classInfo.p, (const unsigned*)h->classMapA.p,
                       (const int*)h->classBigIdx.p, (const unsigned*)h->classBigMap.p, (const int*)h->classRel.p,
                       (const int*)h->Cp.p + r0, out_cj(h), out_cx(h), accStride, stageCap, descCap, rmax, range, r0);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

// LDS of a wave of the ring kernel: slots of a row of C (one spare for idle lanes), a run's A values, (longest chain + 1)
// slabs of the neediest class; all in multiples of 16 bytes
struct RingLds { int accStride, stageCap, ringCap; size_t bytes; };
RingLds class_ring_lds(bhs_handle* h)
{
    RingLds l;
    l.accStride = (h->ps.classMaxNnz + 1 + 3) & ~3;
    l.stageCap = (kClassRun * h->ps.classMaxNA + 3) & ~3;
    l.ringCap = (int)(((long long)h->ps.classMaxRing + 3) & ~3ll);
    l.bytes = (size_t)(l.accStride + l.stageCap) * sizeof(acc_t) + (size_t)l.ringCap * sizeof(value_t);
    return l;
}

// rows of a few thousand entries window by window, a wave each (bhs_row_window.hip.h): the windows and the index of B
// (on h->stream, before the bins fork: every bin's stream waits for it)
int ensure_b_windows(bhs_handle* h)
{
    // (rebuilt by every multiply that uses it, 0.12 ms: borrowed arrays may change between multiplies -- every other hint kept
    // from bhs_set_data time is verified on the device where it is used, a stale index of B's windows could not be)
    if (h->ps.bWinBuilt) return BHS_SUCCESS;
    BHS_TRY(ensure(h, h->bWinTab, (kWwBuckets + kWwTabInts) * sizeof(int)));
    BHS_TRY(ensure(h, h->bWin, (size_t)std::max(h->k, 1) * (size_t)kWwStride * sizeof(unsigned short)));
    unsigned* hist = (unsigned*)h->bWinTab.p;
    int* tab = (int*)h->bWinTab.p + kWwBuckets;
    EventPair* ep = nullptr;
    BHS_TRY(timed_begin(h, "b_windows", &ep));
    BHS_HIP(hipMemsetAsync(hist, 0, kWwBuckets * sizeof(unsigned), h->stream));
    const long long gh = std::max<long long>(1, std::min<long long>(((long long)h->nnzB + 4095) / 4096, (long long)h->numCU * 4));
    hipLaunchKernelGGL(k_window_hist, dim3((unsigned)gh), dim3(256), 0, h->stream, (long long)h->nnzB, h->dBj, hist);
    hipLaunchKernelGGL(k_window_pick, dim3(1), dim3(64), 0, h->stream, h->n, (long long)h->nnzB, (const unsigned*)hist, tab);
    hipLaunchKernelGGL(k_b_windows16, dim3((unsigned)((h->k + 255) / 256)), dim3(256), 0, h->stream, h->k, (const int*)tab, h->dBp, h->dBj,
                       (unsigned short*)h->bWin.p);
    BHS_HIP(hipGetLastError());
    BHS_TRY(timed_end(h, ep));
    h->stats[ep->stat].launches++;
    h->ps.bWinBuilt = true;
    return BHS_SUCCESS;
}

// one wave per row, windows of 2^16 columns: the numeric bins between the hash tables and the long rows
template <bool WG>   // false: one wave per row (k_row_wave_window); true: 256 lanes per row (k_row_wg_window, the long rows)
int launch_row_window(bhs_handle* h, const int4* queue, int qn, int* Cp, int reverse = 0)
{
    auto kern = WG ? k_row_wg_window : k_row_wave_window;
    const int block = WG ? kWgLanes : 64;
    const size_t smem = WG ? wg_window_smem() : wave_window_smem();
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), block, smem, &perCU));
    perCU = std::max(1, std::min(perCU, 32));
    const long long grid = std::max<long long>(1, std::min<long long>(qn, (long long)h->numCU * perCU));
    int* small = (int*)h->small.p;
    // the rows it hands on (long rows of A, rows crowded into one window): a list of its own per launch -- the bins run
    // side by side -- then k_row_bitmap_lds on that list, its length read on the device
    BHS_TRY(ensure(h, h->bWinSpill, ((size_t)std::max(h->m, 1) + 2 * kMaxBins + 2) * sizeof(int4)));
    int4* spill = (int4*)h->bWinSpill.p + (queue - (const int4*)h->queue.p) + 2 * (h->ticketSlot - S_TICKETS + 1);
    BHS_HIP(hipMemsetAsync(spill, 0, sizeof(int4), h->ls));
    BHS_HIP(hipMemsetAsync(small + h->ticketSlot, 0, sizeof(int), h->ls));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), smem, h->ls, queue, qn, (const int*)h->bWinTab.p + kWwBuckets, h->dAj, h->dAx,
                       h->dBp, (const unsigned short*)h->bWin.p, h->dBj, h->dBx, out_cj(h), out_cx(h), small + h->ticketSlot, reverse, spill);
    BHS_HIP(hipGetLastError());
    return launch_row_bitmap_lds<true>(h, spill + 1, h->numCU, Cp, 0, (const int*)spill);
}

// Numeric pass by row classes on the rows [r0, r1): the ring kernel (bhs_class_wg.hip.h)
template <int MAXU, int MAXV, int SE, int MAXJ>
int launch_class_numeric_impl(bhs_handle* h, int r0, int r1)
{
    auto kern = k_class_numeric<MAXU, MAXV, SE, MAXJ>;
    const RingLds lds = class_ring_lds(h);
    const int accStride = lds.accStride, stageCap = lds.stageCap, ringCap = lds.ringCap;
    const size_t smem = lds.bytes;
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64, smem, &perCU));
    perCU = std::max(1, std::min(perCU, 32));
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    const int mR = r1 - r0;
    const int superRows = std::max(4 * kClassRun, h->classSuperRows > 0 ? h->classSuperRows : (h->lineA > 0 ? h->lineA : kClassSuper));
    const long long nSuper = ((long long)mR + superRows - 1) / superRows;
    long long grid = std::min<long long>(nSuper, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    if (h->verbose > 1) printf("  [class numeric (ring): %d waves per CU by the occupancy API, %d used, %zu bytes of LDS each, grid %lld]\n", perCU, useCU, smem, grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), smem, h->ls, mR, h->dAp + r0, h->dAj, h->dAx,
                       (long long)h->nnzA, h->dBp, h->dBx, (long long)h->nnzB, (const int*)h->classC.p + r0, (const int4*)h->classInfo.p,
                       (const unsigned*)h->classMap.p, (const int*)h->classRel.p, (const int*)h->classLane.p, (const int*)h->Cp.p + r0, out_cj(h),
                       out_cx(h), accStride, stageCap, ringCap, r0, superRows);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

template <int MAXU, int MAXV>
int launch_class_numeric_uv(bhs_handle* h, int r0, int r1)
{
    const int passes = (kClassRun * h->ps.classMaxNA + 63) / 64;   // 64-entry passes that stage the A entries of a run
    const bool smallSlab = h->ps.classMaxSlab <= 2 * 64 * kClassEpl;
    if (passes <= 4) return smallSlab ? launch_class_numeric_impl<MAXU, MAXV, 4, 2>(h, r0, r1) : launch_class_numeric_impl<MAXU, MAXV, 4, kClassMaxJ>(h, r0, r1);
    return smallSlab ? launch_class_numeric_impl<MAXU, MAXV, kClassRun, 2>(h, r0, r1) : launch_class_numeric_impl<MAXU, MAXV, kClassRun, kClassMaxJ>(h, r0, r1);
}

// (false: some class's slab or ring is beyond what the ring kernel keeps in LDS -- the caller takes the atomic kernel)
bool class_ring_fits(bhs_handle* h)
{
    if (h->ps.classMaxRing < 0 || h->ps.classMaxRing == 0x7fffffff) return false;
    return class_ring_lds(h).bytes <= 40 * 1024;
}

int launch_class_numeric(bhs_handle* h, int r0, int r1)
{
    const int U = (h->ps.classMaxP + 63) / 64, V = (h->ps.classMaxNnz + 63) / 64;
    if (U <= 1 && V <= 1) return launch_class_numeric_uv<1, 1>(h, r0, r1);
    if (U <= 2 && V <= 1) return launch_class_numeric_uv<2, 1>(h, r0, r1);
    if (U <= 4 && V <= 2) return launch_class_numeric_uv<4, 2>(h, r0, r1);
    if (U <= 8 && V <= 4) return launch_class_numeric_uv<8, 4>(h, r0, r1);
    if (U <= 12 && V <= 2) return launch_class_numeric_uv<12, 2>(h, r0, r1);
    return launch_class_numeric_uv<16, 8>(h, r0, r1);
}

// ... round 5's ring kernel (bhs_class_ring.hip.h): the ring a power of two of bytes at LDS address 0, then the slots of a
// row of C, then the row's A values with a zero behind them
struct Ring2Lds { int ringBytes, accStride, afixCap; size_t bytes; };   // (the ring: what the neediest class keeps, bhs_class.hip.h CS_RINGFULL / CS_RINGONE)
Ring2Lds class_ring2_lds(bhs_handle* h)
{
    Ring2Lds l;
    l.ringBytes = (int)std::min<long long>((((long long)h->ps.classMaxRing2 * (long long)sizeof(value_t)) + 15) & ~15ll, 1 << 30);
    l.accStride = (h->ps.classMaxNnz + 2) & ~1;
    l.afixCap = (h->ps.classMaxNA + 2) & ~1;
    l.bytes = (size_t)l.ringBytes + (size_t)(l.accStride + l.afixCap) * sizeof(acc_t);
    return l;
}
// (false: some class's slab is beyond a slab's load instructions, or the ring beyond the 16 bits of a product's place)
bool class_ring2_fits(bhs_handle* h)
{
    if (h->ps.classMaxRing2 <= 0 || h->ps.classMaxRing2 == 0x7fffffff) return false;
    const Ring2Lds l = class_ring2_lds(h);
    return l.ringBytes <= 32 * 1024 && l.bytes <= 40 * 1024;
}

template <int MAXU, int MAXV, int MAXJ>
int launch_class_ring_impl(bhs_handle* h, int r0, int r1)
{
    auto kern = k_class_ring<MAXU, MAXV, MAXJ>;
    const Ring2Lds lds = class_ring2_lds(h);
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64, lds.bytes, &perCU));
    perCU = std::max(1, std::min(perCU, 32));
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    const int mR = r1 - r0;
    const int superRows = std::max(32, h->classSuperRows > 0 ? h->classSuperRows : (h->lineA > 0 ? h->lineA : kClassSuper));
    const long long nSuper = ((long long)mR + superRows - 1) / superRows;
    long long grid = std::min<long long>(nSuper, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    const int chunkRows = std::max(1, std::min(64, 128 / std::max(1, h->ps.classMaxNA)));   // whole rows, <= 128 entries of A
    if (h->verbose > 1) printf("  [class numeric (ring, round 5): %d waves per CU by the occupancy API, %d used, %zu bytes of LDS each, grid %lld, %d rows per chunk]\n", perCU, useCU, lds.bytes, grid, chunkRows);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds.bytes, h->ls, mR, h->dAp + r0, h->dAj, h->dAx,
                       (long long)h->nnzA, h->dBp, h->dBx, (long long)h->nnzB, (const int*)h->classC.p + r0, (const int4*)h->classInfo.p,
                       (const unsigned*)h->classRing.p, (const int*)h->classRel.p, (const int*)h->classLane.p, (const int*)h->Cp.p + r0, out_cj(h),
                       out_cx(h), lds.ringBytes, lds.accStride, r0, superRows, chunkRows);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

template <int MAXU, int MAXV>
int launch_class_ring_uv(bhs_handle* h, int r0, int r1)
{
    const bool smallSlab = h->ps.classMaxSlab <= 2 * 64 * kClassEpl;
    return smallSlab ? launch_class_ring_impl<MAXU, MAXV, 2>(h, r0, r1) : launch_class_ring_impl<MAXU, MAXV, kClassMaxJ>(h, r0, r1);
}

int launch_class_ring(bhs_handle* h, int r0, int r1)
{
    const int U = (h->ps.classMaxP + 63) / 64, V = (h->ps.classMaxNnz + 63) / 64;
    if (U <= 1 && V <= 1) return launch_class_ring_uv<1, 1>(h, r0, r1);
    if (U <= 2 && V <= 1) return launch_class_ring_uv<2, 1>(h, r0, r1);
    if (U <= 4 && V <= 2) return launch_class_ring_uv<4, 2>(h, r0, r1);
    if (U <= 8 && V <= 4) return launch_class_ring_uv<8, 4>(h, r0, r1);
    if (U <= 12 && V <= 2) return launch_class_ring_uv<12, 2>(h, r0, r1);
    return launch_class_ring_uv<16, 8>(h, r0, r1);
}

// Hub rows: plan -> mark -> count [-> emit -> place], in batches of as many rows as there are bitmap slots.
template <bool NUM>
int launch_hub(bhs_handle* h, const int4* hubQ, int nHub, int* CpOrCnt)
{
    const HubGeom g = hub_geom(h->n);
    size_t freeB = 0, totalB = 0;
    BHS_HIP(hipMemGetInfo(&freeB, &totalB));
    const size_t perSlot = ((size_t)g.slotWords + (size_t)g.nW) * sizeof(int);
    long long slots = (long long)(std::min(totalB / 16, freeB / 2) / perSlot);
    if (h->hubMaxSlots > 0) slots = std::min<long long>(slots, h->hubMaxSlots);
    slots = std::min<long long>(slots, nHub);
    if (slots < 1) return BHS_ERR_ALLOC;
    // every chunk of 512 A entries yields ceil(products / item) items
    const long long cap = (long long)h->nnzA / kHubChunk + h->nnzCt / h->hubItemProducts + 2LL * nHub + 16;
    if (cap > 0x7fffffffLL) return BHS_ERR_ALLOC;
    BHS_TRY(ensure(h, h->hubBits, (size_t)slots * (size_t)g.slotWords * sizeof(unsigned)));
    if (NUM) {
        BHS_TRY(ensure(h, h->hubRank, (size_t)slots * (size_t)g.nW * sizeof(int)));
        BHS_TRY(ensure(h, h->hubSeg, (size_t)slots * (size_t)g.seg * sizeof(int)));
    }
    BHS_TRY(ensure(h, h->hubItems, (size_t)cap * sizeof(int4)));
    BHS_TRY(ensure(h, h->hubCtl, 16 * sizeof(int)));
    int* ctl = (int*)h->hubCtl.p;                 // [0] item count, [1] ticket of mark, [2] ticket of place
    int* err = (int*)h->small.p + S_ERR;
    const unsigned grid = (unsigned)(h->numCU * 2);
    for (int b0 = 0; b0 < nHub; b0 += (int)slots) {
        const int nb = std::min<int>((int)slots, nHub - b0);
        const int4* q = hubQ + b0;
        BHS_HIP(hipMemsetAsync(h->hubBits.p, 0, (size_t)nb * (size_t)g.slotWords * sizeof(unsigned), h->ls));
        BHS_HIP(hipMemsetAsync(ctl, 0, 16 * sizeof(int), h->ls));
        hipLaunchKernelGGL(k_hub_plan, dim3((unsigned)nb * kHubPlanWG), dim3(256), 0, h->ls, q, h->dAj, h->dBp, (int4*)h->hubItems.p,
                           ctl, (int)cap, h->hubItemProducts, NUM ? (int*)nullptr : CpOrCnt, err);
        hipLaunchKernelGGL(k_hub_mark<NUM>, dim3(grid), dim3(kHubBlock), 0, h->ls, (const int4*)h->hubItems.p,
                           (const int*)ctl, q, h->dAj, h->dBp, h->dBj, (unsigned*)h->hubBits.p, g.slotWords, g.nW, ctl + 1,
                           (h->hubAggregate && h->bSorted) ? 1 : 0);
        hipLaunchKernelGGL(k_hub_count<NUM>, dim3((unsigned)(nb * g.seg)), dim3(kHubBlock), 0, h->ls, q,
                           (const unsigned*)h->hubBits.p, g.slotWords, g.seg, g.segW, (int*)h->hubSeg.p, CpOrCnt);
        if constexpr (NUM) {
            hipLaunchKernelGGL(k_hub_emit, dim3((unsigned)(nb * g.seg)), dim3(kHubBlock), 0, h->ls, q,
                               (const unsigned*)h->hubBits.p, g.slotWords, g.nW, g.seg, g.segW, (const int*)h->hubSeg.p,
                               (int*)h->hubRank.p, out_cj(h), out_cx(h));
            hipLaunchKernelGGL(k_hub_place, dim3(grid), dim3(kHubBlock), 0, h->ls, (const int4*)h->hubItems.p,
                               (const int*)ctl, q, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, (const unsigned*)h->hubBits.p,
                               g.slotWords, g.nW, (const int*)h->hubRank.p, out_cx(h), ctl + 2);
        }
        BHS_HIP(hipGetLastError());
    }
    return BHS_SUCCESS;
}

template <int LOG2TS, bool NUM, bool PACK32, bool SMALLB>
int launch_row_wave_impl(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt)
{
    constexpr int TS = 1 << LOG2TS;
    auto kern = k_row_wave<TS, LOG2TS, NUM, PACK32, SMALLB>;
    constexpr int WPB = kWavesPerBlock;
    const size_t smem = sizeof(WaveSmem<TS, NUM, PACK32>) * WPB;
    int perCU = 1;    // resident 64-lane workgroups per CU: registers, LDS and the 32-wave cap all count
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64 * WPB, smem, &perCU));
    perCU = std::max(1, std::min(perCU, 32 / WPB));
    if (h->verbose > 1) printf("  [%s TS=%d] occupancy API: %d workgroups/CU, smem %zu B\n", NUM ? "numeric" : "symbolic", TS, perCU, smem);
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    long long grid = std::min<long long>(((long long)qn + WPB - 1) / WPB, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);       // XCD-aware schedule needs a multiple of 8
    // XCD chunk: BHS_XCD_CHUNK entries for long queues; short queues get >= 8 chunks per XCD
    int chunkLog2 = 0;
    while ((2 << chunkLog2) <= BHS_XCD_CHUNK && (128LL << chunkLog2) <= (long long)qn) ++chunkLog2;
    const bool wf = !NUM && queue == nullptr;             // wave-first symbolic pass: rows straight from rowPtrA
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WPB), smem, h->ls, queue, qn, chunkLog2, h->dAj, h->dAx,
                       h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h), out_cx(h), h->dAp,
                       wf ? (int*)h->ub.p : (int*)nullptr,
                       wf ? (unsigned long long*)((int*)h->small.p + S_CT_SLOTS) : (unsigned long long*)nullptr,
                       (int*)h->small.p + S_ERR);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

template <int LOG2TS, bool NUM>
int launch_row_wave(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt)
{
    // byte offsets into colIndB / valB fit 32 bits: the common case gets its own instantiation
    const bool smallB = h->allowSmallB && h->nnzB < (1 << 29);
    if constexpr (NUM) {
        // 32-bit sort keys when every column index fits beside the slot index
        const bool pack32 = (long long)h->n <= (1LL << (32 - LOG2TS)) && !h->noPack32;
        if (pack32) {
            if (smallB) return launch_row_wave_impl<LOG2TS, true, true, true>(h, queue, qn, CpOrCnt);
            return launch_row_wave_impl<LOG2TS, true, true, false>(h, queue, qn, CpOrCnt);
        }
        if (smallB) return launch_row_wave_impl<LOG2TS, true, false, true>(h, queue, qn, CpOrCnt);
        return launch_row_wave_impl<LOG2TS, true, false, false>(h, queue, qn, CpOrCnt);
    } else {
        if (smallB) return launch_row_wave_impl<LOG2TS, false, false, true>(h, queue, qn, CpOrCnt);
        return launch_row_wave_impl<LOG2TS, false, false, false>(h, queue, qn, CpOrCnt);
    }
}

template <int LOG2TS>
int launch_row_wave_csym(bhs_handle* h, const int4* queue, int qn, int* cnt)
{
    constexpr int TS = 1 << LOG2TS;
    auto kern = k_row_wave_csym<TS, LOG2TS>;
    constexpr int WPB = kWavesPerBlock;
    const size_t smem = sizeof(CsymSmem<TS>) * WPB;
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64 * WPB, smem, &perCU));
    perCU = std::max(1, std::min(perCU, 32 / WPB));
    if (h->verbose > 1) printf("  [symbolic/compressed TS=%d] occupancy API: %d workgroups/CU, smem %zu B\n", TS, perCU, smem);
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    long long grid = std::min<long long>(((long long)qn + WPB - 1) / WPB, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    int chunkLog2 = 0;
    while ((2 << chunkLog2) <= BHS_XCD_CHUNK && (128LL << chunkLog2) <= (long long)qn) ++chunkLog2;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WPB), smem, h->ls, queue, qn, chunkLog2, h->dAj,
                       (const int2*)h->cExt.p, (const int2*)h->cPair.p, cnt, (int*)h->small.p + S_ERR);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

int launch_compress_b(bhs_handle* h)
{
    int G = 1 << h->logL;                       // lanes per row of B: its average length, 2..16
    G = std::max(2, std::min(G, 16));
    const int rowsPerBlock = 256 / G * 4;       // 4 rows in flight per lane group
    long long grid = ((long long)h->k + rowsPerBlock - 1) / rowsPerBlock;
    grid = std::max<long long>(1, std::min<long long>(grid, (long long)h->numCU * 8));
    int* small = (int*)h->small.p;
#define BHS_CB(GG)                                                                                          \
    case GG:                                                                                                \
        hipLaunchKernelGGL(k_compress_b<GG>, dim3((unsigned)grid), dim3(256), 0, h->stream, h->k, h->dBp,   \
                           h->dBj, (int2*)h->cExt.p, (int2*)h->cLen.p, (int2*)h->cPair.p,                                     \
                           (unsigned long long*)(small + S_PAIRS));                                         \
        break;
    switch (G) {
        BHS_CB(2) BHS_CB(4) BHS_CB(8) BHS_CB(16)
        default: return BHS_ERR_INTERNAL;
    }
#undef BHS_CB
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

template <bool NUM, bool PACK32>
int launch_row_quad_impl(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt)
{
    auto kern = k_row_quad<NUM, PACK32>;
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64, 0, &perCU));
    perCU = std::min(perCU, 32);
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    long long grid = std::min<long long>(((long long)qn + 3) / 4, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, h->ls, queue, qn, h->dAp, h->dAj, h->dAx, h->dBp, h->dBj,
                       h->dBx, CpOrCnt, out_cj(h), out_cx(h), (int*)h->small.p + S_ERR);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

template <bool NUM>
int launch_row_quad(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt)
{
    if constexpr (NUM) {
        if ((long long)h->n <= (1LL << 26) && !h->noPack32) return launch_row_quad_impl<true, true>(h, queue, qn, CpOrCnt);
        return launch_row_quad_impl<true, false>(h, queue, qn, CpOrCnt);
    } else {
        return launch_row_quad_impl<false, false>(h, queue, qn, CpOrCnt);
    }
}

template <bool NUM>
int launch_row_lane(bhs_handle* h, int K, const int4* queue, int qn, int* CpOrCnt, int* ubOut = nullptr,
                    unsigned long long* ctSlots = nullptr)
{
    const unsigned grid = (unsigned)(((long long)qn + 255) / 256);
    const bool smallB = h->allowSmallB && h->nnzB < (1 << 29);
#define BHS_LANE(KK)                                                                                          \
    case KK:                                                                                                  \
        if (smallB)                                                                                           \
            hipLaunchKernelGGL((k_row_lane<KK, NUM, true>), dim3(grid), dim3(256), 0, h->ls, queue, qn,       \
                               h->dAp, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h),                    \
                               out_cx(h), ubOut, ctSlots, (int*)h->small.p + S_ERR);                  \
        else                                                                                                  \
            hipLaunchKernelGGL((k_row_lane<KK, NUM, false>), dim3(grid), dim3(256), 0, h->ls, queue, qn,      \
                               h->dAp, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h),                    \
                               out_cx(h), ubOut, ctSlots, (int*)h->small.p + S_ERR);                  \
        break;
    switch (K) {
        BHS_LANE(4) BHS_LANE(6) BHS_LANE(8) BHS_LANE(10) BHS_LANE(12)
        default: return BHS_ERR_INTERNAL;
    }
#undef BHS_LANE
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}

// does this bin run k_row_bitmap_lds? (long rows, and the numeric workgroup bins from ldsBitmapMinLog2 up)
template <bool NUM>
bool bin_takes_lds_bitmap(const bhs_handle* h, const KernelCfg& c)
{
    if (!(h->useSpa && h->maxTableLog2 >= 15 && h->useLdsBitmap && h->n <= kLdsBitmapCols)) return false;
    if (c.win) return true;
    return NUM && c.block > 64 && c.log2ts >= h->ldsBitmapMinLog2 && h->forcePath == 0;
}

// (only when the multiply has enough such rows to fill the device's wave slots several times over -- h->ps.midRows, set
// before the numeric bins fork: the index of B costs 0.12 ms to build, and a few thousand rows are not worth it.  Measured:
// R-MAT 2^20 rows, 38 k such rows, 18.4 -> 15.1 ms; the two web-graph stand-ins, 1.2 k / 5 k such rows, 1.8 -> 2.1 / 3.1 -> 3.2 ms.)
template <bool NUM>
bool bin_takes_wave_window(const bhs_handle* h, const KernelCfg& c)
{
    if (!(NUM && bin_takes_lds_bitmap<NUM>(h, c) && h->useWindowBitmap && h->bSorted && h->maxRowB < 65536 &&
          (long long)h->n <= ((long long)kWwBuckets << kWwBucketLog2)))
        return false;
    if (c.win) return h->useWindowBitmap >= 2 || h->ps.longRows >= 16LL * h->numCU;   // (256 lanes per row: k_row_wg_window)
    return h->useWindowBitmap >= 2 || h->ps.midRows >= 32LL * h->numCU;
}

template <bool NUM>
int dispatch_bin(bhs_handle* h, const KernelCfg& c, const int4* queue, int qn, int* CpOrCnt, int reverse = 0)
{
    if (c.block == 16) return launch_row_quad<NUM>(h, queue, qn, CpOrCnt);
    if (queue != nullptr && bin_takes_wave_window<NUM>(h, c)) return c.win ? launch_row_window<true>(h, queue, qn, CpOrCnt, reverse) : launch_row_window<false>(h, queue, qn, CpOrCnt, reverse);
    if (bin_takes_lds_bitmap<NUM>(h, c)) return launch_row_bitmap_lds<NUM>(h, queue, qn, CpOrCnt, reverse);
    if (c.win && h->useSpa && h->maxTableLog2 >= 15 && h->spaSlots > 0) return launch_row_spa<NUM>(h, queue, qn, CpOrCnt);
    const int lg = std::min(c.log2ts, h->maxTableLog2);
    const bool win = c.win || lg < c.log2ts;   // a capped table can overflow => window variant
    if constexpr (!NUM) {
        if (h->cmpActive && c.block == 64 && !win && h->forcePath != 2) {
            switch (lg) {
                case 6: return launch_row_wave_csym<6>(h, queue, qn, CpOrCnt);
                case 7: return launch_row_wave_csym<7>(h, queue, qn, CpOrCnt);
                case 8: return launch_row_wave_csym<8>(h, queue, qn, CpOrCnt);
                case 9: return launch_row_wave_csym<9>(h, queue, qn, CpOrCnt);
                case 10: return launch_row_wave_csym<10>(h, queue, qn, CpOrCnt);
                case 11: return launch_row_wave_csym<11>(h, queue, qn, CpOrCnt);
                case 12: return launch_row_wave_csym<12>(h, queue, qn, CpOrCnt);
                default: break;
            }
        }
    }
#define BHS_WAVE(LG) \
    if (lg == LG && c.block == 64 && !win && h->forcePath != 2) return launch_row_wave<LG, NUM>(h, queue, qn, CpOrCnt)
    BHS_WAVE(6); BHS_WAVE(7); BHS_WAVE(8); BHS_WAVE(9); BHS_WAVE(10); BHS_WAVE(11);
    if constexpr (!NUM) { BHS_WAVE(12); }
#undef BHS_WAVE
    // long rows: workgroup per row (every instantiation carries the column-window loop)
    (void)win;
    if constexpr (!NUM) {
        if (lg >= 15) return launch_row_block<15, 1024, false>(h, queue, qn, CpOrCnt);
        if (lg >= 13) return launch_row_block<13, 256, false>(h, queue, qn, CpOrCnt);
        return launch_row_block<8, 256, false>(h, queue, qn, CpOrCnt);       // capped tables (tests): many windows
    } else {
        if (lg >= 13) return launch_row_block<13, 512, true>(h, queue, qn, CpOrCnt);
        if (lg >= 12) return launch_row_block<12, 256, true>(h, queue, qn, CpOrCnt);
        if (lg >= 11) return launch_row_block<11, 256, true>(h, queue, qn, CpOrCnt);
        return launch_row_block<8, 256, true>(h, queue, qn, CpOrCnt);
    }
}

const char* kSymNames[kNumSymBins] = {"", "symbolic_quad<64>", "symbolic_wave<64>", "symbolic_wave<128>", "symbolic_wave<256>",
                                      "symbolic_wave<512>", "symbolic_wave<1024>", "symbolic_wave<2048>",
                                      "symbolic_wave<4096>", "symbolic_wg<8192>", "symbolic_wg<32768>",
                                      "symbolic_long_rows"};
const char* kNumNames[kNumNumBins] = {"", "numeric_quad<64>", "numeric_wave<64>", "numeric_wave<128>", "numeric_wave<256>",
                                      "numeric_wave<512>", "numeric_wave<1024>",
                                      "numeric_wg<2048>", "numeric_wg<4096>", "numeric_wg<8192>", "numeric_long_rows"};

int launch_upper_bound(bhs_handle* h, const BinSpec& spec, bool cmp, int keyMax)
{
    const int G = h->ubG;
    const int rowsPerBlock = 256 / G;
    const int R = ub_rows_in_flight(G);
    long long grid = ((long long)h->m + rowsPerBlock * R - 1) / (rowsPerBlock * R);   // R rows per lane group per pass
    // every block ends with a handful of same-address atomics (nnzCt, bin histogram): short-row inputs, whose blocks
    // cover many rows each, run fewer and longer blocks (poisson5pt 1024^2: 0.066 -> 0.048 ms)
    grid = std::max<long long>(1, std::min<long long>(grid, (long long)h->numCU * (G <= 8 ? 4 : 32)));
    int* small = (int*)h->small.p;
    // rows of A beyond kUbLongA entries (if the data set has any: maxRowA is the hint) are listed and summed by
    // k_upper_bound_long, 16 workgroups per row
    const bool useLong = h->maxRowA > h->ubLong;
    int2* longList = nullptr;
    if (useLong) {
        const size_t cap = (size_t)h->nnzA / h->ubLong + 2;       // a listed row of len entries takes <= len / ubLong entries
        BHS_TRY(ensure(h, h->longList, cap * sizeof(int2)));
        BHS_TRY(ensure(h, h->longPart, cap * 2 * sizeof(long long)));
        longList = (int2*)h->longList.p;
    }
#define BHS_UB(GG)                                                                                       \
    case GG:                                                                                             \
        if (cmp)                                                                                         \
            hipLaunchKernelGGL((k_upper_bound<GG, true>), dim3((unsigned)grid), dim3(256), 0, h->stream, \
                               h->m, h->dAp, h->dAj, h->dBp, (int*)h->ub.p, (int*)h->Cp.p,               \
                               (unsigned long long*)(small + S_TOTAL_CT), small + S_SYM_COUNT, spec,     \
                               (const int2*)h->cLen.p, (int*)h->symKey.p, keyMax, longList,              \
                               small + S_UB_LONG, h->ubLong);                                            \
        else                                                                                             \
            hipLaunchKernelGGL((k_upper_bound<GG, false>), dim3((unsigned)grid), dim3(256), 0, h->stream,\
                               h->m, h->dAp, h->dAj, h->dBp, (int*)h->ub.p, (int*)h->Cp.p,               \
                               (unsigned long long*)(small + S_TOTAL_CT), small + S_SYM_COUNT, spec,     \
                               (const int2*)nullptr, (int*)nullptr, 0, longList, small + S_UB_LONG, h->ubLong); \
        break;
    switch (G) {
        BHS_UB(1) BHS_UB(2) BHS_UB(4) BHS_UB(8) BHS_UB(16) BHS_UB(32) BHS_UB(64)
        default: return BHS_ERR_INTERNAL;
    }
#undef BHS_UB
    BHS_HIP(hipGetLastError());
    if (useLong) {
        const unsigned g1 = (unsigned)(h->numCU * 4), g2 = (unsigned)std::min<size_t>(((size_t)h->nnzA / h->ubLong + 257) / 256, 1024);
        if (cmp) {
            hipLaunchKernelGGL(k_upper_bound_long<true>, dim3(g1), dim3(256), 0, h->stream, (const int2*)longList,
                               (const int*)(small + S_UB_LONG), h->dAp, h->dAj, h->dBp, (const int2*)h->cLen.p,
                               (long long*)h->longPart.p);
            hipLaunchKernelGGL(k_upper_bound_long_finish<true>, dim3(g2), dim3(256), 0, h->stream, (const int2*)longList,
                               (const int*)(small + S_UB_LONG), h->dAp, (const long long*)h->longPart.p, (int*)h->ub.p,
                               (int*)h->Cp.p, (unsigned long long*)(small + S_TOTAL_CT), small + S_SYM_COUNT, spec,
                               (int*)h->symKey.p, keyMax);
        } else {
            hipLaunchKernelGGL(k_upper_bound_long<false>, dim3(g1), dim3(256), 0, h->stream, (const int2*)longList,
                               (const int*)(small + S_UB_LONG), h->dAp, h->dAj, h->dBp, (const int2*)nullptr,
                               (long long*)h->longPart.p);
            hipLaunchKernelGGL(k_upper_bound_long_finish<false>, dim3(g2), dim3(256), 0, h->stream, (const int2*)longList,
                               (const int*)(small + S_UB_LONG), h->dAp, (const long long*)h->longPart.p, (int*)h->ub.p,
                               (int*)h->Cp.p, (unsigned long long*)(small + S_TOTAL_CT), small + S_SYM_COUNT, spec,
                               (int*)nullptr, 0);
        }
        BHS_HIP(hipGetLastError());
    }
    return BHS_SUCCESS;
}

int pow2_at_least(double x, int lo, int hi)
{
    int v = lo;
    while (v < hi && (double)v < x) v <<= 1;
    return v;
}

// Concurrent bins: fork the side streams from `stream`, give every bin its own stream (round robin) and ticket
// word, join them back.  Otherwise everything stays on `stream`, one kernel after another.
int fork_bins(bhs_handle* h, const int* count, int nbins)
{
    h->ls = h->stream;
    int used = 0;
    for (int b = 1; b < nbins; ++b) used += count[b] > 0;
    // forking and joining four streams costs ~70 us of event traffic: it pays for power-law matrices whose rows
    // spread over many small bins, not for a stencil with one dominant bin
    h->binsForked = h->concurrentBins == 1 || (h->concurrentBins == 2 && used >= 8);
    if (!h->binsForked) return BHS_SUCCESS;
    BHS_HIP(hipEventRecord(h->evFork, h->stream));
    for (int i = 0; i < bhs_handle::kBinStreams; ++i) BHS_HIP(hipStreamWaitEvent(h->binStream[i], h->evFork, 0));
    return BHS_SUCCESS;
}

void bin_stream(bhs_handle* h, int bin)
{
    h->ticketSlot = S_TICKETS + bin;
    h->ls = h->binsForked ? h->binStream[bin % bhs_handle::kBinStreams] : h->stream;
}

int join_bins(bhs_handle* h)
{
    h->ls = h->stream;
    h->ticketSlot = S_TICKET;
    if (!h->binsForked) return BHS_SUCCESS;
    h->binsForked = false;
    for (int i = 0; i < bhs_handle::kBinStreams; ++i) {
        BHS_HIP(hipEventRecord(h->evJoin[i], h->binStream[i]));
        BHS_HIP(hipStreamWaitEvent(h->stream, h->evJoin[i], 0));
    }
    return BHS_SUCCESS;
}

// Stages 1 and 2 of the general pipeline: upper bound, symbolic bins and queues, the symbolic kernels.  Leaves the
// per-row counts in Cp and tells stage 3 which choices it made.
struct SymChoices {
    bool noUpperBound = false, symDirect = false;
    int laneK = 0, hubRows = 0;
    BinSpec numSpec;
};

int symbolic_general(bhs_handle* h, SymChoices& out)
{
    const int m = h->m;
    int* small = (int*)h->small.p;
    int* hs = h->hostSmall;
    // ------------------------------------------------------------ stage 1
    // lane bin (k_row_lane): matrices whose A rows are all tiny, B rows strictly ascending
    int laneK = 0;
    if (h->bSorted && h->forcePath == 0 && h->laneRows && (h->laneRows == 2 || (h->maxRowA <= kLaneMaxK && h->localA)))
        laneK = h->laneRows == 2 ? kLaneMaxK : std::max(4, (h->maxRowA + 1) & ~1);
    // hub bin: rows with hubMin products or more are split across workgroups (bhs_hub.hip.h) in both stages
    const int hubMin = (h->hubMin > 0 && h->useSpa && h->forcePath == 0 && h->maxTableLog2 >= 15 && h->n <= (1 << 25))
                           ? h->hubMin : 0;
    const BinSpec symSpec = make_spec(kSymCfg, kNumSymBins, h->maxTableLog2, h->symLoadPct, h->forcePath == 0, laneK, hubMin);
    BinSpec numSpec = make_spec(kNumCfg, kNumNumBins, std::min(h->maxTableLog2, 13), h->numLoadPct, h->forcePath == 0,
                                (h->laneNumeric == 1 || (h->laneNumeric == 2 && laneK <= 8)) ? laneK : 0, hubMin);
    BHS_HIP(hipMemsetAsync(small, 0, sizeof(int) * S_ZERO_END, h->stream));
    EventPair* ep;
    h->cmpActive = false;
    // (the undecided first multiply on a data set only measures the ratio: bins and symbolic pass stay plain)
    const bool cmpRun = h->compressB && h->bSorted && h->forcePath == 0 && h->maxTableLog2 >= 15 &&
                        (h->compressB == 2 || h->cmpState >= 0);
    const bool cmpBins = cmpRun && (h->compressB == 2 || h->cmpState > 0);
    if (cmpRun) {
        BHS_TRY(ensure(h, h->cExt, sizeof(int2) * (size_t)std::max(h->k, 1)));
        BHS_TRY(ensure(h, h->cPair, sizeof(int2) * (size_t)std::max(h->nnzB, 1)));
        BHS_TRY(ensure(h, h->cLen, sizeof(int2) * (size_t)std::max(h->k, 1)));
        BHS_TRY(ensure(h, h->symKey, sizeof(int) * (size_t)m));
        BHS_TRY(timed_begin(h, "compress_b", &ep));
        BHS_TRY(launch_compress_b(h));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += h->k;
    }
    const int* symKeys = cmpBins ? (const int*)h->symKey.p : (const int*)h->ub.p;
    // "Lane-first": every row of A has <= laneK entries and every row of B is short, so every row can go through
    // the lane-per-row symbolic kernel whatever its product count (the kernel has no table to overflow).  The
    // upper-bound pass, its host round trip and the symbolic queue all disappear; the lane kernel writes ub[] and
    // the product total on the side.
    const bool laneFirst = laneK > 0 && h->maxRowA <= laneK && h->laneFirst && h->directBins && !cmpRun && h->maxRowB <= 64 &&
                           !h->specFailed;
    // "Wave-first": maxRow(A) x maxRow(B) bounds every row's product count; when that bound fits a wave-per-row
    // table and is not far above the average row (stencils, FEM meshes: poisson27pt 27 x 27 = 729 for every interior
    // row), every row can run the symbolic wave kernel of that one table size -- again without upper-bound pass,
    // host round trip or queue; the kernel delivers ub[] and the product total.
    int wfBin = 0;
    if (!laneFirst && h->waveFirst && h->directBins && !cmpRun && h->forcePath == 0 && h->maxTableLog2 >= 15 && !h->specFailed) {
        const long long bound = (long long)h->maxRowA * h->maxRowB;
        if (bound > 0 && bound <= symSpec.upper[8] && (double)bound <= 4.0 * h->avgRowA * h->avgRowB)
            for (int b = 2; b <= 8 && !wfBin; ++b) if (bound <= symSpec.upper[b]) wfBin = b;
    }
    const bool noUpperBound = laneFirst || wfBin > 0;
    if (noUpperBound) numSpec.hubMin = 0;     // (every row is bounded by maxRow(A) x maxRow(B), far below the hub bin)
    int symCount[kMaxBins], symStart[kMaxBins + 1];
    if (noUpperBound) {
        BHS_HIP(hipMemsetAsync(small + S_CT_SLOTS, 0, sizeof(int) * 128, h->stream));
        for (int b = 0; b < kMaxBins; ++b) { symCount[b] = 0; symStart[b] = 0; }
        symStart[kMaxBins] = 0;
        symCount[laneFirst ? kLaneBin : wfBin] = m;
    }
    bool symDirect = noUpperBound;
    if (!noUpperBound) {
    BHS_TRY(timed_begin(h, "upper_bound", &ep));
    BHS_TRY(launch_upper_bound(h, symSpec, cmpBins, symSpec.upper[8]));
    BHS_TRY(timed_end(h, ep));
    h->stats[ep->stat].launches++;
    h->stats[ep->stat].rows += m;
    BHS_HIP(hipMemcpyAsync(hs, small, sizeof(int) * S_SMALL_INTS, hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    symStart[0] = 0;
    for (int b = 0; b < kMaxBins; ++b) {
        symCount[b] = hs[S_SYM_COUNT + b];
        symStart[b + 1] = symStart[b] + (b == 0 ? 0 : symCount[b]);
    }
    unsigned long long tot;
    memcpy(&tot, hs + S_TOTAL_CT, 8);
    h->nnzCt = (long long)tot;
    if (cmpRun) {
        unsigned long long pairs;
        memcpy(&pairs, hs + S_PAIRS, 8);
        if (h->cmpState == 0) {
            const double avgP = h->avgRowA * h->avgRowB;          // (the rule of bhs_set_data's count)
            h->cmpState = ((avgP > 1536.0 && (double)pairs <= 0.6 * (double)h->nnzB) || (double)pairs <= 0.25 * (double)h->nnzB) ? 1 : -1;
        }
        h->cmpActive = cmpBins;
        if (h->verbose > 1) printf("  [compress_b] %llu pairs for %d entries: %s\n", pairs, h->nnzB, h->cmpActive ? "used" : "not used");
    }
    // "Direct" stages: when EVERY row of the matrix sits in the lane bin or the quad bin (stencils: poisson5pt,
    // 7pt, 9pt), that bin's queue would list the rows 0..m-1 in order -- the fill pass is skipped and the kernel
    // derives its descriptors from rowPtrA (and rowPtrC) itself.
    symDirect = h->directBins && (symCount[kLaneBin] == m || symCount[1] == m);
    if (!symDirect) {
    memcpy(hs + S_SMALL_INTS, symStart, sizeof(int) * kMaxBins);        // pinned staging: a truly asynchronous H2D
    BHS_HIP(hipMemcpyAsync(small + S_SYM_START, hs + S_SMALL_INTS, sizeof(int) * kMaxBins, hipMemcpyHostToDevice, h->stream));
    {
        long long grid = std::min<long long>(((long long)m + kFillTile - 1) / kFillTile, (long long)h->numCU * 8);
        BHS_TRY(timed_begin(h, "fill_queues", &ep));
        hipLaunchKernelGGL(k_fill_queues<false>, dim3((unsigned)grid), dim3(256), 0, h->stream, m,
                           symKeys, h->dAp, (const int*)h->ub.p, (const int*)(small + S_SYM_START),
                           small + S_SYM_CURSOR, (int4*)h->queue.p, symSpec,
                           (unsigned long long*)(small + S_SYM_SUMS));
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
    }
    }
    }   // !noUpperBound
    const int4* symQueue = symDirect ? nullptr : (const int4*)h->queue.p;
    BHS_HIP(hipEventRecord(h->ev[1], h->stream));

    // ------------------------------------------------------------ stage 2: symbolic
    int (&symStat)[kMaxBins] = h->ps.symStat;
    for (int b = 0; b < kMaxBins; ++b) h->ps.symStat[b] = h->ps.numStat[b] = -1;
    BHS_TRY(fork_bins(h, symCount, kNumSymBins));
    if (symCount[kLaneBin]) {
        bin_stream(h, kLaneBin);
        BHS_TRY(timed_begin(h, "symbolic_lane", &ep));
        BHS_TRY(launch_row_lane<false>(h, laneK, symQueue ? symQueue + symStart[kLaneBin] : nullptr, symCount[kLaneBin], (int*)h->Cp.p,
                                       laneFirst ? (int*)h->ub.p : nullptr,
                                       laneFirst ? (unsigned long long*)(small + S_CT_SLOTS) : nullptr));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += symCount[kLaneBin];
        symStat[kLaneBin] = ep->stat;
    }
    if (symCount[kHubBin]) {
        bin_stream(h, kHubBin);
        BHS_TRY(timed_begin(h, "symbolic_hub_rows", &ep));
        int rc = launch_hub<false>(h, symQueue + symStart[kHubBin], symCount[kHubBin], (int*)h->Cp.p);
        if (rc) { h->ls = h->stream; return rc; }
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += symCount[kHubBin];
        symStat[kHubBin] = ep->stat;
    }
    for (int i = 1; i < kNumSymBins; ++i) {
        const int b = kNumSymBins - i;                              // longest rows first: they have the longest tails
        if (!symCount[b]) continue;
        bin_stream(h, b);
        BHS_TRY(timed_begin(h, kSymNames[b], &ep));
        int rc = dispatch_bin<false>(h, kSymCfg[b], symQueue ? symQueue + symStart[b] : nullptr, symCount[b], (int*)h->Cp.p);
        if (rc) { h->ls = h->stream; return rc; }
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += symCount[b];
        symStat[b] = ep->stat;
    }
    BHS_TRY(join_bins(h));
    BHS_HIP(hipEventRecord(h->ev[2], h->stream));

    out.noUpperBound = noUpperBound;
    out.symDirect = symDirect;
    out.laneK = laneK;
    out.hubRows = noUpperBound ? 0 : symCount[kHubBin];
    out.numSpec = numSpec;
    return BHS_SUCCESS;
}

// The classifier kernels are instantiated for G lanes per row and E entries per lane: E follows the longest row (a
// hint from bhs_set_data; a longer row finds no class and sends the multiply to the general pipeline) over the G lanes,
// as a quarter, a half or all of what the block's class cache holds; rows beyond kClassMaxRow: 64 lanes, 1 / 2 / 4 entries.
int class_entries_per_lane(int G, int maxRow)
{
    if (G >= 64) return maxRow <= kClassMaxRow ? 1 : (maxRow <= 2 * kClassMaxRow ? 2 : 4);
    const int full = kClassMaxRow / G;
    if (full >= 4 && maxRow <= kClassMaxRow / 4) return full / 4;
    if (full >= 2 && maxRow <= kClassMaxRow / 2) return full / 2;
    return full;
}
template <typename F>
int class_dispatch(int G, int E, F&& f)
{
    switch (G * 100 + E) {
        case 404: return f(template_int<4>{}, template_int<4>{});
        case 408: return f(template_int<4>{}, template_int<8>{});
        case 416: return f(template_int<4>{}, template_int<16>{});
        case 802: return f(template_int<8>{}, template_int<2>{});
        case 804: return f(template_int<8>{}, template_int<4>{});
        case 808: return f(template_int<8>{}, template_int<8>{});
        case 1601: return f(template_int<16>{}, template_int<1>{});
        case 1602: return f(template_int<16>{}, template_int<2>{});
        case 1604: return f(template_int<16>{}, template_int<4>{});
        case 3201: return f(template_int<32>{}, template_int<1>{});
        case 3202: return f(template_int<32>{}, template_int<2>{});
        case 6401: return f(template_int<64>{}, template_int<1>{});
        case 6402: return f(template_int<64>{}, template_int<2>{});
        case 6404: return f(template_int<64>{}, template_int<4>{});
    }
    return BHS_ERR_INTERNAL;
}

// Stages 1 and 2 by row classes (bhs_class.hip.h): classify the rows of B and A, work out every class's pattern,
// write the per-row counts.  Everything is launched without a host round trip; stage 3's read-back tells whether every
// row found a class (otherwise the multiply starts over on the general pipeline).
int symbolic_class(bhs_handle* h)
{
    const int m = h->m, k = h->k;
    int* small = (int*)h->small.p;
    EventPair* ep;
    BHS_TRY(ensure(h, h->classB, sizeof(int) * (size_t)std::max(k, 1)));
    BHS_TRY(ensure(h, h->classC, sizeof(int) * (size_t)std::max(m, 1)));
    BHS_TRY(ensure(h, h->classTab, sizeof(unsigned long long) * 2 * kClassSlots));
    BHS_TRY(ensure(h, h->classInfo, sizeof(int4) * kClassSlots));
    BHS_TRY(ensure(h, h->classMap, sizeof(unsigned) * (size_t)kClassSlots * kClassMaxP));
    BHS_TRY(ensure(h, h->classMapA, sizeof(unsigned) * (size_t)kClassSlots * kClassMaxP));
    BHS_TRY(ensure(h, h->classRing, sizeof(unsigned) * (size_t)kClassSlots * kClassRingStride));
    BHS_TRY(ensure(h, h->classRel, sizeof(int) * (size_t)kClassSlots * kClassMaxNnz));
    BHS_TRY(ensure(h, h->classLane, sizeof(int) * (size_t)kClassSlots * kClassLaneInts));
    BHS_TRY(ensure(h, h->classHeads, sizeof(int) * ((size_t)std::max(std::max(m, k), 1) + (size_t)kClassHeadSegs * (kClassHeadsBlock / 64) * kClassHeadPiece)));
    // classes beyond the register kernels' tables are possible: their lists and the big numeric kernel (bhs_class_big.hip.h)
    const bool bigPossible = h->maxRowA > kClassMaxRow || h->maxRowB > kClassMaxRow || (long long)h->maxRowA * h->maxRowB > kClassMaxP;
    BHS_TRY(ensure(h, h->classBigIdx, sizeof(int) * kClassSlots));
    if (bigPossible) BHS_TRY(ensure(h, h->classBigMap, sizeof(unsigned) * (size_t)kClassBigCap * kClassBigMaxP));
    BHS_TRY(ensure(h, h->classHeadCnt, sizeof(int) * 2 * 16 * kClassHeadSegs));
    const int nScanTiles = (m + kClassScanTile - 1) / kClassScanTile;           // (k_class_scan's tile words live in blockSum)
    BHS_TRY(ensure(h, h->blockSum, sizeof(unsigned long long) * (size_t)std::max(nScanTiles, (int)(((long long)m + 1 + kScanTile - 1) / kScanTile)), true));
    hipLaunchKernelGGL(k_class_reset, dim3(32), dim3(256), 0, h->stream, small, (int)S_ZERO_END, small + S_CT_SLOTS, (int)CS_INTS,
                       (int*)h->classHeadCnt.p, 2 * 16 * kClassHeadSegs, (unsigned long long*)h->classTab.p, 2 * kClassSlots,
                       (int*)h->classBigIdx.p, bigPossible ? kClassSlots : 0, (unsigned long long*)h->blockSum.p, nScanTiles);
    BHS_HIP(hipGetLastError());
    for (int b = 0; b < kMaxBins; ++b) h->ps.symStat[b] = h->ps.numStat[b] = -1;
    unsigned long long* tabB = (unsigned long long*)h->classTab.p;
    unsigned long long* tabA = tabB + kClassSlots;
    int* cstats = small + S_CT_SLOTS;
    BHS_TRY(timed_begin(h, "classify_rows", &ep));
    // lanes per row: the average row, rounded up to a power of two
    // Three launches per matrix: k_class_heads lists the rows that differ from the row before them (and notes for
    // every other row which head it follows), k_class_rows classifies the listed rows, k_class_propagate hands the
    // classes on.  (class_heads = 0: k_class_rows over all rows, round 2's form.)
    auto rows_grid = [&](int n, int G) {
        return (unsigned)std::max<long long>(1, std::min<long long>(((long long)n + kClassRowsBlock / G - 1) / (kClassRowsBlock / G), (long long)h->numCU * h->classGridMul));
    };
    auto heads_grid = [&](int n, int G) { const long long perBlock = (long long)(kClassHeadsBlock / 64) * class_head_piece(G); return (unsigned)std::max<long long>(1, ((long long)n + perBlock - 1) / perBlock); };
    auto heads_cap = [&](int n, int G) { return (int)(((long long)heads_grid(n, G) + kClassHeadSegs - 1) / kClassHeadSegs) * (kClassHeadsBlock / 64) * class_head_piece(G); };    // slots per list
    // A as a row block of a larger product (multi-GPU): only the rows of B that A points at need a class
    const int* bRange = nullptr;
    if ((long long)m * 2 <= (long long)k) {
        int* rg = cstats + CS_RANGE;
        BHS_HIP(hipMemsetD32Async((hipDeviceptr_t)rg, 0x7fffffff, 1, h->stream));
        BHS_HIP(hipMemsetD32Async((hipDeviceptr_t)(rg + 1), -1, 1, h->stream));
        const unsigned gr = (unsigned)std::max<long long>(1, std::min<long long>(((long long)h->nnzA + 255) / 256, (long long)h->numCU * 8));
        hipLaunchKernelGGL(k_class_col_range, dim3(gr), dim3(256), 0, h->stream, (long long)h->nnzA, h->dAj, rg);
        bRange = rg;
    }
    int* headsL = (int*)h->classHeads.p;                            // (one list area: B's is used up before A's is written)
    int* nHeadsB = (int*)h->classHeadCnt.p;
    int* nHeadsA = nHeadsB + 16 * kClassHeadSegs;
    // (~2 entries per lane in flight; a data set with rows of more than kClassMaxRow entries: 64 lanes, 2 or 4 entries each)
    const int GB = h->maxRowB > kClassMaxRow ? 64 : pow2_at_least(h->avgRowB / h->classPerLane, 4, 64);
    const int GA = h->maxRowA > kClassMaxRow ? 64 : pow2_at_least(h->avgRowA / h->classPerLane, 4, 64);
    const int periodA = std::max(1, std::min(8, h->periodA)), periodB = std::max(1, std::min(8, h->periodB));
    const unsigned rowsGridList = (unsigned)std::max(1, h->numCU / (2 * kClassHeadSegs));
    const unsigned propGrid = (unsigned)std::max<long long>(1, std::min<long long>(((long long)std::max(m, k) + 255) / 256, (long long)h->numCU * 8));
    // one matrix: its heads, their classes, the classes handed on -- or, without heads, every row through the table
    auto classify = [&](auto isA, int n, const int* Rp, const int* Rj, const int* cb, unsigned long long* tab, int* out,
                        const int* rng, int* nHeads, int G, int maxRow, int period) {
        constexpr bool IS_A = decltype(isA)::value != 0;
        return class_dispatch(G, class_entries_per_lane(G, maxRow), [&](auto gc, auto ec) {
            constexpr int GG = decltype(gc)::value, E = decltype(ec)::value;
            if (h->classHeadsOn >= 2) {
                hipLaunchKernelGGL((k_class_fused<IS_A, GG, E>), dim3(heads_grid(n, GG)), dim3(kClassHeadsBlock), 0, h->stream, n, Rp, Rj, cb, out,
                                   tab, cstats, (long long)(IS_A ? h->nnzA : h->nnzB), rng, period);
            } else if (h->classHeadsOn) {
                hipLaunchKernelGGL((k_class_heads<IS_A, GG, E>), dim3(heads_grid(n, GG)), dim3(kClassHeadsBlock), 0, h->stream, n, Rp, Rj, cb, out,
                                   headsL, nHeads, heads_cap(n, GG), rng, period);
                hipLaunchKernelGGL((k_class_rows<IS_A, GG, E>), dim3(rowsGridList, kClassHeadSegs), dim3(kClassRowsBlock), 0, h->stream, n, Rp, Rj, cb,
                                   tab, out, cstats, (const int*)nullptr, (const int*)headsL, (const int*)nHeads, heads_cap(n, GG));
                hipLaunchKernelGGL(k_class_propagate, dim3(propGrid), dim3(256), 0, h->stream, n, out, rng);
            } else
                hipLaunchKernelGGL((k_class_rows<IS_A, GG, E>), dim3(rows_grid(n, GG)), dim3(kClassRowsBlock), 0, h->stream, n, Rp, Rj, cb, tab, out,
                                   cstats, rng, (const int*)nullptr, (const int*)nullptr, 0);
            return (int)BHS_SUCCESS;
        });
    };
    int rc = classify(template_int<0>{}, k, h->dBp, h->dBj, (const int*)nullptr, tabB, (int*)h->classB.p, bRange, nHeadsB, GB, h->maxRowB, periodB);
    if (rc == BHS_SUCCESS)
        rc = classify(template_int<1>{}, m, h->dAp, h->dAj, (const int*)h->classB.p, tabA, (int*)h->classC.p, (const int*)nullptr, nHeadsA, GA, h->maxRowA, periodA);
    if (rc != BHS_SUCCESS) return rc;
    BHS_HIP(hipGetLastError());
    BHS_TRY(timed_end(h, ep));
    h->stats[ep->stat].launches += h->classHeadsOn == 1 ? 6 : 2;
    h->stats[ep->stat].rows += (int64_t)m + k;
    BHS_HIP(hipEventRecord(h->ev[1], h->stream));
    BHS_TRY(timed_begin(h, "class_patterns", &ep));
    hipLaunchKernelGGL(k_class_patterns, dim3(kClassSlots), dim3(256), 0, h->stream, (const unsigned long long*)tabA,
                       h->dAp, h->dAj, h->dBp, h->dBj, (int4*)h->classInfo.p,
                       (unsigned*)h->classMap.p, (unsigned*)h->classMapA.p, (int*)h->classRel.p, (int*)h->classLane.p,
                       (unsigned*)h->classRing.p, cstats);
    if (bigPossible) {
        const size_t smemBig = sizeof(int) * 2 * kClassBigMaxP;
        int unused = 0;
        BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(k_class_patterns_big), kClassBigPatThreads, smemBig, &unused));   // (raises its LDS limit)
        hipLaunchKernelGGL(k_class_patterns_big, dim3(kClassSlots), dim3(kClassBigPatThreads), smemBig, h->stream, (const unsigned long long*)tabA,
                           h->dAp, h->dAj, h->dBp, h->dBj, (int4*)h->classInfo.p, (int*)h->classBigIdx.p,
                           (unsigned*)h->classBigMap.p, (int*)h->classRel.p, cstats);
    }
    BHS_HIP(hipGetLastError());
    BHS_TRY(timed_end(h, ep));
    h->stats[ep->stat].launches += bigPossible ? 2 : 1;
    BHS_HIP(hipEventRecord(h->ev[2], h->stream));
    return BHS_SUCCESS;
}

// restart: the same multiply starting over on another path (a refuted speculation, rows without a class): the
// timers and kernel statistics of the abandoned attempt stay in -- it ran inside this multiply.
int pipeline_symbolic(bhs_handle* h, bool restart = false)
{
    h->ls = h->stream;
    const int m = h->m;
    if (!restart) {
        h->evUsed = 0;
        for (auto& s : h->stats) { s.launches = 0; s.ms = 0; s.rows = s.products = s.nnz_out = s.nnzA_rows = 0; }
        BHS_HIP(hipEventRecord(h->ev[0], h->stream));
    }
    int* small = (int*)h->small.p;
    int* hs = h->hostSmall;
    h->nnzC = 0;
    h->nnzCt = 0;
    h->hasC = false;
    h->rowPtrStaged = false;          // (an empty product returns early: the previous multiply's staging must not be read)
    h->ps = bhs_handle::PipeState();

    BHS_TRY(ensure(h, h->Cp, sizeof(int) * ((size_t)m + 1)));
    if (m == 0 || h->nnzA == 0 || h->nnzB == 0) {
        BHS_HIP(hipMemsetAsync(h->Cp.p, 0, sizeof(int) * ((size_t)m + 1), h->stream));
        for (int i = 1; i < 5; ++i) BHS_HIP(hipEventRecord(h->ev[i], h->stream));
        BHS_HIP(hipStreamSynchronize(h->stream));
        h->hasC = true;
        h->ps.open = true;
        h->ps.empty = true;
        return BHS_SUCCESS;
    }
    BHS_TRY(ensure(h, h->ub, sizeof(int) * (size_t)m));
    BHS_TRY(ensure(h, h->queue, sizeof(int4) * (size_t)m));
    const int nScanBlocks = (int)(((long long)m + 1 + kScanTile - 1) / kScanTile);
    BHS_TRY(ensure(h, h->blockSum, sizeof(long long) * (size_t)nScanBlocks, true));

    EventPair* ep;
    SymChoices sc;
    // Row classes first, for data sets whose rows are short on both sides (the hint from bhs_set_data time is
    // verified on the device row by row)
    // ... and long enough for the classification passes to pay (round 3's kernels, same box): poisson27pt (729 products
    // per row) 4.85 -> 2.1 ms on the class kernels, poisson9pt 1024^2 (81) 0.57 -> 0.47 ms, poisson7pt 128^3 (49) 0.73 ->
    // 0.78 ms, poisson5pt 1024^2 (25) 0.23 -> 0.42 ms: from class_min_products = 64 products per row on
    const bool useClass = h->classPath && h->classState >= 0 && h->forcePath == 0 && h->maxTableLog2 >= 15 &&
                          h->maxRowA <= kClassMaxRowBig && h->maxRowB <= kClassMaxRowBig &&
                          (h->classPath == 2 || (h->avgRowA * h->avgRowB >= (double)h->classMinProducts &&
                                                 // ... and enough of them: every block of the classifier meets every class once
                                                 // (poisson27pt 51^3, 90 M products: 0.44 ms general, 0.41 ms by classes;
                                                 // poisson9pt 512^2, 21 M: 0.20 against 0.25)
                                                 (double)h->m * h->avgRowA * h->avgRowB >= 6e7));
    if (useClass) {
        BHS_TRY(symbolic_class(h));
        sc.noUpperBound = true;                 // (no ub[] either: the numeric bins are never built)
        sc.numSpec = make_spec(kNumCfg, kNumNumBins, std::min(h->maxTableLog2, 13), h->numLoadPct, true, 0, 0);
    } else {
        BHS_TRY(symbolic_general(h, sc));
    }
    const bool noUpperBound = sc.noUpperBound, symDirect = sc.symDirect;
    const int laneK = sc.laneK;
    const BinSpec& numSpec = sc.numSpec;

    // ------------------------------------------------------------ stage 3: scan, allocate C, numeric queues
    BHS_TRY(timed_begin(h, "scan_rowptr", &ep));
    if (useClass) {
        // one pass: every row's count from its class, scanned with look-back over the tiles before (k_class_scan)
        const int nTiles = (m + kClassScanTile - 1) / kClassScanTile;
        // (blockSum holds the tile words, cleared by k_class_reset)
        hipLaunchKernelGGL(k_class_scan, dim3((unsigned)nTiles), dim3(kClassScanBlock), 0, h->stream, m, (const int*)h->classC.p,
                           (const int4*)h->classInfo.p, (int*)h->Cp.p, (unsigned long long*)h->blockSum.p,
                           (long long*)(small + S_TOTAL_C), small + S_CT_SLOTS);
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches += 1;
    } else {
    if (h->scanOnePass) {
        // one pass with look-back over the tiles before (k_scan_onepass); the tile words carry this multiply's epoch
        const int nTiles = (m + kScan1Tile - 1) / kScan1Tile;
        h->scanEpoch = (h->scanEpoch + 1) & 0x3FFFFu;
        if (h->scanEpoch == 0) {                                    // (every 2^18 multiplies the words of 2^18 multiplies ago could match)
            BHS_HIP(hipMemsetAsync(h->blockSum.p, 0, sizeof(unsigned long long) * (size_t)std::max(nTiles, 1), h->stream));
            h->scanEpoch = 1;
        }
        hipLaunchKernelGGL(k_scan_onepass, dim3((unsigned)nTiles), dim3(kScan1Block), 0, h->stream, m, (int*)h->Cp.p, h->dAp,
                           (unsigned long long*)h->blockSum.p, h->scanEpoch, small + S_SCAN_TICKET, (long long*)(small + S_TOTAL_C),
                           small + S_NUM_COUNT, numSpec, small + S_MAXCNT, (const int*)h->ub.p);
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches += 1;
    } else {
    hipLaunchKernelGGL(k_scan_reduce, dim3(nScanBlocks), dim3(256), 0, h->stream, m, (const int*)h->Cp.p, h->dAp,
                       (long long*)h->blockSum.p, small + S_NUM_COUNT, numSpec, small + S_MAXCNT, (const int*)h->ub.p);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, h->stream, nScanBlocks,
                       (long long*)h->blockSum.p, (long long*)(small + S_TOTAL_C));
    hipLaunchKernelGGL(k_scan_apply, dim3(nScanBlocks), dim3(256), 0, h->stream, m, (int*)h->Cp.p,
                       (const long long*)h->blockSum.p);
    BHS_HIP(hipGetLastError());
    BHS_TRY(timed_end(h, ep));
    h->stats[ep->stat].launches += 3;
    }
    }
    BHS_HIP(hipMemcpyAsync(hs, small, sizeof(int) * S_SMALL_INTS, hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    if (useClass) {
        const int* cs = hs + S_CT_SLOTS;
        if (cs[CS_FLAGS] || cs[CS_CLASSES] == 0) {
            // a row without a class, or a class beyond the tables: this data set is for the general pipeline
            h->classState = -1;
            if (h->verbose > 1) printf("  [row classes: flags %d, %d classes: general pipeline]\n", cs[CS_FLAGS], cs[CS_CLASSES]);
            return pipeline_symbolic(h, true);
        }
        // The class kernels take rows in stretches -- consecutive rows of one class; a matrix whose rows classify but
        // each for itself (block-diagonal with dense blocks: every row of a block has its own relative pattern) makes
        // them change class every row: 3.4 ms against 1.9 ms on the general pipeline for 2^20 rows in blocks of 4..32.
        // More than a quarter of the rows through the table: this data set goes to the general pipeline (class_path = 2
        // insists on the classes).
        if (h->classHeadsOn && h->classPath != 2 && !cs[CS_BIGCOUNT] && (long long)cs[CS_HEADS] * 4 > (long long)m) {
            h->classState = -1;
            if (h->verbose > 1) printf("  [row classes: %d of %d rows start a stretch: general pipeline]\n", cs[CS_HEADS], m);
            return pipeline_symbolic(h, true);
        }
        unsigned long long t = 0, v;
        for (int i = 0; i < kClassSumSlots; ++i) { memcpy(&v, cs + CS_SUMS + 2 * i, 8); t += v; }
        h->nnzCt = (long long)t;
        h->ps.useClass = true;
        h->ps.classMaxP = cs[CS_MAXP];
        h->ps.classMaxNnz = cs[CS_MAXNNZ];
        h->ps.classMaxNA = cs[CS_MAXNA];
        h->ps.classMaxLB = cs[CS_MAXLB];
        h->ps.classMaxRing = cs[CS_MAXRING];
        h->ps.classMaxRing2 = std::max(cs[CS_RINGFULL], cs[CS_RINGONE]);
        h->ps.classMaxSlab = cs[CS_MAXSLAB];
        h->ps.classBig = cs[CS_BIGCOUNT];
        h->ps.classBigMaxP = cs[CS_BIGMAXP];
        if (h->verbose > 1) printf("  [row classes: %d classes, <= %d products and <= %d entries per row; slabs of <= %d values]\n", cs[CS_CLASSES], cs[CS_MAXP], cs[CS_MAXNNZ], cs[CS_MAXSLAB]);
    } else if (noUpperBound) {                           // product count: the symbolic kernel's 64 partial sums
        unsigned long long t = 0, v;
        for (int i = 0; i < 64; ++i) { memcpy(&v, hs + S_CT_SLOTS + 2 * i, 8); t += v; }
        h->nnzCt = (long long)t;
    }
    long long nnzC;
    memcpy(&nnzC, hs + S_TOTAL_C, 8);
    if (hs[S_ERR] & 2) {
        // The lane-first / wave-first launch was chosen from the row bounds seen at bhs_set_data time and the
        // kernels found a row beyond them (borrowed arrays changed since): this multiply starts over on the
        // general pipeline, which assumes nothing, and the data set stays there.
        if (!noUpperBound || h->specFailed) return BHS_ERR_INTERNAL;
        h->specFailed = true;
        if (h->verbose > 1) printf("  [speculative direct launch refuted on the device: general pipeline]\n");
        return pipeline_symbolic(h, true);
    }
    if (hs[S_ERR]) return BHS_ERR_INTERNAL;
    if (nnzC > 0x7fffffffLL) return BHS_ERR_NNZ_OVERFLOW;
    h->nnzC = nnzC;
    h->ps.noUpperBound = noUpperBound;
    h->ps.symDirect = symDirect;
    h->ps.laneK = laneK;
    h->ps.numSpec = numSpec;
    h->ps.maxCnt = hs[S_MAXCNT];
    h->ps.hubRows = sc.hubRows;
    for (int b = 0; b < kMaxBins; ++b) h->ps.fullCount[b] = hs[S_NUM_COUNT + b];
    memcpy(h->ps.symSums, hs + S_SYM_SUMS, sizeof(h->ps.symSums));
    if (h->extCj) {
        if (nnzC > h->extCap) return BHS_ERR_ALLOC;
    } else if (!h->lazyOut) {
        BHS_TRY(ensure(h, h->Cj, sizeof(int) * (size_t)std::max<long long>(nnzC, 1)));
        BHS_TRY(ensure(h, h->Cx, sizeof(value_t) * (size_t)std::max<long long>(nnzC, 1)));
    }
    BHS_HIP(hipEventRecord(h->ev[3], h->stream));
    h->rowPtrStaged = false;
    if (h->wantHostRowPtr) {
        // rowPtrC is final after the scan: ship it to pinned host memory on a second stream while the
        // numeric kernels run (the reference does this D2H inside its timed region too, bhsparse_cuda.h:2787)
        const size_t bytes = sizeof(int) * ((size_t)m + 1);
        BHS_TRY(ensure_host_rowptr(h, bytes));
        BHS_HIP(hipEventRecord(h->evScanDone, h->stream));
        BHS_HIP(hipStreamWaitEvent(h->copyStream, h->evScanDone, 0));
        BHS_HIP(hipMemcpyAsync(h->hostRowPtr, h->Cp.p, bytes, hipMemcpyDeviceToHost, h->copyStream));
        BHS_HIP(hipEventRecord(h->evCopyDone, h->copyStream));
        h->rowPtrStaged = true;
    }
    h->ps.open = true;
    return BHS_SUCCESS;
}

// Stage 4 on the rows [r0, r1) of A / C.  A row range is the same multiply seen through shifted row pointers (the
// kernels index rowPtrA / rowPtrC / ub / the pattern array by row), so the handle's views are shifted for the
// duration of the call; bins and queues are rebuilt for the range.
int numeric_stage(bhs_handle* h, int r0, int r1)
{
    if (!h->ps.open) return BHS_ERR_NOT_READY;
    if (h->ps.empty) return BHS_SUCCESS;
    if (r0 < 0 || r1 > h->m || r0 > r1) return BHS_ERR_INVALID_ARG;
    if (r0 == r1) return BHS_SUCCESS;
    const bool full = r0 == 0 && r1 == h->m;
    int* small = (int*)h->small.p;
    int* hs = h->hostSmall;
    EventPair* ep;
    const BinSpec& numSpec = h->ps.numSpec;
    const int laneK = h->ps.laneK;
    int (&numStat)[kMaxBins] = h->ps.numStat;
    h->ls = h->stream;
    if (h->ps.useClass) {
        h->ps.rangesRun++;
        BHS_TRY(timed_begin(h, "numeric_class", &ep));
        if (h->ps.classBig) BHS_TRY(launch_class_numeric_big(h, r0, r1));
        else if (h->classNumeric >= 2 && class_ring2_fits(h)) BHS_TRY(launch_class_ring(h, r0, r1));
        else BHS_TRY(h->classNumeric && class_ring_fits(h) ? launch_class_numeric(h, r0, r1) : launch_class_numeric_atomic(h, r0, r1));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += r1 - r0;
        if (full) { h->stats[ep->stat].products += h->nnzCt; h->stats[ep->stat].nnz_out += h->nnzC; h->stats[ep->stat].nnzA_rows += h->nnzA; }
        return BHS_SUCCESS;
    }
    // ---- the range as a view
    struct View {
        bhs_handle* h; int m; const int* dAp; void *cp, *ub;
        View(bhs_handle* h_, int r0_, int mR) : h(h_), m(h_->m), dAp(h_->dAp), cp(h_->Cp.p), ub(h_->ub.p)
        {
            h->m = mR;
            h->dAp = dAp + r0_;
            h->Cp.p = (int*)cp + r0_;
            h->ub.p = (int*)ub + r0_;
        }
        ~View() { h->m = m; h->dAp = dAp; h->Cp.p = cp; h->ub.p = ub; }
    } view(h, r0, r1 - r0);
    const int m = r1 - r0;
    int numCount[kMaxBins], numStart[kMaxBins + 1];
    int maxCnt = h->ps.maxCnt;
    if (full) {
        for (int b = 0; b < kMaxBins; ++b) numCount[b] = h->ps.fullCount[b];
    } else {
        // bins of the range: histogram of its rows (one small round trip per range)
        int* hr = hs + S_SMALL_INTS + 2 * kMaxBins;
        BHS_HIP(hipMemsetAsync(small + S_NUM_COUNT, 0, sizeof(int) * 3 * kMaxBins, h->stream));     // counts, starts, cursors
        BHS_HIP(hipMemsetAsync(small + S_NUM_SUMS, 0, sizeof(unsigned long long) * 3 * kMaxBins, h->stream));
        BHS_HIP(hipMemsetAsync(small + S_MAXCNT, 0, sizeof(int), h->stream));
        const long long grid = std::min<long long>(((long long)m + 255) / 256, (long long)h->numCU * 4);
        hipLaunchKernelGGL(k_bin_hist, dim3((unsigned)grid), dim3(256), 0, h->stream, m, (const int*)h->Cp.p, h->dAp,
                           numSpec, small + S_NUM_COUNT, small + S_MAXCNT, (const int*)h->ub.p);
        BHS_HIP(hipGetLastError());
        BHS_HIP(hipMemcpyAsync(hr, small + S_NUM_COUNT, sizeof(int) * kMaxBins, hipMemcpyDeviceToHost, h->stream));
        BHS_HIP(hipMemcpyAsync(hr + kMaxBins, small + S_MAXCNT, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        BHS_HIP(hipStreamSynchronize(h->stream));
        for (int b = 0; b < kMaxBins; ++b) numCount[b] = hr[b];
        maxCnt = hr[kMaxBins];
    }
    numStart[0] = 0;
    for (int b = 0; b < kMaxBins; ++b) numStart[b + 1] = numStart[b] + (b == 0 ? 0 : numCount[b]);
    bool numDirect = h->directBins && (numCount[kLaneBin] == m || numCount[1] == m);
    // "Numeric-first": the longest row of C fits a wave-per-row table that is not oversized for the average row
    // (poisson27pt: longest 125, average 121): every row runs that one kernel straight from rowPtrA / rowPtrC -- no
    // queue, and the few short boundary rows no longer pay for kernels of their own.
    if (!numDirect && h->waveFirst && h->directBins && h->forcePath == 0 && h->maxTableLog2 >= 15 &&
        h->ps.hubRows == 0) {
        int nb = 0;
        for (int b = 2; b <= 6 && !nb; ++b) if (maxCnt <= numSpec.upper[b]) nb = b;
        if (nb && maxCnt > 0 && (double)h->nnzC / std::max(view.m, 1) * 4.0 >= (double)numSpec.upper[nb]) {
            for (int b = 0; b < kMaxBins; ++b) { numCount[b] = 0; numStart[b] = 0; }
            numStart[kMaxBins] = 0;
            numCount[nb] = m;
            numDirect = true;
        }
    }
    if (!numDirect) {
        memcpy(hs + S_SMALL_INTS + kMaxBins, numStart, sizeof(int) * kMaxBins);
        BHS_HIP(hipMemcpyAsync(small + S_NUM_START, hs + S_SMALL_INTS + kMaxBins, sizeof(int) * kMaxBins, hipMemcpyHostToDevice, h->stream));
        long long grid = std::min<long long>(((long long)m + kFillTile - 1) / kFillTile, (long long)h->numCU * 8);
        BHS_TRY(timed_begin(h, "fill_queues", &ep));
        hipLaunchKernelGGL(k_fill_queues<true>, dim3((unsigned)grid), dim3(256), 0, h->stream, m,
                           (const int*)h->Cp.p, h->dAp, (const int*)h->ub.p, (const int*)(small + S_NUM_START),
                           small + S_NUM_CURSOR, (int4*)h->queue.p, numSpec,
                           (unsigned long long*)(small + S_NUM_SUMS));
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
    }
    if (full) h->ps.numDirectFull = numDirect;
    h->ps.rangesRun++;
    const int4* numQueue = numDirect ? nullptr : (const int4*)h->queue.p;
    h->ps.midRows = h->ps.longRows = 0;
    for (int b = 2; b < kNumNumBins; ++b)
        if (bin_takes_lds_bitmap<true>(h, kNumCfg[b])) (kNumCfg[b].win ? h->ps.longRows : h->ps.midRows) += numCount[b];
    for (int b = 2; b < kNumNumBins; ++b)
        if (numCount[b] && numQueue && bin_takes_wave_window<true>(h, kNumCfg[b])) BHS_TRY(ensure_b_windows(h));
    BHS_TRY(fork_bins(h, numCount, kNumNumBins));
    if (numCount[kHubBin]) {
        bin_stream(h, kHubBin);
        BHS_TRY(timed_begin(h, "numeric_hub_rows", &ep));
        BHS_TRY(launch_hub<true>(h, numQueue + numStart[kHubBin], numCount[kHubBin], (int*)h->Cp.p));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += numCount[kHubBin];
        numStat[kHubBin] = ep->stat;
    }
    if (numCount[kLaneBin]) {
        bin_stream(h, kLaneBin);
        BHS_TRY(timed_begin(h, "numeric_lane", &ep));
        BHS_TRY(launch_row_lane<true>(h, laneK, numQueue ? numQueue + numStart[kLaneBin] : nullptr, numCount[kLaneBin], (int*)h->Cp.p));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += numCount[kLaneBin];
        numStat[kLaneBin] = ep->stat;
    }
    for (int i = 1; i < kNumNumBins; ++i) {
        const int b = kNumNumBins - i;
        if (!numCount[b]) continue;
        // Neighbouring bins that all run the LDS-bitmap kernel (one workgroup per CU: two such kernels side by side
        // only take CUs from each other, and the shorter bins' launch would trail behind) go as ONE queue, taken
        // from its end so that the longest rows start first.
        int lo = b, rows = numCount[b];
        auto kernel_of = [&](int bb) { return !bin_takes_lds_bitmap<true>(h, kNumCfg[bb]) ? 0 : bin_takes_wave_window<true>(h, kNumCfg[bb]) ? (kNumCfg[bb].win ? 3 : 2) : 1; };
        if (numQueue && h->mergeBitmapBins && kernel_of(b))
            while (lo - 1 >= 2 && kernel_of(lo - 1) == kernel_of(b)) { --lo; rows += numCount[lo]; }
        bin_stream(h, b);
        BHS_TRY(timed_begin(h, kNumNames[b], &ep));
        BHS_TRY(dispatch_bin<true>(h, kNumCfg[b], numQueue ? numQueue + numStart[lo] : nullptr, rows, (int*)h->Cp.p, lo < b));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += rows;
        for (int bb = lo; bb <= b; ++bb) { if (numCount[bb]) numStat[bb] = ep->stat; numCount[bb] = 0; }
    }
    BHS_TRY(join_bins(h));
    return BHS_SUCCESS;
}

// End of a multiply: everything launched has run, errors raised on the device are collected, timers are read.
int pipeline_finish(bhs_handle* h)
{
    if (!h->ps.open) return BHS_ERR_NOT_READY;
    h->ps.open = false;
    if (h->ps.empty) { for (int i = 0; i < 4; ++i) h->stageMs[i] = 0.0; return BHS_SUCCESS; }
    int* small = (int*)h->small.p;
    int* hs = h->hostSmall;
    BHS_HIP(hipMemcpyAsync(hs, small, sizeof(int) * S_SMALL_INTS, hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipEventRecord(h->ev[4], h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    if (hs[S_ERR]) return BHS_ERR_INTERNAL;
    const bool oneRange = h->ps.rangesRun == 1;
    for (int b = 1; b < kMaxBins; ++b) {
        unsigned long long v[3];
        if (h->ps.symStat[b] >= 0) {
            memcpy(v, h->ps.symSums + 3 * b, sizeof(v));
            if (h->ps.symDirect) { v[0] = (unsigned long long)h->nnzCt; v[2] = (unsigned long long)h->nnzA; }   // no fill pass counted them
            StatRec& r = h->stats[h->ps.symStat[b]];
            r.products += (int64_t)v[0]; r.nnzA_rows += (int64_t)v[2];
        }
        if (h->ps.numStat[b] >= 0 && oneRange) {             // (per-bin sums of the last range only: reported for whole multiplies)
            memcpy(v, hs + S_NUM_SUMS + 6 * b, sizeof(v));
            if (h->ps.numDirectFull) { v[0] = (unsigned long long)h->nnzCt; v[1] = (unsigned long long)h->nnzC; v[2] = (unsigned long long)h->nnzA; }
            StatRec& r = h->stats[h->ps.numStat[b]];
            r.products += (int64_t)v[0]; r.nnz_out += (int64_t)v[1]; r.nnzA_rows += (int64_t)v[2];
        }
    }
    for (int i = 0; i < 4; ++i) {
        float ms = 0;
        BHS_HIP(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        h->stageMs[i] = ms;
    }
    for (size_t i = 0; i < h->evUsed; ++i) {
        float ms = 0;
        BHS_HIP(hipEventElapsedTime(&ms, h->evPool[i].a, h->evPool[i].b));
        h->stats[h->evPool[i].stat].ms += ms;
    }
    h->hasC = true;
    h->resCj = out_cj(h);
    return BHS_SUCCESS;
}

int run_pipeline_impl(bhs_handle* h)
{
    BHS_TRY(pipeline_symbolic(h));
    BHS_TRY(numeric_stage(h, 0, h->m));
    return pipeline_finish(h);
}

// Every exit of the pipeline leaves the handle quiescent: an error taken while the bins of a stage are forked
// onto the side streams would otherwise leave kernels queued there -- still writing Cp / Cj / the counters while
// the next bhs_spgemm starts on `stream` -- and stale launch state (ls, ticket slot) behind.
void quiesce(bhs_handle* h)
{
    for (int i = 0; i < bhs_handle::kBinStreams; ++i)
        if (h->binStream[i]) (void)hipStreamSynchronize(h->binStream[i]);
    if (h->copyStream) (void)hipStreamSynchronize(h->copyStream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipGetLastError();
    h->ls = h->stream;
    h->ticketSlot = S_TICKET;
    h->binsForked = false;
    h->rowPtrStaged = false;
}

int run_pipeline(bhs_handle* h)
{
    const int rc = run_pipeline_impl(h);
    if (rc != BHS_SUCCESS) { quiesce(h); h->ps.open = false; }
    return rc;
}

// per-row sort of a device CSR by column, in place (bhs_csr_sort_indices_device; also applied to unsorted B)
int sort_rows_device(bhs_handle* h, int n_row, const int* d_rowPtr, int* d_colInd, value_t* d_val)
{
    BHS_TRY(ensure(h, h->sortCnt, 16));
    BHS_TRY(ensure(h, h->sortList, sizeof(int) * (size_t)n_row));
    int* cnt = (int*)h->sortCnt.p;                      // [0] long rows, [1] longest row
    BHS_HIP(hipMemsetAsync(cnt, 0, 16, h->stream));
    const long long gmr = std::min<long long>(((long long)n_row + 255) / 256, (long long)h->numCU * 2);
    hipLaunchKernelGGL(k_max_row, dim3((unsigned)gmr), dim3(256), 0, h->stream, n_row, d_rowPtr, cnt + 1);
    BHS_HIP(hipGetLastError());
    int host[2] = {0, 0};
    int nnz = 0;
    BHS_HIP(hipMemcpyAsync(host, cnt, 8, hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipMemcpyAsync(&nnz, d_rowPtr + n_row, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    if (host[1] > kSortLdsMax) {                        // rows beyond the LDS buffer sort in HBM scratch
        BHS_TRY(ensure(h, h->sortK, sizeof(unsigned long long) * (size_t)std::max(nnz, 1)));
        BHS_TRY(ensure(h, h->sortV, sizeof(value_t) * (size_t)std::max(nnz, 1)));
    }
    const long long gw = std::min<long long>(((long long)n_row + 3) / 4, (long long)h->numCU * 32);
    hipLaunchKernelGGL(k_sort_rows_wave, dim3((unsigned)std::max<long long>(gw, 1)), dim3(256), 0, h->stream, n_row,
                       d_rowPtr, d_colInd, (value_t*)d_val, (int*)h->sortList.p, cnt);
    BHS_HIP(hipGetLastError());
    if (host[1] > 1024) {
        hipLaunchKernelGGL(k_sort_rows_block, dim3((unsigned)(h->numCU * 2)), dim3(256), 0, h->stream, d_rowPtr, d_colInd,
                           (value_t*)d_val, (const int*)h->sortList.p, (const int*)cnt,
                           (unsigned long long*)h->sortK.p, (value_t*)h->sortV.p);
        BHS_HIP(hipGetLastError());
    }
    BHS_HIP(hipStreamSynchronize(h->stream));
    return BHS_SUCCESS;
}

int finish_set_data(bhs_handle* h)
{
    // derived launch parameters
    const double avgA = h->m > 0 ? (double)h->nnzA / h->m : 1.0;
    const double avgB = h->k > 0 ? (double)h->nnzB / h->k : 1.0;
    BHS_TRY(ensure(h, h->small, sizeof(int) * S_SMALL_INTS));
    // the scans of the data set (longest rows, the period hint, sortedness of B's rows) are queued together and read
    // back with ONE synchronisation
    int* small0 = (int*)h->small.p;
    BHS_HIP(hipMemsetAsync(small0 + S_SCAN, 0, sizeof(int) * (S_SMALL_INTS - S_SCAN), h->stream));
    h->periodA = h->periodB = 1;
    if (h->m > 0) {
        const long long gmr = std::min<long long>(((long long)h->m + 255) / 256, (long long)h->numCU * 2);
        hipLaunchKernelGGL(k_max_row, dim3((unsigned)gmr), dim3(256), 0, h->stream, h->m, h->dAp, small0 + S_SCAN);
        hipLaunchKernelGGL(k_row_period, dim3(1), dim3(64), 0, h->stream, h->m, h->dAp, h->dAj, small0 + S_SCAN + 1, small0 + S_SCAN + 4, h->k, small0 + S_SCAN + 5);
        BHS_HIP(hipGetLastError());
    }
    if (h->k > 0) {
        const long long gmb = std::min<long long>(((long long)h->k + 255) / 256, (long long)h->numCU * 2);
        hipLaunchKernelGGL(k_max_row, dim3((unsigned)gmb), dim3(256), 0, h->stream, h->k, h->dBp, small0 + S_SCAN + 2);
        hipLaunchKernelGGL(k_row_period, dim3(1), dim3(64), 0, h->stream, h->k, h->dBp, h->dBj, small0 + S_SCAN + 3, (int*)nullptr, 0);
        BHS_HIP(hipGetLastError());
    }
    h->avgRowA = avgA;
    h->avgRowB = avgB;
    int L = pow2_at_least(avgB, 1, 64);
    int lg = 0;
    while ((1 << lg) < L) ++lg;
    h->logL = lg;
    h->bSorted = 1;
    h->cmpState = 0;
    h->specFailed = false;
    h->classState = 0;
    const bool checkB = h->nnzB > 1 && h->k > 0;
    // rows of B beyond kSortedLongB entries are listed and checked by k_check_sorted_long, 16 workgroups per row
    int2* longB = nullptr;
    const int logG = std::min(h->logL, 6);                      // lanes per row of B: its average length
    const long long sortGrid = std::max<long long>(1, std::min<long long>(((long long)h->k + (256 >> logG) - 1) / (256 >> logG), (long long)h->numCU * 16));
    auto check_sorted = [&]() -> int {
        BHS_HIP(hipMemsetAsync(small0 + S_SORTED, 0, sizeof(int), h->stream));
        BHS_HIP(hipMemsetAsync(small0 + S_LONG_B, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_check_sorted, dim3((unsigned)sortGrid), dim3(256), 0, h->stream, h->k, logG, h->dBp, h->dBj,
                           small0 + S_SORTED, longB, small0 + S_LONG_B);
        hipLaunchKernelGGL(k_check_sorted_long, dim3((unsigned)(h->numCU * 4)), dim3(256), 0, h->stream,
                           (const int2*)longB, (const int*)(small0 + S_LONG_B), h->dBp, h->dBj, small0 + S_SORTED);
        BHS_HIP(hipGetLastError());
        return BHS_SUCCESS;
    };
    if (checkB) {
        BHS_TRY(ensure(h, h->longList, ((size_t)h->nnzB / 2048 + 2) * sizeof(int2)));
        longB = (int2*)h->longList.p;
        BHS_TRY(check_sorted());
    }
    int* hscan = (int*)h->hostSmall;                                // (pinned)
    BHS_HIP(hipMemcpyAsync(hscan, small0 + S_SCAN, sizeof(int) * 6, hipMemcpyDeviceToHost, h->stream));
    if (checkB) BHS_HIP(hipMemcpyAsync(hscan + 6, small0 + S_SORTED, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    const int maxRowA = hscan[0];
    h->maxRowA = maxRowA;
    h->maxRowB = hscan[2];
    if (h->m > 0) h->periodA = hscan[1];
    h->localA = h->m > 0 ? hscan[4] : 1;
    // a wave of the ring kernel takes whole grid lines when A has them (rows whose lengths repeat with that period,
    // the matrix a whole number of them) -- a stretch of rows ends where a line ends anyway
    h->lineA = 0;
    if (h->m > 0 && hscan[5] >= 16 && h->m % hscan[5] == 0) {
        // (measured on poisson27pt n^3 against 64 rows: n = 96 -8 %, 110 -6 %, 160 -4 %, 200 -4 %, 128 -1 %; n = 100, whose
        // line ends in half a run, +1 %; n = 72, 1.7 lines per wave, +3 %)
        int line = hscan[5];
        while (line < 48 && h->m % (2 * line) == 0) line *= 2;      // (short lines: two, four at a time -- still a whole number of them)
        const int whole = (line + kClassRun - 1) / kClassRun * kClassRun;
        if (line <= 256 && (whole - line) * 50 <= line && h->m / line >= 32LL * h->numCU) h->lineA = line;
    }
    if (h->k > 0) h->periodB = hscan[3];
    // lanes per row of A in k_upper_bound: the average row for regular inputs, widened for skewed ones so
    // that the longest row is walked in <= 32 passes
    // (round 4: the lanes follow the AVERAGE row and rows of more than 32 passes go to k_upper_bound_long -- a web graph's
    // rows of 3 entries were walked by 16 lanes each because a few rows have hundreds: 0.12 ms for 3 M entries)
    // Measured on the two web-graph stand-ins (weblike / power-law, avg 3 entries, longest row 4.7 k): 16 lanes 0.138 / 0.187 ms,
    // 8 lanes and rows beyond 128 entries listed 0.078 / 0.179, 4 lanes 0.066 / 0.223.
    h->ubG = pow2_at_least(avgA, 1, 64);
    h->ubLong = kUbLongA;
    if (maxRowA > 32 * h->ubG) {                                // skewed: twice the lanes, rows beyond 16 passes listed
        h->ubG = std::min(64, 2 * h->ubG);
        h->ubLong = std::max(64, std::min(kUbLongA, 16 * h->ubG));
    }
    if (checkB) {
        int* small = small0;
        int flag = hscan[6];
        h->bSorted = flag ? 0 : 1;
        if (!h->bSorted && h->sortB) {
            // Unsorted rows of B: sort them once here (the reference's driver does this on the host before
            // initData, main.cu:62-64) so that the multiply can take the kernels that want ascending rows.
            // Borrowed device arrays are never written: the sort runs on a private copy.
            if (!h->ownAB) {
                BHS_TRY(ensure(h, h->ownB[1], sizeof(int) * (size_t)h->nnzB));
                BHS_TRY(ensure(h, h->ownB[2], sizeof(value_t) * (size_t)h->nnzB));
                BHS_HIP(hipMemcpyAsync(h->ownB[1].p, h->dBj, sizeof(int) * (size_t)h->nnzB, hipMemcpyDeviceToDevice, h->stream));
                BHS_HIP(hipMemcpyAsync(h->ownB[2].p, h->dBx, sizeof(value_t) * (size_t)h->nnzB, hipMemcpyDeviceToDevice, h->stream));
                h->dBj = (const int*)h->ownB[1].p;
                h->dBx = (const value_t*)h->ownB[2].p;
            }
            BHS_TRY(sort_rows_device(h, h->k, h->dBp, (int*)h->ownB[1].p, (value_t*)h->ownB[2].p));
            BHS_TRY(check_sorted());
            BHS_HIP(hipMemcpyAsync(&flag, small + S_SORTED, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            BHS_HIP(hipStreamSynchronize(h->stream));
            h->bSorted = flag ? 0 : 1;          // (duplicate columns inside a row still count as "not ascending")
        }
    }
    // compressed pattern of B: decide now whether it pays (the multiply itself re-runs the compression inside its
    // timed region; this pass only yields the pair count)
    // (round 4: the pair count is also taken for rows of 256 to 1536 products: where B's entries come in long runs -- banded
    // matrices, dense diagonal blocks: a tenth as many pairs as entries -- the compressed pass pays from there on)
    // A data set that will try the row classes first (pipeline_symbolic's test) and has rows below the old gate leaves the
    // count to its first multiply on the general pipeline, if it ever gets there (cmpState 0: that multiply measures the
    // ratio, the ones after it use the verdict) -- poisson27pt's hand-over does not pay a pass over B for nothing.
    const bool classFirst = h->classPath && h->forcePath == 0 && h->maxTableLog2 >= 15 && h->maxRowA <= kClassMaxRowBig &&
                            h->maxRowB <= kClassMaxRowBig &&
                            (h->classPath == 2 || (avgA * avgB >= (double)h->classMinProducts && (double)h->m * avgA * avgB >= 6e7));
    if (h->compressB == 1 && (avgA * avgB < 256.0 || !h->bSorted)) h->cmpState = -1;
    else if (h->compressB == 1 && classFirst && avgA * avgB <= 1536.0) h->cmpState = 0;
    else if (h->compressB == 1 && h->nnzB > 0 && h->k > 0) {
        int* small = (int*)h->small.p;
        BHS_TRY(ensure(h, h->cExt, sizeof(int2) * (size_t)h->k));
        BHS_TRY(ensure(h, h->cLen, sizeof(int2) * (size_t)h->k));
        BHS_TRY(ensure(h, h->cPair, sizeof(int2) * (size_t)h->nnzB));
        BHS_HIP(hipMemsetAsync(small + S_PAIRS, 0, 8, h->stream));
        BHS_TRY(launch_compress_b(h));
        unsigned long long pairs = 0;
        BHS_HIP(hipMemcpyAsync(&pairs, small + S_PAIRS, 8, hipMemcpyDeviceToHost, h->stream));
        BHS_HIP(hipStreamSynchronize(h->stream));
        h->cmpState = ((avgA * avgB > 1536.0 && (double)pairs <= 0.6 * (double)h->nnzB) || (double)pairs <= 0.25 * (double)h->nnzB) ? 1 : -1;
        if (h->verbose > 1) printf("  [compress_b] %llu pairs for %d entries: %s\n", pairs, h->nnzB, h->cmpState > 0 ? "used" : "not used");
    }
    if (h->useSpa) BHS_TRY(ensure_spa(h));
    h->hasData = true;
    h->hasC = false;
    return BHS_SUCCESS;
}

}  // namespace

// ============================================================== C-ABI
extern "C" {

int bhs_create(bhs_handle** out, int device_count, const int* device_ids)
{
    if (!out || device_count != 1) return BHS_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return BHS_ERR_NO_DEVICE; }
    const int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= ndev) return BHS_ERR_INVALID_ARG;
    bhs_handle* h = new (std::nothrow) bhs_handle();
    if (!h) return BHS_ERR_ALLOC;
    h->device = dev;
    if (hipSetDevice(dev) != hipSuccess) { delete h; return BHS_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return BHS_ERR_NO_DEVICE; }
    h->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "[bhsparse_hip] device %d is %s; this library carries gfx950 code objects only\n", dev,
                prop.gcnArchName);
        delete h;
        return BHS_ERR_NO_DEVICE;
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return BHS_ERR_LAUNCH; }
    for (int i = 0; i < 5; ++i)
        if (hipEventCreate(&h->ev[i]) != hipSuccess) { delete h; return BHS_ERR_LAUNCH; }
    if (hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evScanDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evCopyDone, hipEventDisableTiming) != hipSuccess) { delete h; return BHS_ERR_LAUNCH; }
    if (hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess) { delete h; return BHS_ERR_LAUNCH; }
    for (int i = 0; i < bhs_handle::kBinStreams; ++i)
        if (hipStreamCreateWithFlags(&h->binStream[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->evJoin[i], hipEventDisableTiming) != hipSuccess) { delete h; return BHS_ERR_LAUNCH; }
    h->ls = h->stream;
    if (hipHostMalloc((void**)&h->hostSmall, sizeof(int) * (S_SMALL_INTS + 4 * kMaxBins + 16), hipHostMallocDefault) != hipSuccess) {
        delete h;
        return BHS_ERR_ALLOC;
    }
    h->stats.reserve(64);
    *out = h;
    return BHS_SUCCESS;
}

int bhs_set_verbose(bhs_handle* h, int level)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (level && !h->bannerDone) {
        h->bannerDone = true;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess)
            printf("Device [ %d ] %s (%s) @ %.0f MHz, %d CUs, %.0f GB HBM\n", h->device,
                   prop.name[0] ? prop.name : "AMD Instinct", prop.gcnArchName, prop.clockRate * 1e-3,
                   prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0);
    }
    h->verbose = level;
    return BHS_SUCCESS;
}

// (keepOutput: bhs_set_data[_device] replaces the data set but keeps the output arrays of the grow-only pool -- a hipFree
// and hipMalloc of 3 GB cost 0.5 ms per hand-over on poisson27pt 128^3, and where the new arrays land moves the numeric
// kernel's time by several per cent; the caller's bhs_free_data releases them as the reference's
// free_mem does, bhsparse_cuda.h:3006-3020)
static int free_data(bhs_handle* h, bool keepOutput)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 3; ++i) { release(h->ownA[i]); release(h->ownB[i]); }
    if (!keepOutput) {
        release(h->Cj);
        release(h->Cx);
    }
    h->dAp = h->dAj = h->dBp = h->dBj = nullptr;
    h->dAx = h->dBx = nullptr;
    h->hasData = h->hasC = h->ownAB = false;
    h->extCj = nullptr; h->extCx = nullptr; h->extCap = 0;
    h->ps.open = false;
    return BHS_SUCCESS;
}

int bhs_free_data(bhs_handle* h) { return free_data(h, false); }

int bhs_destroy(bhs_handle* h)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    bhs_free_data(h);
    release(h->Cp);
    release(h->ub);
    release(h->queue);
    release(h->cExt);
    release(h->cPair);
    release(h->sortList);
    release(h->sortCnt);
    release(h->sortK);
    release(h->sortV);
    release(h->cLen);
    release(h->symKey);
    release(h->blockSum);
    release(h->small);
    release(h->spaRank);
    release(h->longList); release(h->longPart);
    release(h->classB); release(h->classC); release(h->classTab); release(h->classInfo);
    release(h->classHeads); release(h->classHeadCnt); release(h->classMap); release(h->classMapA); release(h->classRing); release(h->classRel); release(h->classLane);
    release(h->classBigIdx); release(h->classBigMap);
    release(h->bWin); release(h->bWinTab); release(h->bWinSpill);
    release(h->hubBits); release(h->hubRank); release(h->hubItems); release(h->hubSeg); release(h->hubCtl);
    release(h->spaBits);
    if (h->hostSmall) (void)hipHostFree(h->hostSmall);
    if (h->hostRowPtr) (void)hipHostFree(h->hostRowPtr);
    for (int i = 0; i < bhs_handle::kBinStreams; ++i) {
        if (h->binStream[i]) (void)hipStreamDestroy(h->binStream[i]);
        if (h->evJoin[i]) (void)hipEventDestroy(h->evJoin[i]);
    }
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
    if (h->evScanDone) (void)hipEventDestroy(h->evScanDone);
    if (h->evCopyDone) (void)hipEventDestroy(h->evCopyDone);
    for (auto& p : h->evPool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (int i = 0; i < 5; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return BHS_SUCCESS;
}

static int check_dims(int m, int k, int n, int nnzA, int nnzB)
{
    return (m < 0 || k < 0 || n < 0 || nnzA < 0 || nnzB < 0) ? BHS_ERR_INVALID_ARG : BHS_SUCCESS;
}

int bhs_set_data(bhs_handle* h, int m, int k, int n, int nnzA, const bhs_value_t* csrValA, const int* csrRowPtrA,
                 const int* csrColIndA, int nnzB, const bhs_value_t* csrValB, const int* csrRowPtrB,
                 const int* csrColIndB)
{
    if (!h || check_dims(m, k, n, nnzA, nnzB)) return BHS_ERR_INVALID_ARG;
    if (!csrRowPtrA || !csrRowPtrB || (nnzA && (!csrValA || !csrColIndA)) || (nnzB && (!csrValB || !csrColIndB)))
        return BHS_ERR_INVALID_ARG;
    BHS_HIP(hipSetDevice(h->device));
    free_data(h, true);
    h->m = m; h->k = k; h->n = n; h->nnzA = nnzA; h->nnzB = nnzB;
    BHS_TRY(ensure(h, h->ownA[0], sizeof(int) * ((size_t)m + 1)));
    BHS_TRY(ensure(h, h->ownA[1], sizeof(int) * (size_t)std::max(nnzA, 1)));
    BHS_TRY(ensure(h, h->ownA[2], sizeof(value_t) * (size_t)std::max(nnzA, 1)));
    BHS_TRY(ensure(h, h->ownB[0], sizeof(int) * ((size_t)k + 1)));
    BHS_TRY(ensure(h, h->ownB[1], sizeof(int) * (size_t)std::max(nnzB, 1)));
    BHS_TRY(ensure(h, h->ownB[2], sizeof(value_t) * (size_t)std::max(nnzB, 1)));
    BHS_HIP(hipMemcpyAsync(h->ownA[0].p, csrRowPtrA, sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice, h->stream));
    if (nnzA) {
        BHS_HIP(hipMemcpyAsync(h->ownA[1].p, csrColIndA, sizeof(int) * (size_t)nnzA, hipMemcpyHostToDevice, h->stream));
        BHS_HIP(hipMemcpyAsync(h->ownA[2].p, csrValA, sizeof(value_t) * (size_t)nnzA, hipMemcpyHostToDevice, h->stream));
    }
    BHS_HIP(hipMemcpyAsync(h->ownB[0].p, csrRowPtrB, sizeof(int) * ((size_t)k + 1), hipMemcpyHostToDevice, h->stream));
    if (nnzB) {
        BHS_HIP(hipMemcpyAsync(h->ownB[1].p, csrColIndB, sizeof(int) * (size_t)nnzB, hipMemcpyHostToDevice, h->stream));
        BHS_HIP(hipMemcpyAsync(h->ownB[2].p, csrValB, sizeof(value_t) * (size_t)nnzB, hipMemcpyHostToDevice, h->stream));
    }
    BHS_HIP(hipStreamSynchronize(h->stream));
    h->dAp = (const int*)h->ownA[0].p; h->dAj = (const int*)h->ownA[1].p; h->dAx = (const value_t*)h->ownA[2].p;
    h->dBp = (const int*)h->ownB[0].p; h->dBj = (const int*)h->ownB[1].p; h->dBx = (const value_t*)h->ownB[2].p;
    h->ownAB = true;
    BHS_TRY(ensure_host_rowptr(h, sizeof(int) * ((size_t)m + 1)));   // pinned staging, outside the timed region
    return finish_set_data(h);
}

int bhs_set_data_device(bhs_handle* h, int m, int k, int n, int nnzA, const bhs_value_t* d_valA, const int* d_rowPtrA,
                        const int* d_colIndA, int nnzB, const bhs_value_t* d_valB, const int* d_rowPtrB,
                        const int* d_colIndB)
{
    if (!h || check_dims(m, k, n, nnzA, nnzB)) return BHS_ERR_INVALID_ARG;
    if (!d_rowPtrA || !d_rowPtrB || (nnzA && (!d_valA || !d_colIndA)) || (nnzB && (!d_valB || !d_colIndB)))
        return BHS_ERR_INVALID_ARG;
    BHS_HIP(hipSetDevice(h->device));
    free_data(h, true);
    h->m = m; h->k = k; h->n = n; h->nnzA = nnzA; h->nnzB = nnzB;
    h->dAp = d_rowPtrA; h->dAj = d_colIndA; h->dAx = d_valA;
    h->dBp = d_rowPtrB; h->dBj = d_colIndB; h->dBx = d_valB;
    h->ownAB = false;
    return finish_set_data(h);
}

int bhs_warmup(bhs_handle* h)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->hasData) return BHS_ERR_NOT_READY;
    BHS_HIP(hipSetDevice(h->device));
    h->wantHostRowPtr = h->ownAB;      // host-pointer callers get rowPtrC back: warm that path up too
    const int rc = run_pipeline(h);
    h->wantHostRowPtr = false;
    if (rc == BHS_SUCCESS && h->rowPtrStaged) BHS_HIP(hipEventSynchronize(h->evCopyDone));
    return rc;
}

int bhs_spgemm(bhs_handle* h, int* rowPtrC_out, int64_t* nnzCt_out, int* nnzC_out, double stage_ms_out[4])
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->hasData) return BHS_ERR_NOT_READY;
    BHS_HIP(hipSetDevice(h->device));
    h->wantHostRowPtr = rowPtrC_out != nullptr;
    if (h->useSpa && (h->spaDirty || h->spaCols != h->n)) BHS_TRY(ensure_spa(h));
    const int rc = run_pipeline(h);
    h->wantHostRowPtr = false;
    if (rc) { h->spaDirty = true; return rc; }
    if (h->verbose) {
        printf("STAGE 1 time: %g ms.\n", h->stageMs[0]);
        printf("STAGE 2 time: %g ms.\n", h->stageMs[1]);
        printf("exact size %lld out of full size %lld\n", h->nnzC, h->nnzCt);
        printf("STAGE 3 time: %g ms.\n", h->stageMs[2]);
        printf("STAGE 4 time: %g ms.\n", h->stageMs[3]);
    }
    if (rowPtrC_out) {
        if (h->rowPtrStaged) {
            BHS_HIP(hipEventSynchronize(h->evCopyDone));
            memcpy(rowPtrC_out, h->hostRowPtr, sizeof(int) * ((size_t)h->m + 1));
        } else {
            BHS_HIP(hipMemcpyAsync(rowPtrC_out, h->Cp.p, sizeof(int) * ((size_t)h->m + 1), hipMemcpyDeviceToHost, h->stream));
            BHS_HIP(hipStreamSynchronize(h->stream));
        }
    }
    if (nnzCt_out) *nnzCt_out = h->nnzCt;
    if (nnzC_out) *nnzC_out = (int)h->nnzC;
    if (stage_ms_out) for (int i = 0; i < 4; ++i) stage_ms_out[i] = h->stageMs[i];
    return BHS_SUCCESS;
}

// ---- a multiply in two halves (multi-GPU: the counts of every rank are exchanged between the halves, and the
// numeric half runs in row ranges so that the all-gatherv of one range overlaps the numeric kernels of the next)
int bhs_spgemm_symbolic(bhs_handle* h, int64_t* nnzCt_out, int* nnzC_out)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->hasData) return BHS_ERR_NOT_READY;
    BHS_HIP(hipSetDevice(h->device));
    h->wantHostRowPtr = false;
    if (h->useSpa && (h->spaDirty || h->spaCols != h->n)) BHS_TRY(ensure_spa(h));
    const long long savedCap = h->extCap;
    h->extCap = h->extCj ? (1LL << 62) : 0;          // the output arrays are (re)bound between the halves: no capacity check yet
    h->lazyOut = true;                               // ... and a caller that binds its own never makes the library allocate C
    int rc = pipeline_symbolic(h);
    h->lazyOut = false;
    h->extCap = savedCap;
    if (rc) { quiesce(h); h->ps.open = false; h->spaDirty = true; return rc; }
    if (nnzCt_out) *nnzCt_out = h->nnzCt;
    if (nnzC_out) *nnzC_out = (int)h->nnzC;
    return BHS_SUCCESS;
}

int bhs_spgemm_numeric(bhs_handle* h, int row_begin, int row_end)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->ps.open) return BHS_ERR_NOT_READY;
    if (row_begin < 0 || row_end > h->m || row_begin > row_end) return BHS_ERR_INVALID_ARG;   // (the multiply stays open)
    BHS_HIP(hipSetDevice(h->device));
    if (h->extCj && h->nnzC > h->extCap) return BHS_ERR_ALLOC;
    if (!h->extCj && !h->ps.empty) {                               // the library's own output arrays (no-ops once they are large enough)
        BHS_TRY(ensure(h, h->Cj, sizeof(int) * (size_t)std::max<long long>(h->nnzC, 1)));
        BHS_TRY(ensure(h, h->Cx, sizeof(value_t) * (size_t)std::max<long long>(h->nnzC, 1)));
    }
    const int rc = numeric_stage(h, row_begin, row_end);
    if (rc) { quiesce(h); h->ps.open = false; h->spaDirty = true; }
    return rc;
}

int bhs_spgemm_finish(bhs_handle* h, double stage_ms_out[4])
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->ps.open) return BHS_ERR_NOT_READY;
    BHS_HIP(hipSetDevice(h->device));
    const int rc = pipeline_finish(h);
    if (rc) { quiesce(h); h->spaDirty = true; return rc; }
    if (stage_ms_out) for (int i = 0; i < 4; ++i) stage_ms_out[i] = h->stageMs[i];
    return BHS_SUCCESS;
}

int bhs_set_output_device(bhs_handle* h, int* d_colIndC, bhs_value_t* d_valC, int64_t capacity)
{
    if (!h || capacity < 0 || ((d_colIndC == nullptr) != (d_valC == nullptr))) return BHS_ERR_INVALID_ARG;
    h->extCj = d_colIndC;
    h->extCx = (value_t*)d_valC;
    h->extCap = d_colIndC ? (long long)capacity : 0;
    return BHS_SUCCESS;
}

int bhs_get_stream(bhs_handle* h, void** stream_out)
{
    if (!h || !stream_out) return BHS_ERR_INVALID_ARG;
    *stream_out = (void*)h->stream;
    return BHS_SUCCESS;
}

int bhs_get_nnzC(bhs_handle* h, int* nnzC_out)
{
    if (!h || !nnzC_out) return BHS_ERR_INVALID_ARG;
    if (!h->hasC) return BHS_ERR_NOT_READY;
    *nnzC_out = (int)h->nnzC;
    return BHS_SUCCESS;
}

int bhs_get_C(bhs_handle* h, int* csrColIndC, bhs_value_t* csrValC)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->hasC) return BHS_ERR_NOT_READY;
    if (h->nnzC && out_cj(h) != h->resCj) return BHS_ERR_NOT_READY;   // (the result went to arrays that were unbound since: it lives there)
    if (h->nnzC && (!csrColIndC || !csrValC)) return BHS_ERR_INVALID_ARG;
    BHS_HIP(hipSetDevice(h->device));
    if (h->nnzC) {
        BHS_HIP(hipMemcpyAsync(csrColIndC, out_cj(h), sizeof(int) * (size_t)h->nnzC, hipMemcpyDeviceToHost, h->stream));
        BHS_HIP(hipMemcpyAsync(csrValC, out_cx(h), sizeof(value_t) * (size_t)h->nnzC, hipMemcpyDeviceToHost, h->stream));
    }
    BHS_HIP(hipStreamSynchronize(h->stream));
    return BHS_SUCCESS;
}

int bhs_get_rowptrC(bhs_handle* h, int* csrRowPtrC)
{
    if (!h || !csrRowPtrC) return BHS_ERR_INVALID_ARG;
    if (!h->hasC && !h->ps.open) return BHS_ERR_NOT_READY;         // (between the halves rowPtrC is already final)
    BHS_HIP(hipSetDevice(h->device));
    BHS_HIP(hipMemcpyAsync(csrRowPtrC, h->Cp.p, sizeof(int) * ((size_t)h->m + 1), hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    return BHS_SUCCESS;
}

int bhs_get_C_device(bhs_handle* h, const int** d_rowPtrC, const int** d_colIndC, const bhs_value_t** d_valC)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    if (!h->hasC && !h->ps.open) return BHS_ERR_NOT_READY;        // (between the halves rowPtrC is already final)
    if (h->hasC && h->nnzC && out_cj(h) != h->resCj && (d_colIndC || d_valC)) return BHS_ERR_NOT_READY;   // (see bhs_get_C)
    if (d_rowPtrC) *d_rowPtrC = (const int*)h->Cp.p;
    if (d_colIndC) *d_colIndC = (const int*)out_cj(h);
    if (d_valC) *d_valC = (const bhs_value_t*)out_cx(h);
    return BHS_SUCCESS;
}

int bhs_csr_sort_indices_device(bhs_handle* h, int n_row, const int* d_rowPtr, int* d_colInd, bhs_value_t* d_val)
{
    if (!h || n_row < 0 || (n_row > 0 && (!d_rowPtr || !d_colInd || !d_val))) return BHS_ERR_INVALID_ARG;
    if (n_row == 0) return BHS_SUCCESS;
    BHS_HIP(hipSetDevice(h->device));
    return sort_rows_device(h, n_row, d_rowPtr, d_colInd, (value_t*)d_val);
}

int bhs_get_kernel_stats(bhs_handle* h, bhs_kernel_stat* out, int cap)
{
    if (!h) return BHS_ERR_INVALID_ARG;
    int nrec = 0;
    for (auto& s : h->stats) {
        if (!s.launches) continue;
        if (out && nrec < cap) {
            out[nrec].name = s.name;
            out[nrec].launches = s.launches;
            out[nrec].ms = s.ms;
            out[nrec].rows = s.rows;
            out[nrec].products = s.products;
            out[nrec].nnz_out = s.nnz_out;
            out[nrec].nnzA_rows = s.nnzA_rows;
        }
        ++nrec;
    }
    return nrec;
}

int bhs_set_option(bhs_handle* h, const char* key, int64_t value)
{
    if (!h || !key) return BHS_ERR_INVALID_ARG;
    if (!strcmp(key, "force_path")) { h->forcePath = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "max_table_log2")) {
        if (value < 6 || value > 15) return BHS_ERR_INVALID_ARG;
        h->maxTableLog2 = (int)value;
        return BHS_SUCCESS;
    }
    if (!strcmp(key, "no_pack32")) { h->noPack32 = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "wg_per_cu")) { h->wgPerCU = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "class_super_rows")) {                          // (whole runs of round 4's ring kernel)
        h->classSuperRows = (int)std::max<long long>(0, std::min<long long>(value, 1 << 15)) / kClassRun * kClassRun;
        return BHS_SUCCESS;
    }
    if (!strcmp(key, "spa")) { h->useSpa = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "small_b")) { h->allowSmallB = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "wave_first")) { h->waveFirst = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "lane_first")) { h->laneFirst = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "direct_bins")) { h->directBins = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "sort_b")) { h->sortB = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "lane_rows")) { h->laneRows = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "lane_numeric")) { h->laneNumeric = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "compress_b")) { h->compressB = (int)value; h->cmpState = 0; return BHS_SUCCESS; }
    if (!strcmp(key, "kernel_stats")) { h->kernelStats = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "concurrent_bins")) { h->concurrentBins = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "window_bitmap")) { h->useWindowBitmap = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "lds_bitmap_min_log2")) { h->ldsBitmapMinLog2 = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "lds_bitmap")) { h->useLdsBitmap = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "hub_min_products")) { h->hubMin = (int)std::min<int64_t>(value, 0x7fffffff); return BHS_SUCCESS; }
    if (!strcmp(key, "hub_item_products")) { if (value < 64) return BHS_ERR_INVALID_ARG; h->hubItemProducts = (int)std::min<int64_t>(value, 1 << 30); return BHS_SUCCESS; }
    if (!strcmp(key, "scan_one_pass")) { h->scanOnePass = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "ub_lanes")) {      // (tuning hook) lanes per row of A in k_upper_bound, a power of two; rows beyond 32 passes go to its long list
        int g = 1;
        while (g < value && g < 64) g <<= 1;
        h->ubG = g;
        h->ubLong = std::max(64, std::min(kUbLongA, 16 * g));
        return BHS_SUCCESS;
    }
    if (!strcmp(key, "ub_long")) { h->ubLong = (int)std::max<int64_t>(16, std::min<int64_t>(value, kUbLongA)); return BHS_SUCCESS; }
    if (!strcmp(key, "class_grid_mul")) { h->classGridMul = (int)std::max<int64_t>(1, value); return BHS_SUCCESS; }
    if (!strcmp(key, "class_per_lane")) { h->classPerLane = (int)std::max<int64_t>(1, value); return BHS_SUCCESS; }
    if (!strcmp(key, "class_path")) { h->classPath = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); h->classState = 0; return BHS_SUCCESS; }
    if (!strcmp(key, "class_heads")) { h->classHeadsOn = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); return BHS_SUCCESS; }
    if (!strcmp(key, "class_numeric")) { h->classNumeric = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); return BHS_SUCCESS; }
    if (!strcmp(key, "class_min_products")) { h->classMinProducts = (int)std::max<int64_t>(0, value); return BHS_SUCCESS; }
    if (!strcmp(key, "merge_bitmap_bins")) { h->mergeBitmapBins = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "hub_aggregate")) { h->hubAggregate = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "hub_slots")) { h->hubMaxSlots = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "spa_slots")) { h->spaMaxSlots = (int)value; h->spaDirty = true; return BHS_SUCCESS; }
    if (!strcmp(key, "sym_load_pct") || !strcmp(key, "num_load_pct")) {
        if (value < 5 || value > 75) return BHS_ERR_INVALID_ARG;
        (key[0] == 's' ? h->symLoadPct : h->numLoadPct) = (int)value;
        return BHS_SUCCESS;
    }
    if (!strcmp(key, "verbose")) return bhs_set_verbose(h, (int)value);
    return BHS_ERR_INVALID_ARG;
}

int bhs_get_info(bhs_handle* h, const char* key, int64_t* value_out)
{
    if (!h || !key || !value_out) return BHS_ERR_INVALID_ARG;
    if (!h->hasData) return BHS_ERR_NOT_READY;
    if (!strcmp(key, "b_sorted")) { *value_out = h->bSorted; return BHS_SUCCESS; }
    if (!strcmp(key, "max_row_a")) { *value_out = h->maxRowA; return BHS_SUCCESS; }
    if (!strcmp(key, "max_row_b")) { *value_out = h->maxRowB; return BHS_SUCCESS; }
    if (!strcmp(key, "local_a")) { *value_out = h->localA; return BHS_SUCCESS; }
    if (!strcmp(key, "line_a")) { *value_out = h->lineA; return BHS_SUCCESS; }
    if (!strcmp(key, "compress_b_used")) { *value_out = h->cmpState > 0 ? 1 : 0; return BHS_SUCCESS; }
    return BHS_ERR_INVALID_ARG;
}

int bhs_get_class_tables_device(bhs_handle* h, const int** d_classC, const void** d_classInfo, const int** d_classRel,
                                int* slots_out, int* rel_stride_out, int* usable_out)
{
    if (!h || !usable_out) return BHS_ERR_INVALID_ARG;
    *usable_out = 0;
    if (slots_out) *slots_out = kClassSlots;                       // (the table geometry is a property of the build)
    if (rel_stride_out) *rel_stride_out = kClassMaxNnz;
    if (!h->hasC && !h->ps.open) return BHS_ERR_NOT_READY;
    const bool usable = h->ps.useClass && !h->ps.classBig && !h->ps.empty;
    *usable_out = usable ? 1 : 0;
    if (d_classC) *d_classC = usable ? (const int*)h->classC.p : nullptr;
    if (d_classInfo) *d_classInfo = usable ? (const void*)h->classInfo.p : nullptr;
    if (d_classRel) *d_classRel = usable ? (const int*)h->classRel.p : nullptr;
    if (slots_out) *slots_out = kClassSlots;
    if (rel_stride_out) *rel_stride_out = kClassMaxNnz;
    return BHS_SUCCESS;
}

int bhs_expand_class_columns_device(void* stream, int n, int row0, const int* d_classC, const void* d_classInfo,
                                    const int* d_classRel, int rel_stride, const int* d_rowPtrC, int* d_colIndC)
{
    if (n < 0 || rel_stride <= 0 || (n > 0 && (!d_classC || !d_classInfo || !d_classRel || !d_rowPtrC || !d_colIndC))) return BHS_ERR_INVALID_ARG;
    if (n == 0) return BHS_SUCCESS;
    const unsigned grid = (unsigned)std::min<long long>(((long long)n + 3) / 4, 1 << 16);
    hipLaunchKernelGGL(k_class_expand_columns, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, row0, d_classC,
                       (const int4*)d_classInfo, d_classRel, rel_stride, d_rowPtrC, d_colIndC);
    return hipGetLastError() == hipSuccess ? (int)BHS_SUCCESS : (int)BHS_ERR_LAUNCH;
}

const char* bhs_strerror(int status)
{
    switch (status) {
        case BHS_SUCCESS: return "success";
        case BHS_ERR_INVALID_ARG: return "invalid argument";
        case BHS_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
        case BHS_ERR_ALLOC: return "device memory allocation failed";
        case BHS_ERR_LAUNCH: return "HIP runtime / kernel launch error";
        case BHS_ERR_NNZ_OVERFLOW: return "nnz(C) exceeds int32 index_type";
        case BHS_ERR_NOT_READY: return "call order violated (no data / no result yet)";
        case BHS_ERR_INTERNAL: return "accumulator overflow not resolved";
        case BHS_ERR_PEER: return "another rank of the multi-GPU job failed";
        default: return "unknown bhsparse_hip status";
    }
}

#ifdef BHS_VALUE_FLOAT
const char* bhs_version(void) { return "bhsparse_hip 0.1 (gfx950, value_type float)"; }
#else
const char* bhs_version(void) { return "bhsparse_hip 0.1 (gfx950, value_type double)"; }
#endif

#if BHS_PHASES || BHS_PHASES_SPA || BHS_PHASES_CLS
// measurement-only builds (tools/build_variants.sh -DBHS_PHASES=1): read and reset the phase counters
__attribute__((visibility("default"))) int bhs_debug_phases(unsigned long long* out)
{
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

}  // extern "C"
