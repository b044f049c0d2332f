// bhsparse_hip.hip — libbhsparse_hip.so (see include/bhsparse_hip.h): the handle and its helpers here; the launch helpers,
// the pipeline, the hand-over of a data set and the C-ABI in bhs_host_{launch,pipeline,setdata,cabi}.inc.h (one translation unit).
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
#if BHS_LAB               // (kernels that were built, measured slower and left out of the product: tools/lab_tests.sh builds and tests them)
#include "bhs_row_span.hip.h"
#endif
#include "bhs_row_window.hip.h"
#include "bhs_row_quad.hip.h"
#include "bhs_compress.hip.h"
#include "bhs_row_lane.hip.h"
#if BHS_LAB
#include "bhs_row_tiny.hip.h"
#endif
#include "bhs_sort.hip.h"
#include "bhs_hub.hip.h"
#include "bhs_class.hip.h"
#include "bhs_class_mix.hip.h"
#include "bhs_class_wg.hip.h"
#if BHS_LAB && defined(BHS_RING_LAB)       // (the ring kernel with its ablation switches: a copy outside the product's sources)
#include "../../tools/lab/bhs_class_ring_lab.hip.h"
#else
#include "bhs_class_ring.hip.h"
#endif
#include "bhs_class_fused.hip.h"
#include "bhs_class_tile.hip.h"
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
// What a lane's walk over its row costs grows with the LENGTH of the B rows, not with the products alone (round 6,
// tools/toeplitz_case.py, 400 k rows of nA x nB entries, symbolic lane kernel against the wave kernels: 8 x 16 0.23 / 0.32 ms,
// 4 x 20 0.20 / 0.29, but 6 x 20 0.37 / 0.28, 4 x 30 0.82 / 0.27, 9 x 31 1.98 / 0.52; ~ nA nB^2): the lane kernels take rows with
// nA nB^2 <= kLaneCost -- a row's products p stand in for nA nB: p^2 <= kLaneCost nA.  The numeric lane kernel stages a row's
// entries through LDS 16 at a time: beyond ~50 entries a row the wave kernels' tables win (7 x 7: 0.19 / 0.17 ms even,
// 8 x 8 0.31 / 0.25, 8 x 16 0.92 / 0.43, 4 x 30 1.42 / 0.38).
constexpr int kLaneCost = 2048;
constexpr int kLaneNumMax = 56;

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

// The ladder with fewer steps, for a handful of rows (mixed mode's irregular rows, bhs_class_mix.hip.h): a launch costs more
// than a table that is too large for a few hundred rows.  keep: the bins that stay; a row of a dropped bin moves up.
BinSpec coarse_spec(BinSpec s, unsigned keep)
{
    s.quadMax = 0;
    int prev = 0;
    for (int b = 2; b < s.nbins - 1; ++b) {
        if (!((keep >> b) & 1u)) s.upper[b] = prev;
        prev = s.upper[b];
    }
    return s;
}
constexpr unsigned kCoarseSym = (1u << 4) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10) | (1u << 11);
constexpr unsigned kCoarseNum = (1u << 4) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10);

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
    int spanPath = 0;                    // rows accumulated over their column span (bhs_row_span.hip.h) where the hand-over's scans say every row's span fits: 1 when they do, 0 (default) never -- measured 2.7 + 4.6 ms against the hash kernels' 0.66 + 1.68 on the banded input (profiles/r05_experiments.md)
    int spanState = 0;                   //   -1: a row beyond the bitmap was met on the device, the data set stays on the hash kernels
    int reachL = 0, reachR = 0, widthA = 0;   //   how far left / right of its own number a row of B reaches, the widest row of A (first to last entry)
    // the classes' figures of this data set's last whole multiply on the ring kernel: the next one launches its numeric kernel on
    // them before it has seen its own (pipeline_symbolic), checked on the device (k_class_spec_check)
    struct ClassSpec { bool valid = false; int cs[CS_INTS]; long long nnzC = 0, nnzCt = 0; } classSpec;
    // ... the same for a lane-first multiply whose numeric stage ran k_row_lane on every row (k_lane_spec_check)
    struct LaneSpec { bool valid = false; int laneK = 0; long long nnzC = 0, nnzCt = 0; } laneSpec;
    // ... and rowPtrC made by the numeric kernel itself from the symbolic kernel's counts and block sums (k_row_lane), no scan
    // kernel, no check kernel: option "lane_from_counts" (1; up to kLaneFromCountsBlocks blocks of 256 rows)
    int laneFromCounts = 1;
    DevBuf laneBlockSums;
    int classTilePiece = 0;              // option "class_tile_piece": rows per wave of that classifier (0: one piece per wave slot)
    int classTile = 1;                   // option "class_tile": the classifier with a lane per row (bhs_class_tile.hip.h) where rows have at most 32 entries
    int specNumeric = 1;                 // option "spec_numeric": 0 never launch speculatively
    int spinWait = 1;                    // option "spin_wait": the multiply's waits for its stream poll (wait_stream)
    int spinWaitUs = 0;                  // option "spin_wait_us": the longest a wait polls before it sleeps (0: four times the last multiply's wall time, 0.5 .. 5 ms)
    double lastMultiplyMs = 0.5;         // wall time of this handle's last bhs_spgemm
    long long specLaunches = 0, specRefuted = 0;
    int numDirectHint = -1;              // this data set's last whole multiply ran its numeric stage without queues (1), with them (0); -1: none yet
    int earlyFill = 1;                   // general pipeline: the queues filled while the host waits for the bin counts (starts computed on the device)
    int sortedScan = 1;                  // the sortedness scan of B at hand-over: 1 element-parallel (k_sorted_flat + k_sorted_starts), 0 row by row (k_check_sorted)
    int bSorted = 1;
    int logL = 5, ubG = 8, ubLong = kUbLongA;   // k_upper_bound: lanes per row of A, rows beyond ubLong entries go to its long list
    // C
    DevBuf Cp, Cj, Cx;
    long long nnzC = 0;
    long long nnzCt = 0;
    // workspace
    DevBuf ub, queue, blockSum, small;   // small: counters (see layout below)
    DevBuf spaRank, spaBits;             // bitmap-accumulator slots for rows beyond the LDS tables (bitmaps kept all-zero)
    int spaSlots = 0, spaCols = -1, useSpa = 1, spaMaxSlots = 0, useLdsBitmap = 1, ldsBitmapMinLog2 = 12, symBitmapMinLog2 = 13;
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
    int classGridMul = 4, classPerLane = 4, classMinProducts = 64;    // tuning hooks of the classifier (entries per lane: 2 until round 5; the one-pass classifier measured 0.295 -> 0.268 ms with 4 on poisson27pt 128^3, 0.537 -> 0.477 on 160^3)
    int scanOnePass = 1;                 // stage 3 of the general pipeline: k_scan_onepass (0: the three scan kernels of rounds 1-3)
    unsigned scanEpoch = 0;              // tag of this multiply's tile words
    int classHeadsOn = 2;                // 2: one pass per matrix (k_class_fused: the wave that finds a row differing from the row before it takes it through the class table itself); 1: rounds 3-4's three launches (k_class_heads, k_class_rows on its lists, k_class_propagate); 0: every row through the table
    int classNumeric = 2;                // numeric kernel of the class path: 2 round 5's ring kernel (bhs_class_ring.hip.h) where its LDS fits, 1 round 4's (bhs_class_wg.hip.h), 0 k_class_numeric_atomic (round 2) always
    int classPath = 1;                   // 0 never; 1 for data sets whose rows of A and B have <= 64 entries and >= classMinProducts products on average (2: any), until one multiply finds
    int classState = 0;                  //   rows it cannot classify (classState -1: the data set stays on the general pipeline)
    // Round 6, mixed mode (bhs_class_mix.hip.h): rows without a class go through the general pipeline's kernels, the others
    // stay on the class kernels.  classMixed: this data set's last multiply had such rows -- the next one runs the mixed
    // flow from the start (everything it decides it works out anew on the device); option "class_mixed" 0: one row without
    // a class sends the data set to the general pipeline, as until round 5.
    int mixOn = 1, classMixed = 0;
    bool mixProbed = false;              // the mixed flow found every class of this data set worth its pattern: many classes alone no longer ask for it
    // option "ring_dynamic": the ring kernel's super-runs handed out by a counter per XCD -- 1 always, 0 never (a fixed share per
    // wave), 2 (default) where other kernels run beside it (mixed mode with a handful of irregular rows: its workgroups then
    // start late on the CUs those kernels hold).  Measured in one process: poisson27pt 160^3 2.5689 against 2.5683 ms,
    // 128^3 1.3652 / 1.3642, poisson9pt 1024^2 0.3499 / 0.3414 -- an even share costs nothing where the memory system is the
    // limit (the waves with a line more run while the others' bandwidth is free), the counter costs a round trip per line.
    int ringDynamic = 2;
    int mixFork = 0;                     // option "class_mixed_fork": the irregular rows' bins side by side on the side streams whatever their number (measurement)
    int mixMaxPct = 30;                  // option "class_mixed_max_pct": more irregular rows than this share of all rows -> the general pipeline
    int lenStatsA[4] = {0, 0, 0, 0}, lenStatsB[4] = {0, 0, 0, 0};   // bhs_set_data's scan (k_max_row): rows beyond 64 entries, the longest within 64, rows beyond 256, the longest within 256
    DevBuf mixList, classCount;          // the irregular rows of A; rows per class (B's table, then A's)
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
    hipStream_t besideStream = nullptr;  // mixed mode with a handful of irregular rows: their kernels' stream while the ring kernel runs on `stream`
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
    int tinyRows = 0;                    // rows of <= 32 products with every product in registers (bhs_row_tiny.hip.h) where the hand-over's longest rows allow: 1 yes, 0 (default) never -- measured 0.044 + 0.150 ms against k_row_lane's 0.036 + 0.113 on poisson5pt 1024^2
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
        bool numQueueFilled = false;     // the numeric queues of the whole multiply were filled behind the scan
        int spanWPL = 0;                 // this multiply's wave bins run k_row_span with this many bitmap words per lane (0: hash kernels)
        bool bWinBuilt = false;          // bWin / bWinTab belong to this multiply
        bool specLaunched = false;       // the numeric kernel goes out on the last multiply's figures (classSpec), k_class_spec_check decides
        bool specLane = false;           // ... a lane-first multiply's (laneSpec, k_lane_spec_check)
        bool fromCounts = false;         // ... whose numeric kernel makes rowPtrC from the counts (no scan ran)
        bool laneFirst = false;          // this multiply's symbolic stage was the lane kernel on every row, no upper-bound pass
        bool mixed = false;              // class path with irregular rows on the general kernels (bhs_class_mix.hip.h)
        int mixRows = 0;                 // ... how many
        bool mixNumFilled = false;       // ... their numeric queue (all rows) was filled behind the scan
        int mixSymCount[kMaxBins], mixNumCount[kMaxBins];
        bool ringBeside = false;         // this launch of the ring kernel runs beside other kernels (mixed mode, a handful of irregular rows)
        int ringLaunches = 0;            // launches of the ring kernel in this multiply (its tickets are zero for the first)
        long long mixProducts = 0;       // ... their products
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
       S_MIX_COUNT = 104 + 195 /* rows on the mixed mode's list of irregular rows (bhs_class_mix.hip.h) */,
       S_MIX_SYM2 = 104 + 196 /* kMaxBins: their symbolic bins on the coarse ladder (S_SYM_COUNT: on the fine one) */,
       S_RING_TICKETS = 104 + 212 /* 8: the ring kernel's next super-run per XCD (bhs_class_ring.hip.h) */,
       S_ZERO_END = 104 + 220,   /* everything below is zeroed at the start of every spgemm (324 ints: a multiple of 16 bytes -- hipMemsetAsync is ONE fill kernel then, three otherwise) */
       S_SORTED = 324, S_MAXROW = 325, S_SPEC = 326 /* k_class_spec_check's word: 1 the speculative numeric launch stands, 2 refuted */,
       S_LONG_B = 327 /* rows on k_check_sorted's long list */,
       S_TICKETS = 328 /* kMaxBins: one scheduler ticket per bin, bins run concurrently */,
       S_CT_SLOTS = 344 /* 64 x u64: product count of a lane-first multiply, spread over 64 counters */,
       S_SCAN = 472 /* bhs_set_data's scans: longest row of A, its period hint, the same for B, A's entries near the diagonal,
                       the length of A's grid lines */,
       S_SPAN = 478 /* bhs_set_data's scans for bhs_row_span.hip.h: left reach of B's rows, right reach, widest row of A */,
       S_ROWLEN = 482 /* bhs_set_data's scans for the classifier's sizes (k_max_row): per matrix rows beyond 64 entries, the
                         longest row within 64, rows beyond 256, the longest within 256 -- A's four, then B's */,
       S_SMALL_INTS = 490 };

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
    if (!h->kernelStats || (h->kernelStats == 2 && strncmp(name, "numeric", 7) != 0)) {   // no events: the record still counts launches / rows
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

#include "bhs_host_launch.inc.h"
#include "bhs_host_pipeline.inc.h"
#include "bhs_host_setdata.inc.h"

}  // namespace

#include "bhs_host_cabi.inc.h"
