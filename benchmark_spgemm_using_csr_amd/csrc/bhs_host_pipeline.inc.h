// bhs_host_pipeline.inc.h -- one multiply: the class path and the general pipeline, stage by stage; the multiply in two halves
// (A part of bhsparse_hip.hip's translation unit: included there, inside its unnamed namespace where that applies.)

constexpr int kSpecRefuted = -1000;      // pipeline_finish to run_pipeline_impl only: never leaves the library

// The host waits for h->stream in the middle of a multiply (the bins' counts, nnzC) and at its end.  hipStreamSynchronize puts
// the thread to sleep when the wait is long, and waking it costs ~20 us -- of a multiply of 0.2 .. 2 ms, two or three times.
// Poll instead (option "spin_wait", on by default), with a pause between polls, for as long as a multiply of this data set
// may plausibly take (four times the last one's wall time, between 0.5 and 5 ms; option "spin_wait_us" sets a fixed cap) --
// then sleep as before: a drop-in library must not keep a core busy for the 10 - 20 ms of a large multiply, let alone eight
// ranks' cores next to RCCL's proxy threads.
int wait_stream(bhs_handle* h)
{
    if (h->spinWait) {
        const long long capUs = h->spinWaitUs > 0 ? h->spinWaitUs : std::max<long long>(500, std::min<long long>(5000, (long long)(4000.0 * h->lastMultiplyMs)));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; ++i) {
            const hipError_t e = hipStreamQuery(h->stream);
            if (e == hipSuccess) return BHS_SUCCESS;
            if (e != hipErrorNotReady) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? (int)BHS_ERR_ALLOC : (int)BHS_ERR_LAUNCH; }
            __builtin_ia32_pause();
            if ((i & 15) == 15 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(capUs)) break;
        }
    }
    BHS_HIP(hipStreamSynchronize(h->stream));
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
int fork_bins(bhs_handle* h, const int* count, int nbins, bool always = false)
{
    h->ls = h->stream;
    int used = 0;
    for (int b = 1; b < nbins; ++b) used += count[b] > 0;
    // forking and joining four streams costs ~70 us of event traffic: it pays for power-law matrices whose rows
    // spread over many small bins, not for a stencil with one dominant bin
    // (always: mixed mode's numeric stage -- the bins of the irregular rows run beside the ring kernel)
    h->binsForked = always || h->concurrentBins == 1 || (h->concurrentBins == 2 && used >= 8);
    if (!h->binsForked) return BHS_SUCCESS;
    BHS_HIP(hipEventRecord(h->evFork, h->stream));
    for (int i = 0; i < bhs_handle::kBinStreams; ++i) BHS_HIP(hipStreamWaitEvent(h->binStream[i], h->evFork, 0));
    return BHS_SUCCESS;
}

void bin_stream(bhs_handle* h, int bin)
{
    h->ticketSlot = S_TICKETS + bin;
    h->ls = h->binsForked ? h->binStream[bin % bhs_handle::kBinStreams] : (h->besideStream ? h->besideStream : h->stream);
}

int join_bins(bhs_handle* h)
{
    h->ls = h->stream;                    // (the "beside" mode's side stream is joined by its caller)
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
    bool noUpperBound = false, symDirect = false, laneFirst = false;
    bool blockSums = false;           // the lane kernel left the entries of every block of 256 rows (laneBlockSums)
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
    // Rows over their column span (bhs_row_span.hip.h): chosen when the hand-over's scans put every row's span inside the
    // bitmap (<= 8192 columns: up to four words per lane); no table can overflow, so -- direct launches allowed -- EVERY such
    // multiply goes without upper-bound pass, round trip and queue.  Not where the lane kernels apply; instead of the
    // compressed symbolic pass where both do.
    h->ps.spanWPL = 0;
    if (h->spanPath && h->spanState >= 0 && h->bSorted && h->forcePath == 0 && laneK == 0 && h->maxRowA <= 64 && !h->specFailed &&
        h->maxTableLog2 >= 15) {
        const long long need = (long long)h->widthA + h->reachL + h->reachR + 1;
        if (need > 0 && need <= 8192) h->ps.spanWPL = need <= 2048 ? 1 : (need <= 4096 ? 2 : 4);
    }
    // hub bin: rows with hubMin products or more are split across workgroups (bhs_hub.hip.h) in both stages
    const int hubMin = (h->hubMin > 0 && h->useSpa && h->forcePath == 0 && h->maxTableLog2 >= 15 && h->n <= (1 << 25))
                           ? h->hubMin : 0;
    BinSpec symSpec = make_spec(kSymCfg, kNumSymBins, h->maxTableLog2, h->symLoadPct, h->forcePath == 0, laneK, hubMin);
    BinSpec numSpec = make_spec(kNumCfg, kNumNumBins, std::min(h->maxTableLog2, 13), h->numLoadPct, h->forcePath == 0,
                                (h->laneNumeric == 1 || (h->laneNumeric == 2 && laneK <= 8)) ? laneK : 0, hubMin);
    if (h->laneRows != 2) {                                      // (lane_rows = 2: the lane kernels wherever they can run, for the tests)
        symSpec.laneCost = kLaneCost;
        numSpec.laneMax = std::min(numSpec.laneMax, kLaneNumMax);
    }
    BHS_HIP(hipMemsetAsync(small, 0, sizeof(int) * S_ZERO_END, h->stream));
    EventPair* ep;
    h->cmpActive = false;
    // (the undecided first multiply on a data set only measures the ratio: bins and symbolic pass stay plain)
    const bool cmpRun = h->compressB && h->bSorted && h->forcePath == 0 && h->maxTableLog2 >= 15 && h->ps.spanWPL == 0 &&
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
                           (h->laneRows == 2 || (long long)h->maxRowA * h->maxRowB * h->maxRowB <= kLaneCost) && !h->specFailed;
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
    if (h->ps.spanWPL && !laneFirst && h->waveFirst && h->directBins && !wfBin) wfBin = 2;
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
    // the queues are filled while the host waits for the bin counts (their starts: k_bin_starts, the host's own sum below)
    const bool earlyFill = h->earlyFill != 0;
    auto fill_sym = [&]() -> int {
        long long grid = std::min<long long>(((long long)m + kFillTile - 1) / kFillTile, (long long)h->numCU * 8);
        BHS_TRY(timed_begin(h, "fill_queues", &ep));
        hipLaunchKernelGGL(k_fill_queues<false>, dim3((unsigned)grid), dim3(256), 0, h->stream, m,
                           symKeys, h->dAp, (const int*)h->ub.p, (const int*)(small + S_SYM_START),
                           small + S_SYM_CURSOR, (int4*)h->queue.p, symSpec,
                           (unsigned long long*)(small + S_SYM_SUMS));
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        return BHS_SUCCESS;
    };
    if (earlyFill) {
        hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(64), 0, h->stream, (const int*)(small + S_SYM_COUNT), small + S_SYM_START);
        BHS_TRY(fill_sym());
    }
    BHS_HIP(hipMemcpyAsync(hs, small, sizeof(int) * S_SMALL_INTS, hipMemcpyDeviceToHost, h->stream));
    BHS_TRY(wait_stream(h));
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
    if (!symDirect && !earlyFill) {
    memcpy(hs + S_SMALL_INTS, symStart, sizeof(int) * kMaxBins);        // pinned staging: a truly asynchronous H2D
    BHS_HIP(hipMemcpyAsync(small + S_SYM_START, hs + S_SMALL_INTS, sizeof(int) * kMaxBins, hipMemcpyHostToDevice, h->stream));
    BHS_TRY(fill_sym());
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
        int* blockSums = nullptr;                          // (lane-first: the numeric kernel may make rowPtrC from these, pipeline_symbolic)
        if (laneFirst && h->laneFromCounts && (long long)m <= 256LL * kLaneFromCountsBlocks) {
            // (the blocks' sums, then -- 8-byte aligned -- their exclusive scan and the total)
            BHS_TRY(ensure(h, h->laneBlockSums, sizeof(int) * (size_t)kLaneFromCountsBlocks + sizeof(long long) * ((size_t)kLaneFromCountsBlocks + 1)));
            blockSums = (int*)h->laneBlockSums.p;
        }
        BHS_TRY(launch_row_lane<false>(h, laneK, symQueue ? symQueue + symStart[kLaneBin] : nullptr, symCount[kLaneBin], (int*)h->Cp.p,
                                       laneFirst ? (int*)h->ub.p : nullptr,
                                       laneFirst ? (unsigned long long*)(small + S_CT_SLOTS) : nullptr, nullptr, blockSums));
        out.blockSums = blockSums != nullptr;
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
        // (neighbouring bins that all run the LDS-bitmap kernel go as ONE queue, taken from its end: as in the numeric stage)
        int lo = b, rows = symCount[b];
        if (symQueue && h->mergeBitmapBins && bin_takes_lds_bitmap<false>(h, kSymCfg[b]))
            while (lo - 1 >= 2 && bin_takes_lds_bitmap<false>(h, kSymCfg[lo - 1])) { --lo; rows += symCount[lo]; }
        bin_stream(h, b);
        BHS_TRY(timed_begin(h, kSymNames[b], &ep));
        int rc = dispatch_bin<false>(h, kSymCfg[b], symQueue ? symQueue + symStart[lo] : nullptr, rows, (int*)h->Cp.p, lo < b);
        if (rc) { h->ls = h->stream; return rc; }
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += rows;
        for (int bb = lo; bb <= b; ++bb) { if (symCount[bb]) symStat[bb] = ep->stat; symCount[bb] = 0; }
    }
    BHS_TRY(join_bins(h));
    BHS_HIP(hipEventRecord(h->ev[2], h->stream));

    out.noUpperBound = noUpperBound;
    out.laneFirst = laneFirst;
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
// rows with hubMin products or more are split across workgroups (bhs_hub.hip.h) in both stages
int hub_min_products(const bhs_handle* h)
{
    return (h->hubMin > 0 && h->useSpa && h->forcePath == 0 && h->maxTableLog2 >= 15 && h->n <= (1 << 25)) ? h->hubMin : 0;
}

// mixed: the flow of bhs_class_mix.hip.h -- B's single-row classes are pruned before A is classified, classes of a few rows
// are not worked out, the irregular rows of A are listed and get their product counts (symSpec: their symbolic bins)
int symbolic_class(bhs_handle* h, bool mixed, const BinSpec& symSpec)
{
    const int m = h->m, k = h->k;
    const int clsA = cls_row_a(h), clsB = cls_row_b(h);           // the longest rows the classifier is sized for
    int* small = (int*)h->small.p;
    EventPair* ep;
    BHS_TRY(ensure(h, h->classB, sizeof(int) * (size_t)std::max(k, 1)));
    BHS_TRY(ensure(h, h->classC, sizeof(int) * (size_t)std::max(m, 1)));
    BHS_TRY(ensure(h, h->classTab, sizeof(unsigned long long) * 2 * kClassSlots));
    // (one slot more than the table has: kClassDummy, the class of the rows without one in mixed mode -- all zeros, written by nobody)
    BHS_TRY(ensure(h, h->classInfo, sizeof(int4) * (kClassSlots + 1), true));
    BHS_TRY(ensure(h, h->classMap, sizeof(unsigned) * (size_t)kClassSlots * kClassMaxP));
    BHS_TRY(ensure(h, h->classMapA, sizeof(unsigned) * (size_t)kClassSlots * kClassMaxP));
    BHS_TRY(ensure(h, h->classRing, sizeof(unsigned) * (size_t)(kClassSlots + 1) * kClassRingStride, true));
    BHS_TRY(ensure(h, h->classRel, sizeof(int) * (size_t)kClassSlots * kClassMaxNnz));
    BHS_TRY(ensure(h, h->classLane, sizeof(int) * (size_t)(kClassSlots + 1) * kClassLaneInts, true));
    BHS_TRY(ensure(h, h->classHeads, sizeof(int) * ((size_t)std::max(std::max(m, k), 1) + (size_t)kClassHeadSegs * (kClassHeadsBlock / 64) * kClassHeadPiece)));
    // classes beyond the register kernels' tables are possible: their lists and the big numeric kernel (bhs_class_big.hip.h)
    const bool bigPossible = clsA > kClassMaxRow || clsB > kClassMaxRow || (long long)clsA * clsB > kClassMaxP;
    if (mixed) {
        BHS_TRY(ensure(h, h->mixList, sizeof(int) * (size_t)std::max(m, 1)));
        BHS_TRY(ensure(h, h->classCount, sizeof(int) * 2 * kClassSlots));
        BHS_HIP(hipMemsetAsync(h->classCount.p, 0, sizeof(int) * 2 * kClassSlots, h->stream));
    }
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
    const int GB = clsB > kClassMaxRow ? 64 : pow2_at_least(h->avgRowB / h->classPerLane, 4, 64);
    const int GA = clsA > kClassMaxRow ? 64 : pow2_at_least(h->avgRowA / h->classPerLane, 4, 64);
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
                // a wave's piece: 512 rows where that still leaves every CU 16 waves (one row in 512 instead of one in 256
                // goes through the class table for being a piece's first: classify_rows 0.269 -> 0.258 ms on poisson27pt
                // 128^3; 1024: 0.355), the old kernels' piece otherwise
                int piece = class_head_piece(GG);
                if (GG < 32 && (long long)n >= 512LL * 16 * h->numCU) piece = 512;
                if constexpr (GG * E <= 32) {
                    // a lane per row, the rows' column indices through a tile in LDS (bhs_class_tile.hip.h)
                    if (h->classTile && period == 1) {
                        // as many waves as the device has slots for them (16 per CU), ONE piece each, in whole steps of 63 new rows: 2 M
                        // rows in pieces of 504 were 4161 waves for 4096 slots -- a second round for 65 of them (classify_rows 0.227 ->
                        // 0.176 ms on poisson27pt 128^3).  The kernel works the piece out itself: with a row range (A a row block of a
                        // larger product: only the rows of B that A points at) the rows in question are known on the device only.
                        const long long slots = (long long)h->numCU * 16;
                        // (at most 4096 steps per piece: a row's place in its piece has 18 bits)
                        const long long wavesT = std::max<long long>(std::max<long long>(1, ((long long)n + 63 * 4096 - 1) / (63 * 4096)), std::min<long long>(slots, ((long long)n + 125) / 126));
                        long long blocksT = (wavesT + kClassTileBlock / 64 - 1) / (kClassTileBlock / 64);
                        int pieceT = 0;
                        if (h->classTilePiece > 0) {
                            pieceT = std::max(63, h->classTilePiece / 63 * 63);
                            const long long perBlockT = (long long)(kClassTileBlock / 64) * pieceT;
                            blocksT = std::max<long long>(1, ((long long)n + perBlockT - 1) / perBlockT);
                        }
                        hipLaunchKernelGGL((k_class_tile<IS_A, GG, E>), dim3((unsigned)blocksT), dim3(kClassTileBlock), 0,
                                           h->stream, n, Rp, Rj, cb, out, tab, cstats, (long long)(IS_A ? h->nnzA : h->nnzB), pieceT, rng);
                        return (int)BHS_SUCCESS;
                    }
                }
                const long long perBlock = (long long)(kClassHeadsBlock / 64) * piece;
                hipLaunchKernelGGL((k_class_fused<IS_A, GG, E>), dim3((unsigned)std::max<long long>(1, ((long long)n + perBlock - 1) / perBlock)), dim3(kClassHeadsBlock), 0,
                                   h->stream, n, Rp, Rj, cb, out, tab, cstats, (long long)(IS_A ? h->nnzA : h->nnzB), piece, rng, period);
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
    int* countB = (int*)h->classCount.p;
    int* countA = mixed ? countB + kClassSlots : nullptr;
    const unsigned histGridB = (unsigned)std::max<long long>(1, std::min<long long>(((long long)k + kMixHistBlock - 1) / kMixHistBlock, (long long)h->numCU * 2));
    const unsigned histGridA = (unsigned)std::max<long long>(1, std::min<long long>(((long long)m + kMixHistBlock - 1) / kMixHistBlock, (long long)h->numCU * 2));
    int rc = classify(template_int<0>{}, k, h->dBp, h->dBj, (const int*)nullptr, tabB, (int*)h->classB.p, bRange, nHeadsB, GB, clsB, periodB);
    if (rc == BHS_SUCCESS && mixed) {
        // a row of B that is the only one of its kind (a perturbed row) loses its class: the rows of A that point at it
        // would each claim a class of their own in A's table
        hipLaunchKernelGGL(k_mix_class_hist, dim3(histGridB), dim3(kMixHistBlock), 0, h->stream, k, (const int*)h->classB.p, countB, bRange);
        hipLaunchKernelGGL(k_mix_prune, dim3(propGrid), dim3(256), 0, h->stream, k, (int*)h->classB.p, (const int*)countB, bRange);
    }
    if (rc == BHS_SUCCESS)
        rc = classify(template_int<1>{}, m, h->dAp, h->dAj, (const int*)h->classB.p, tabA, (int*)h->classC.p, (const int*)nullptr, nHeadsA, GA, clsA, periodA);
    if (rc != BHS_SUCCESS) return rc;
    if (mixed) hipLaunchKernelGGL(k_mix_class_hist, dim3(histGridA), dim3(kMixHistBlock), 0, h->stream, m, (const int*)h->classC.p, countA, (const int*)nullptr);
    BHS_HIP(hipGetLastError());
    BHS_TRY(timed_end(h, ep));
    h->stats[ep->stat].launches += h->classHeadsOn == 1 ? 6 : 2;
    h->stats[ep->stat].rows += (int64_t)m + k;
    BHS_HIP(hipEventRecord(h->ev[1], h->stream));
    BHS_TRY(timed_begin(h, "class_patterns", &ep));
    hipLaunchKernelGGL(k_class_patterns, dim3(kClassSlots), dim3(256), 0, h->stream, (const unsigned long long*)tabA,
                       h->dAp, h->dAj, h->dBp, h->dBj, (int4*)h->classInfo.p,
                       (unsigned*)h->classMap.p, (unsigned*)h->classMapA.p, (int*)h->classRel.p, (int*)h->classLane.p,
                       (unsigned*)h->classRing.p, cstats, (const int*)countA);
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
    if (mixed) {
        // the irregular rows: listed, their product counts and symbolic bins
        BHS_TRY(timed_begin(h, "irregular_rows", &ep));
        const unsigned gc = (unsigned)std::max<long long>(1, std::min<long long>(((long long)m + kMixCollectTile - 1) / kMixCollectTile, (long long)h->numCU * 8));
        hipLaunchKernelGGL(k_mix_collect, dim3(gc), dim3(256), 0, h->stream, m, (int*)h->classC.p, (const int4*)h->classInfo.p,
                           (int*)h->mixList.p, small + S_MIX_COUNT);
        hipLaunchKernelGGL(k_mix_upper_bound, dim3((unsigned)(h->numCU * 4)), dim3(256), 0, h->stream, (const int*)h->mixList.p,
                           (const int*)(small + S_MIX_COUNT), h->dAp, h->dAj, h->dBp, (int*)h->ub.p, (int*)h->Cp.p,
                           (unsigned long long*)(small + S_TOTAL_CT), small + S_SYM_COUNT, symSpec, small + S_MIX_SYM2, coarse_spec(symSpec, kCoarseSym));
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches += 2;
    }
    BHS_HIP(hipEventRecord(h->ev[2], h->stream));
    return BHS_SUCCESS;
}

// restart: the same multiply starting over on another path (a refuted speculation, rows without a class): the
// timers and kernel statistics of the abandoned attempt stay in -- it ran inside this multiply.
int stage_rowptr_and_open(bhs_handle* h);

// May this multiply's numeric kernel go out on the figures of the last one?  Sets h->ps's class fields from them if so.
bool class_spec_try(bhs_handle* h)
{
    const bhs_handle::ClassSpec& sp = h->classSpec;
    if (!h->specNumeric || !sp.valid || h->lazyOut || h->classNumeric < 2) return false;
    const int* cs = sp.cs;
    if (cs[CS_FLAGS] || cs[CS_CLASSES] == 0 || cs[CS_BIGCOUNT]) return false;
    if (h->classHeadsOn && h->classPath != 2 && (long long)cs[CS_HEADS] * 4 > (long long)h->m) return false;
    const size_t need = (size_t)std::max<long long>(sp.nnzC, 1);
    if (h->extCj ? sp.nnzC > h->extCap : (!h->Cj.p || !h->Cx.p || h->Cj.cap < need * sizeof(int) || h->Cx.cap < need * sizeof(value_t))) return false;
    h->ps.useClass = true;
    h->ps.classMaxP = cs[CS_MAXP];
    h->ps.classMaxNnz = cs[CS_MAXNNZ];
    h->ps.classMaxNA = cs[CS_MAXNA];
    h->ps.classMaxLB = cs[CS_MAXLB];
    h->ps.classMaxRing = cs[CS_MAXRING];
    h->ps.classMaxRing2 = std::max(cs[CS_RINGFULL], cs[CS_RINGONE]);
    h->ps.classMaxSlab = cs[CS_MAXSLAB];
    h->ps.classBig = 0;
    h->ps.classBigMaxP = cs[CS_BIGMAXP];
    if (!class_ring2_fits(h)) { h->ps.useClass = false; return false; }
    return true;
}

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
        BHS_TRY(wait_stream(h));
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
                          cls_row_a(h) <= kClassMaxRowBig && cls_row_b(h) <= kClassMaxRowBig &&
                          (h->classPath == 2 || (h->avgRowA * h->avgRowB >= (double)h->classMinProducts &&
                                                 // ... and enough of them: every block of the classifier meets every class once.
                                                 // (Round 6, class kernels against general pipeline: 16 x 16 entries a row, 25 k
                                                 // rows -- 5.8 M products -- 0.182 / 0.179 ms, 50 k rows 0.192 / 0.266, 100 k rows
                                                 // 0.209 / 0.435; 27 x 8 entries, 50 k rows 0.192 / 0.240; poisson27pt 40^3, 43 M
                                                 // products, 0.219 / 0.249; poisson9pt 512^2, 21 M, 0.192 / 0.194.  Round 3's
                                                 // kernels had put the line at 6e7.)
                                                 (double)h->m * h->avgRowA * h->avgRowB >= 1e7));
    // Mixed mode (bhs_class_mix.hip.h): this data set's last multiply met rows without a class -- they go through the general
    // pipeline's kernels, everything else stays on the class kernels.  mixRows: how many this multiply found.
    const bool mixedFlow = useClass && h->mixOn && h->classMixed;
    int mixRows = 0;
    int mixSymCount[kMaxBins], mixSymStart[kMaxBins + 1];
    unsigned long long mixProducts = 0;
    if (useClass) {
        const int hubMin = mixedFlow ? hub_min_products(h) : 0;
        BinSpec symSpec = make_spec(kSymCfg, kNumSymBins, h->maxTableLog2, h->symLoadPct, true, 0, hubMin);
        BHS_TRY(symbolic_class(h, mixedFlow, symSpec));
        sc.noUpperBound = true;                 // (no ub[] either: the numeric bins are never built)
        sc.numSpec = make_spec(kNumCfg, kNumNumBins, std::min(h->maxTableLog2, 13), h->numLoadPct, true, 0, hubMin);
        if (mixedFlow) {
            // what the classes look like and how many rows have none: the first of the mixed flow's two read-backs
            BHS_HIP(hipMemcpyAsync(hs, small, sizeof(int) * S_SMALL_INTS, hipMemcpyDeviceToHost, h->stream));
            BHS_TRY(wait_stream(h));
            const int* cs = hs + S_CT_SLOTS;
            mixRows = hs[S_MIX_COUNT];
            memcpy(&mixProducts, hs + S_TOTAL_CT, 8);
            auto to_general = [&](const char* why) {
                h->classState = -1;
                if (h->verbose > 1) printf("  [row classes, mixed: %s (%d of %d rows without a class, %d classes, %d rows start a stretch): general pipeline]\n", why, mixRows, m, cs[CS_CLASSES], cs[CS_HEADS]);
                return pipeline_symbolic(h, true);
            };
            if (hs[S_ERR]) return BHS_ERR_INTERNAL;
            if (cs[CS_CLASSES] == 0) return to_general("no class");
            if (mixRows > 0) {
                if ((long long)mixRows * 100 > (long long)h->mixMaxPct * m) return to_general("too many rows without a class");
                h->ps.classMaxNnz = cs[CS_MAXNNZ];
                h->ps.classMaxNA = cs[CS_MAXNA];
                h->ps.classMaxRing2 = std::max(cs[CS_RINGFULL], cs[CS_RINGONE]);
                // (the kernels that pass an irregular row by: the ring kernel, and k_class_numeric_big for block-structured grids)
                if (!cs[CS_BIGCOUNT] && (h->classNumeric < 2 || !class_ring2_fits(h))) return to_general("classes beyond the ring kernel");
            }
            // (an irregular row is a head, and so is the row behind it: they are not what the verdict below is about)
            if (h->classHeadsOn && h->classPath != 2 && !cs[CS_BIGCOUNT] && ((long long)cs[CS_HEADS] - 2LL * mixRows) * 4 > (long long)m)
                return to_general("rows classify each for itself");
            if (mixRows == 0) { h->classMixed = 0; h->mixProbed = true; }   // (every row has a class worth its pattern: the clean flow from the next multiply on)
            if (h->verbose > 1) printf("  [row classes, mixed: %d of %d rows without a class -> the general pipeline's kernels; %d classes]\n", mixRows, m, cs[CS_CLASSES]);
        }
        if (mixRows > 0) {
            // their symbolic pass, bin by bin: exact counts into Cp[row] (the general pipeline's stage 2 on the listed rows)
            // a handful of rows (fewer than one in sixteen): the ladder with fewer steps -- a launch costs more than a table
            // that is too large for a few hundred rows
            const bool coarse = (long long)mixRows * 16 <= (long long)m && h->maxTableLog2 >= 15;
            if (coarse) {
                symSpec = coarse_spec(symSpec, kCoarseSym);
                sc.numSpec = coarse_spec(sc.numSpec, kCoarseNum);
            }
            mixSymStart[0] = 0;
            for (int b = 0; b < kMaxBins; ++b) {
                mixSymCount[b] = hs[(coarse ? S_MIX_SYM2 : S_SYM_COUNT) + b];
                mixSymStart[b + 1] = mixSymStart[b] + (b == 0 ? 0 : mixSymCount[b]);
            }
            h->nnzCt = (long long)mixProducts;                    // (launch_hub sizes its item list by it)
            h->cmpActive = false;
            const long long gridF = std::max<long long>(1, std::min<long long>(((long long)mixRows + 255) / 256, (long long)h->numCU * 8));
            BHS_TRY(timed_begin(h, "fill_queues", &ep));
            hipLaunchKernelGGL(k_mix_fill<false>, dim3((unsigned)gridF), dim3(256), 0, h->stream, (const int*)h->mixList.p, (const int*)(small + S_MIX_COUNT),
                               0, 0x7fffffff, (const int*)h->ub.p, h->dAp, (const int*)h->ub.p, (const int*)(small + (coarse ? S_MIX_SYM2 : S_SYM_COUNT)), small + S_SYM_CURSOR,
                               (int4*)h->queue.p, symSpec, (unsigned long long*)(small + S_SYM_SUMS));
            BHS_HIP(hipGetLastError());
            BHS_TRY(timed_end(h, ep));
            h->stats[ep->stat].launches++;
            const int4* symQueue = (const int4*)h->queue.p;
            BHS_TRY(fork_bins(h, mixSymCount, kNumSymBins, h->mixFork != 0));
            if (mixSymCount[kHubBin]) {
                bin_stream(h, kHubBin);
                BHS_TRY(timed_begin(h, "symbolic_hub_rows", &ep));
                int rc = launch_hub<false>(h, symQueue + mixSymStart[kHubBin], mixSymCount[kHubBin], (int*)h->Cp.p);
                if (rc) { h->ls = h->stream; return rc; }
                BHS_TRY(timed_end(h, ep));
                h->stats[ep->stat].launches++;
                h->stats[ep->stat].rows += mixSymCount[kHubBin];
                h->ps.symStat[kHubBin] = ep->stat;
            }
            for (int i = 1; i < kNumSymBins; ++i) {
                const int b = kNumSymBins - i;
                if (!mixSymCount[b]) continue;
                bin_stream(h, b);
                BHS_TRY(timed_begin(h, kSymNames[b], &ep));
                int rc = dispatch_bin<false>(h, kSymCfg[b], symQueue + mixSymStart[b], mixSymCount[b], (int*)h->Cp.p);
                if (rc) { h->ls = h->stream; return rc; }
                BHS_TRY(timed_end(h, ep));
                h->stats[ep->stat].launches++;
                h->stats[ep->stat].rows += mixSymCount[b];
                h->ps.symStat[b] = ep->stat;
            }
            BHS_TRY(join_bins(h));
        }
    } else {
        BHS_TRY(symbolic_general(h, sc));
    }
    const bool noUpperBound = sc.noUpperBound, symDirect = sc.symDirect;
    const int laneK = sc.laneK;
    const BinSpec& numSpec = sc.numSpec;

    // ------------------------------------------------------------ stage 3: scan, allocate C, numeric queues
    // Round 6, a lane-first multiply whose numeric kernel may go out on the last multiply's nnz(C) (laneSpec): no scan kernel and
    // no check kernel either -- the numeric kernel's blocks sum the symbolic kernel's block sums, compare the total with the
    // nnz(C) C was sized for (all of them the same verdict; nothing is written otherwise) and make rowPtrC on the way.  Not for
    // callers that want rowPtrC on the host while the numeric kernel runs.
    h->ps.laneFirst = !useClass && sc.laneFirst;
    if (!useClass && !restart && sc.laneFirst && sc.blockSums && !h->wantHostRowPtr && h->specNumeric && h->laneSpec.valid &&
        h->laneSpec.laneK == sc.laneK && !h->lazyOut && h->directBins && (h->laneNumeric == 1 || (h->laneNumeric == 2 && sc.laneK <= 8))) {
        const long long need = std::max<long long>(h->laneSpec.nnzC, 1);
        const bool room = h->extCj ? h->laneSpec.nnzC <= h->extCap
                                   : (h->Cj.p && h->Cx.p && h->Cj.cap >= (size_t)need * sizeof(int) && h->Cx.cap >= (size_t)need * sizeof(value_t));
        if (room) {
            h->ps.specLaunched = true;
            h->ps.specLane = true;
            h->ps.fromCounts = true;
            h->specLaunches++;
            h->nnzC = h->laneSpec.nnzC;
            h->nnzCt = h->laneSpec.nnzCt;                  // (pipeline_finish puts this multiply's own count here)
            h->ps.noUpperBound = sc.noUpperBound;
            h->ps.symDirect = sc.symDirect;
            h->ps.laneK = sc.laneK;
            h->ps.numSpec = sc.numSpec;
            BHS_HIP(hipEventRecord(h->ev[3], h->stream));
            return stage_rowptr_and_open(h);
        }
    }
    BHS_TRY(timed_begin(h, "scan_rowptr", &ep));
    if (useClass) {
        // one pass: every row's count from its class, scanned with look-back over the tiles before (k_class_scan)
        const int nTiles = (m + kClassScanTile - 1) / kClassScanTile;
        // (blockSum holds the tile words, cleared by k_class_reset)
        hipLaunchKernelGGL(k_class_scan, dim3((unsigned)nTiles), dim3(kClassScanBlock), 0, h->stream, m, (const int*)h->classC.p,
                           (const int4*)h->classInfo.p, (int*)h->Cp.p, (unsigned long long*)h->blockSum.p,
                           (long long*)(small + S_TOTAL_C), small + S_CT_SLOTS, mixRows > 0, h->dAp, (const int*)h->ub.p, sc.numSpec, small + S_NUM_COUNT);
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches += 1;
        if (mixRows > 0) {
            // the irregular rows' numeric bins and queue (rowPtrC is final), on the device while the host waits for nnz(C)
            const BinSpec& ns = sc.numSpec;
            const long long gridF = std::max<long long>(1, std::min<long long>(((long long)mixRows + 255) / 256, (long long)h->numCU * 8));
            BHS_TRY(timed_begin(h, "fill_queues", &ep));
            hipLaunchKernelGGL(k_mix_fill<true>, dim3((unsigned)gridF), dim3(256), 0, h->stream, (const int*)h->mixList.p, (const int*)(small + S_MIX_COUNT),
                               0, 0x7fffffff, (const int*)h->Cp.p, h->dAp, (const int*)h->ub.p, (const int*)(small + S_NUM_COUNT), small + S_NUM_CURSOR,
                               (int4*)h->queue.p, ns, (unsigned long long*)(small + S_NUM_SUMS));
            BHS_HIP(hipGetLastError());
            BHS_TRY(timed_end(h, ep));
            h->stats[ep->stat].launches++;
        }
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
    // ... the numeric queues likewise, where the symbolic stage went by queues (then the numeric stage will): k_scan_onepass
    // has left the numeric bins' counts
    h->ps.numQueueFilled = false;
    // (not where this data set's last multiply ran its numeric stage straight from the row pointers: a hint, it costs or saves a launch)
    if (!useClass && h->earlyFill && h->scanOnePass && !sc.noUpperBound && !sc.symDirect && h->numDirectHint != 1) {
        hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(64), 0, h->stream, (const int*)(small + S_NUM_COUNT), small + S_NUM_START);
        long long grid = std::min<long long>(((long long)m + kFillTile - 1) / kFillTile, (long long)h->numCU * 8);
        BHS_TRY(timed_begin(h, "fill_queues", &ep));
        hipLaunchKernelGGL(k_fill_queues<true>, dim3((unsigned)grid), dim3(256), 0, h->stream, m,
                           (const int*)h->Cp.p, h->dAp, (const int*)h->ub.p, (const int*)(small + S_NUM_START),
                           small + S_NUM_CURSOR, (int4*)h->queue.p, numSpec,
                           (unsigned long long*)(small + S_NUM_SUMS));
        BHS_HIP(hipGetLastError());
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->ps.numQueueFilled = true;
    }
    // The classes' figures of the data set's last multiply stand in for this one's (bhs_class.hip.h, k_class_spec_check): no
    // round trip to the host between the scan and the numeric kernel.  Only for a whole multiply on the ring kernel whose C
    // fits the arrays at hand; pipeline_finish sees the device's verdict.
    if (useClass && !restart && !mixedFlow && class_spec_try(h)) {
        ClassSpecKey key;
        memcpy(key.cs, h->classSpec.cs, sizeof(key.cs));
        key.nnzC = h->classSpec.nnzC;
        hipLaunchKernelGGL(k_class_spec_check, dim3(1), dim3(64), 0, h->stream, key, (const int*)(small + S_CT_SLOTS),
                           (const long long*)(small + S_TOTAL_C), (const int*)(small + S_ERR), small + S_SPEC);
        BHS_HIP(hipGetLastError());
        h->ps.specLaunched = true;
        h->specLaunches++;
        h->nnzC = h->classSpec.nnzC;
        h->nnzCt = h->classSpec.nnzCt;                 // (pipeline_finish puts this multiply's own count here)
        h->ps.noUpperBound = noUpperBound;
        h->ps.symDirect = symDirect;
        h->ps.laneK = laneK;
        h->ps.numSpec = numSpec;
        BHS_HIP(hipEventRecord(h->ev[3], h->stream));
        return stage_rowptr_and_open(h);
    }
    // ... and a lane-first multiply likewise (round 6): the last multiply's nnz(C) and "every row in the lane bin" stand in,
    // k_lane_spec_check compares on the device, pipeline_finish sees the verdict
    h->ps.laneFirst = !useClass && sc.laneFirst;
    if (!useClass && !restart && sc.laneFirst && h->specNumeric && h->laneSpec.valid && h->laneSpec.laneK == sc.laneK && !h->lazyOut &&
        h->directBins && h->scanOnePass && (h->laneNumeric == 1 || (h->laneNumeric == 2 && sc.laneK <= 8))) {
        const long long need = std::max<long long>(h->laneSpec.nnzC, 1);
        const bool room = h->extCj ? h->laneSpec.nnzC <= h->extCap
                                   : (h->Cj.p && h->Cx.p && h->Cj.cap >= (size_t)need * sizeof(int) && h->Cx.cap >= (size_t)need * sizeof(value_t));
        if (room) {
            hipLaunchKernelGGL(k_lane_spec_check, dim3(1), dim3(64), 0, h->stream, h->laneSpec.nnzC, m, (const long long*)(small + S_TOTAL_C),
                               (const int*)(small + S_ERR), (const int*)(small + S_NUM_COUNT), small + S_SPEC);
            BHS_HIP(hipGetLastError());
            h->ps.specLaunched = true;
            h->ps.specLane = true;
            h->specLaunches++;
            h->nnzC = h->laneSpec.nnzC;
            h->nnzCt = h->laneSpec.nnzCt;                  // (pipeline_finish puts this multiply's own count here)
            h->ps.noUpperBound = noUpperBound;
            h->ps.symDirect = symDirect;
            h->ps.laneK = laneK;
            h->ps.numSpec = numSpec;
            BHS_HIP(hipEventRecord(h->ev[3], h->stream));
            return stage_rowptr_and_open(h);
        }
    }
    BHS_HIP(hipMemcpyAsync(hs, small, sizeof(int) * S_SMALL_INTS, hipMemcpyDeviceToHost, h->stream));
    BHS_TRY(wait_stream(h));
    if (useClass) {
        const int* cs = hs + S_CT_SLOTS;
        // (... or a great many classes: single rows that found room in the table, each with a pattern of its own to work out --
        // the mixed flow counts the rows of every class and leaves the classes of a few rows out; asked once per data set)
        bool manyClasses = false;
        if (!mixedFlow && !cs[CS_FLAGS] && cs[CS_CLASSES] > kClassManyClasses && !h->mixProbed && !cs[CS_BIGCOUNT] && h->classNumeric >= 2) {
            h->ps.classMaxNnz = cs[CS_MAXNNZ];                    // (only where the mixed flow's ring kernel can take the classes)
            h->ps.classMaxNA = cs[CS_MAXNA];
            h->ps.classMaxRing2 = std::max(cs[CS_RINGFULL], cs[CS_RINGONE]);
            manyClasses = class_ring2_fits(h);
        }
        if (!mixedFlow && (cs[CS_FLAGS] || manyClasses) && cs[CS_CLASSES] > 0 && h->mixOn) {
            // a row without a class, or a class beyond the tables: the same multiply once more, with those rows on the general
            // pipeline's kernels (mixed mode, bhs_class_mix.hip.h) -- and this data set's next multiplies that way from the start
            h->classMixed = 1;
            if (h->verbose > 1) printf("  [row classes: flags %d, %d classes: the multiply again in mixed mode]\n", cs[CS_FLAGS], cs[CS_CLASSES]);
            return pipeline_symbolic(h, true);
        }
        if (!mixedFlow && (cs[CS_FLAGS] || cs[CS_CLASSES] == 0)) {
            // ... with mixed mode off, or no class at all: this data set is for the general pipeline
            h->classState = -1;
            if (h->verbose > 1) printf("  [row classes: flags %d, %d classes: general pipeline]\n", cs[CS_FLAGS], cs[CS_CLASSES]);
            return pipeline_symbolic(h, true);
        }
        // The class kernels take rows in stretches -- consecutive rows of one class; a matrix whose rows classify but
        // each for itself (block-diagonal with dense blocks: every row of a block has its own relative pattern) makes
        // them change class every row: 3.4 ms against 1.9 ms on the general pipeline for 2^20 rows in blocks of 4..32.
        // More than a quarter of the rows through the table: this data set goes to the general pipeline (class_path = 2
        // insists on the classes).
        if (!mixedFlow && h->classHeadsOn && h->classPath != 2 && !cs[CS_BIGCOUNT] && (long long)cs[CS_HEADS] * 4 > (long long)m) {
            h->classState = -1;
            if (h->verbose > 1) printf("  [row classes: %d of %d rows start a stretch: general pipeline]\n", cs[CS_HEADS], m);
            return pipeline_symbolic(h, true);
        }
        unsigned long long t = 0, v;
        for (int i = 0; i < kClassSumSlots; ++i) { memcpy(&v, cs + CS_SUMS + 2 * i, 8); t += v; }
        h->nnzCt = (long long)t + (long long)mixProducts;         // (the rows with a class, the rows without)
        h->ps.useClass = true;
        h->ps.mixed = mixRows > 0;
        h->ps.mixRows = mixRows;
        h->ps.mixNumFilled = mixRows > 0;
        h->ps.mixProducts = (long long)mixProducts;
        for (int b = 0; b < kMaxBins; ++b) { h->ps.mixSymCount[b] = mixRows > 0 ? mixSymCount[b] : 0; h->ps.mixNumCount[b] = mixRows > 0 ? hs[S_NUM_COUNT + b] : 0; }
        h->ps.classMaxP = cs[CS_MAXP];
        h->ps.classMaxNnz = cs[CS_MAXNNZ];
        h->ps.classMaxNA = cs[CS_MAXNA];
        h->ps.classMaxLB = cs[CS_MAXLB];
        h->ps.classMaxRing = cs[CS_MAXRING];
        h->ps.classMaxRing2 = std::max(cs[CS_RINGFULL], cs[CS_RINGONE]);
        h->ps.classMaxSlab = cs[CS_MAXSLAB];
        h->ps.classBig = cs[CS_BIGCOUNT];
        h->ps.classBigMaxP = cs[CS_BIGMAXP];
        memcpy(h->classSpec.cs, cs, sizeof(h->classSpec.cs));          // (valid once nnzC is known, below)
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
        if (h->ps.spanWPL > 0 && h->spanState >= 0) {             // (a row beyond the span bitmap: the hash kernels from here on)
            h->spanState = -1;
            if (h->verbose > 1) printf("  [a row's column span is beyond the bitmap: hash kernels]\n");
            return pipeline_symbolic(h, true);
        }
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
    if (useClass) { h->classSpec.nnzC = nnzC; h->classSpec.nnzCt = h->nnzCt; h->classSpec.valid = !mixedFlow; }
    BHS_HIP(hipEventRecord(h->ev[3], h->stream));
    return stage_rowptr_and_open(h);
}

// the end of pipeline_symbolic: rowPtrC on its way to the host where the caller wants it there, the multiply open
int stage_rowptr_and_open(bhs_handle* h)
{
    const int m = h->m;
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
    if (h->ps.specLane) {
        // (a whole multiply: bhs_spgemm_symbolic's lazyOut keeps the two-halves API off this path)
        h->ps.rangesRun++;
        BHS_TRY(timed_begin(h, "numeric_lane", &ep));
        if (h->ps.fromCounts) {
            const int nb = (int)(((long long)h->m + 255) / 256);
            hipLaunchKernelGGL(k_lane_block_prefix, dim3(1), dim3(1024), 0, h->ls, nb, (const int*)h->laneBlockSums.p,
                               reinterpret_cast<long long*>((int*)h->laneBlockSums.p + kLaneFromCountsBlocks));
            BHS_HIP(hipGetLastError());
        }
        if (h->ps.fromCounts)
            BHS_TRY(launch_row_lane<true>(h, laneK, nullptr, h->m, (int*)h->Cp.p, nullptr, nullptr, nullptr, (int*)h->laneBlockSums.p, h->laneSpec.nnzC,
                                          (int*)h->small.p + S_SPEC));
        else
        BHS_TRY(launch_row_lane<true>(h, laneK, nullptr, h->m, (int*)h->Cp.p, nullptr, nullptr, (const int*)h->small.p + S_SPEC));
        BHS_TRY(timed_end(h, ep));
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += h->m;
        numStat[kLaneBin] = ep->stat;
        h->ps.numDirectFull = true;
        return BHS_SUCCESS;
    }
    if (h->ps.useClass && h->ps.mixed) {
        // Mixed mode (bhs_class_mix.hip.h): the ring kernel on the rows with a class -- it skips the others -- and, beside it on
        // the side streams, the general pipeline's numeric kernels on the queue of the rows without
        h->ps.rangesRun++;
        int numCount[kMaxBins], numStart[kMaxBins + 1];
        if (full && h->ps.mixNumFilled) {
            for (int b = 0; b < kMaxBins; ++b) numCount[b] = h->ps.mixNumCount[b];
        } else {
            // the listed rows inside [r0, r1): their bins and queue anew (one small round trip per range, as in the general pipeline)
            int* hr = hs + S_SMALL_INTS + 2 * kMaxBins;
            BHS_HIP(hipMemsetAsync(small + S_NUM_COUNT, 0, sizeof(int) * 3 * kMaxBins, h->stream));
            BHS_HIP(hipMemsetAsync(small + S_NUM_SUMS, 0, sizeof(unsigned long long) * 3 * kMaxBins, h->stream));
            const long long gridH = std::max<long long>(1, std::min<long long>(((long long)h->ps.mixRows + 255) / 256, (long long)h->numCU * 4));
            const long long gridF = std::max<long long>(1, std::min<long long>(((long long)h->ps.mixRows + 255) / 256, (long long)h->numCU * 8));
            hipLaunchKernelGGL(k_bin_hist, dim3((unsigned)gridH), dim3(256), 0, h->stream, h->m, (const int*)h->Cp.p, h->dAp, numSpec, small + S_NUM_COUNT,
                               small + S_MAXCNT, (const int*)h->ub.p, (const int*)h->mixList.p, (const int*)(small + S_MIX_COUNT), r0, r1);
            hipLaunchKernelGGL(k_mix_fill<true>, dim3((unsigned)gridF), dim3(256), 0, h->stream, (const int*)h->mixList.p, (const int*)(small + S_MIX_COUNT),
                               r0, r1, (const int*)h->Cp.p, h->dAp, (const int*)h->ub.p, (const int*)(small + S_NUM_COUNT), small + S_NUM_CURSOR,
                               (int4*)h->queue.p, numSpec, (unsigned long long*)(small + S_NUM_SUMS));
            BHS_HIP(hipGetLastError());
            BHS_HIP(hipMemcpyAsync(hr, small + S_NUM_COUNT, sizeof(int) * kMaxBins, hipMemcpyDeviceToHost, h->stream));
            BHS_TRY(wait_stream(h));
            for (int b = 0; b < kMaxBins; ++b) numCount[b] = hr[b];
            h->ps.mixNumFilled = false;                               // (the queue now holds this range's rows)
        }
        numStart[0] = 0;
        for (int b = 0; b < kMaxBins; ++b) numStart[b + 1] = numStart[b] + (b == 0 ? 0 : numCount[b]);
        const int4* numQueue = (const int4*)h->queue.p;
        h->ps.midRows = h->ps.longRows = 0;
        for (int b = 2; b < kNumNumBins; ++b)
            if (bin_takes_lds_bitmap<true>(h, kNumCfg[b])) (kNumCfg[b].win ? h->ps.longRows : h->ps.midRows) += numCount[b];
        for (int b = 2; b < kNumNumBins; ++b)
            if (numCount[b] && bin_takes_wave_window<true>(h, kNumCfg[b])) BHS_TRY(ensure_b_windows(h));
        int anyBin = 0;
        for (int b = 1; b < kMaxBins; ++b) anyBin += numCount[b] > 0;
        // The bins first, side by side on the side streams, THEN the ring kernel: its 16 waves per CU take all of a CU's LDS and
        // all of its vector registers, and whatever is launched beside it waits for its waves to finish (measured on
        // poisson27pt 128^3 with 55 k irregular rows: their bins, 0.2 ms of work, ended 0.8 ms after the ring kernel beside
        // which they were launched; launched just in front of it, the last of them still did).
        // ... A HANDFUL of irregular rows (their kernels: a few workgroups, each a chain of round trips -- one row of 8100 products
        // takes k_row_block 0.09 ms) go to ONE side stream and the ring kernel starts beside them at once: its workgroups on
        // the few CUs those hold start late, and with its super-runs handed out by the XCDs' counters (ring_dynamic) nobody
        // waits for them.
        BHS_TRY(fork_bins(h, numCount, kNumNumBins, anyBin > 0 && h->mixFork != 0));
        const bool beside = !h->binsForked && anyBin > 0 && h->ps.mixRows <= kMixBesideRows && h->ringDynamic != 0 && !h->ps.classBig;
        if (beside) {                                            // (their kernels FIRST: what the ring kernel has taken it keeps until it ends)
            BHS_HIP(hipEventRecord(h->evFork, h->stream));
            BHS_HIP(hipStreamWaitEvent(h->binStream[0], h->evFork, 0));
            h->besideStream = h->binStream[0];
        }
        if (numCount[kHubBin]) {
            bin_stream(h, kHubBin);
            BHS_TRY(timed_begin(h, "numeric_hub_rows", &ep));
            BHS_TRY(launch_hub<true>(h, numQueue + numStart[kHubBin], numCount[kHubBin], (int*)h->Cp.p));
            BHS_TRY(timed_end(h, ep));
            h->stats[ep->stat].launches++;
            h->stats[ep->stat].rows += numCount[kHubBin];
            numStat[kHubBin] = ep->stat;
        }
        for (int i = 1; i < kNumNumBins; ++i) {
            const int b = kNumNumBins - i;
            if (!numCount[b]) continue;
            bin_stream(h, b);
            BHS_TRY(timed_begin(h, kNumNames[b], &ep));
            BHS_TRY(dispatch_bin<true>(h, kNumCfg[b], numQueue + numStart[b], numCount[b], (int*)h->Cp.p, 0));
            BHS_TRY(timed_end(h, ep));
            h->stats[ep->stat].launches++;
            h->stats[ep->stat].rows += numCount[b];
            numStat[b] = ep->stat;
        }
        BHS_TRY(join_bins(h));
        h->besideStream = nullptr;
        h->ps.ringBeside = beside;
        BHS_TRY(timed_begin(h, "numeric_class", &ep));
        if (h->ps.classBig) BHS_TRY(launch_class_numeric_big(h, r0, r1));
        else BHS_TRY(launch_class_ring(h, r0, r1));
        BHS_TRY(timed_end(h, ep));
        h->ps.ringBeside = false;
        h->stats[ep->stat].launches++;
        h->stats[ep->stat].rows += r1 - r0;
        if (full) { h->stats[ep->stat].products += h->nnzCt - h->ps.mixProducts; h->stats[ep->stat].nnzA_rows += h->nnzA; }
        if (beside) {
            BHS_HIP(hipEventRecord(h->evJoin[0], h->binStream[0]));
            BHS_HIP(hipStreamWaitEvent(h->stream, h->evJoin[0], 0));
        }
        return BHS_SUCCESS;
    }
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
        BHS_TRY(wait_stream(h));
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
        if (nb && maxCnt > 0 && ((double)h->nnzC / std::max(view.m, 1) * 4.0 >= (double)numSpec.upper[nb] || h->ps.spanWPL > 0)) {
            for (int b = 0; b < kMaxBins; ++b) { numCount[b] = 0; numStart[b] = 0; }
            numStart[kMaxBins] = 0;
            numCount[nb] = m;
            numDirect = true;
        }
    }
    if (!numDirect && !(full && h->ps.numQueueFilled)) {
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
    if (full) { h->ps.numDirectFull = numDirect; h->numDirectHint = numDirect ? 1 : 0; }
    // (what the next multiply of this data set may assume: every row through the lane kernels, this nnz(C))
    h->laneSpec.valid = full && h->ps.laneFirst && numDirect && numCount[kLaneBin] == m && !h->lazyOut;
    if (h->laneSpec.valid) { h->laneSpec.laneK = laneK; h->laneSpec.nnzC = h->nnzC; h->laneSpec.nnzCt = h->nnzCt; }
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
    BHS_TRY(wait_stream(h));
    if (h->ps.specLaunched && h->ps.specLane) {
        if (hs[S_SPEC] != 1) return kSpecRefuted;
        unsigned long long t = 0, v;
        for (int i = 0; i < 64; ++i) { memcpy(&v, hs + S_CT_SLOTS + 2 * i, 8); t += v; }
        h->nnzCt = h->laneSpec.nnzCt = (long long)t;
    } else if (h->ps.specLaunched) {
        if (hs[S_SPEC] != 1) return kSpecRefuted;       // (the numeric kernel has written nothing: run_pipeline_impl starts over)
        unsigned long long t = 0, v;
        for (int i = 0; i < kClassSumSlots; ++i) { memcpy(&v, hs + S_CT_SLOTS + CS_SUMS + 2 * i, 8); t += v; }
        // (numeric_stage booked the numeric kernel's products on the figure of the multiply before: put this multiply's there)
        h->stats[stat_index(h, "numeric_class")].products += (long long)t - h->nnzCt;
        h->nnzCt = h->classSpec.nnzCt = (long long)t;
    }
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
    const int rc = pipeline_finish(h);
    if (rc != kSpecRefuted) return rc;
    // the arrays are not what they were a multiply ago: once more, every decision from this multiply's own figures
    h->classSpec.valid = false;
    h->laneSpec.valid = false;
    h->specRefuted++;
    if (h->verbose > 1) printf("  [speculative numeric launch refuted on the device: the multiply again]\n");
    BHS_TRY(pipeline_symbolic(h, true));            // (restart: the refuted attempt's timers and kernel records stay in -- it ran inside this multiply)
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
    h->besideStream = nullptr;
    h->rowPtrStaged = false;
}

int run_pipeline(bhs_handle* h)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = run_pipeline_impl(h);
    h->lastMultiplyMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != BHS_SUCCESS) { quiesce(h); h->ps.open = false; }
    return rc;
}
