// bhs_host_launch.inc.h -- the launch helpers: grid, LDS and template instance of every kernel family
// (A part of bhsparse_hip.hip's translation unit: included there, inside its unnamed namespace where that applies.)

// The classifier's kernels are sized by the longest row.  Where only a few rows are long (a constraint row, a handful of
// boundary rows: fewer than one in 64) the others' longest row sizes them instead: a longer row finds no class and goes
// through the general pipeline's kernels, mixed mode (bhs_class_mix.hip.h).  A hint like every other: checked per row.
int cls_row(const bhs_handle* h, int maxRow, const int* st, int rows)
{
    int eff = maxRow;
    if (!h->mixOn) return eff;
    if (eff > kClassMaxRowBig && st[2] > 0 && (long long)st[2] * 64 <= rows) eff = st[3];
    if (eff > kClassMaxRow && st[0] > 0 && (long long)st[0] * 64 <= rows) eff = st[1];
    return eff;
}
int cls_row_a(const bhs_handle* h) { return cls_row(h, h->maxRowA, h->lenStatsA, h->m); }
int cls_row_b(const bhs_handle* h) { return cls_row(h, h->maxRowB, h->lenStatsB, h->k); }

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
    const int chunkRows = std::max(1, std::min(64, 128 / std::max(1, h->ps.classMaxNA)));   // whole rows of a class, <= 128 entries of A (a row without a class is a chunk of its own: chunk_end)
    if (h->verbose > 1) printf("  [class numeric (ring, round 5): %d waves per CU by the occupancy API, %d used, %zu bytes of LDS each, grid %lld, %d rows per chunk]\n", perCU, useCU, lds.bytes, grid, chunkRows);
    // (the XCDs' counters of super-runs: zero from the multiply's start for its first launch, cleared for a later row range's)
    if (h->ps.ringLaunches++ > 0) BHS_HIP(hipMemsetAsync((int*)h->small.p + S_RING_TICKETS, 0, sizeof(int) * 8, h->ls));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds.bytes, h->ls, mR, h->dAp + r0, h->dAj, h->dAx,
                       (long long)h->nnzA, h->dBp, h->dBx, (long long)h->nnzB, (const int*)h->classC.p + r0, (const int4*)h->classInfo.p,
                       (const unsigned*)h->classRing.p, (const int*)h->classRel.p, (const int*)h->classLane.p, (const int*)h->Cp.p + r0, out_cj(h),
                       out_cx(h), lds.ringBytes, lds.accStride, r0, superRows, chunkRows,
                       h->ps.specLaunched ? (const int*)h->small.p + S_SPEC : (const int*)nullptr, (h->ringDynamic == 1 || (h->ringDynamic == 2 && h->ps.ringBeside)) ? (int*)h->small.p + S_RING_TICKETS : (int*)nullptr);
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

#if BHS_LAB
// one wave per row, the row's columns as a bitmap over its span (bhs_row_span.hip.h); VCAP follows the bin's table size
template <int WPL, int VCAP, bool NUM>
int launch_row_span_impl(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt)
{
    auto kern = k_row_span<WPL, VCAP, NUM>;
    const size_t smem = sizeof(SpanSmem<WPL, VCAP, NUM>);
    int perCU = 1;
    BHS_TRY(kernel_occupancy(h, reinterpret_cast<const void*>(kern), 64, smem, &perCU));
    perCU = std::max(1, std::min(perCU, 32));
    const int useCU = h->wgPerCU > 0 ? h->wgPerCU : perCU;
    long long grid = std::min<long long>(qn, (long long)h->numCU * useCU);
    grid = std::max<long long>(8, (grid + 7) / 8 * 8);
    int chunkLog2 = 0;
    while ((2 << chunkLog2) <= BHS_XCD_CHUNK && (128LL << chunkLog2) <= (long long)qn) ++chunkLog2;
    const bool wf = !NUM && queue == nullptr;
    if (h->verbose > 1) printf("  [%s span: %d words per lane, %d sums, %d waves per CU, smem %zu B]\n", NUM ? "numeric" : "symbolic", WPL, VCAP, perCU, smem);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), smem, h->ls, queue, qn, chunkLog2, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx,
                       CpOrCnt, out_cj(h), out_cx(h), h->dAp, wf ? (int*)h->ub.p : (int*)nullptr,
                       wf ? (unsigned long long*)((int*)h->small.p + S_CT_SLOTS) : (unsigned long long*)nullptr,
                       (int*)h->small.p + S_ERR, h->reachL, h->reachR);
    BHS_HIP(hipGetLastError());
    return BHS_SUCCESS;
}
template <bool NUM>
int launch_row_span(bhs_handle* h, const int4* queue, int qn, int* CpOrCnt, int lg)
{
    const int wpl = h->ps.spanWPL;
    if constexpr (!NUM) {
        if (wpl <= 1) return launch_row_span_impl<1, 1, false>(h, queue, qn, CpOrCnt);
        if (wpl <= 2) return launch_row_span_impl<2, 1, false>(h, queue, qn, CpOrCnt);
        return launch_row_span_impl<4, 1, false>(h, queue, qn, CpOrCnt);
    } else {
#define BHS_SPAN(W)                                                                                   \
        if (wpl <= W) {                                                                                \
            if (lg <= 7) return launch_row_span_impl<W, 128, true>(h, queue, qn, CpOrCnt);            \
            if (lg <= 9) return launch_row_span_impl<W, 512, true>(h, queue, qn, CpOrCnt);            \
            return launch_row_span_impl<W, 1024, true>(h, queue, qn, CpOrCnt);                        \
        }
        BHS_SPAN(1) BHS_SPAN(2) BHS_SPAN(4)
#undef BHS_SPAN
        return BHS_ERR_INTERNAL;
    }
}
#endif

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
                    unsigned long long* ctSlots = nullptr, const int* specWord = nullptr,
                    int* blockSums = nullptr, long long assumedNnzC = 0, int* specOut = nullptr)   // (k_row_lane: "rowPtrC on the way")
{
    const int nBlocks = (int)(((long long)qn + 255) / 256);
    const unsigned grid = (unsigned)(((long long)qn + 255) / 256);
    const bool smallB = h->allowSmallB && h->nnzB < (1 << 29);
    // every row's products in registers where the longest rows of A and B seen at hand-over keep them to 32 (checked per
    // row on the device): bhs_row_tiny.hip.h
#if BHS_LAB
    if (h->tinyRows && h->forcePath == 0 && h->maxRowA > 0 && h->maxRowB > 0 && (long long)h->maxRowA * h->maxRowB <= 32) {
#define BHS_TINY(KA, LB)                                                                                              \
        if (h->maxRowA <= KA && h->maxRowB <= LB) {                                                                    \
            hipLaunchKernelGGL((k_row_tiny<KA, LB, NUM>), dim3(grid), dim3(256), 0, h->ls, queue, qn, h->dAp, h->dAj, h->dAx, h->dBp, \
                               h->dBj, h->dBx, CpOrCnt, out_cj(h), out_cx(h), ubOut, ctSlots, (int*)h->small.p + S_ERR);        \
            BHS_HIP(hipGetLastError());                                                                                \
            return BHS_SUCCESS;                                                                                        \
        }
        BHS_TINY(5, 5) BHS_TINY(4, 8) BHS_TINY(8, 4) BHS_TINY(2, 16) BHS_TINY(16, 2)
#undef BHS_TINY
    }
#endif
#define BHS_LANE(KK)                                                                                          \
    case KK:                                                                                                  \
        if (smallB)                                                                                           \
            hipLaunchKernelGGL((k_row_lane<KK, NUM, true>), dim3(grid), dim3(256), 0, h->ls, queue, qn,       \
                               h->dAp, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h),                    \
                               out_cx(h), ubOut, ctSlots, (int*)h->small.p + S_ERR, specWord, blockSums, nBlocks, assumedNnzC, specOut); \
        else                                                                                                  \
            hipLaunchKernelGGL((k_row_lane<KK, NUM, false>), dim3(grid), dim3(256), 0, h->ls, queue, qn,      \
                               h->dAp, h->dAj, h->dAx, h->dBp, h->dBj, h->dBx, CpOrCnt, out_cj(h),                    \
                               out_cx(h), ubOut, ctSlots, (int*)h->small.p + S_ERR, specWord, blockSums, nBlocks, assumedNnzC, specOut); \
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
    // (symbolic: the workgroup-per-row bin of 32768 slots -- 128 KB of LDS for a hash table where the bitmap's first pass and a
    // popcount do: option "sym_bitmap_min_log2", 16: never)
    if (!NUM) return c.block > 64 && c.log2ts >= h->symBitmapMinLog2 && h->forcePath == 0;
    return c.block > 64 && c.log2ts >= h->ldsBitmapMinLog2 && h->forcePath == 0;
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
#if BHS_LAB
    // rows accumulated over their column span where the data set's scans say every row fits (checked per row on the device)
    if (h->ps.spanWPL > 0 && c.block == 64 && !win && lg <= 10 && h->forcePath == 0) return launch_row_span<NUM>(h, queue, qn, CpOrCnt, lg);
    if constexpr (!NUM) {
        if (h->ps.spanWPL > 0 && c.block == 64 && !win && h->forcePath == 0) return launch_row_span<NUM>(h, queue, qn, CpOrCnt, lg);
    }
#endif
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
