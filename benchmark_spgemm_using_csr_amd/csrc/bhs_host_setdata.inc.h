// bhs_host_setdata.inc.h -- the hand-over of a data set: the device row sort, the scans whose hints choose the launches
// (A part of bhsparse_hip.hip's translation unit: included there, inside its unnamed namespace where that applies.)

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
        hipLaunchKernelGGL(k_max_row, dim3((unsigned)gmr), dim3(256), 0, h->stream, h->m, h->dAp, small0 + S_SCAN, small0 + S_ROWLEN);
        hipLaunchKernelGGL(k_row_period, dim3(1), dim3(64), 0, h->stream, h->m, h->dAp, h->dAj, small0 + S_SCAN + 1, small0 + S_SCAN + 4, h->k, small0 + S_SCAN + 5);
        BHS_HIP(hipGetLastError());
    }
    if (h->k > 0) {
        const long long gmb = std::min<long long>(((long long)h->k + 255) / 256, (long long)h->numCU * 2);
        hipLaunchKernelGGL(k_max_row, dim3((unsigned)gmb), dim3(256), 0, h->stream, h->k, h->dBp, small0 + S_SCAN + 2, small0 + S_ROWLEN + 4);
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
    h->numDirectHint = -1;
    h->classSpec.valid = false;
    h->laneSpec.valid = false;
    h->classState = 0;
    const bool checkB = h->nnzB > 1 && h->k > 0;
    // rows of B beyond kSortedLongB entries are listed and checked by k_check_sorted_long, 16 workgroups per row
    int2* longB = nullptr;
    const int logG = std::min(h->logL, 6);                      // lanes per row of B: its average length
    const long long sortGrid = std::max<long long>(1, std::min<long long>(((long long)h->k + (256 >> logG) - 1) / (256 >> logG), (long long)h->numCU * 16));
    // (element-parallel scan: 32 counters of positions of colIndB not above their predecessor at S_CT_SLOTS, 32 of those of
    // them that are first entries of rows behind them -- free between multiplies --; equal sums = every row strictly
    // ascending.  h->sortedScan 0: round 2's row-by-row scan, flag in S_SORTED)
    auto check_sorted = [&]() -> int {
        BHS_HIP(hipMemsetAsync(small0 + S_SORTED, 0, sizeof(int), h->stream));
        BHS_HIP(hipMemsetAsync(small0 + S_LONG_B, 0, sizeof(int), h->stream));
        if (h->sortedScan) {
            BHS_HIP(hipMemsetAsync(small0 + S_CT_SLOTS, 0, sizeof(int) * 64, h->stream));
            const long long gf = std::max<long long>(1, std::min<long long>(((long long)h->nnzB + 1023) / 1024, (long long)h->numCU * 16));
            const long long gs = std::max<long long>(1, std::min<long long>(((long long)h->k + 255) / 256, (long long)h->numCU * 16));
            hipLaunchKernelGGL(k_sorted_flat, dim3((unsigned)gf), dim3(256), 0, h->stream, (long long)h->nnzB, h->dBj, small0 + S_CT_SLOTS);
            hipLaunchKernelGGL(k_sorted_starts, dim3((unsigned)gs), dim3(256), 0, h->stream, h->k, h->dBp, h->dBj, small0 + S_CT_SLOTS + 32);
        } else {
            hipLaunchKernelGGL(k_check_sorted, dim3((unsigned)sortGrid), dim3(256), 0, h->stream, h->k, logG, h->dBp, h->dBj,
                               small0 + S_SORTED, longB, small0 + S_LONG_B);
            hipLaunchKernelGGL(k_check_sorted_long, dim3((unsigned)(h->numCU * 4)), dim3(256), 0, h->stream,
                               (const int2*)longB, (const int*)(small0 + S_LONG_B), h->dBp, h->dBj, small0 + S_SORTED);
        }
        BHS_HIP(hipGetLastError());
        return BHS_SUCCESS;
    };
    // w: S_SORTED as read back, then the 64 counters
    auto unsorted = [&](const int* w) {
        if (!h->sortedScan) return w[0] != 0;
        long long flat = 0, starts = 0;
        for (int i = 0; i < 32; ++i) { flat += w[1 + i]; starts += w[33 + i]; }
        return flat != starts;
    };
    if (checkB) {
        BHS_TRY(ensure(h, h->longList, ((size_t)h->nnzB / 2048 + 2) * sizeof(int2)));
        longB = (int2*)h->longList.p;
        BHS_TRY(check_sorted());
    }
    // the scans bhs_row_span.hip.h's kernels are chosen from (rows of B strictly ascending: first / last entry = smallest / largest column)
#if BHS_LAB
    const bool spanScan = h->spanPath && h->m > 0 && h->k > 0 && h->nnzA > 0 && h->nnzB > 0;
#else
    const bool spanScan = false;
#endif
#if BHS_LAB
    if (spanScan) {
        BHS_HIP(hipMemsetD32Async((hipDeviceptr_t)(small0 + S_SPAN), (int)0x80000000, 2, h->stream));
        BHS_HIP(hipMemsetAsync(small0 + S_SPAN + 2, 0, sizeof(int), h->stream));
        const long long gb = std::max<long long>(1, std::min<long long>(((long long)h->k + 255) / 256, (long long)h->numCU * 8));
        const long long ga = std::max<long long>(1, std::min<long long>(((long long)h->m + 255) / 256, (long long)h->numCU * 8));
        hipLaunchKernelGGL(k_b_reach, dim3((unsigned)gb), dim3(256), 0, h->stream, h->k, h->dBp, h->dBj, small0 + S_SPAN);
        hipLaunchKernelGGL(k_a_width, dim3((unsigned)ga), dim3(256), 0, h->stream, h->m, h->dAp, h->dAj, small0 + S_SPAN + 2);
        BHS_HIP(hipGetLastError());
    }
#endif
    int* hscan = (int*)h->hostSmall;                                // (pinned)
    BHS_HIP(hipMemcpyAsync(hscan, small0 + S_SCAN, sizeof(int) * 6, hipMemcpyDeviceToHost, h->stream));
    if (checkB) {
        BHS_HIP(hipMemcpyAsync(hscan + 6, small0 + S_SORTED, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        BHS_HIP(hipMemcpyAsync(hscan + 7, small0 + S_CT_SLOTS, sizeof(int) * 64, hipMemcpyDeviceToHost, h->stream));
    }
    if (spanScan) BHS_HIP(hipMemcpyAsync(hscan + 72, small0 + S_SPAN, sizeof(int) * 3, hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipMemcpyAsync(hscan + 76, small0 + S_ROWLEN, sizeof(int) * 8, hipMemcpyDeviceToHost, h->stream));
    BHS_HIP(hipStreamSynchronize(h->stream));
    h->spanState = 0;
    h->reachL = h->reachR = h->widthA = 0x3fffffff;              // (no scan: never fits)
    if (spanScan && hscan[72] != (int)0x80000000) { h->reachL = hscan[72]; h->reachR = hscan[73]; h->widthA = hscan[74]; }
    const int maxRowA = hscan[0];
    h->maxRowA = maxRowA;
    h->maxRowB = hscan[2];
    if (h->m > 0) h->periodA = hscan[1];
    memcpy(h->lenStatsA, hscan + 76, sizeof(h->lenStatsA));
    memcpy(h->lenStatsB, hscan + 80, sizeof(h->lenStatsB));
    h->classMixed = 0;
    h->mixProbed = false;
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
        h->bSorted = unsorted(hscan + 6) ? 0 : 1;
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
            BHS_HIP(hipMemcpyAsync(hscan + 6, small + S_SORTED, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            BHS_HIP(hipMemcpyAsync(hscan + 7, small + S_CT_SLOTS, sizeof(int) * 64, hipMemcpyDeviceToHost, h->stream));
            BHS_HIP(hipStreamSynchronize(h->stream));
            h->bSorted = unsorted(hscan + 6) ? 0 : 1;          // (duplicate columns inside a row still count as "not ascending")
        }
    }
    // compressed pattern of B: decide now whether it pays (the multiply itself re-runs the compression inside its
    // timed region; this pass only yields the pair count)
    // (round 4: the pair count is also taken for rows of 256 to 1536 products: where B's entries come in long runs -- banded
    // matrices, dense diagonal blocks: a tenth as many pairs as entries -- the compressed pass pays from there on)
    // A data set that will try the row classes first (pipeline_symbolic's test) and has rows below the old gate leaves the
    // count to its first multiply on the general pipeline, if it ever gets there (cmpState 0: that multiply measures the
    // ratio, the ones after it use the verdict) -- poisson27pt's hand-over does not pay a pass over B for nothing.
    const bool classFirst = h->classPath && h->forcePath == 0 && h->maxTableLog2 >= 15 && cls_row_a(h) <= kClassMaxRowBig &&
                            cls_row_b(h) <= kClassMaxRowBig &&
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
