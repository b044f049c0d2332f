// bhs_host_cabi.inc.h -- the C-ABI of include/bhsparse_hip.h
// (A part of bhsparse_hip.hip's translation unit: included there, inside its unnamed namespace where that applies.)

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
    release(h->mixList); release(h->classCount); release(h->laneBlockSums);
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
    // (2: event pairs around the numeric kernels only -- what a roofline of the dominant kernel needs, at a fifth of the events;
    // timers decide nothing: what the next multiply may assume of the last one stands)
    if (!strcmp(key, "kernel_stats")) { h->kernelStats = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); return BHS_SUCCESS; }
    h->classSpec.valid = false;                                     // (any other option may change what a multiply decides)
    h->laneSpec.valid = false;
    if (!strcmp(key, "spec_numeric")) { h->specNumeric = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "class_tile_piece")) { h->classTilePiece = (int)std::max<long long>(0, std::min<long long>(value, 1 << 17)); return BHS_SUCCESS; }
    if (!strcmp(key, "class_tile")) { h->classTile = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "class_mixed")) { h->mixOn = value ? 1 : 0; h->classMixed = 0; if (h->classState < 0) h->classState = 0; return BHS_SUCCESS; }
    if (!strcmp(key, "ring_dynamic")) { h->ringDynamic = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); return BHS_SUCCESS; }
    if (!strcmp(key, "class_mixed_fork")) { h->mixFork = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "class_mixed_max_pct")) { h->mixMaxPct = (int)std::max<int64_t>(0, std::min<int64_t>(value, 100)); return BHS_SUCCESS; }
    if (!strcmp(key, "spin_wait")) { h->spinWait = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "spin_wait_us")) { h->spinWaitUs = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1000000)); return BHS_SUCCESS; }
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
#if BHS_LAB               // (bhs_row_tiny.hip.h / bhs_row_span.hip.h are not in a product build: the options do not exist there)
    if (!strcmp(key, "tiny_rows")) { h->tinyRows = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "span_path")) { h->spanPath = value ? 1 : 0; h->spanState = 0; return BHS_SUCCESS; }
#endif
    if (!strcmp(key, "lane_numeric")) { h->laneNumeric = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "compress_b")) { h->compressB = (int)value; h->cmpState = 0; return BHS_SUCCESS; }
    if (!strcmp(key, "concurrent_bins")) { h->concurrentBins = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "window_bitmap")) { h->useWindowBitmap = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "sym_bitmap_min_log2")) { h->symBitmapMinLog2 = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "lds_bitmap_min_log2")) { h->ldsBitmapMinLog2 = (int)value; return BHS_SUCCESS; }
    if (!strcmp(key, "lds_bitmap")) { h->useLdsBitmap = value != 0; return BHS_SUCCESS; }
    if (!strcmp(key, "hub_min_products")) { h->hubMin = (int)std::min<int64_t>(value, 0x7fffffff); return BHS_SUCCESS; }
    if (!strcmp(key, "hub_item_products")) { if (value < 64) return BHS_ERR_INVALID_ARG; h->hubItemProducts = (int)std::min<int64_t>(value, 1 << 30); return BHS_SUCCESS; }
    if (!strcmp(key, "lane_from_counts")) { h->laneFromCounts = value ? 1 : 0; return BHS_SUCCESS; }
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
    if (!strcmp(key, "early_fill")) { h->earlyFill = value ? 1 : 0; return BHS_SUCCESS; }
    if (!strcmp(key, "sorted_scan")) { h->sortedScan = value ? 1 : 0; return BHS_SUCCESS; }
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
    if (!strcmp(key, "span_words")) { *value_out = h->ps.spanWPL; return BHS_SUCCESS; }   // bitmap words per lane of the last multiply's span kernels (0: hash kernels)
    if (!strcmp(key, "spec_launches")) { *value_out = h->specLaunches; return BHS_SUCCESS; }   // multiplies whose numeric kernel went out before the host saw the classes, so far
    if (!strcmp(key, "spec_refuted")) { *value_out = h->specRefuted; return BHS_SUCCESS; }     // ... of them, refuted on the device and run again
    if (!strcmp(key, "mixed_rows")) { *value_out = h->ps.mixed ? h->ps.mixRows : 0; return BHS_SUCCESS; }   // rows of the last multiply that went through the general kernels beside the class kernels
    if (!strcmp(key, "class_state")) { *value_out = h->classState < 0 ? -1 : (h->classMixed ? 2 : 1); return BHS_SUCCESS; }   // -1 general pipeline for good, 1 row classes, 2 row classes with irregular rows
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
    const bool usable = h->ps.useClass && !h->ps.classBig && !h->ps.empty && !h->ps.mixed;   // (a row without a class has no columns to rebuild)
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
