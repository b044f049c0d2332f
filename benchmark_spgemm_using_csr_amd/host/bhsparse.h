// bhsparse.h — the bhSPARSE facade class for the MI355X backend.
//
// Public interface = SpGEMM_cuda/bhsparse.h:17-33, verbatim signatures, same
// call order (main.cu:104-135), same `int` error convention (0 ==
// BHSPARSE_SUCCESS), same borrowed-pointer ownership (bhsparse.h:205-216), same
// stdout lines (stage times, "[ HIP ] SpGEMM time: T ms. Gflops = G",
// bhsparse.h:287-289).  Every method forwards to libbhsparse_hip.so through the
// C-ABI; there is no device code and no HIP header on this side.
// The OpenCL variant's trailing `use_host_mem` flag (SpGEMM_opencl/bhsparse.h:44-47)
// is accepted and ignored (MI355X is a discrete HBM part).
#ifndef BHSPARSE_AMD_BHSPARSE_H
#define BHSPARSE_AMD_BHSPARSE_H

#include <chrono>

#include "../../include/bhsparse_hip.h"
#include "common.h"

class bhsparse
{
public:
    bhsparse() : _spgemm_platform(0), _h(0), _m(0), _k(0), _n(0), _nnzCt_full(0), _nnzC(0), _h_csrRowPtrC(0) {}
    int initPlatform(bool *spgemm_platform);
    int initData(int m, int k, int n,
                 int nnzA, value_type *csrValA, index_type *csrRowPtrA, index_type *csrColIndA,
                 int nnzB, value_type *csrValB, index_type *csrRowPtrB, index_type *csrColIndB,
                 index_type *csrRowPtrC, bool use_host_mem = false);
    int spgemm();
    int warmup();

    int get_nnzC();
    int get_C(index_type *csrColIndC, value_type *csrValC);

    int freePlatform();
    int free_mem();

    // additions (not in the reference): product count and device stage times of the last spgemm()
    long long get_nnzCt() const { return _nnzCt_full; }
    const double *get_stage_ms() const { return _stage_ms; }
    // the C-ABI handle behind this object, for the multi-GPU layer (include/bhsparse_dist.h)
    bhs_handle *handle() const { return _h; }

private:
    bool       *_spgemm_platform;
    bhs_handle *_h;
    int         _m, _k, _n;
    long long   _nnzCt_full;      // size_t in the reference (bhsparse.h:57)
    int         _nnzC;
    index_type *_h_csrRowPtrC;    // caller-owned, filled by spgemm()
    double      _stage_ms[4];
};

inline int bhsparse::initPlatform(bool *spgemm_platform)
{
    _spgemm_platform = spgemm_platform;
    if (!spgemm_platform) return BHS_ERR_INVALID_ARG;
    if (!(spgemm_platform[BHSPARSE_HIP] || spgemm_platform[BHSPARSE_CUDA] || spgemm_platform[BHSPARSE_OPENCL]))
        return BHS_ERR_INVALID_ARG;
    int dev = 0;                                  // the reference hard-codes device 0 (bhsparse_cuda.h:100-101)
    if (const char *e = getenv("BHSPARSE_DEVICE")) dev = atoi(e);
    int err = bhs_create(&_h, 1, &dev);
    if (err != BHSPARSE_SUCCESS) return err;
    return bhs_set_verbose(_h, 1);                // device banner + stage lines, as the reference prints
}

inline int bhsparse::initData(int m, int k, int n,
                              int nnzA, value_type *csrValA, index_type *csrRowPtrA, index_type *csrColIndA,
                              int nnzB, value_type *csrValB, index_type *csrRowPtrB, index_type *csrColIndB,
                              index_type *csrRowPtrC, bool /*use_host_mem*/)
{
    if (!_h) return BHS_ERR_NOT_READY;
    _m = m; _k = k; _n = n;
    _nnzC = 0;
    _h_csrRowPtrC = csrRowPtrC;
    return bhs_set_data(_h, m, k, n, nnzA, csrValA, csrRowPtrA, csrColIndA, nnzB, csrValB, csrRowPtrB, csrColIndB);
}

inline int bhsparse::warmup()
{
    if (!_h) return BHS_ERR_NOT_READY;
    int err = bhs_set_verbose(_h, 0);             // warm-ups are silent in the reference too
    if (err == BHSPARSE_SUCCESS) err = bhs_warmup(_h);
    bhs_set_verbose(_h, 1);
    if (err != BHSPARSE_SUCCESS) std::cout << "warmup error = " << err << std::endl;
    return err;
}

inline int bhsparse::spgemm()
{
    if (!_h) return BHS_ERR_NOT_READY;
    const auto t0 = std::chrono::steady_clock::now();
    int64_t nnzCt = 0;
    int err = bhs_spgemm(_h, _h_csrRowPtrC, &nnzCt, &_nnzC, _stage_ms);
    const double time = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (err != BHSPARSE_SUCCESS) { std::cout << "spgemm error = " << err << std::endl; return err; }
    _nnzCt_full = nnzCt;
    std::cout << "[ HIP ] SpGEMM time: " << time << " ms. Gflops = "
              << 2.0 * (double)_nnzCt_full / (time * 1.0e+6) << std::endl;
    return err;
}

inline int bhsparse::get_nnzC()
{
    int v = 0;
    if (!_h || bhs_get_nnzC(_h, &v) != BHSPARSE_SUCCESS) return 0;
    return v;
}

inline int bhsparse::get_C(index_type *csrColIndC, value_type *csrValC)
{
    if (!_h) return BHS_ERR_NOT_READY;
    int err = bhs_get_C(_h, csrColIndC, csrValC);
    if (err == BHSPARSE_SUCCESS && _h_csrRowPtrC) err = bhs_get_rowptrC(_h, _h_csrRowPtrC);   // bhsparse_cuda.h:3016
    return err;
}

inline int bhsparse::free_mem()
{
    if (!_h) return BHS_ERR_NOT_READY;
    return bhs_free_data(_h);
}

inline int bhsparse::freePlatform()
{
    if (!_h) return BHSPARSE_SUCCESS;
    int err = bhs_destroy(_h);
    _h = 0;
    return err;
}

#endif
