/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the contract that the reference's SpGEMM path pins
 * (bhSPARSE, weifengliu-ssslab/Benchmark_SpGEMM_using_CSR).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libbhsparse_hip.so) never links or calls it.
 *
 * PARITY STATUS: **pinned to the reference itself** (round 3).
 *   - The reference's CUDA branch and its checker (cusp::multiply on device COO, CUSP v0.4.0, README.md:91,
 *     un-vendored) cannot be built here (no nvcc, no CUSP, no helper_cuda.h).  Its OpenCL branch can: oracle/Makefile
 *     `_ref` compiles SpGEMM_opencl/{bhsparse,bhsparse_opencl,basiccl}.cpp unmodified with g++ around a small dump
 *     driver (oracle/ref_opencl_dump.cpp), and the MI355X boxes expose an OpenCL device (profiles/
 *     r03_gpu_box_clinfo.txt).  oracle/make_ref_golden.py ran 15 inputs through it ON AN MI355X and compared this
 *     oracle with the reference's C on the spot: rowPtr, colInd and values identical in all 15 (the reference's two
 *     fixed inputs, the gallery stencils up to its default sizes poisson5pt 256^2 / poisson27pt 51^3, rectangular,
 *     every bin of its table, its multi-round merge on power-law rows, exact cancellation -- 168 structural zeros
 *     kept).  The reference's outputs are committed as tests/golden/ref_opencl_*.npz (+ digests of the three large
 *     cases) and tests/test_oracle.py re-checks this file against them on every CPU run.
 *   - Beside that: hand-derived known answers for the reference's two fixed inputs, scipy
 *     fixtures (tests/golden/make_golden.py; valid because all driver values are > 0), the stencils' closed forms.
 *
 * What is restated, and from where:
 *   - result definition: for each row i the multiset {(colB, valA*valB)} over
 *     j in A(i,:), colB in B(j,:) is sorted by column and equal columns are
 *     summed; every distinct column is emitted even when the sum is 0.0
 *     (membership is by key only: bhsparse_cuda.h:608-648 heap pop/merge,
 *     :1311-1397 compression_scan, :2051-2053 binary-search hit => val+=).
 *   - nnzCt ("intermediate products", the GFLOPs numerator):
 *     bhsparse_cuda.h:210-237 (compute_nnzCt_cudakernel) and
 *     bhsparse.h:365-406 (_nnzCt_full accumulation).
 *   - csr_sort_indices: ref_spgemm.h:37-62 (per-row sort of (col,val) by col).
 *   - compare: ref_spgemm.h:79-126 (nnzC, then rowPtr, then col exact + value
 *     tolerance; the reference's 10% is tightened to the caller's rel_tol).
 *
 * Algorithm: Gustavson row-wise two-pass with a dense marker/SPA per thread,
 * then a per-row column sort.  Deliberately independent of the hashing /
 * sorting-network scheme used by the HIP kernels.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ref_spgemm.h:37-62 — sort every row's (col,val) pairs by column.
 * Insertion sort for short rows, heap-free merge sort otherwise; stable. */
typedef struct { int32_t c; double v; } cv_pair;

static int cv_cmp(const void *a, const void *b)
{
    int32_t x = ((const cv_pair *)a)->c, y = ((const cv_pair *)b)->c;
    return (x > y) - (x < y);
}

static void sort_row(int32_t *cols, double *vals, int64_t len, cv_pair *tmp)
{
    if (len < 2) return;
    int sorted = 1;
    for (int64_t t = 1; t < len; t++) if (cols[t - 1] > cols[t]) { sorted = 0; break; }
    if (sorted) return;
    if (len <= 32) {
        for (int64_t t = 1; t < len; t++) {
            int32_t c = cols[t]; double v = vals[t]; int64_t u = t - 1;
            while (u >= 0 && cols[u] > c) { cols[u + 1] = cols[u]; vals[u + 1] = vals[u]; u--; }
            cols[u + 1] = c; vals[u + 1] = v;
        }
        return;
    }
    for (int64_t t = 0; t < len; t++) { tmp[t].c = cols[t]; tmp[t].v = vals[t]; }
    qsort(tmp, (size_t)len, sizeof(cv_pair), cv_cmp);   /* columns are unique in C rows */
    for (int64_t t = 0; t < len; t++) { cols[t] = tmp[t].c; vals[t] = tmp[t].v; }
}

ORACLE_API void oracle_csr_sort_indices(int32_t n_row, const int32_t *Ap, int32_t *Aj, double *Ax)
{
    int64_t maxlen = 0;
    for (int32_t i = 0; i < n_row; i++) if (Ap[i + 1] - Ap[i] > maxlen) maxlen = Ap[i + 1] - Ap[i];
    cv_pair *tmp = (cv_pair *)malloc((size_t)(maxlen > 0 ? maxlen : 1) * sizeof(cv_pair));
    for (int32_t i = 0; i < n_row; i++) {
        /* stable for duplicate columns: insertion sort path or index-tagged */
        int64_t s = Ap[i], len = Ap[i + 1] - Ap[i];
        if (len <= 32) { sort_row(Aj + s, Ax + s, len, tmp); continue; }
        /* stable merge via (col, original position) ordering */
        for (int64_t t = 0; t < len; t++) { tmp[t].c = Aj[s + t]; tmp[t].v = Ax[s + t]; }
        /* simple bottom-up merge sort for stability */
        cv_pair *buf = (cv_pair *)malloc((size_t)len * sizeof(cv_pair));
        cv_pair *src = tmp, *dst = buf;
        for (int64_t w = 1; w < len; w *= 2) {
            for (int64_t lo = 0; lo < len; lo += 2 * w) {
                int64_t mid = lo + w < len ? lo + w : len, hi = lo + 2 * w < len ? lo + 2 * w : len;
                int64_t a = lo, b = mid, o = lo;
                while (a < mid && b < hi) dst[o++] = (src[b].c < src[a].c) ? src[b++] : src[a++];
                while (a < mid) dst[o++] = src[a++];
                while (b < hi) dst[o++] = src[b++];
            }
            cv_pair *t2 = src; src = dst; dst = t2;
        }
        for (int64_t t = 0; t < len; t++) { Aj[s + t] = src[t].c; Ax[s + t] = src[t].v; }
        free(buf);
    }
    free(tmp);
}

/* bhsparse_cuda.h:210-237 + bhsparse.h:365-406 — per-row upper bound
 * ub[i] = sum_{j in A(i,:)} len(B(j,:)); returns nnzCt_full = sum ub[i]. */
ORACLE_API int64_t oracle_nnzCt(int32_t m, const int32_t *Ap, const int32_t *Aj,
                                const int32_t *Bp, int64_t *ub_out /* m or NULL */)
{
    int64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (int32_t i = 0; i < m; i++) {
        int64_t s = 0;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int32_t j = Aj[jj];
            s += Bp[j + 1] - Bp[j];
        }
        if (ub_out) ub_out[i] = s;
        total += s;
    }
    return total;
}

/* Symbolic pass: Cp[i+1] = number of distinct columns in row i of A*B
 * (structural; zeros are entries).  Writes exclusive-scan row pointer Cp[0..m]
 * as int64 so the caller can detect int32 overflow.  Returns nnzC. */
ORACLE_API int64_t oracle_spgemm_symbolic(int32_t m, int32_t k, int32_t n,
                                          const int32_t *Ap, const int32_t *Aj,
                                          const int32_t *Bp, const int32_t *Bj,
                                          int64_t *Cp, int nthreads)
{
    (void)k;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    Cp[0] = 0;
#pragma omp parallel num_threads(nthreads)
    {
        int32_t *marker = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
        for (int32_t c = 0; c < n; c++) marker[c] = -1;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < m; i++) {
            int64_t cnt = 0;
            for (int32_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                int32_t j = Aj[jj];
                for (int32_t kk = Bp[j]; kk < Bp[j + 1]; kk++) {
                    int32_t c = Bj[kk];
                    if (marker[c] != i) { marker[c] = i; cnt++; }
                }
            }
            Cp[i + 1] = cnt;
        }
        free(marker);
    }
    for (int32_t i = 0; i < m; i++) Cp[i + 1] += Cp[i];
    return Cp[m];
}

/* Numeric pass into caller-allocated Cj/Cx (sized by the symbolic pass).
 * Products are accumulated in A-row order then B-row order (the natural
 * Gustavson order); rows are column-sorted afterwards (ref_spgemm.h:77). */
ORACLE_API void oracle_spgemm_numeric(int32_t m, int32_t k, int32_t n,
                                      const int32_t *Ap, const int32_t *Aj, const double *Ax,
                                      const int32_t *Bp, const int32_t *Bj, const double *Bx,
                                      const int64_t *Cp, int32_t *Cj, double *Cx, int nthreads)
{
    (void)k;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        int64_t *pos = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
        for (int32_t c = 0; c < n; c++) pos[c] = -1;
        int64_t maxlen = 0;
        for (int32_t i = 0; i < m; i++) if (Cp[i + 1] - Cp[i] > maxlen) maxlen = Cp[i + 1] - Cp[i];
        cv_pair *tmp = (cv_pair *)malloc((size_t)(maxlen > 0 ? maxlen : 1) * sizeof(cv_pair));
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < m; i++) {
            int64_t base = Cp[i], fill = base;
            for (int32_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                int32_t j = Aj[jj];
                double a = Ax[jj];
                for (int32_t kk = Bp[j]; kk < Bp[j + 1]; kk++) {
                    int32_t c = Bj[kk];
                    double p = a * Bx[kk];
                    if (pos[c] < base) { pos[c] = fill; Cj[fill] = c; Cx[fill] = p; fill++; }
                    else Cx[pos[c]] += p;
                }
            }
            sort_row(Cj + base, Cx + base, fill - base, tmp);
            /* rows reach a thread in arbitrary order: reset this row's marks */
            for (int64_t t = base; t < fill; t++) pos[Cj[t]] = -1;
        }
        free(tmp);
        free(pos);
    }
}

/* ref_spgemm.h:79-126 restated as a function that returns what it would
 * print.  out[0]=stage reached (0 nnzC mismatch, 1 rowPtr mismatch, 2 col/val
 * mismatch, 3 PASS), out[1]=rowPtr error count, out[2]=col error count,
 * out[3]=value error count (|d| > rel_tol*|ref|). */
ORACLE_API void oracle_compare(int32_t m, int64_t ref_nnzC, const int64_t *ref_Cp,
                               const int32_t *ref_Cj, const double *ref_Cx,
                               int64_t nnzC, const int32_t *Cp, const int32_t *Cj, const double *Cx,
                               double rel_tol, int64_t *out)
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if (ref_nnzC != nnzC) return;
    int64_t e = 0;
    for (int32_t i = 0; i <= m; i++) if (ref_Cp[i] != (int64_t)Cp[i]) e++;
    out[1] = e;
    if (e) { out[0] = 1; return; }
    int64_t ec = 0, ev = 0;
#pragma omp parallel for reduction(+ : ec, ev) schedule(static)
    for (int64_t t = 0; t < nnzC; t++) {
        if (ref_Cj[t] != Cj[t]) ec++;
        else if (fabs(ref_Cx[t] - Cx[t]) > fabs(rel_tol * ref_Cx[t])) ev++;
    }
    out[2] = ec; out[3] = ev;
    out[0] = (ec || ev) ? 2 : 3;
}

/* Size-independent digest of a CSR result, used for full-size parity where the
 * whole C is too big to ship as a fixture:
 * d[0]=nnz, d[1]=sum rowPtr (mod 2^64), d[2]=sum col*(pos%8191+1) (mod 2^64),
 * d[3]=bit pattern of sum of values (exact for integer-valued inputs < 2^53). */
ORACLE_API void oracle_digest(int32_t m, const int64_t *Cp, const int32_t *Cj, const double *Cx,
                              uint64_t *d)
{
    uint64_t s1 = 0, s2 = 0; double sv = 0.0;
    int64_t nnz = Cp[m];
    for (int32_t i = 0; i <= m; i++) s1 += (uint64_t)Cp[i];
#pragma omp parallel for reduction(+ : s2, sv) schedule(static)
    for (int64_t t = 0; t < nnz; t++) {
        s2 += (uint64_t)Cj[t] * (uint64_t)(t % 8191 + 1);
        sv += Cx[t];
    }
    d[0] = (uint64_t)nnz; d[1] = s1; d[2] = s2; memcpy(&d[3], &sv, 8);
}
