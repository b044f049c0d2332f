/* C declarations of the CPU oracle (TEST INFRASTRUCTURE ONLY; see ref_spgemm_oracle.c). */
#ifndef REF_SPGEMM_ORACLE_H
#define REF_SPGEMM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int     oracle_max_threads(void);
void    oracle_csr_sort_indices(int32_t n_row, const int32_t *Ap, int32_t *Aj, double *Ax);
int64_t oracle_nnzCt(int32_t m, const int32_t *Ap, const int32_t *Aj, const int32_t *Bp, int64_t *ub_out);
int64_t oracle_spgemm_symbolic(int32_t m, int32_t k, int32_t n, const int32_t *Ap, const int32_t *Aj,
                               const int32_t *Bp, const int32_t *Bj, int64_t *Cp, int nthreads);
void    oracle_spgemm_numeric(int32_t m, int32_t k, int32_t n, const int32_t *Ap, const int32_t *Aj,
                              const double *Ax, const int32_t *Bp, const int32_t *Bj, const double *Bx,
                              const int64_t *Cp, int32_t *Cj, double *Cx, int nthreads);
void    oracle_compare(int32_t m, int64_t ref_nnzC, const int64_t *ref_Cp, const int32_t *ref_Cj,
                       const double *ref_Cx, int64_t nnzC, const int32_t *Cp, const int32_t *Cj,
                       const double *Cx, double rel_tol, int64_t *out);
void    oracle_digest(int32_t m, const int64_t *Cp, const int32_t *Cj, const double *Cx, uint64_t *d);
#ifdef __cplusplus
}
#endif
#endif
