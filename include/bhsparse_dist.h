/* bhsparse_dist.h — C-ABI of libbhsparse_dist.so: the multi-GPU layer of the
 * CSR SpGEMM hot path (one process per GPU, RCCL over xGMI).
 *
 * The reference is single-device (device 0 is hard-coded, SpGEMM_cuda/
 * bhsparse_cuda.h:100-101); this layer is new.  Rows of A -- and therefore of C
 * -- shard across the GPUs of one node in contiguous blocks, B is replicated,
 * every rank runs the single-GPU pipeline (include/bhsparse_hip.h) on its block
 * and ONE all-gatherv assembles the CSR of C on every rank.  The all-gatherv is
 * a group of point-to-point transfers (ncclGroupStart / ncclSend / ncclRecv /
 * ncclGroupEnd): block sizes differ per rank, and on a fully connected xGMI
 * node every GPU then sends its block straight to each of its 7 peers.
 *
 * A reference-style C++ driver binds it like this (tests/driver/spgemm_main.cpp,
 * option -gpus N: one child process per GPU):
 *     bhs_dist_unique_id(id) on rank 0, id handed to the other ranks;
 *     bhs_create / bhs_set_data with the rank's row block of A and all of B;
 *     bhs_dist_create(&d, handle, world, rank, id);
 *     bhs_dist_spgemm_allgatherv(d, ...);      // multiply + assemble
 * Plain C types only; every function returns a bhs_status (0 = success).       */
#ifndef BHSPARSE_DIST_H
#define BHSPARSE_DIST_H
#include "bhsparse_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bhs_dist bhs_dist;
#define BHS_DIST_ID_BYTES 128              /* sizeof(ncclUniqueId) */

/* rank 0 creates the rendezvous id and hands the bytes to every other rank (pipe, file, MPI, torch broadcast ..) */
BHS_API int bhs_dist_unique_id(char id_out[BHS_DIST_ID_BYTES]);
/* collective: every rank of the job calls it with the same id; `h` is the rank's bhs_handle (its device and stream) */
BHS_API int bhs_dist_create(bhs_dist **out, bhs_handle *h, int world, int rank, const char id[BHS_DIST_ID_BYTES]);
BHS_API int bhs_dist_destroy(bhs_dist *d);

/* Contiguous row blocks of A balanced by WORK: starts_out[r] .. starts_out[r+1] are the rows of rank r, chosen so that
 * every block carries about the same number of intermediate products (the prefix of the per-row upper bound that
 * compute_nnzCt produces, bhsparse_cuda.h:210-237).  For stencil matrices this is the equal-rows split; for power-law
 * matrices it is not.  Host arrays; starts_out has world + 1 entries.                                              */
BHS_API int bhs_dist_partition_rows(int m, const int *rowPtrA, const int *colIndA, const int *rowPtrB, int world,
                                    int *starts_out);

/* One multiply of this rank's row block (already bound to the handle with bhs_set_data[_device]) and the all-gatherv
 * of C.  Collective.
 *   m_local / m_total      rows of this rank's block / of the whole matrix
 *   sub_blocks             >= 1: the numeric half runs in this many row ranges, and the transfers of range s (on a
 *                          second stream) overlap the numeric kernels of range s + 1
 *   d_rowPtrC / d_colIndC / d_valC   device arrays of m_total + 1 / capacity / capacity entries on THIS rank that
 *                          receive the assembled C (every rank ends with the same content).  The rank's own block is
 *                          written there by the numeric kernels themselves (bhs_set_output_device): no staging copy.
 *   nnzCt_total_out, nnzC_total_out  products / entries of the whole job; may be NULL
 *   ms_out                 [0] symbolic half + size exchange, [1] numeric half with the overlapped transfers,
 *                          [2] wait for the remaining transfers (host wall clock, ms); may be NULL
 * Returns after the assembled C is complete on this rank.  BHS_ERR_NNZ_OVERFLOW when nnz(C) of the job does not fit
 * the int32 index_type, BHS_ERR_ALLOC when it exceeds some rank's `capacity`.
 * Failures are collective: every rank reaches three agreements whatever happened to it locally -- the sizes all-gather
 * (it carries each rank's status and capacity, so sizes, totals and capacities are judged identically everywhere
 * before any transfer is posted), a one-word all-reduce once the output arrays are in place, and one after the numeric
 * half (a rank whose kernels failed still posts its sends and receives).  A rank that failed returns its own error,
 * the others BHS_ERR_PEER; nobody is left waiting.  Only an RCCL error inside a send / receive group is fatal: the
 * group is closed, the communicator aborted, and this object refuses further calls (BHS_ERR_LAUNCH).  On every
 * return the handle has no open multiply and no output arrays bound (a later bhs_spgemm on it uses its own).        */
BHS_API int bhs_dist_spgemm_allgatherv(bhs_dist *d, int m_local, int m_total, int sub_blocks, int *d_rowPtrC,
                                       int *d_colIndC, bhs_value_t *d_valC, int64_t capacity,
                                       int64_t *nnzCt_total_out, int64_t *nnzC_total_out, double ms_out[3]);

/* The same for host-pointer callers (the reference-style driver): the assembled C is kept in device memory owned by
 * `d`; rowPtrC_out (m_total + 1 ints, host) is filled, the entry count comes back in nnzC_total_out, and
 * bhs_dist_get_C_host copies colIndC / valC out afterwards -- the get_nnzC / malloc / get_C sequence of the
 * reference's driver (main.cu:123-135) at job scope.                                                              */
BHS_API int bhs_dist_spgemm_allgatherv_host(bhs_dist *d, int m_local, int m_total, int sub_blocks, int *rowPtrC_out,
                                            int64_t *nnzCt_total_out, int64_t *nnzC_total_out, double ms_out[3]);
BHS_API int bhs_dist_get_C_host(bhs_dist *d, int *csrColIndC, bhs_value_t *csrValC);

/* The point-to-point operations bhs_dist_spgemm_allgatherv issues on `rank`, in order, for a job whose ranks own
 * rows[r] rows and whose rowPtrC at the sub-block boundaries are cuts[r * (sub_blocks + 1) + s]: 6 int64 per operation
 * {kind 0 send / 1 recv, peer, array 0 colInd / 1 val / 2 rowPtr, element offset in the assembled array, element count,
 * group = sub-block}.  Returns the number of operations (writes at most cap_ops).  No device, no communicator: the CPU
 * tests replay the plans of all ranks against each other.                                                          */
BHS_API int bhs_dist_plan(int world, int rank, int sub_blocks, const int64_t *rows, const int64_t *cuts,
                          int64_t *ops_out, int cap_ops);

/* The same plan in the values-only mode (option "values_only"): no colInd operations; with the first group the classes
 * of a block's rows (array 3, offsets in rows) and its rank's class tables (array 4: table_ints int32 per rank, rank
 * r's at r * table_ints) travel instead.                                                                            */
BHS_API int bhs_dist_plan_values_only(int world, int rank, int sub_blocks, const int64_t *rows, const int64_t *cuts,
                                      int64_t table_ints, int64_t *ops_out, int cap_ops);

/* Options of the multi-GPU layer:
 *   "values_only"  1: when EVERY rank's multiply went by row classes (grid matrices; bhs_get_class_tables_device), the
 *                  column indices of the other ranks' blocks are not transferred but rebuilt on this GPU from their
 *                  classes -- 4 bytes per row and 8.4 MB of tables per rank cross xGMI instead of 4 bytes per entry:
 *                  8 instead of 12 bytes per entry of C on every link (poisson27pt 256^3 at 8 GPUs: 13.6 instead of
 *                  20.3 ms per link).  Decided collectively per call from the sizes exchange; any rank that cannot
 *                  serve it sends everyone to the ordinary all-gatherv.  Default 0.
 * bhs_dist_last_values_only: 1 when the last call ran that way.                                                   */
BHS_API int bhs_dist_set_option(bhs_dist *d, const char *key, int64_t value);
BHS_API int bhs_dist_last_values_only(bhs_dist *d);

/* per-link lower bound of the all-gatherv in ms: bytes this rank receives from its largest peer / 153 GB/s (one xGMI
 * link; /opt/skills/guides/MI355X_MICROARCH.md), for the sizes of the last bhs_dist_spgemm_allgatherv               */
BHS_API double bhs_dist_last_link_floor_ms(bhs_dist *d);

/* Number of ranks RCCL itself counts in the communicator (ncclCommCount): the bench line reports it so that a scaling
 * record shows that the collective really spanned N processes.  BHS_ERR_LAUNCH once the communicator was aborted. */
BHS_API int bhs_dist_nranks(bhs_dist *d, int *nranks_out);

#ifdef __cplusplus
}
#endif
#endif /* BHSPARSE_DIST_H */
