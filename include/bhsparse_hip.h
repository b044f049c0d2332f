/*
 * bhsparse_hip.h — C-ABI of libbhsparse_hip.so, the MI355X (gfx950) CSR SpGEMM
 * backend that sits behind the bhSPARSE `bhsparse` class API.
 *
 * This is the drop-in boundary: the C++ facade (host/bhsparse.h, same public
 * signatures as the reference's SpGEMM_cuda/bhsparse.h:17-33) and the Python
 * mirror (facade.py) call ONLY these entry points.  Plain C types, opaque
 * handle, caller-owned buffers, every function returns 0 (BHS_SUCCESS ==
 * BHSPARSE_SUCCESS, SpGEMM_cuda/common.h:26) or a negative bhs_status /
 * positive hipError_t; nothing throws or aborts.  One handle = one device =
 * one in-flight multiply; calls on a handle must be externally serialised
 * (same threading contract as the reference: single host thread, synchronous
 * calls).
 *
 * Types: index_type = int32 (SpGEMM_cuda/common.h:30), value_type = bhs_value_t: double
 * (common.h:31) or float in the f32 build.  Intermediate-product counts are int64 (the reference's int
 * overflows beyond 2^31 products, bhsparse.h:367,431).
 */
#ifndef BHSPARSE_HIP_H
#define BHSPARSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define BHS_API __attribute__((visibility("default")))
#else
#define BHS_API
#endif

/* value_type of A, B and C: double (libbhsparse_hip.so; SpGEMM_cuda/common.h:31) or float
 * (libbhsparse_hip_f32.so, the same sources built with -DBHS_VALUE_FLOAT; the reference supports it by
 * editing the same typedef, README.md:84-86).  Callers of the f32 library define BHS_VALUE_FLOAT too. */
#ifdef BHS_VALUE_FLOAT
typedef float bhs_value_t;
#else
typedef double bhs_value_t;
#endif

typedef struct bhs_handle bhs_handle;

enum bhs_status {
    BHS_SUCCESS            = 0,
    BHS_ERR_INVALID_ARG    = -1,   /* NULL / negative size / bad state          */
    BHS_ERR_NO_DEVICE      = -2,   /* no HIP device, or not a gfx950 code object */
    BHS_ERR_ALLOC          = -3,   /* hipMalloc failed                           */
    BHS_ERR_LAUNCH         = -4,   /* kernel launch / runtime error (reference returns -1 here,
                                      bhsparse_cuda.h:251-253)                   */
    BHS_ERR_NNZ_OVERFLOW   = -5,   /* nnz(C) does not fit index_type (int32)     */
    BHS_ERR_NOT_READY      = -6,   /* get_C before spgemm, spgemm before set_data */
    BHS_ERR_INTERNAL       = -7,   /* accumulator overflow that the retry logic could not resolve */
    BHS_ERR_PEER           = -8    /* multi-GPU calls: another rank of the job failed (bhsparse_dist.h) */
};

/* ---- lifecycle -----------------------------------------------------------
 * replaces bhsparse::initPlatform -> bhsparse_cuda::initPlatform
 *   (SpGEMM_cuda/bhsparse.h:91-125, bhsparse_cuda.h:92-119: picks device 0,
 *   prints the device banner)   and freePlatform (bhsparse.h:127-149).
 * device_count must be 1 (one process per GPU; multi-GPU runs shard rows of A
 * across processes, see bhs_set_data on a row block).  device_ids may be NULL
 * (=> device 0).  Prints the reference-style device banner when verbose.      */
BHS_API int bhs_create(bhs_handle **out, int device_count, const int *device_ids);
BHS_API int bhs_destroy(bhs_handle *h);

/* verbosity: 0 silent, 1 reference-style stage prints (default 0 for the C-ABI;
 * the C++ facade sets 1 to reproduce bhsparse.h:307-336 stdout).              */
BHS_API int bhs_set_verbose(bhs_handle *h, int level);

/* ---- data ----------------------------------------------------------------
 * replaces bhsparse::initData -> bhsparse_cuda::initData
 *   (bhsparse.h:180-258, bhsparse_cuda.h:151-203: cudaMalloc + H2D of A and B).
 * Host pointers are read during the call only (copied to HBM); the reference
 * keeps borrowing them, which stays legal.  A is m x k, B is k x n, 0-based
 * CSR.  Rows of B should be column-sorted (reference precondition, SURVEY.md
 * §8b); unsorted B is detected and still multiplied correctly.               */
BHS_API int bhs_set_data(bhs_handle *h, int m, int k, int n,
                         int nnzA, const bhs_value_t *csrValA, const int *csrRowPtrA, const int *csrColIndA,
                         int nnzB, const bhs_value_t *csrValB, const int *csrRowPtrB, const int *csrColIndB);

/* Same, but the six arrays are DEVICE pointers on the handle's device (borrowed
 * until bhs_free_data; never written).  This is the entry the benchmark uses so
 * that inputs are HBM-resident when the timed region starts, and the entry a
 * multi-GPU host uses to hand each rank its row block of A with B replicated.
 * The library works on its own stream and starts reading the arrays inside this
 * call: they must be COMPLETE (producer kernels / copies on other streams
 * synchronised) before it is made, and stay unchanged until bhs_free_data.     */
BHS_API int bhs_set_data_device(bhs_handle *h, int m, int k, int n,
                                int nnzA, const bhs_value_t *d_valA, const int *d_rowPtrA, const int *d_colIndA,
                                int nnzB, const bhs_value_t *d_valB, const int *d_rowPtrB, const int *d_colIndB);

/* replaces bhsparse::free_mem -> bhsparse_cuda::free_mem (bhsparse.h:151-178,
 * bhsparse_cuda.h:121-149).  Drops A, B, C; keeps the workspace pool.         */
BHS_API int bhs_free_data(bhs_handle *h);

/* ---- compute -------------------------------------------------------------
 * replaces bhsparse::warmup (bhsparse.h:341-363, bhsparse_cuda.h:239-256:
 * re-runs the nnzCt kernel).  Here: runs the upper-bound kernel and pre-sizes
 * the workspace pool so the timed spgemm() does no hipMalloc.                 */
BHS_API int bhs_warmup(bhs_handle *h);

/* replaces bhsparse::spgemm / spgemm_cuda (bhsparse.h:260-339): the whole timed
 * region — stage 1 upper bound + row binning, stage 2 symbolic (exact nnz per
 * row), stage 3 scan + allocation of C, stage 4 numeric (C written once, in
 * place, rows column-sorted, duplicates summed, explicit zeros kept).
 *   rowPtrC_out : caller buffer of m+1 ints, filled with the exclusive-scan row
 *                 pointer (as bhsparse_cuda.h:2787-2799 does); may be NULL.
 *   nnzCt_out   : number of intermediate products (GFLOPs numerator,
 *                 bhsparse.h:287-289); may be NULL.
 *   nnzC_out    : nnz(C); may be NULL.
 *   stage_ms_out: 4 doubles, device time of the 4 stages in ms; may be NULL.
 * Synchronous: returns after C is complete in HBM.                            */
BHS_API int bhs_spgemm(bhs_handle *h, int *rowPtrC_out, int64_t *nnzCt_out, int *nnzC_out,
                       double stage_ms_out[4]);

/* The same multiply in two halves, for callers that place C themselves (the multi-GPU layer, include/bhsparse_dist.h):
 *   bhs_spgemm_symbolic  stages 1-3 of bhsparse::spgemm_cuda (bhsparse.h:297-325: compute_nnzCt, binning, and here the
 *                        exact symbolic pass + scan): afterwards nnz(C) and rowPtrC (bhs_get_C_device / bhs_get_rowptrC)
 *                        are final and the library's own C arrays are allocated;
 *   bhs_set_output_device (optional, between the halves) binds caller-owned device arrays of `capacity` entries: the
 *                        numeric half then writes colIndC / valC there (entry k of C at index k) instead of into the
 *                        library's pool -- e.g. straight into this rank's slice of the assembled global C.  Stays bound
 *                        until bhs_free_data or a call with NULL pointers;
 *   bhs_spgemm_numeric   stage 4 (bhsparse.h:327-335, compute_nnzC_Ct_* + copy) for rows [row_begin, row_end) of C,
 *                        enqueued on the handle's stream (bhs_get_stream): returns without waiting for the kernels, so
 *                        the caller can overlap the transfer of one row range with the numeric kernels of the next;
 *   bhs_spgemm_finish    waits for everything enqueued, collects errors raised on the device, reads the stage timers.
 * bhs_spgemm == symbolic + numeric(0, m) + finish.                                                              */
BHS_API int bhs_spgemm_symbolic(bhs_handle *h, int64_t *nnzCt_out, int *nnzC_out);
BHS_API int bhs_set_output_device(bhs_handle *h, int *d_colIndC, bhs_value_t *d_valC, int64_t capacity);
BHS_API int bhs_spgemm_numeric(bhs_handle *h, int row_begin, int row_end);
BHS_API int bhs_spgemm_finish(bhs_handle *h, double stage_ms_out[4]);
/* the HIP stream (hipStream_t) every kernel of this handle is enqueued on */
BHS_API int bhs_get_stream(bhs_handle *h, void **stream_out);

/* replaces bhsparse::get_nnzC (bhsparse.h: get_nnzC -> bhsparse_cuda::get_nnzC). */
BHS_API int bhs_get_nnzC(bhs_handle *h, int *nnzC_out);

/* replaces bhsparse::get_C -> bhsparse_cuda::get_C (bhsparse_cuda.h:3006-3020:
 * D2H of colIndC / valC; the reference also re-copies rowPtrC there — pass
 * rowPtrC_out to bhs_spgemm or use bhs_get_rowptrC).  Caller buffers hold
 * nnzC entries.                                                               */
BHS_API int bhs_get_C(bhs_handle *h, int *csrColIndC, bhs_value_t *csrValC);
BHS_API int bhs_get_rowptrC(bhs_handle *h, int *csrRowPtrC /* m+1 */);

/* Device-resident result for callers that keep C on the GPU (multi-GPU
 * all-gatherv of row blocks, chained products).  Pointers stay valid until the
 * next bhs_spgemm / bhs_free_data / bhs_destroy on this handle.               */
BHS_API int bhs_get_C_device(bhs_handle *h, const int **d_rowPtrC, const int **d_colIndC,
                             const bhs_value_t **d_valC);

/* ---- input preparation ----------------------------------------------------
 * Per-row sort of a DEVICE-resident CSR matrix by column index, in place and
 * stable: ref_spgemm::csr_sort_indices (SpGEMM_cuda/ref_spgemm.h:37-62), which
 * the reference's driver runs on the host over every Matrix Market input before
 * the multiply (main.cu:62-64).  Rows already in order are left untouched.
 * Synchronous.  (bhs_set_data / bhs_set_data_device accept unsorted rows as they
 * are; sorted rows of B let the multiply take its fastest kernels.)            */
BHS_API int bhs_csr_sort_indices_device(bhs_handle *h, int n_row, const int *d_rowPtr, int *d_colInd,
                                        bhs_value_t *d_val);

/* ---- measurement ----------------------------------------------------------
 * Per-kernel-family device times of the LAST bhs_spgemm (option "kernel_stats" = 1), measured with
 * hipEvents on the stream the kernels were launched on (what bench.py reports
 * as roofline.achieved; replaces the reference's never-enabled `_profiling`
 * prints, bhsparse_cuda.h:728-733).  Returns the number of records; fills up to
 * `cap` of them.  `name` points to a static string.                           */
typedef struct bhs_kernel_stat {
    const char *name;      /* e.g. "numeric_wave<256>"                        */
    int         launches;  /* launches of this family in the last spgemm      */
    double      ms;        /* summed device time of those launches            */
    int64_t     rows;      /* rows of C processed by them                     */
    int64_t     products;  /* intermediate products processed by them         */
    int64_t     nnz_out;   /* entries of C produced (numeric) / counted (symbolic) */
    int64_t     nnzA_rows; /* nnz of the A rows processed (for algorithmic bytes) */
} bhs_kernel_stat;
BHS_API int bhs_get_kernel_stats(bhs_handle *h, bhs_kernel_stat *out, int cap);

/* Tunables (tests use them to force a particular accumulator path; defaults are the measured best):
 *   "force_path"      0 auto | 1 no quarter-wave bin (tiny rows take the wave kernel) |
 *                     2 wave bins are served by the workgroup-per-row kernel
 *   "max_table_log2"  cap (6..15) on the LDS table size => forces the column-window path
 *   "no_pack32"       1: always 64-bit sort keys
 *   "sym_load_pct", "num_load_pct"  table load factor (5..75 %) that decides a row's bin
 *   "spa"             0: rows beyond the LDS tables use column windows instead of the bitmap accumulators
 *   "lds_bitmap"      0: keep the long-row bitmap in HBM even when the matrix has <= 2^20 columns
 *   "lds_bitmap_min_log2"  numeric workgroup bins with tables of at least 2^v slots go to the LDS bitmap
 *                     kernel when n <= 2^20 (default 12; 99: only rows beyond every table)
 *   "window_bitmap"   rows of thousands of entries of C column window by column window, a wave (2 k .. 8 k entries) or 256
 *                     lanes (beyond) per row, several rows per CU (bhs_row_window.hip.h): 1 (default) when the multiply has
 *                     >= 32 resp. >= 16 such rows per CU, 2 always, 0 never; needs ascending rows of B shorter than 2^16
 *                     and n <= 2^20
 *   "class_super_rows"  consecutive rows a wave of the class numeric kernel takes (0, default: a grid line of A where
 *                     "line_a" found one, else 64)
 *   "small_b"         0: always 64-bit address arithmetic for colIndB / valB (default: 32-bit byte offsets when
 *                     nnz(B) < 2^29)
 *   "lane_first"      1 (default): when every row of A has <= 12 entries and every row of B <= 64 (stencils), the
 *                     upper-bound pass and its host round trip are skipped; the lane-per-row symbolic kernel handles
 *                     every row and counts the products on the side
 *   "wave_first"      1 (default): when maxRow(A) x maxRow(B) fits a wave-per-row table and is within 4x of the average
 *                     row's product count (poisson27pt: 27 x 27), every row runs the symbolic wave kernel of that table
 *                     size straight from rowPtrA -- no upper-bound pass, host round trip or queue
 *   "direct_bins"     1 (default): a stage whose rows ALL sit in the lane bin or the quad bin (stencils) skips the
 *                     queue-fill pass; the kernel derives row q's descriptor from rowPtrA / rowPtrC
 *   "sort_b"          1 (default): rows of B that are not ascending are sorted at bhs_set_data[_device] time (device
 *                     pointers are borrowed and never written: the sort runs on a private copy); 0: multiply them as
 *                     they are (general kernels only).  Set it before bhs_set_data.
 *   "lane_rows"       lane-per-row symbolic kernel (k_row_lane: one row per lane, K-way merge of the sorted B rows in
 *                     registers) for rows with <= 12 entries and <= 144 products: 0 never, 1 (default) when EVERY row of
 *                     A has <= 12 entries (stencils), 2 for any matrix.  Needs B with strictly ascending rows.
 *   "lane_numeric"    the numeric stage of those rows goes through k_row_lane as well: 0 never, 1 always, 2 (default)
 *                     when no row of A has more than 8 entries (poisson5pt, 7pt; measured slower beyond)
 *   "compress_b"      symbolic pass on the compressed pattern of B ((column >> 5, mask) pairs; rows binned by their
 *                     pair count): 0 never, 1 (default) when the average row has more than 1536 products and the data
 *                     has <= 60 % as many pairs as entries (FEM-like inputs: keeps rows out of the workgroup-per-row
 *                     symbolic kernels), 2 always.  Only for B with ascending rows.  Set it before bhs_set_data for
 *                     mode 1 to be decided there.
 *   "kernel_stats"    1: every kernel family of a multiply is bracketed by a hipEvent pair for bhs_get_kernel_stats
 *                     (times of the families; launches / rows are always counted); 0 (default): only the four stage
 *                     timers are recorded -- the pairs cost 34 us of a 0.24 ms poisson5pt 1024^2 multiply, 0.16 ms of
 *                     the 3.3 ms power-law stand-in (many bins).  The reference's own `_profiling` prints are off by
 *                     default too (bhsparse_cuda.h:728-733).
 *   "concurrent_bins" the kernels of a stage's bins run concurrently on side streams: 0 never, 1 always,
 *                     2 (default) when the stage has >= 8 non-empty bins (power-law matrices)
 *   "spa_slots"       HBM bitmap slots (default: one per CU)
 *   "class_path"      row classes (bhs_class.hip.h): rows that repeat one another's relative pattern -- stencils, anything
 *                     assembled on a regular grid -- get their structure (sorted columns, entry count, product ->
 *                     position map) worked out once per class instead of once per row.  1 (default): tried on data
 *                     sets whose rows of A and B have at most 256 entries and, on average, at least
 *                     "class_min_products" (default 64) products per row of C and 1e7 products in all; every row is classified and verified
 *                     on the device, and a data set with rows that find no class (or a class of more than 8192
 *                     products / 512 entries per row of C) goes back to the general pipeline for good.
 *                     2: tried whatever the average; 0: never.
 *   "class_numeric"   numeric kernel of the classes whose product list fits a wave's registers (<= 64 entries per row
 *                     of A and B, <= 1024 products): 2 (default) round 5's ring kernel (bhs_class_ring.hip.h: sums in
 *                     registers, stored where an entry of C ends; B's values through a ring of slabs in LDS; 16 waves
 *                     per CU) wherever its LDS fits, 1 round 4's ring kernel (bhs_class_wg.hip.h), 0 always the
 *                     LDS-atomic kernel of round 2.  Multiplies with bigger classes (several unknowns per grid node)
 *                     run bhs_class_big.hip.h whatever this says.
 *   "class_heads"     2 (default): one pass per matrix -- the wave that finds a row differing from the row `period`
 *                     rows before it takes it through the class table itself (bhs_class_fused.hip.h; period: sampled
 *                     at bhs_set_data time, the unknowns per node); 1: rounds 3-4's three launches per matrix (list
 *                     the rows that differ, classify the list, hand the classes on); 0: every row through the table
 *   "class_tile"      1 (default): the classifier gives a row to ONE lane where rows have at most 32 entries (63 rows'
 *                     column indices as 16-byte loads through a tile in LDS, bhs_class_tile.hip.h); 0: G lanes per row
 *                     everywhere.  "class_tile_piece": rows a wave of it walks (0, default: one piece per wave slot)
 *   "spec_numeric"    1 (default): from a data set's second multiply on, the class path launches its numeric kernel on
 *                     the classes' figures of the multiply before (k_class_spec_check compares them with this multiply's
 *                     on the device; a refuted launch writes nothing and the multiply runs again); 0: always wait for
 *                     the read-back.  bhs_get_info: "spec_launches", "spec_refuted"
 *                     (round 6: a lane-first multiply -- every row through the lane kernels -- likewise, k_lane_spec_check;
 *                     "lane_from_counts" 1 (default): its numeric kernel then makes rowPtrC from the symbolic kernel's counts
 *                     and block sums, no scan kernel)
 *   "class_mixed"     1 (default, round 6): a row without a class -- or of a class beyond the tables, or of a class with fewer
 *                     than four rows -- goes through the general pipeline's kernels inside the same multiply while the other
 *                     rows stay on the class kernels (bhs_class_mix.hip.h; the reference bins every row for itself,
 *                     SpGEMM_cuda/bhsparse.h:483-586); 0: one such row sends the data set to the general pipeline, as until
 *                     round 5.  "class_mixed_max_pct" (default 30): more irregular rows than this share of all rows send
 *                     the data set to the general pipeline.  bhs_get_info: "class_state", "mixed_rows"
 *   "kernel_stats"    0 (default) no per-kernel timers; 1 hipEvent pairs around every kernel family (bhs_get_kernel_stats);
 *                     2 around the numeric kernels only
 *   "ring_dynamic"    the ring kernel's super-runs handed out by a counter per XCD: 0 never, 1 always, 2 (default) where other
 *                     kernels run beside it
 *   "spin_wait"       1 (default): a multiply's waits for its stream poll, with a pause between polls, for at most
 *                     "spin_wait_us" microseconds (default 0: four times the last multiply's wall time, 0.5 .. 5 ms) before
 *                     they sleep; 0: sleep at once
 *   "hub_min_products"  rows with at least this many intermediate products are split across workgroups
 *                     (bhs_hub.hip.h: items of "hub_item_products" products handed out to the whole device, one shared
 *                     bitmap slot per row); default 131072, 0 never.  "hub_item_products" (default 8192, >= 64),
 *                     "hub_slots" (rows per batch; default: as many as fit 1/16 of the device memory)
 *   "wg_per_cu"       persistent workgroups per CU of the wave kernels (default: occupancy API)
 *   "verbose"         same as bhs_set_verbose
 * Returns BHS_ERR_INVALID_ARG for unknown keys.                               */
BHS_API int bhs_set_option(bhs_handle *h, const char *key, int64_t value);

/* What the library found out about the bound data set (after bhs_set_data[_device]):
 *   "b_sorted"   1 when every row of B (as multiplied: after the optional sort) is strictly ascending
 *   "max_row_a", "max_row_b"   longest row of A / B
 *   "local_a"    1 when sampled rows of A keep their entries near the diagonal (mean |column - row| < columns / 16): the
 *                lane-per-row kernels are only chosen then
 *   "line_a"     rows per grid line of A when the places where a row's length changes repeat with a fixed period and the
 *                matrix is a whole number of such lines (a wave of the class numeric kernel then takes whole lines), else 0
 *   "compress_b_used"  1 when the symbolic pass of the general pipeline runs on B's pattern compressed to (column block,
 *                mask) pairs for this data set
 *   "class_state"  which pipeline this data set's multiplies take: 1 row classes, 2 row classes with irregular rows on the
 *                general pipeline's kernels (mixed mode), -1 the general pipeline (for good: until the next bhs_set_data)
 *   "mixed_rows"   rows of the last multiply that had no class and went through the general pipeline's kernels (0: none)
 * Returns BHS_ERR_INVALID_ARG for unknown keys, BHS_ERR_NOT_READY without data.  */
BHS_API int bhs_get_info(bhs_handle *h, const char *key, int64_t *value_out);

/* Row classes of the open / last multiply, for a caller that rebuilds column indices itself instead of moving them
 * (bhs_dist's values-only all-gatherv: on a grid matrix colIndC of a row is its class's relative column list plus the
 * row number, so only valC has to cross xGMI).  No reference counterpart (the reference has no classes, no second GPU).
 *   d_classC      int32[m]: class of every row of A (= row of C); valid from bhs_spgemm_symbolic until the next multiply
 *   d_classInfo   16 bytes per class slot: int32 {entries of the A row, products, entries of the C row, representative}
 *   d_classRel    int32[slots * rel_stride]: the class's columns relative to the row, ascending
 *   usable_out    0: that multiply did not go by row classes with register-sized tables -- nothing above is valid
 * Device pointers into the handle's workspace.  BHS_ERR_NOT_READY without an open or finished multiply.            */
BHS_API int bhs_get_class_tables_device(bhs_handle *h, const int **d_classC, const void **d_classInfo,
                                        const int **d_classRel, int *slots_out, int *rel_stride_out, int *usable_out);

/* colIndC of n consecutive rows from their classes, on `stream` (a hipStream_t; NULL: the device's null stream) of the
 * CURRENT device: row i of the n rows has class d_classC[i], classInfo[class].z entries, and its entry s is column
 * d_classRel[class * rel_stride + s] + row0 + i, written at d_colIndC[d_rowPtrC[i] + s].  The tables may be another
 * handle's or another GPU's, copied over (bhs_dist).  Asynchronous.                                                  */
BHS_API int bhs_expand_class_columns_device(void *stream, int n, int row0, const int *d_classC, const void *d_classInfo,
                                            const int *d_classRel, int rel_stride, const int *d_rowPtrC, int *d_colIndC);

BHS_API const char *bhs_strerror(int status);
BHS_API const char *bhs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BHSPARSE_HIP_H */
