"""Python mirror of the reference's `bhsparse` facade class
(SpGEMM_cuda/bhsparse.h:17-33): same method names, argument order, `int`
return codes (0 == BHSPARSE_SUCCESS, common.h:26) and call sequence

    initPlatform -> initData -> [warmup x3] -> spgemm -> get_nnzC -> get_C
                 -> free_mem -> freePlatform            (main.cu:104-135)

so that the parity tests read like the reference's own driver.  Every method
is a thin call into the C-ABI of libbhsparse_hip.so; arrays are numpy buffers
(host entry, like the reference) or raw device pointers / torch tensors
(`initData_device`, used by the benchmark and the multi-GPU path).
"""
import ctypes as C

import numpy as np

from . import _lib

BHSPARSE_SUCCESS = 0
NUM_PLATFORMS = 9          # common.h:33
BHSPARSE_CUDA = 1          # common.h:36  (accepted as an alias of the HIP backend)
BHSPARSE_OPENCL = 2        # common.h:37  (alias)
BHSPARSE_HIP = 3           # new slot; indices 3..8 are free in the reference


class BhsparseError(RuntimeError):
    def __init__(self, where, code):
        self.code = int(code)
        super().__init__("%s failed: %d (%s)" % (where, code, _lib.strerror(code)))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):          # torch tensor (device or host)
        return C.c_void_p(a.data_ptr())
    raise TypeError("unsupported buffer type %r" % type(a))


class bhsparse(object):
    """index_type = int32, value_type = float64 (common.h:30-31), or float32 with
    `bhsparse(value_dtype=np.float32)` (libbhsparse_hip_f32.so, the reference's float build)."""

    def __init__(self, value_dtype=np.float64):
        self._vdt = np.dtype(value_dtype)
        if self._vdt not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("value_type is double or float")
        self._h = None
        self._lib = None
        self._m = 0
        self._rowptrC = None
        self._keep = None
        self.nnzCt = 0
        self.nnzC = 0
        self.stage_ms = [0.0] * 4
        self.time_ms = 0.0
        self.quiet = True

    # -- bhsparse.h:91-125 -------------------------------------------------
    def initPlatform(self, spgemm_platform, device=0):
        plats = list(spgemm_platform)
        if not any(plats[i] for i in (BHSPARSE_CUDA, BHSPARSE_OPENCL, BHSPARSE_HIP) if i < len(plats)):
            return _lib.BHS_ERR_INVALID_ARG
        self._lib = _lib.load(f32=self._vdt == np.dtype(np.float32))   # raises if missing: no fallback
        h = C.c_void_p()
        dev = C.c_int(int(device))
        err = self._lib.bhs_create(C.byref(h), 1, C.byref(dev))
        if err != BHSPARSE_SUCCESS:
            return err
        self._h = h
        # this mirror reports per-kernel times (kernel_stats()): tests and bench.py read them; C callers leave it off
        self._lib.bhs_set_option(self._h, b"kernel_stats", 1)
        if not self.quiet:
            self._lib.bhs_set_verbose(self._h, 1)
        return BHSPARSE_SUCCESS

    # -- bhsparse.h:180-258 ------------------------------------------------
    def initData(self, m, k, n, nnzA, csrValA, csrRowPtrA, csrColIndA,
                 nnzB, csrValB, csrRowPtrB, csrColIndB, csrRowPtrC, use_host_mem=False):
        """Host buffers (numpy), as the reference.  csrRowPtrC: caller-allocated
        int32[m+1], filled by spgemm().  `use_host_mem` mirrors the OpenCL
        variant's trailing flag (SpGEMM_opencl/bhsparse.h:44-47) and is ignored."""
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        for a, dt in ((csrValA, self._vdt), (csrRowPtrA, np.int32), (csrColIndA, np.int32),
                      (csrValB, self._vdt), (csrRowPtrB, np.int32), (csrColIndB, np.int32)):
            if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous):
                return _lib.BHS_ERR_INVALID_ARG
        if csrRowPtrC is not None and not (isinstance(csrRowPtrC, np.ndarray) and
                                           csrRowPtrC.dtype == np.int32 and csrRowPtrC.size >= m + 1):
            return _lib.BHS_ERR_INVALID_ARG
        self._m = m
        self._rowptrC = csrRowPtrC
        return self._lib.bhs_set_data(self._h, m, k, n, nnzA, _ptr(csrValA), _ptr(csrRowPtrA), _ptr(csrColIndA),
                                      nnzB, _ptr(csrValB), _ptr(csrRowPtrB), _ptr(csrColIndB))

    def initData_device(self, m, k, n, nnzA, d_valA, d_rowPtrA, d_colIndA,
                        nnzB, d_valB, d_rowPtrB, d_colIndB):
        """Device-resident inputs (torch tensors on this handle's GPU, or raw
        device addresses).  Borrowed until free_mem()."""
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        self._m = m
        self._rowptrC = None
        self._keep = (d_valA, d_rowPtrA, d_colIndA, d_valB, d_rowPtrB, d_colIndB)
        # The library reads these arrays on ITS stream, starting inside this call (row-length and sortedness
        # scans).  torch tensors may still be being written by kernels queued on torch's stream: wait for them.
        # (Without this a multiply could read half-written row pointers: observed as a memory fault or a hang
        # when freshly generated inputs landed in recycled memory.)
        if any(hasattr(t, "is_cuda") and t.is_cuda for t in self._keep):
            import torch
            torch.cuda.synchronize(self._keep[1].device if hasattr(self._keep[1], "device") else None)
        return self._lib.bhs_set_data_device(self._h, m, k, n, nnzA, _ptr(d_valA), _ptr(d_rowPtrA),
                                             _ptr(d_colIndA), nnzB, _ptr(d_valB), _ptr(d_rowPtrB),
                                             _ptr(d_colIndB))

    # -- bhsparse.h:341-363 ------------------------------------------------
    def warmup(self):
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        return self._lib.bhs_warmup(self._h)

    # -- bhsparse.h:260-295 ------------------------------------------------
    def spgemm(self):
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        import time
        nnzCt, nnzC = C.c_int64(0), C.c_int(0)
        st = (C.c_double * 4)()
        t0 = time.perf_counter()
        err = self._lib.bhs_spgemm(self._h, _ptr(self._rowptrC), C.byref(nnzCt), C.byref(nnzC), st)
        self.time_ms = (time.perf_counter() - t0) * 1e3
        if err != BHSPARSE_SUCCESS:
            if not self.quiet:
                print("spgemm error = %d" % err)
            return err
        self.nnzCt, self.nnzC, self.stage_ms = int(nnzCt.value), int(nnzC.value), list(st)
        if not self.quiet:
            # bhsparse.h:287-289, tag changed from [ CUDA ] to [ HIP ]
            print("[ HIP ] SpGEMM time: %g ms. Gflops = %g" %
                  (self.time_ms, 2.0 * self.nnzCt / (self.time_ms * 1.0e6)))
        return BHSPARSE_SUCCESS

    def get_nnzC(self):
        if self._h is None:
            return 0
        v = C.c_int(0)
        err = self._lib.bhs_get_nnzC(self._h, C.byref(v))
        return int(v.value) if err == BHSPARSE_SUCCESS else 0

    # -- bhsparse_cuda.h:3006-3020 ------------------------------------------
    def get_C(self, csrColIndC, csrValC):
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        nnz = self.get_nnzC()
        for a, dt in ((csrColIndC, np.int32), (csrValC, self._vdt)):
            if nnz and not (isinstance(a, np.ndarray) and a.dtype == dt and a.size >= nnz and
                            a.flags.c_contiguous):
                return _lib.BHS_ERR_INVALID_ARG
        err = self._lib.bhs_get_C(self._h, _ptr(csrColIndC), _ptr(csrValC))
        if err == BHSPARSE_SUCCESS and self._rowptrC is not None:
            err = self._lib.bhs_get_rowptrC(self._h, _ptr(self._rowptrC))   # reference re-copies rowPtrC here
        return err

    # -- the multiply in two halves (include/bhsparse_hip.h): multi-GPU callers place C themselves
    def spgemm_symbolic(self):
        nnzCt, nnzC = C.c_int64(0), C.c_int(0)
        err = self._lib.bhs_spgemm_symbolic(self._h, C.byref(nnzCt), C.byref(nnzC))
        if err == BHSPARSE_SUCCESS:
            self.nnzCt, self.nnzC = int(nnzCt.value), int(nnzC.value)
        return err

    def set_output_device(self, d_colIndC, d_valC, capacity):
        return self._lib.bhs_set_output_device(self._h, _ptr(d_colIndC), _ptr(d_valC), int(capacity))

    def spgemm_numeric(self, row_begin, row_end):
        return self._lib.bhs_spgemm_numeric(self._h, int(row_begin), int(row_end))

    def spgemm_finish(self):
        st = (C.c_double * 4)()
        err = self._lib.bhs_spgemm_finish(self._h, st)
        if err == BHSPARSE_SUCCESS:
            self.stage_ms = list(st)
        return err

    def get_C_device(self):
        """(rowPtrC, colIndC, valC) device addresses of the last result."""
        p = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
        err = self._lib.bhs_get_C_device(self._h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]))
        if err != BHSPARSE_SUCCESS:
            raise BhsparseError("bhs_get_C_device", err)
        return tuple(int(x.value or 0) for x in p)

    def class_tables_device(self):
        """bhs_get_class_tables_device: (classC, classInfo, classRel) device addresses, slots, rel stride, usable."""
        p = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
        slots, stride, usable = C.c_int(0), C.c_int(0), C.c_int(0)
        err = self._lib.bhs_get_class_tables_device(self._h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(slots),
                                                    C.byref(stride), C.byref(usable))
        if err != BHSPARSE_SUCCESS:
            raise BhsparseError("bhs_get_class_tables_device", err)
        return tuple(int(x.value or 0) for x in p) + (slots.value, stride.value, bool(usable.value))

    def expand_class_columns_device(self, n, row0, d_classC, d_classInfo, d_classRel, rel_stride, d_rowPtrC, d_colIndC, stream=0):
        """bhs_expand_class_columns_device: colIndC of n rows from their classes (asynchronous on `stream`)."""
        return self._lib.bhs_expand_class_columns_device(C.c_void_p(stream), n, row0, C.c_void_p(d_classC), C.c_void_p(d_classInfo),
                                                         C.c_void_p(d_classRel), rel_stride, C.c_void_p(d_rowPtrC), C.c_void_p(d_colIndC))

    def csr_sort_indices_device(self, n_row, d_rowPtr, d_colInd, d_val):
        """In-place, stable per-row sort by column of a device-resident CSR matrix
        (ref_spgemm::csr_sort_indices, SpGEMM_cuda/ref_spgemm.h:37-62, on the GPU)."""
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        if any(hasattr(t, "is_cuda") and t.is_cuda for t in (d_rowPtr, d_colInd, d_val)):
            import torch
            torch.cuda.synchronize()                       # the library works on its own stream (see initData_device)
        return self._lib.bhs_csr_sort_indices_device(self._h, n_row, _ptr(d_rowPtr), _ptr(d_colInd), _ptr(d_val))

    def get_rowptrC(self, out=None):
        out = np.empty(self._m + 1, np.int32) if out is None else out
        err = self._lib.bhs_get_rowptrC(self._h, _ptr(out))
        if err != BHSPARSE_SUCCESS:
            raise BhsparseError("bhs_get_rowptrC", err)
        return out

    def kernel_stats_raw(self, arr):
        """bhs_get_kernel_stats into a caller's (_lib.KernelStat * 64)(): the number of records (decode_kernel_stats reads
        them later -- a timed loop pays one C call per multiply, not a dozen dictionaries)"""
        return self._lib.bhs_get_kernel_stats(self._h, arr, 64)

    @staticmethod
    def decode_kernel_stats(arr, nrec):
        return [{"name": arr[i].name.decode(), "launches": arr[i].launches, "ms": arr[i].ms, "rows": arr[i].rows,
                 "products": arr[i].products, "nnz_out": arr[i].nnz_out, "nnzA_rows": arr[i].nnzA_rows}
                for i in range(min(nrec, 64))]

    def kernel_stats(self):
        arr = (_lib.KernelStat * 64)()
        nrec = self._lib.bhs_get_kernel_stats(self._h, arr, 64)
        out = []
        for i in range(min(nrec, 64)):
            s = arr[i]
            out.append({"name": s.name.decode(), "launches": s.launches, "ms": s.ms, "rows": s.rows,
                        "products": s.products, "nnz_out": s.nnz_out, "nnzA_rows": s.nnzA_rows})
        return out

    def set_option(self, key, value):
        return self._lib.bhs_set_option(self._h, key.encode(), int(value))

    def get_info(self, key):
        """bhs_get_info: what the library found out about the bound data set ("b_sorted", "max_row_a", "max_row_b")"""
        v = C.c_int64(0)
        err = self._lib.bhs_get_info(self._h, key.encode(), C.byref(v))
        if err:
            raise BhsparseError("get_info(%s)" % key, err)
        return v.value

    # -- bhsparse.h:151-178 / 127-149 ----------------------------------------
    def free_mem(self):
        if self._h is None:
            return _lib.BHS_ERR_NOT_READY
        self._keep = None
        return self._lib.bhs_free_data(self._h)

    def freePlatform(self):
        if self._h is None:
            return BHSPARSE_SUCCESS
        err = self._lib.bhs_destroy(self._h)
        self._h = None
        return err


def spgemm_csr(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, device=0, warmups=0, options=None, value_dtype=np.float64):
    """Convenience: run the reference call sequence once on host CSR arrays and
    return (rowPtrC int32[m+1], colIndC int32[nnzC], valC value_dtype[nnzC], info)."""
    plats = [False] * NUM_PLATFORMS
    plats[BHSPARSE_HIP] = True
    bh = bhsparse(value_dtype=value_dtype)
    err = bh.initPlatform(plats, device=device)
    if err:
        raise BhsparseError("initPlatform", err)
    try:
        for key, val in (options or {}).items():
            err = bh.set_option(key, val)
            if err:
                raise BhsparseError("set_option(%s)" % key, err)
        Ap, Aj, Ax = (np.ascontiguousarray(Ap, np.int32), np.ascontiguousarray(Aj, np.int32),
                      np.ascontiguousarray(Ax, value_dtype))
        Bp, Bj, Bx = (np.ascontiguousarray(Bp, np.int32), np.ascontiguousarray(Bj, np.int32),
                      np.ascontiguousarray(Bx, value_dtype))
        Cp = np.zeros(m + 1, np.int32)
        err = bh.initData(m, k, n, len(Aj), Ax, Ap, Aj, len(Bj), Bx, Bp, Bj, Cp)
        if err:
            raise BhsparseError("initData", err)
        for _ in range(warmups):
            err = bh.warmup()
            if err:
                raise BhsparseError("warmup", err)
        err = bh.spgemm()
        if err:
            raise BhsparseError("spgemm", err)
        nnzC = bh.get_nnzC()
        Cj = np.empty(nnzC, np.int32)
        Cx = np.empty(nnzC, value_dtype)
        err = bh.get_C(Cj, Cx)
        if err:
            raise BhsparseError("get_C", err)
        info = {"nnzCt": bh.nnzCt, "nnzC": nnzC, "stage_ms": bh.stage_ms, "time_ms": bh.time_ms,
                "kernels": bh.kernel_stats(), "mixed_rows": bh.get_info("mixed_rows"),
                "class_state": bh.get_info("class_state")}
        err = bh.free_mem()
        if err:
            raise BhsparseError("free_mem", err)
    finally:
        bh.freePlatform()
    return Cp, Cj, Cx, info
