"""ctypes wrapper of the CPU oracle (oracle/ref_spgemm_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of ref_spgemm_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
package `benchmark_spgemm_using_csr_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_spgemm.so")
_lib = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "ref_spgemm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle_spgemm.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_max_threads.restype = C.c_int
        L.oracle_csr_sort_indices.argtypes = [C.c_int32, _i32p, _i32p, _f64p]
        L.oracle_csr_sort_indices.restype = None
        L.oracle_nnzCt.argtypes = [C.c_int32, _i32p, _i32p, _i32p, C.c_void_p]
        L.oracle_nnzCt.restype = C.c_int64
        L.oracle_spgemm_symbolic.argtypes = [C.c_int32] * 3 + [_i32p] * 4 + [_i64p, C.c_int]
        L.oracle_spgemm_symbolic.restype = C.c_int64
        L.oracle_spgemm_numeric.argtypes = ([C.c_int32] * 3 + [_i32p, _i32p, _f64p] * 2 +
                                            [_i64p, _i32p, _f64p, C.c_int])
        L.oracle_spgemm_numeric.restype = None
        L.oracle_compare.argtypes = [C.c_int32, C.c_int64, _i64p, _i32p, _f64p,
                                     C.c_int64, _i32p, _i32p, _f64p, C.c_double, _i64p]
        L.oracle_compare.restype = None
        L.oracle_digest.argtypes = [C.c_int32, _i64p, _i32p, _f64p, _u64p]
        L.oracle_digest.restype = None
        _lib = L
    return _lib


def max_threads():
    return int(lib().oracle_max_threads())


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def csr_sort_indices(rowptr, col, val):
    """In-place per-row sort by column (ref_spgemm.h:37-62)."""
    assert col.dtype == np.int32 and val.dtype == np.float64 and col.flags.c_contiguous
    lib().oracle_csr_sort_indices(len(rowptr) - 1, _c(rowptr, np.int32), col, val)


def nnzCt(Ap, Aj, Bp, want_ub=False):
    m = len(Ap) - 1
    ub = np.empty(m, np.int64) if want_ub else None
    tot = lib().oracle_nnzCt(m, _c(Ap, np.int32), _c(Aj, np.int32), _c(Bp, np.int32),
                             ub.ctypes.data if want_ub else None)
    return (int(tot), ub) if want_ub else int(tot)


def spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, nthreads=0):
    """C = A*B, structural, rows column-sorted.  Returns (Cp int64[m+1], Cj int32, Cx f64)."""
    Ap, Aj, Ax = _c(Ap, np.int32), _c(Aj, np.int32), _c(Ax, np.float64)
    Bp, Bj, Bx = _c(Bp, np.int32), _c(Bj, np.int32), _c(Bx, np.float64)
    Cp = np.zeros(m + 1, np.int64)
    nnzC = lib().oracle_spgemm_symbolic(m, k, n, Ap, Aj, Bp, Bj, Cp, nthreads)
    Cj = np.empty(max(nnzC, 1), np.int32)[:nnzC]
    Cx = np.empty(max(nnzC, 1), np.float64)[:nnzC]
    Cj = np.ascontiguousarray(Cj)
    Cx = np.ascontiguousarray(Cx)
    lib().oracle_spgemm_numeric(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, Cp, Cj, Cx, nthreads)
    return Cp, Cj, Cx


STAGES = {0: "nnzC NO PASS", 1: "RowPtrC NO PASS", 2: "ColIndC/csrValC NO PASS", 3: "PASS"}


def compare(ref, got, rel_tol=1e-6):
    """ref=(Cp int64, Cj, Cx) from spgemm(); got=(rowPtrC int32, colIndC, valC) from the HIP path.
    Mirrors compData's order of checks (ref_spgemm.h:79-126). Returns dict."""
    rCp, rCj, rCx = ref
    gCp, gCj, gCx = got
    m = len(rCp) - 1
    out = np.zeros(4, np.int64)
    nn = int(gCp[-1]) if len(gCp) else 0
    lib().oracle_compare(m, int(rCp[-1]), _c(rCp, np.int64), _c(rCj, np.int32), _c(rCx, np.float64),
                         nn, _c(gCp, np.int32), _c(gCj, np.int32), _c(gCx, np.float64),
                         float(rel_tol), out)
    return {"stage": int(out[0]), "verdict": STAGES[int(out[0])], "rowptr_err": int(out[1]),
            "col_err": int(out[2]), "val_err": int(out[3]), "ok": int(out[0]) == 3}


def digest(Cp, Cj, Cx):
    d = np.zeros(4, np.uint64)
    lib().oracle_digest(len(Cp) - 1, _c(Cp, np.int64), _c(Cj, np.int32), _c(Cx, np.float64), d)
    return [int(x) for x in d]
