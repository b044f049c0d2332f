"""ctypes binding of libbhsparse_hip.so (include/bhsparse_hip.h).

No fallback: if the HIP library is missing or fails to load this raises — the
product path never routes through a CPU implementation.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# BHSPARSE_HIP_LIB: alternate build of the same library (measurement variants); never a fallback
SO_PATH = os.environ.get("BHSPARSE_HIP_LIB") or os.path.join(CSRC, "libbhsparse_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "bhsparse_hip.h")

BHS_SUCCESS = 0
BHS_ERR_INVALID_ARG = -1
BHS_ERR_NO_DEVICE = -2
BHS_ERR_ALLOC = -3
BHS_ERR_LAUNCH = -4
BHS_ERR_NNZ_OVERFLOW = -5
BHS_ERR_NOT_READY = -6
BHS_ERR_INTERNAL = -7
BHS_ERR_PEER = -8


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int), ("ms", C.c_double),
                ("rows", C.c_int64), ("products", C.c_int64), ("nnz_out", C.c_int64),
                ("nnzA_rows", C.c_int64)]


# every symbol include/bhsparse_hip.h declares: (restype, argtypes)
_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
SYMBOLS = {
    "bhs_create": (_i, [C.POINTER(_vp), _i, C.POINTER(_i)]),
    "bhs_destroy": (_i, [_vp]),
    "bhs_set_verbose": (_i, [_vp, _i]),
    "bhs_set_data": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "bhs_set_data_device": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "bhs_free_data": (_i, [_vp]),
    "bhs_warmup": (_i, [_vp]),
    "bhs_spgemm": (_i, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i), C.POINTER(C.c_double)]),
    "bhs_spgemm_symbolic": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i)]),
    "bhs_set_output_device": (_i, [_vp, _vp, _vp, _i64]),
    "bhs_spgemm_numeric": (_i, [_vp, _i, _i]),
    "bhs_spgemm_finish": (_i, [_vp, C.POINTER(C.c_double)]),
    "bhs_get_stream": (_i, [_vp, C.POINTER(_vp)]),
    "bhs_get_nnzC": (_i, [_vp, C.POINTER(_i)]),
    "bhs_get_C": (_i, [_vp, _vp, _vp]),
    "bhs_get_rowptrC": (_i, [_vp, _vp]),
    "bhs_get_C_device": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "bhs_csr_sort_indices_device": (_i, [_vp, _i, _vp, _vp, _vp]),
    "bhs_get_kernel_stats": (_i, [_vp, C.POINTER(KernelStat), _i]),
    "bhs_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "bhs_get_info": (_i, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "bhs_get_class_tables_device": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "bhs_expand_class_columns_device": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "bhs_strerror": (C.c_char_p, [_i]),
    "bhs_version": (C.c_char_p, []),
}

# value_type float build of the same sources (-DBHS_VALUE_FLOAT; README.md:84-86 of the reference)
SO_PATH_F32 = os.path.join(CSRC, "libbhsparse_hip_f32.so")
_lib = None
_libs = {}


SOURCES = ("bhsparse_hip.hip", "bhs_host_launch.inc.h", "bhs_host_pipeline.inc.h", "bhs_host_setdata.inc.h", "bhs_host_cabi.inc.h", "bhs_kernels.hip.h", "bhs_row_wg.hip.h", "bhs_row_window.hip.h", "bhs_row_wave.hip.h", "bhs_row_quad.hip.h", "bhs_compress.hip.h", "bhs_row_lane.hip.h", "bhs_sort.hip.h", "bhs_hub.hip.h", "bhs_class.hip.h", "bhs_class_mix.hip.h", "bhs_class_wg.hip.h", "bhs_class_ring.hip.h", "bhs_class_fused.hip.h", "bhs_class_tile.hip.h", "bhs_class_big.hip.h", "bhs_wave.hip.h", "bhs_lab.hip.h")


def source_digest():
    """Short hash of the device sources: keys measurements (profiles/hbm_traffic.json) to the build they came from."""
    import hashlib
    hsh = hashlib.sha256()
    for f in SOURCES:
        path = os.path.join(CSRC, f)
        if os.path.exists(path):
            hsh.update(f.encode())
            hsh.update(open(path, "rb").read())
    return hsh.hexdigest()[:16]


def build(force=False):
    """Compile libbhsparse_hip.so and libbhsparse_hip_f32.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))] + [HEADER]
    outs = [os.path.join(CSRC, "libbhsparse_hip.so"), SO_PATH_F32]
    stale = any(not os.path.exists(o) or any(os.path.getmtime(s) > os.path.getmtime(o) for s in srcs) for o in outs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j2"] + (["-B"] if force else []))
    return SO_PATH


def load(f32=False):
    """The double library (default) or the float one.  Raises if it is not built: no fallback."""
    global _lib
    path = SO_PATH_F32 if f32 else SO_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError(
                "%s is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C %s`. There is no CPU fallback." % (os.path.basename(path), path, CSRC))
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64, and two HSA runtimes in one
        # process cannot both own the GPU (the one initialised second reports no device).  When torch is
        # installed, let it load its runtime first: this library then binds to the same copy by soname.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)      # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _libs[path] = L
    if not f32:
        _lib = _libs[path]
    return _libs[path]


def strerror(code):
    return load().bhs_strerror(int(code)).decode()
