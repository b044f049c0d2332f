"""CPU tests: the C-ABI library builds for gfx950, loads, and exports every
symbol include/bhsparse_hip.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from benchmark_spgemm_using_csr_amd import _lib


def _declared_symbols():
    txt = open(_lib.HEADER).read()
    return sorted(set(re.findall(r"BHS_API\s+[\w\s\*]+?\b(bhs_\w+)\s*\(", txt)))


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert len(decl) >= 14
    assert sorted(_lib.SYMBOLS) == decl


def test_library_exports_every_declared_symbol(hiplib):
    raw = C.CDLL(_lib.SO_PATH)
    for name in _declared_symbols():
        assert getattr(raw, name) is not None


def test_float_library_exports_the_same_abi(hiplib):
    assert os.path.exists(_lib.SO_PATH_F32)
    raw = C.CDLL(_lib.SO_PATH_F32)
    for name in _declared_symbols():
        assert getattr(raw, name) is not None
    raw.bhs_version.restype = C.c_char_p
    assert b"float" in raw.bhs_version() and b"double" in hiplib.bhs_version()


def test_code_object_is_gfx950():
    so = _lib.SO_PATH
    assert os.path.exists(so)
    blob = open(so, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"k_row_wave", b"k_row_quad", b"k_row_block", b"k_row_spa", b"k_row_bitmap_lds", b"k_upper_bound"):
        assert kern in blob           # the accumulator kernels are in the fat binary


def test_strerror_and_version(hiplib):
    assert hiplib.bhs_strerror(0) == b"success"
    assert b"int32" in hiplib.bhs_strerror(_lib.BHS_ERR_NNZ_OVERFLOW)
    assert b"gfx950" in hiplib.bhs_version() and b"double" in hiplib.bhs_version()


def test_null_handle_is_rejected(hiplib):
    assert hiplib.bhs_spgemm(None, None, None, None, None) == _lib.BHS_ERR_INVALID_ARG
    assert hiplib.bhs_destroy(None) == _lib.BHS_ERR_INVALID_ARG
    assert hiplib.bhs_create(None, 1, None) == _lib.BHS_ERR_INVALID_ARG


def test_python_facade_has_reference_api():
    # SpGEMM_cuda/bhsparse.h:20-33
    from benchmark_spgemm_using_csr_amd import bhsparse
    for meth in ("initPlatform", "initData", "spgemm", "warmup", "get_nnzC", "get_C", "freePlatform", "free_mem"):
        assert callable(getattr(bhsparse, meth))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_dist_library_exports_every_declared_symbol(hiplib):
    """libbhsparse_dist.so (include/bhsparse_dist.h): the multi-GPU layer loads next to the core library and exports
    every entry the header declares; no RCCL call is made here (no GPU)."""
    hdr = os.path.join(os.path.dirname(_lib.HEADER), "bhsparse_dist.h")
    decl = sorted(set(re.findall(r"BHS_API\s+[\w\s\*]+?\b(bhs_dist_\w+)\s*\(", open(hdr).read())))
    assert len(decl) >= 7
    so = os.path.join(_lib.CSRC, "libbhsparse_dist.so")
    assert os.path.exists(so), "make -C %s" % _lib.CSRC
    raw = C.CDLL(so)
    for name in decl:
        assert getattr(raw, name) is not None
    # argument checks that need no device
    assert raw.bhs_dist_destroy(None) == _lib.BHS_ERR_INVALID_ARG
    out = (C.c_int * 3)()
    assert raw.bhs_dist_partition_rows(-1, None, None, None, 2, out) == _lib.BHS_ERR_INVALID_ARG
    assert raw.bhs_dist_partition_rows(0, None, None, None, 2, out) == 0 and list(out) == [0, 0, 0]


def test_measurement_tools_still_build(tmp_path):
    """tools/ holds stand-alone probes (.hip) and measurement scripts (.py) that nothing else imports: every probe must
    still cross-compile for gfx950 and every script must still parse, or they rot silently."""
    import glob
    import py_compile
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    probes = sorted(glob.glob(os.path.join(tools, "*.hip")))
    assert probes

    def build(src):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        p = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-I", os.path.join(os.path.dirname(tools), "benchmark_spgemm_using_csr_amd", "csrc"), "-c", "-o", obj, src],
                           capture_output=True, text=True, timeout=600)
        return src, p.returncode, p.stderr[-800:]
    with ThreadPoolExecutor(4) as ex:
        for src, rc, err in ex.map(build, probes):
            assert rc == 0, (src, err)
    for f in sorted(glob.glob(os.path.join(tools, "*.py"))):
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)
