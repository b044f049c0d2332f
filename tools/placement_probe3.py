#!/usr/bin/env python3
"""Measurement helper (round 5): numeric_class on output arrays from hipExtMallocWithFlags(hipDeviceMallocContiguous) against plain
hipMalloc, alternating, one process (poisson27pt 128^3)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
Bp, Bj = gallery.poisson_csr_torch("poisson27pt", 128, 128, 128, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0   # (BHSPARSE_HIP_LIB picks the build)
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
assert bh.spgemm() == 0
nnzC = bh.get_nnzC()
rng = np.random.default_rng(3)
keep = []
def alloc(nbytes, contiguous):
    p = C.c_void_p()
    e = hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4) if contiguous else hip.hipMalloc(C.byref(p), nbytes)
    return (p.value if e == 0 else None), e
for trial in range(16):
    contiguous = trial % 2 == 1
    pj, e1 = alloc(nnzC * 4, contiguous); px, e2 = alloc(nnzC * 8, contiguous)
    if pj is None or px is None:
        print("trial %2d: %s allocation failed (%d, %d)" % (trial, "contiguous" if contiguous else "plain", e1, e2), flush=True)
        for p in (pj, px):
            if p: hip.hipFree(C.c_void_p(p))
        continue
    assert bh.set_output_device(pj, px, nnzC) == 0
    for _ in range(2): assert bh.spgemm() == 0
    nc = []
    for _ in range(5):
        assert bh.spgemm() == 0
        nc.append([s["ms"] for s in bh.kernel_stats() if s["name"] == "numeric_class"][0])
    print("trial %2d: %-10s numeric_class %.3f ms   colIndC %#x valC %#x" % (trial, "contiguous" if contiguous else "hipMalloc", np.median(nc), pj, px), flush=True)
    assert bh.set_output_device(None, None, 0) == 0
    torch.cuda.synchronize()
    hip.hipFree(C.c_void_p(pj)); hip.hipFree(C.c_void_p(px))
    if trial % 4 == 3: keep.append(torch.empty(int(rng.integers(1 << 24, 1 << 27)), dtype=torch.float64, device=dev))
    if len(keep) > 2: keep.pop(0); torch.cuda.empty_cache()
