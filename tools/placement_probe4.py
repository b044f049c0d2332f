#!/usr/bin/env python3
"""Measurement helper (round 5): numeric_class on output arrays whose physical chunks are mapped in shuffled order (tools/vmm_alloc.hip)
against plain hipMalloc, one process (poisson27pt 128^3)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
vmm = C.CDLL(os.path.join(ROOT, "gpurun_variants", "libvmm.so"))
vmm.vmm_alloc.restype = C.c_void_p
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, C.c_int]
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
dev = torch.device("cuda", 0)
Bp, Bj = gallery.poisson_csr_torch("poisson27pt", 128, 128, 128, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
assert bh.spgemm() == 0
nnzC = bh.get_nnzC()
def run(tag, pj, px):
    assert bh.set_output_device(pj, px, nnzC) == 0
    for _ in range(2): assert bh.spgemm() == 0
    nc = []
    for _ in range(5):
        assert bh.spgemm() == 0
        nc.append([s["ms"] for s in bh.kernel_stats() if s["name"] == "numeric_class"][0])
    print("%-34s numeric_class %.3f ms" % (tag, np.median(nc)), flush=True)
    assert bh.set_output_device(None, None, 0) == 0
    torch.cuda.synchronize()
for trial in range(3):
    p1, p2 = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(p1), nnzC * 4) == 0 and hip.hipMalloc(C.byref(p2), nnzC * 8) == 0
    run("hipMalloc", p1.value, p2.value)
    for chunk, shuffle in ((2 << 20, 1), (2 << 20, 0), (64 << 20, 1), (256 << 10, 1)):
        pj = vmm.vmm_alloc(nnzC * 4, chunk, 11 + trial, shuffle); px = vmm.vmm_alloc(nnzC * 8, chunk, 23 + trial, shuffle)
        if not pj or not px: print("vmm failed"); continue
        run("vmm chunk %d KB shuffle %d" % (chunk >> 10, shuffle), pj, px)
