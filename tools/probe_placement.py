#!/usr/bin/env python3
"""Measurement helper (round 4): numeric_class against WHERE the output arrays are.  One process, one data set
(poisson27pt 128^3); every trial frees the library's colIndC / valC (bhs_free_data), optionally perturbs the allocator with
a dummy allocation, lets the next multiply allocate them again, and times the ring kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
dev = torch.device("cuda", 0)
Bp, Bj = gallery.poisson_csr_torch("poisson27pt", 128, 128, 128, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
rng = np.random.default_rng(1)
keep = []
for trial in range(14):
    if trial:
        bh.free_mem()
        if trial % 2 == 0:                       # perturb: hold on to a dummy of 64 MB .. 1.5 GB
            keep.append(torch.empty(int(rng.integers(1 << 23, 1 << 27)) * 2, dtype=torch.float64, device=dev))
        if len(keep) > 3: keep.pop(0)
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    for _ in range(3): assert bh.spgemm() == 0
    nc = []
    for _ in range(7):
        assert bh.spgemm() == 0
        nc.append([s["ms"] for s in bh.kernel_stats() if s["name"] == "numeric_class"][0])
    pr, pc, pv = bh.get_C_device()
    print("trial %2d: numeric_class median %.3f min %.3f ms   colIndC %#x valC %#x  (valC mod 2^30 = %#x, mod 2^21 = %#x)" % (trial, np.median(nc), np.min(nc), pc, pv, pv % (1 << 30), pv % (1 << 21)), flush=True)
