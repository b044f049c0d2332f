#!/usr/bin/env python3
"""Measurement helper (round 5): does a plain fill of the output arrays tell where numeric_class will be slow?  One process, poisson27pt
128^3; every trial allocates colIndC / valC anew (torch.empty: hipMalloc behind the caching allocator, emptied between trials),
binds them (bhs_set_output_device), times a fill of both and the ring kernel on them."""
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
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
assert bh.spgemm() == 0
nnzC = bh.get_nnzC()
rng = np.random.default_rng(3)
keep = []
def timed(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for trial in range(12):
    cj = torch.empty(nnzC, dtype=torch.int32, device=dev); cx = torch.empty(nnzC, dtype=torch.float64, device=dev)
    fill = timed(lambda: (cj.fill_(1), cx.fill_(1.0)))
    copy = timed(lambda: cx.copy_(cx.roll(0)) if False else cx.mul_(1.0))          # read + write of valC in place
    assert bh.set_output_device(cj, cx, nnzC) == 0
    for _ in range(2): assert bh.spgemm() == 0
    nc = []
    for _ in range(5):
        assert bh.spgemm() == 0
        nc.append([s["ms"] for s in bh.kernel_stats() if s["name"] == "numeric_class"][0])
    print("trial %2d: fill %.3f ms  in-place scale of valC %.3f ms  numeric_class %.3f ms   valC %#x" % (trial, fill, copy, np.median(nc), cx.data_ptr()), flush=True)
    assert bh.set_output_device(None, None, 0) == 0
    del cj, cx
    if trial % 2 == 0: keep.append(torch.empty(int(rng.integers(1 << 24, 1 << 27)), dtype=torch.float64, device=dev))
    if len(keep) > 2: keep.pop(0)
    torch.cuda.empty_cache()
