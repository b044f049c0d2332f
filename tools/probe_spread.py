#!/usr/bin/env python3
"""Measurement helper (round 4): where the run-to-run spread of the headline comes from.
  * the same handle with and without the per-kernel event pairs (kernel_stats);
  * fresh handles in one process, with the device addresses of their output arrays;
  * the first multiply after bhs_set_data_device against the ones after it."""
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

def timed(bh, n=10):
    t = []
    for _ in range(n):
        torch.cuda.synchronize(); q = time.perf_counter(); assert bh.spgemm() == 0; t.append((time.perf_counter() - q) * 1e3)
    return np.array(t)

def new_handle(stats):
    bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
    assert bh.set_option("kernel_stats", stats) == 0
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    return bh

bh = new_handle(1)
for _ in range(3): assert bh.spgemm() == 0
for rep in range(3):
    for st in (1, 0):
        assert bh.set_option("kernel_stats", st) == 0
        assert bh.spgemm() == 0
        t = timed(bh)
        ks = {s["name"]: round(s["ms"], 3) for s in bh.kernel_stats()} if st else {}
        print("same handle, kernel_stats=%d: median %.3f min %.3f max %.3f ms  %s" % (st, np.median(t), t.min(), t.max(), ks.get("numeric_class", "")), flush=True)
# first multiply after a hand-over
for rep in range(3):
    torch.cuda.synchronize(); q = time.perf_counter()
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    t_set = (time.perf_counter() - q) * 1e3
    first = timed(bh, 1)[0]; second = timed(bh, 1)[0]; rest = np.median(timed(bh, 5))
    print("set_data %.3f ms, first multiply %.3f, second %.3f, then %.3f (stage_ms of the last %s)" % (t_set, first, second, rest, [round(x, 3) for x in bh.stage_ms]), flush=True)
keep = [bh]
for i in range(4):
    b2 = new_handle(0)
    for _ in range(3): assert b2.spgemm() == 0
    t = timed(b2)
    pr, pc, pv = b2.get_C_device()
    print("fresh handle %d: median %.3f min %.3f ms; rowPtrC %#x colIndC %#x valC %#x" % (i, np.median(t), t.min(), pr, pc, pv), flush=True)
    if i % 2: keep.append(b2)            # (every other one stays allocated: the next one lands elsewhere)
    else: b2.free_mem(); b2.freePlatform()
t = timed(bh)
print("first handle again: median %.3f ms" % np.median(t))
