#!/usr/bin/env python3
"""Measurement helper (round 4): the boxes of the pool run numeric_class of the SAME build in 1.45 or in 1.69 ms while every
other kernel takes the same time on all of them.  One process, poisson27pt 128^3: the ring kernel as shipped, with plain
stores (BHSPARSE_HIP_LIB variant given as argv[1], optional), with 8 instead of 12 waves per CU, round 2's LDS-atomic
kernel, and the general pipeline -- to see which of them follow the box."""
import sys, os, time, subprocess
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

def run(label, **opts):
    for k, v in opts.items(): assert bh.set_option(k, v) == 0
    for _ in range(2): assert bh.spgemm() == 0
    acc = {}
    for _ in range(5):
        assert bh.spgemm() == 0
        for s in bh.kernel_stats(): acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / 5
    print("%-34s %s" % (label, {k: round(v, 3) for k, v in acc.items() if v > 0.3}), flush=True)

run("ring kernel, 12 waves per CU")
run("ring kernel, 8 waves per CU", wg_per_cu=8)
run("ring kernel, 4 waves per CU", wg_per_cu=4)
run("ring kernel again", wg_per_cu=0)
run("round 2's LDS-atomic kernel", class_numeric=0)
run("general pipeline", class_numeric=1, class_path=0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print("stream copy (read + written): %s GB/s" % bench.stream_copy_GBs(dev), flush=True)
# a read-only and a write-only stream as well
x = torch.empty(1 << 28, dtype=torch.float64, device=dev); x.fill_(1.0); torch.cuda.synchronize()
for name, fn in (("read-only sum", lambda: x.sum()), ("write-only fill", lambda: x.fill_(2.0))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): fn()
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f GB/s" % (name, (1 << 28) * 8 / (e0.elapsed_time(e1) / 8 * 1e-3) / 1e9), flush=True)
