#!/usr/bin/env python3
"""Measurement helper (round 4): numeric_class against the number of consecutive rows a wave of the ring kernel takes
(`class_super_rows`), on the SAME output arrays -- the placement of C moves the kernel by 10 % (probe_placement.py),
so variants are compared inside one allocation, interleaved, over several allocations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Bp, Bj = gallery.poisson_csr_torch("poisson27pt", n, n, n, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 64, 128]   # rows; 0: the library's choice
res = {v: [] for v in variants}
for trial in range(3):
    if trial: bh.free_mem()
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    for _ in range(3): assert bh.spgemm() == 0
    line = []
    for rep in range(2):
        for v in variants:
            assert bh.set_option("class_super_rows", v) == 0
            assert bh.spgemm() == 0
            nc = []
            for _ in range(5):
                assert bh.spgemm() == 0
                nc.append([s["ms"] for s in bh.kernel_stats() if s["name"] == "numeric_class"][0])
            res[v].append(np.median(nc))
    print("allocation %d: " % trial + "  ".join("%g:%.3f" % (v, np.mean(res[v][-2:])) for v in variants), flush=True)
print("poisson27pt %d^3 (line_a = %d): rows per super-run -> numeric_class ms, mean over allocations (min .. max)" % (n, bh.get_info("line_a")))
for v in variants:
    a = np.array(res[v]); print("  %4d rows: %.3f  (%.3f .. %.3f)" % (v, a.mean(), a.min(), a.max()))
