#!/usr/bin/env python3
"""Debug helper: one perturbed poisson27pt n^3 multiply with verbose = 2 (the library's own notes about classes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
rp0, col0 = gallery.poisson_csr("poisson27pt", n, n, n)
m = len(rp0) - 1
rp, col = gallery.perturb_rows_csr(rp0, col0, m, frac, seed=11)
val = gallery.fill_values(len(col))
dev = torch.device("cuda", 0)
Ap, Aj, Ax = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
for kv in os.environ.get("BHS_OPTS", "").split(","):
    if kv: assert bh.set_option(kv.split("=")[0], int(kv.split("=")[1])) == 0
assert bh.set_option("verbose", 2) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Aj.numel(), Ax, Ap, Aj) == 0
for _ in range(2): assert bh.spgemm() == 0
