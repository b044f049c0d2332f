#!/usr/bin/env python3
"""Measurement helper: per-phase wave cycles of numeric_wave (library built with -DBHS_PHASES=1)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchmark_spgemm_using_csr_amd import gallery, facade, _lib
dev = torch.device("cuda", 0)
Bp, Bj = gallery.poisson_csr_torch("poisson27pt", 128, 128, 128, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
raw = C.CDLL(_lib.SO_PATH)
buf = (C.c_ulonglong * 16)()
for _ in range(2): assert bh.spgemm() == 0
raw.bhs_debug_phases(buf)
assert bh.spgemm() == 0
raw.bhs_debug_phases(buf)
rows = buf[7]
names = ["setup(clear,scan,sBase)", "marks+index+load issue", "wait loads", "inserts", "compaction", "sort+store"]
tot = sum(buf[i] for i in range(6))
print("rows", rows, "cycles/row/wave total %.0f" % (tot / rows))
for i, n in enumerate(names): print("  %-28s %8.0f cycles/row  %5.1f %%" % (n, buf[i] / rows, 100.0 * buf[i] / tot))
print({s["name"]: round(s["ms"], 3) for s in bh.kernel_stats() if s["ms"] > 0.1})
