#!/usr/bin/env python3
"""Measurement helper: per-phase wave cycles of k_class_numeric (library built with -DBHS_PHASES_CLS=1,
tools/build_variants.sh phcls="-DBHS_PHASES_CLS=1"; BHSPARSE_HIP_LIB=gpurun_variants/phcls.so)."""
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
for kv in os.environ.get('BHS_OPTS', '').split(','):
    if kv: k_, v_ = kv.split('='); assert bh.set_option(k_, int(v_)) == 0
raw = C.CDLL(os.environ.get("BHSPARSE_HIP_LIB", _lib.SO_PATH))
buf = (C.c_ulonglong * 16)()
for _ in range(2): assert bh.spgemm() == 0
raw.bhs_debug_phases(buf)
assert bh.spgemm() == 0
raw.bhs_debug_phases(buf)
rows = buf[7]
names = ["run: requests for the runs behind", "class data (on a change)", "stretch start: first slabs requested",
         "stretch start: wait for them", "row: arithmetic (LDS only)", "row: slab request + write-out", "row: vmcnt wait"]
names += ["(ring, r5) A values placed + 24 LDS reads incl. latency", "(ring, r5) ring moved on", "(ring, r5) chunk switch", "(ring, r5) slab request"]
idx = list(range(7)) + [8, 9, 10, 11]
tot = sum(buf[i] for i in idx)
print("rows", rows, "wave cycles per row: %.0f" % (tot / rows))
for i, n in zip(idx, names): print("  %-52s %8.0f cycles/row  %5.1f %%" % (n, buf[i] / rows, 100.0 * buf[i] / tot))
print({s["name"]: round(s["ms"], 3) for s in bh.kernel_stats() if s["ms"] > 0.1})
