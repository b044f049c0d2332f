#!/usr/bin/env python3
"""Measurement helper: per-phase workgroup cycles of numeric_long_rows (library built with -DBHS_PHASES_SPA=1 by tools/build_variants.sh; BHSPARSE_HIP_LIB=gpurun_variants/phspa.so)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade, _lib
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "powerlaw"
rp, col = gallery.weblike_csr() if which == "weblike" else gallery.rmat_csr() if which == "rmat" else gallery.powerlaw_csr(1000005, 1000005, 3105536, 4700)
val = gallery.fill_values(len(col))
Bp, Bj, Bx = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
raw = C.CDLL(os.environ.get("BHSPARSE_HIP_LIB", _lib.SO_PATH))
buf = (C.c_ulonglong * 16)()
print("data ready", flush=True)
for _ in range(2): assert bh.spgemm() == 0
print("warm done", flush=True)
raw.bhs_debug_phases(buf)
assert bh.spgemm() == 0
raw.bhs_debug_phases(buf)
tot = sum(buf[i] for i in range(8, 14))
for i, n in zip(range(8, 14), ["ticket + descriptor", "pass 1 (bits)", "count sweep + block scan", "ordered sweep (Cj, zero Cx, rank)", "pass 2 (adds)", "clear bitmap"]):
    print("  %-36s %12d cycles  %5.1f %%" % (n, buf[i], 100.0 * buf[i] / max(tot, 1)))
tw = sum(buf[i] for i in range(0, 7))
for i, n in zip(range(0, 7), ["ticket, descriptor, row of A", "stage + pass 1 (bits)", "lane totals, scan, ranks", "zero + columns to LDS", "pass 2 (adds in LDS)", "write-out", "clear bitmap"]):
    print("  wave-window %-28s %12d cycles  %5.1f %%" % (n, buf[i], 100.0 * buf[i] / max(tw, 1)))
print({s["name"]: round(s["ms"], 3) for s in bh.kernel_stats() if s["ms"] > 0.1})
