#!/usr/bin/env python3
"""Measurement helper: one input of tools/suite_table.py through the facade with the kernel-family breakdown
(BHS_OPTS=key=value,.. sets library options).   python tools/run_suite_case.py blockdiag_1m"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
import suite_table
name = sys.argv[1]
gen = {c[0]: c[2] for c in suite_table.CASES}[name]
dev = torch.device("cuda", 0)
rp, col = gen()
val = gallery.fill_values(len(col))
Bp, Bj, Bx = (torch.from_numpy(np.asarray(x)).to(dev) for x in (rp, col, val))
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
for kv in os.environ.get('BHS_OPTS', '').split(','):
    if kv: k_, v_ = kv.split('='); assert bh.set_option(k_, int(v_)) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
for _ in range(2): assert bh.spgemm() == 0
acc = {}; info = {}; st = np.zeros(4); n = 5
for _ in range(n):
    assert bh.spgemm() == 0
    st += np.array(bh.stage_ms) / n
    for s in bh.kernel_stats():
        acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / n
        info[s["name"]] = (s["rows"], s["products"], s["nnz_out"])
print(name, os.environ.get('BHS_OPTS', ''), "stages", np.round(st, 3), "total %.3f ms  %.1f GFLOPs" % (st.sum(), 2 * bh.nnzCt / st.sum() / 1e6))
print("   ", {k: round(v, 3) for k, v in acc.items() if v > 0.01})
print("    rows / products / nnz of the kernels over 0.2 ms:", {k: v for k, v in info.items() if acc[k] > 0.2})
