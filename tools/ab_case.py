#!/usr/bin/env python3
"""Measurement helper: option sets compared in ONE process on one handle, interleaved round by round (boxes drift by several
per cent over seconds; runs one after another cannot tell 2 % apart).   python tools/ab_case.py p27_128 a=1,b=0 a=0,b=0 ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
name = sys.argv[1]
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]]
dev = torch.device("cuda", 0)
if name == "weblike":
    rp, col = gallery.weblike_csr()
    Bp, Bj = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
else:
    st, dims = {"p27_128": ("poisson27pt", (128, 128, 128)), "p5_1024": ("poisson5pt", (1024, 1024, 1)), "p27_160": ("poisson27pt", (160, 160, 160)),
                "p9_1024": ("poisson9pt", (1024, 1024, 1)), "p7_128": ("poisson7pt", (128, 128, 128))}[name]
    Bp, Bj = gallery.poisson_csr_torch(st, *dims, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
assert bh.set_option("kernel_stats", 0) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
for _ in range(5): assert bh.spgemm() == 0
acc = [[] for _ in sets]
for rnd in range(8):
    for i, st_ in enumerate(sets):
        for k_, v_ in st_.items(): assert bh.set_option(k_, v_) == 0
        for _ in range(3): assert bh.spgemm() == 0
        for _ in range(12):
            torch.cuda.synchronize(); q = time.perf_counter(); assert bh.spgemm() == 0; acc[i].append((time.perf_counter() - q) * 1e3)
for i, st_ in enumerate(sets):
    t = np.array(acc[i])
    print("%-8s %-34s wall median %.4f  mean %.4f  min %.4f ms" % (name, sys.argv[2 + i], np.median(t), t.mean(), t.min()), flush=True)
