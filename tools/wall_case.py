#!/usr/bin/env python3
"""Measurement helper: WALL time per bhs_spgemm (median of 30) of a named workload, kernel_stats off -- for host-side
changes that the per-kernel timers do not see.   python tools/wall_case.py p5_1024 [key=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
name = sys.argv[1]
dev = torch.device("cuda", 0)
if name == "weblike":
    rp, col = gallery.weblike_csr()
    Bp, Bj = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
else:
    st, dims = {"p27_128": ("poisson27pt", (128, 128, 128)), "p5_1024": ("poisson5pt", (1024, 1024, 1)), "p5_256": ("poisson5pt", (256, 256, 1)),
                "p27_51": ("poisson27pt", (51, 51, 51)), "p27_160": ("poisson27pt", (160, 160, 160)), "p9_1024": ("poisson9pt", (1024, 1024, 1)), "p7_128": ("poisson7pt", (128, 128, 128))}[name]
    Bp, Bj = gallery.poisson_csr_torch(st, *dims, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
assert bh.set_option("kernel_stats", 0) == 0
for kv in sys.argv[2:]:
    k_, v_ = kv.split("="); assert bh.set_option(k_, int(v_)) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
for _ in range(5): assert bh.spgemm() == 0
t = []
for _ in range(30):
    torch.cuda.synchronize(); q = time.perf_counter(); assert bh.spgemm() == 0; t.append((time.perf_counter() - q) * 1e3)
t = np.array(t)
print("%-8s %-24s wall median %.4f min %.4f ms" % (name, " ".join(sys.argv[2:]), np.median(t), t.min()), flush=True)
