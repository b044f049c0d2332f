#!/usr/bin/env python3
"""Measurement helper: time one named workload through the facade (kernel-family breakdown)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
name = sys.argv[1] if len(sys.argv) > 1 else "webbase"
dev = torch.device("cuda", 0)
if name == "webbase":
    rp, col = gallery.powerlaw_csr(1000005, 1000005, 3105536, 4700)
    val = gallery.fill_values(len(col))
    Bp, Bj, Bx = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
elif name == "weblike":
    rp, col = gallery.weblike_csr()
    val = gallery.fill_values(len(col))
    Bp, Bj, Bx = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
elif name == "denserows":
    rp, col = gallery.dense_rows_csr(1 << 20, 8, 4, 200000)
    val = gallery.fill_values(len(col))
    Bp, Bj, Bx = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
else:
    st, dims = {"p27_128": ("poisson27pt", (128, 128, 128)), "p27_160": ("poisson27pt", (160, 160, 160)), "p27_256": ("poisson27pt", (256, 256, 256)), "p5_1024": ("poisson5pt", (1024, 1024, 1)),
                "p7_128": ("poisson7pt", (128, 128, 128)), "p9_1024": ("poisson9pt", (1024, 1024, 1)),
                "p27_slab16": ("poisson27pt", (16, 16, 8192)), "p27_slab32": ("poisson27pt", (32, 32, 2048)),
                "p27_slab64": ("poisson27pt", (64, 64, 512)), "p27_51": ("poisson27pt", (51, 51, 51)), "p27_72": ("poisson27pt", (72, 72, 72)),
                "p9_256": ("poisson9pt", (256, 256, 1)), "p9_512": ("poisson9pt", (512, 512, 1))}[name]
    Bp, Bj = gallery.poisson_csr_torch(st, *dims, device=dev)
    Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
f32 = os.environ.get('BHS_F32') == '1'
if f32: Bx = Bx.to(torch.float32)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(value_dtype=np.float32 if f32 else np.float64); assert bh.initPlatform(plats) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
for kv in os.environ.get('BHS_OPTS','').split(','):
    if kv: k_, v_ = kv.split('='); assert bh.set_option(k_, int(v_)) == 0
for _ in range(int(os.environ.get("BHS_WARM", "2"))): assert bh.spgemm() == 0
acc = {}; info = {}; st = np.zeros(4); n = 5
for _ in range(n):
    assert bh.spgemm() == 0
    st += np.array(bh.stage_ms) / n
    for s in bh.kernel_stats():
        acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / n
        info[s["name"]] = (s["rows"], s["products"], s["nnz_out"])
print(name, "m=%d nnzA=%d nnzCt=%d nnzC=%d" % (m, Aj.numel(), bh.nnzCt, bh.nnzC), "stages", np.round(st, 3), "total %.3f ms  %.1f GFLOPs" % (st.sum(), 2 * bh.nnzCt / st.sum() / 1e6))
print("   ", {k: round(v, 3) for k, v in acc.items() if v > 0.01})
print("   rows/products/nnz:", {k: v for k, v in info.items() if acc[k] > 0.3})
