#!/usr/bin/env python3
"""Measurement helper: one rank's share of the 8-GPU configuration on one GPU -- the fourth of eight row blocks of
poisson27pt 256^3 as A, the whole matrix as B."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from benchmark_spgemm_using_csr_amd import gallery, facade
dev = torch.device("cuda", 0)
Bp, Bj = gallery.poisson_csr_torch("poisson27pt", 256, 256, 256, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
k = Bp.numel() - 1
r0, r1 = 3 * (k // 8), 4 * (k // 8)
a0, a1 = int(Bp[r0]), int(Bp[r1])
Ap = (Bp[r0:r1 + 1] - a0).to(torch.int32).contiguous(); Aj = Bj[a0:a1].contiguous(); Ax = Bx[a0:a1].contiguous()
m = r1 - r0
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
assert bh.initData_device(m, k, k, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
for _ in range(2): assert bh.spgemm() == 0
acc = {}; st = np.zeros(4)
for _ in range(5):
    assert bh.spgemm() == 0
    st += np.array(bh.stage_ms) / 5
    for s in bh.kernel_stats(): acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / 5
print("block of 1/8 of p27 256^3: m=%d stages %s total %.3f ms" % (m, np.round(st, 3), st.sum()), {k_: round(v, 3) for k_, v in acc.items() if v > 0.01})
