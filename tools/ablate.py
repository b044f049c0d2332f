#!/usr/bin/env python3
"""Measurement helper (not product): kernel times of one workload under ablation masks."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchmark_spgemm_using_csr_amd import gallery, facade
stencil, dims = "poisson27pt", (128, 128, 128)
if len(sys.argv) > 1 and sys.argv[1] == "p5": stencil, dims = "poisson5pt", (1024, 1024, 1)
tag = os.environ.get("BHSPARSE_HIP_LIB", "default")
dev = torch.device("cuda", 0)
Bp, Bj = gallery.poisson_csr_torch(stencil, *dims, device=dev)
Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
import numpy as np
f32 = os.environ.get('BHS_F32') == '1'
if f32: Bx = Bx.to(torch.float32)
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
m = Bp.numel() - 1
plats = [False] * 9; plats[3] = True
bh = facade.bhsparse(value_dtype=np.float32 if f32 else np.float64)
if f32: facade._lib.SO_PATH_F32 = os.environ['BHSPARSE_HIP_LIB']
assert bh.initPlatform(plats) == 0
assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
for kv in os.environ.get('BHS_OPTS','').split(','):
    if kv: k_, v_ = kv.split('='); assert bh.set_option(k_, int(v_)) == 0
for _ in range(2): assert bh.spgemm() == 0
acc = {}
for _ in range(3):
    assert bh.spgemm() == 0
    for s in bh.kernel_stats(): acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / 3
print(os.path.basename(tag), os.environ.get('BHS_OPTS',''), {k: round(v, 3) for k, v in acc.items() if v > 0.05})
