#!/usr/bin/env python3
"""Debug helper: one draw of tests/test_parity_gpu.py's mixed-mode soak, with where C differs from the oracle's.
    python tools/mixed_soak_case.py <seed> [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as orc
from benchmark_spgemm_using_csr_amd import facade as bhmod
import test_parity_gpu as T
seed = int(sys.argv[1])
opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[2:])
(m, k, n, sa, sb, na, nb, noise), (Ap, Aj, Ax), (Bp, Bj, Bx) = T._mixed_soak_inputs(seed)
print("seed", seed, (m, k, n), "steps", sa, sb, "entries", na, nb, "noise", noise, "longest row of A", np.diff(Ap).max(), "of B", np.diff(Bp).max())
ref = orc.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
plats = [False] * bhmod.NUM_PLATFORMS; plats[bhmod.BHSPARSE_HIP] = True
bh = bhmod.bhsparse(); assert bh.initPlatform(plats) == 0
for kk, vv in dict({"class_path": 2}, **opts).items(): assert bh.set_option(kk, vv) == 0
Cp = np.zeros(m + 1, np.int32)
assert bh.initData(m, k, n, len(Aj), Ax, Ap, Aj, len(Bj), Bx, Bp, Bj, Cp) == 0
for it in range(2):
    assert bh.spgemm() == 0
    Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
    assert bh.get_C(Cj, Cx) == 0
    names = sorted(s_["name"] for s_ in bh.kernel_stats() if s_["launches"])
    print("multiply", it, "class_state", bh.get_info("class_state"), "irregular rows", bh.get_info("mixed_rows"), names)
    okp = np.array_equal(Cp, ref[0]); okj = okp and np.array_equal(Cj, ref[1])
    print("  rowptr equal", okp, "columns equal", okj)
    if okj:
        bad = np.flatnonzero(Cx != ref[2])
        rows = np.unique(np.searchsorted(Cp, bad, side="right") - 1)
        print("  entries with other values:", len(bad), "in", len(rows), "rows:", rows[:40])
        lenA = np.diff(Ap)
        for r in rows[:6]:
            e = bad[(bad >= Cp[r]) & (bad < Cp[r + 1])]
            print("   row", r, "entries of A", lenA[r], "of C", Cp[r + 1] - Cp[r], "bad at", (e - Cp[r])[:20], "ours", Cx[e][:8], "oracle", ref[2][e][:8])
            print("      rows of A around it:", lenA[max(0, r - 3):r + 4], " A cols", Aj[Ap[r]:Ap[r + 1]][:10], "B row lengths", np.diff(Bp)[Aj[Ap[r]:Ap[r + 1]]][:40])
