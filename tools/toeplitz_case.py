#!/usr/bin/env python3
"""Measurement helper: C = A B for Toeplitz-like A (na offsets) and B (nb offsets), m rows, class path against the general
pipeline -- which instance of k_class_ring runs and what it is worth.   python tools/toeplitz_case.py m na nb [na nb ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
dev = torch.device("cuda", 0)
m = int(sys.argv[1])
pairs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(2, len(sys.argv) - 1, 2)]
def toeplitz(rows, cols, offs):
    c = np.arange(rows, dtype=np.int64)[:, None] + offs[None, :]
    ok = (c >= 0) & (c < cols)
    rp = np.zeros(rows + 1, np.int32); rp[1:] = np.cumsum(ok.sum(axis=1))
    return rp, c[ok].astype(np.int32)
for na, nb in pairs:
    rng = np.random.default_rng(na * 100 + nb)
    offa = np.sort(rng.choice(np.arange(-750, 751) * 4, na, replace=False)); offb = np.sort(rng.choice(np.arange(-750, 751) * 4, nb, replace=False))
    Ap, Aj = toeplitz(m, m, offa); Bp, Bj = toeplitz(m, m, offb)
    t = [torch.from_numpy(x).to(dev) for x in (Ap, Aj, gallery.fill_values(len(Aj)), Bp, Bj, gallery.fill_values(len(Bj)))]
    sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in o.split(",") if kv) for o in os.environ["TOEP_OPTS"].split(";")] if "TOEP_OPTS" in os.environ else [{}, {"class_path": 0}]
    for opts in sets:
        plats = [False] * 9; plats[3] = True
        bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
        for k_, v_ in opts.items(): assert bh.set_option(k_, v_) == 0
        assert bh.initData_device(m, m, m, len(Aj), t[2], t[0], t[1], len(Bj), t[5], t[3], t[4]) == 0
        for _ in range(3): assert bh.spgemm() == 0
        acc = {}; st = np.zeros(4); n = 5
        for _ in range(n):
            assert bh.spgemm() == 0
            st += np.array(bh.stage_ms) / n
            for s in bh.kernel_stats(): acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / n
        print("%d x %d entries, %d rows, %s: %.3f ms, nnzCt %d nnzC %d (%.1f a row), class_state %d" % (
            na, nb, m, opts or "defaults", st.sum(), bh.nnzCt, bh.nnzC, bh.nnzC / m, bh.get_info("class_state")))
        print("    ", {k: round(v, 3) for k, v in acc.items() if v > 0.02})
        bh.free_mem(); bh.freePlatform()
