#!/usr/bin/env python3
"""Measurement helper (round 6): poisson27pt n^3 clean and with irregular rows -- 0.1 % / 1 % of the rows given one extra random
entry, one 300-entry row -- through the facade: ms per multiply (wall clock around bhs_spgemm, inputs resident), the
kernel-family breakdown, the number of rows that took the general pipeline's kernels, and the digest of C against the
general pipeline's (class_path = 0) on the same input.
usage: mixed_case.py [n=128 | fem40] [cases=clean,p0.1,p1,long]   env BHS_OPTS=key=value,.."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade

arg1 = sys.argv[1] if len(sys.argv) > 1 else "128"
cases = (sys.argv[2] if len(sys.argv) > 2 else "clean,p0.1,p1,long").split(",")
dev = torch.device("cuda", 0)
if arg1.startswith("fem"):                     # "fem40": 3 unknowns per node on poisson27pt 40^3 (the big-class kernels)
    n = int(arg1[3:])
    rp0, col0 = gallery.block_expand_csr(*gallery.poisson_csr("poisson27pt", n, n, n), 3)
else:
    n = int(arg1)
    rp0, col0 = gallery.poisson_csr("poisson27pt", n, n, n)
m = len(rp0) - 1


def digest(bh):
    from benchmark_spgemm_using_csr_amd.dist import device_view
    prp, pcj, pcx = bh.get_C_device()
    nnzC = bh.get_nnzC()
    out = []
    for ptr, cnt, dt in ((prp, m + 1, torch.int32), (pcj, nnzC, torch.int32), (pcx, nnzC, torch.float64)):
        out.append(hashlib.sha256(device_view(ptr, cnt, dt, dev).cpu().numpy().tobytes()).hexdigest()[:16])
    return out


def run(rp, col, opts, reps=10):
    val = gallery.fill_values(len(col))
    Ap, Aj, Ax = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
    plats = [False] * 9; plats[3] = True
    bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
    for k_, v_ in opts.items(): assert bh.set_option(k_, v_) == 0
    assert bh.set_option("kernel_stats", 1) == 0
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Aj.numel(), Ax, Ap, Aj) == 0
    for _ in range(3): assert bh.spgemm() == 0
    acc = {}
    for _ in range(3):
        assert bh.spgemm() == 0
        for s in bh.kernel_stats(): acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / 3
    assert bh.set_option("kernel_stats", 0) == 0
    assert bh.spgemm() == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): assert bh.spgemm() == 0
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    res = dict(ms=round(ms, 4), mixed_rows=bh.get_info("mixed_rows"), state=bh.get_info("class_state"), nnzC=bh.get_nnzC(), nnzCt=bh.nnzCt,
               kernels={k: round(v, 4) for k, v in sorted(acc.items()) if v > 0.002}, digest=digest(bh))
    bh.free_mem(); bh.freePlatform()
    return res


extra = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("BHS_OPTS", "").split(",") if kv}
base = None
for c in cases:
    if c == "clean": rp, col = rp0, col0
    elif c == "long": rp, col = gallery.perturb_rows_csr(rp0, col0, m, 0.0, long_row=(m // 2 + 5, 300))
    else: rp, col = gallery.perturb_rows_csr(rp0, col0, m, float(c[1:]) / 100.0, seed=11)
    r = run(rp, col, dict(extra))
    g = run(rp, col, dict(extra, class_path=0), reps=3) if os.environ.get("BHS_NOGEN") != "1" else None
    if c == "clean": base = r["ms"]
    print(c, "rows=%d nnzA=%d" % (m, len(col)), "ms=%.4f" % r["ms"], "x%.3f of clean" % (r["ms"] / base) if base else "", "irregular rows", r["mixed_rows"], "state", r["state"],
          "| general pipeline %.3f ms, digests equal: %s" % (g["ms"], g["digest"] == r["digest"]) if g else "")
    print("    ", r["kernels"])
    sys.stdout.flush()
