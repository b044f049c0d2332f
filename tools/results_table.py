#!/usr/bin/env python3
"""Results table over the driver's gallery (README.md:40-54 of the reference: datasets 1-4), their full-size versions
in BASELINE.json, and the stand-ins for the inputs that are not in the image (power-law web graph, 3-dof FEM):
C = A^2, fp64, device-resident inputs, median of 10 multiplies after 3 warm-ups.  Prints a Markdown table.
Last column: the REFERENCE ITSELF on the same GPU and matrix -- its OpenCL branch built unmodified into oracle/_ref
(oracle/Makefile), one multiply in its own process, the time its own spgemm() timer prints (bench.reference_opencl_leg);
"-" when the binary is absent, the reason when it cannot run the case.

    python tools/results_table.py > gpurun_out/results_table.md
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
import bench as benchmod

dev = torch.device("cuda", 0)
HBM = 8000.0e9


def fem3(n):
    """poisson27pt on n^3 nodes (x) ones(3,3): 81 entries per row, columns in runs of 3 (3 degrees of freedom per node)"""
    import scipy.sparse as sp
    rp, col = gallery.poisson_csr("poisson27pt", n, n, n)
    P = sp.csr_matrix((np.ones(len(col)), col, rp), shape=(len(rp) - 1,) * 2)
    A = sp.kron(P, np.ones((3, 3)), format="csr")
    A.sort_indices()
    return torch.from_numpy(A.indptr.astype(np.int32)).to(dev), torch.from_numpy(A.indices.astype(np.int32)).to(dev)


CASES = [
    ("poisson5pt 256^2 (driver dataset 1)", lambda: gallery.poisson_csr_torch("poisson5pt", 256, 256, 1, device=dev)),
    ("poisson9pt 256^2 (dataset 2)", lambda: gallery.poisson_csr_torch("poisson9pt", 256, 256, 1, device=dev)),
    ("poisson7pt 51^3 (dataset 3)", lambda: gallery.poisson_csr_torch("poisson7pt", 51, 51, 51, device=dev)),
    ("poisson27pt 51^3 (dataset 4)", lambda: gallery.poisson_csr_torch("poisson27pt", 51, 51, 51, device=dev)),
    ("poisson5pt 1024^2 (BASELINE configs[1])", lambda: gallery.poisson_csr_torch("poisson5pt", 1024, 1024, 1, device=dev)),
    ("poisson9pt 1024^2", lambda: gallery.poisson_csr_torch("poisson9pt", 1024, 1024, 1, device=dev)),
    ("poisson7pt 128^3", lambda: gallery.poisson_csr_torch("poisson7pt", 128, 128, 128, device=dev)),
    ("poisson27pt 128^3 (configs[2], bench default)", lambda: gallery.poisson_csr_torch("poisson27pt", 128, 128, 128, device=dev)),
    ("poisson27pt 160^3 (north_star target size)", lambda: gallery.poisson_csr_torch("poisson27pt", 160, 160, 160, device=dev)),
    ("poisson27pt (x) ones(3,3), 40^3 nodes (3-dof FEM stand-in)", lambda: fem3(40)),
    ("power-law 1 M rows (webbase-1M stand-in, configs[3])",
     lambda: tuple(torch.from_numpy(a).to(dev) for a in gallery.powerlaw_csr(1000005, 1000005, 3105536, 4700))),
]

print("| Workload (C = A^2, fp64) | rows | nnz(A) | products | nnz(C) | ms (median of 10) | GFLOP/s | compulsory bytes / t / 8 TB/s | dominant kernel | reference (SpGEMM_opencl) on this GPU, ms |")
print("|---|---|---|---|---|---|---|---|---|---|")
plats = [False] * 9
plats[3] = True
for name, gen in CASES:
    Bp, Bj = gen()
    Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
    Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
    m = Bp.numel() - 1
    bh = facade.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    for _ in range(3):
        assert bh.spgemm() == 0
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        assert bh.spgemm() == 0
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(ts))
    ks = max(bh.kernel_stats(), key=lambda s: s["ms"])
    alg = 2 * (4 * (m + 1) + 12 * Bj.numel()) + 4 * (m + 1) + 12 * bh.nnzC
    ref = benchmod.reference_opencl_leg(m, Bp.cpu().numpy(), Bj.cpu().numpy(), Bx.cpu().numpy(), bh.nnzCt) if "--no-reference" not in sys.argv else None
    if ref is None:
        refs = "-"
    elif "ms" in ref:
        refs = "%.1f (%.0fx)%s" % (ref["ms"], ref["ms"] / ms, "" if ref.get("nnzC") == bh.nnzC else " nnzC %s!" % ref.get("nnzC"))
    else:
        refs = ref.get("skipped") or ("failed: " + str(ref.get("error")) + " " + " / ".join(ref.get("stdout_tail", []))[:160])
    print("| %s | %d | %d | %d | %d | %.3f | %.1f | %.1f %% | %s (%.3f ms) | %s |" %
          (name, m, Bj.numel(), bh.nnzCt, bh.nnzC, ms, 2.0 * bh.nnzCt / (ms * 1e6), 100.0 * alg / (ms * 1e-3) / HBM,
           ks["name"], ks["ms"], refs), flush=True)
    bh.free_mem(); bh.freePlatform()
    del Ap, Aj, Ax, Bp, Bj, Bx
    torch.cuda.empty_cache()
