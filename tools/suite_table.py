#!/usr/bin/env python3
"""Results table of round 4: structurally distinct inputs of about a million rows -- seeded
stand-ins, no SuiteSparse file is in the image -- each one
  (1) multiplied device-resident through the facade (3 warm-ups, median of 10): time, the path the library took
      (row classes / lane-first / general pipeline with its bins / hub rows), compulsory bytes / t / 8 TB/s;
  (2) written as a Matrix Market file and run through the reference-style driver as a downloaded file would be --
      `spgemm -hip -spgemm file.mtx -cpu`: reader, row sort, 1..9 values, ONE timed multiply after 3 warm-ups, compData
      against the CPU oracle (PASS lines) and the oracle's own time on this host;
  (3) multiplied by the REFERENCE ITSELF (oracle/_ref: SpGEMM_opencl, unmodified) on the same GPU, its own timer, its
      C's digests compared with this library's.

    python tools/suite_table.py [--only name,..] [--no-driver] [--no-reference] > gpurun_out/r04_suite_table.md
"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
import bench as benchmod

dev = torch.device("cuda", 0)
HBM = 8000.0e9
DRIVER = os.path.join(ROOT, "tests", "driver", "spgemm")


def fem3(n):
    return gallery.block_expand_csr(*gallery.poisson_csr("poisson27pt", n, n, n), 3)


CASES = [   # name, what it stands for, generator, driver arguments (None: through a .mtx file)
    ("rmat_s20", "R-MAT / Kronecker graph, 2^20 rows (social network: hubs of 7 k entries, rows of A^2 up to 148 k)", lambda: gallery.rmat_csr(), None),
    ("banded_1m", "banded, irregular bandwidth 3..24 (1-D FEM of varying order)", lambda: gallery.banded_csr(), None),
    ("mesh2d_1m", "Delaunay mesh of 2^20 random points (unstructured 2-D FEM adjacency)", lambda: gallery.mesh2d_csr(), None),
    ("blockdiag_1m", "block-diagonal, dense blocks of 4..32 rows (supernodes)", lambda: gallery.blockdiag_csr(), None),
    ("uniform_1m", "uniform random, 8 per row (no structure)", lambda: gallery.uniform_csr(), None),
    ("roadlike_1m", "1024^2 grid graph with 38 % of the edges removed (road network)", lambda: gallery.roadlike_csr(), None),
    ("weblike_1m", "web graph with webbase-1M's size and compression 1.35 (BASELINE configs[3] stand-in)", lambda: gallery.weblike_csr(), None),
    ("powerlaw_1m", "power-law rows, hub-heavy (rounds 1-3 stand-in for configs[3]; rows of A^2 up to 102 k)", lambda: gallery.powerlaw_csr(1000005, 1000005, 3105536, 4700), None),
    ("fem3_40", "poisson27pt (x) ones(3,3), 40^3 nodes (3-dof elasticity)", lambda: fem3(40), None),
    ("p5_1024", "poisson5pt 1024^2 (BASELINE configs[1])", lambda: gallery.poisson_csr("poisson5pt", 1024, 1024, 1), ["-spgemm", "1", "-grid", "1024", "1024"]),
    ("p27_128", "poisson27pt 128^3 (BASELINE configs[2], bench default)", lambda: gallery.poisson_csr("poisson27pt", 128, 128, 128), ["-spgemm", "4", "-grid", "128", "128", "128"]),
]


def path_of(stats):
    names = {s["name"] for s in stats if s["ms"] > 0}
    if "numeric_class" in names:
        p = "row classes"
    elif "numeric_lane" in names or "symbolic_lane" in names:
        p = "lane-first (K-way merge per lane)"
    else:
        bins = sorted(n.replace("numeric_", "") for n in names if n.startswith("numeric_") and n != "numeric_hub_rows")
        p = "general pipeline: " + ", ".join(bins)
    if "numeric_hub_rows" in names:
        p += " + hub rows split across workgroups"
    return p


def write_mtx(path, m, rp, col):
    import pandas as pd
    rows = np.repeat(np.arange(1, m + 1, dtype=np.int64), np.diff(rp))
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (m, m, len(col)))
    pd.DataFrame({"r": rows, "c": col.astype(np.int64) + 1}).to_csv(path, sep=" ", header=False, index=False, mode="a")


def main():
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only"):
            only = set(sys.argv[sys.argv.index(a) + 1].split(",")) if a == "--only" else set(a.split("=", 1)[1].split(","))
    no_driver, no_ref = "--no-driver" in sys.argv, "--no-reference" in sys.argv
    print("| input | stands for | rows | nnz(A) | products | nnz(C) | products / nnz(C) | ms (median of 10, device-resident) | GFLOP/s | compulsory bytes / t / 8 TB/s | path taken | dominant kernel | driver on the .mtx file: one multiply, ms | compData vs oracle | CPU oracle on this host, ms (threads) | reference (SpGEMM_opencl) on this GPU, ms | reference's C = ours |")
    print("|" + "---|" * 17)
    plats = [False] * 9
    plats[3] = True
    for name, what, gen, dargs in CASES:
        if only and name not in only:
            continue
        rp, col = gen()
        rp, col = np.asarray(rp, np.int32), np.asarray(col, np.int32)
        m = len(rp) - 1
        val = gallery.fill_values(len(col))
        Bp, Bj, Bx = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
        Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
        bh = facade.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
        for _ in range(3):
            assert bh.spgemm() == 0
        ts = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert bh.spgemm() == 0
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.median(ts))
        stats = bh.kernel_stats()
        ks = max(stats, key=lambda s: s["ms"])
        nnzCt, nnzC = bh.nnzCt, bh.nnzC
        alg = 2 * (4 * (m + 1) + 12 * len(col)) + 4 * (m + 1) + 12 * nnzC
        mine = benchmod.device_digest(bh, m, dev)
        bh.free_mem(); bh.freePlatform()
        del Ap, Aj, Ax, Bp, Bj, Bx
        torch.cuda.empty_cache()
        # (2) the driver
        dms = dpass = cpu = "-"
        if not no_driver:
            with tempfile.TemporaryDirectory() as td:
                if dargs is None:
                    f = os.path.join(td, name + ".mtx")
                    write_mtx(f, m, rp, col)
                    dargs2 = ["-spgemm", f]
                else:
                    dargs2 = dargs
                try:
                    p = subprocess.run([DRIVER, "-hip"] + dargs2 + ["-cpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
                    out = p.stdout
                    mt = re.search(r"\[ HIP \] SpGEMM time: ([0-9.eE+-]+) ms", out)
                    mc = re.search(r"\[ CPU oracle, (\d+) threads \] SpGEMM time: ([0-9.eE+-]+) ms", out)
                    dms = "%.3f" % float(mt.group(1)) if mt else "rc %d" % p.returncode
                    ok = "RowPtrC PASS!" in out and "ColIndC/csrValC PASS!" in out and ('"nnzCt": %d, "nnzC": %d, "pass": true' % (nnzCt, nnzC)) in out
                    dpass = "PASS (nnzC, rowPtr, colInd / val)" if ok else "NO PASS: " + " / ".join(out.strip().splitlines()[-3:])[:200]
                    cpu = "%.0f (%s)" % (float(mc.group(2)), mc.group(1)) if mc else "-"
                except Exception as e:
                    dms = "failed: %s" % str(e)[-100:]
        # (3) the reference
        refs = same = "-"
        if not no_ref:
            ref = benchmod.reference_opencl_leg(m, rp, col, val, nnzCt, mine=mine)
            if ref is None:
                refs = "binary absent"
            elif "ms" in ref:
                refs = "%.1f (%.0fx)" % (ref["ms"], ref["ms"] / ms)
                if "hip_digest_equals_reference" in ref:
                    same = "yes" if ref["hip_digest_equals_reference"] else "NO: its nnz(C) = %s (its merge stops at 25 600 entries per row, bhsparse.cpp:469)" % ref.get("nnzC")
            else:
                refs = ref.get("skipped") or ("failed: " + str(ref.get("error")))[:160]
        print("| %s | %s | %d | %d | %d | %d | %.2f | %.3f | %.1f | %.1f %% | %s | %s (%.3f ms) | %s | %s | %s | %s | %s |" %
              (name, what, m, len(col), nnzCt, nnzC, nnzCt / max(nnzC, 1), ms, 2.0 * nnzCt / (ms * 1e6), 100.0 * alg / (ms * 1e-3) / HBM,
               path_of(stats), ks["name"], ks["ms"], dms, dpass, cpu, refs, same), flush=True)


if __name__ == "__main__":
    main()
