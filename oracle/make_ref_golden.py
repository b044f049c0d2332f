"""Generates tests/golden/ref_opencl_*.npz: outputs of the REFERENCE ITSELF (its OpenCL branch, built unmodified by
`make -C oracle _ref`) for fixed inputs, and checks the CPU restatement (oracle/ref_spgemm_oracle.c) against them.

TEST INFRASTRUCTURE ONLY.  Runs where an OpenCL GPU device exists (the MI355X box, through gpurun):

    python oracle/make_ref_golden.py --out gpurun_out/ref_opencl

writes one .npz per small case (inputs + the reference's C, rows as the reference left them, plus `rows_sorted`),
and ref_opencl_digests.json with, for every case incl. the large ones, digests of the reference's C and the verdict of
the comparison with the oracle done on the spot.  The files are then copied into tests/golden/ by hand and committed;
tests/test_oracle.py re-checks the oracle against them on every CPU run.

Inputs come from this repository (tests/golden/*.npz inputs, cage4.mtx, gallery generators): nothing is read from
/root/reference at run time; the binary and the staged .cl files travel in oracle/_ref/ (git-ignored).
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:] = [ROOT] + [p for p in sys.path if os.path.abspath(p or '.') != HERE]   # `oracle` is the package dir, not oracle.py
_spec = importlib.util.spec_from_file_location(
    "gallery", os.path.join(ROOT, "benchmark_spgemm_using_csr_amd", "gallery.py"))
gallery = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gallery)
from oracle import oracle  # noqa: E402

REF_DIR = os.path.join(HERE, "_ref")
REF_BIN = os.path.join(REF_DIR, "ref_opencl_spgemm")
GOLD = os.path.join(ROOT, "tests", "golden")
# The reference's kernels are 2014 OpenCL C: `inline void compression_scan(...)` without `static` has C99 inline
# semantics under today's clang-based AMD compiler (no external definition is emitted), and the program fails to LINK
# (ld.lld: undefined hidden symbol compression_scan; probe: oracle/cl_build_probe.cpp).  The sources stay untouched:
# the AMD OpenCL runtime appends $AMD_OCL_BUILD_OPTIONS_APPEND to the (empty) options the reference passes to
# clBuildProgram (basiccl.cpp:149), and `-Dinline=static` gives those helpers internal linkage.  No code changes.
REF_ENV = dict(os.environ, AMD_OCL_BUILD_OPTIONS_APPEND="-Dinline=static")


def run_reference(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, timeout=600):
    """One C = A*B through the reference binary.  Returns (Cp int32[m+1], Cj, Cx, stdout)."""
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            np.array([m, k, n, len(Aj), len(Bj)], np.int32).tofile(f)
            for a, dt in ((Ap, np.int32), (Aj, np.int32), (Bp, np.int32), (Bj, np.int32),
                          (Ax, np.float64), (Bx, np.float64)):
                np.ascontiguousarray(a, dt).tofile(f)
        p = subprocess.run([REF_BIN, fin, fout], cwd=REF_DIR, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout, text=True, env=REF_ENV)
        if p.returncode != 0:
            raise RuntimeError("reference binary failed (%d):\n%s" % (p.returncode, p.stdout[-2000:]))
        with open(fout, "rb") as f:
            nnzC = int(np.fromfile(f, np.int32, 1)[0])
            Cp = np.fromfile(f, np.int32, m + 1)
            Cj = np.fromfile(f, np.int32, nnzC)
            Cx = np.fromfile(f, np.float64, nnzC)
    assert len(Cx) == nnzC
    return Cp, Cj, Cx, p.stdout


def run_reference_digest(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, timeout=1200):
    """The same through the binary's digest mode ("="): C stays in the reference's process, which prints the sums of
    digest_of() over it (BASELINE.json's full sizes: a 3 GB dump per case would not fit gpurun_out)."""
    import re
    with tempfile.TemporaryDirectory() as td:
        fin = os.path.join(td, "in.bin")
        with open(fin, "wb") as f:
            np.array([m, k, n, len(Aj), len(Bj)], np.int32).tofile(f)
            for a, dt in ((Ap, np.int32), (Aj, np.int32), (Bp, np.int32), (Bj, np.int32),
                          (Ax, np.float64), (Bx, np.float64)):
                np.ascontiguousarray(a, dt).tofile(f)
        p = subprocess.run([REF_BIN, fin, "="], cwd=REF_DIR, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout, text=True, env=REF_ENV)
    mt = re.search(r"ref_opencl_digest: nnzC=(\d+) sum_rowptr=(\d+) wsum_col=(\d+) sum_val=(\S+) wsum_val=(\S+) rows_sorted=(\d)", p.stdout)
    if p.returncode != 0 or not mt:
        raise RuntimeError("reference binary failed (%d):\n%s" % (p.returncode, p.stdout[-2000:]))
    d = {"nnzC": int(mt.group(1)), "sum_rowptr": int(mt.group(2)), "wsum_col": int(mt.group(3)),
         "sum_val": float(mt.group(4)), "wsum_val": float(mt.group(5))}
    return d, bool(int(mt.group(6))), p.stdout


def full_size_cases():
    """BASELINE.json's own sizes (digest only): configs[1], configs[2], the 3-dof FEM stand-in and the web-like
    stand-in for configs[3] (its longest row of C, 13 k entries, stays below the 25 600 where the reference's OpenCL
    merge stops -- bhsparse.cpp:469, 498-502 -- so the reference can do it)."""
    out = []

    def stencil(tag, name, dims, dof=1):
        rp, col = gallery.poisson_csr(name, *dims)
        if dof > 1:
            rp, col = gallery.block_expand_csr(rp, col, dof)
        val = gallery.fill_values(len(col))
        m = len(rp) - 1
        out.append((tag, m, (rp, col, val)))

    stencil("p5_1024", "poisson5pt", (1024, 1024, 1))
    stencil("p27_128", "poisson27pt", (128, 128, 128))
    stencil("fem3_40", "poisson27pt", (40, 40, 40), dof=3)
    rp, col = gallery.weblike_csr()
    out.append(("weblike_1m", len(rp) - 1, (rp, col, gallery.fill_values(len(col)))))
    return out


def full_size(args, report):
    for tag, m, A in full_size_cases():
        if args.only and tag not in args.only.split(","):
            continue
        Ap, Aj, Ax = A
        entry = {"m": m, "k": m, "n": m, "nnzA": int(len(Aj)), "nnzB": int(len(Aj)), "full_size": True}
        try:
            d, was_sorted, log = run_reference_digest(m, m, m, Ap, Aj, Ax, Ap, Aj, Ax)
        except Exception as e:
            entry["error"] = str(e)[-1500:]
            report[tag] = entry
            print(tag, "REFERENCE FAILED:", entry["error"], flush=True)
            continue
        entry.update(d)
        entry["rows_sorted_by_reference"] = was_sorted
        entry["nnzCt"] = oracle.nnzCt(Ap, Aj, Ap)
        # the oracle on the same input (all host cores): digests must agree
        oCp, oCj, oCx = oracle.spgemm(m, m, m, Ap, Aj, Ax, Ap, Aj, Ax)
        od = digest_of(oCp, oCj, oCx)
        entry["oracle_digest_equal"] = bool(all(od[k2] == d[k2] for k2 in od))
        entry["oracle_rowptr_equal"] = entry["oracle_col_equal"] = entry["oracle_digest_equal"]
        if not entry["oracle_digest_equal"]:
            entry["oracle_digest"] = od
        entry["reference_stdout_tail"] = log.strip().splitlines()[-8:]
        report[tag] = entry
        print(tag, {k2: v for k2, v in entry.items() if k2 != "reference_stdout_tail"}, flush=True)
        del oCp, oCj, oCx


def sort_rows(Cp, Cj, Cx):
    """Stable per-row sort by column (ref_spgemm.h:37-62 does the same to inputs); returns copies + 'was sorted'."""
    row = np.repeat(np.arange(len(Cp) - 1, dtype=np.int64), np.diff(Cp.astype(np.int64)))
    order = np.lexsort((Cj, row))
    was = bool(np.array_equal(order, np.arange(len(Cj))))
    return Cj[order], Cx[order], was


def cases():
    """(tag, save_full, m, k, n, A, B).  Values are integers (exact sums in any order) unless stated."""
    out = []

    def gold(tag):
        z = np.load(os.path.join(GOLD, tag + ".npz"))
        return (int(z["m"]), int(z["k"]), int(z["n"]), (z["Ap"], z["Aj"], z["Ax"]), (z["Bp"], z["Bj"], z["Bx"]))

    # the reference's built-in test (main.cu:153-205 / main.cpp:285-350) and the stencil / rectangular fixtures
    for tag in ("small_test", "p5_16", "p27_6", "p9_12", "p7_7", "rect_rand"):
        out.append((tag, True) + gold(tag))
    # cage4 with the file's values (non-integer: compared with the 1e-6 tolerance) and with ones
    z = np.load(os.path.join(GOLD, "cage4_sq.npz"))
    A = (z["Ap"], z["Aj"], z["Ax"])
    out.append(("cage4", True, 9, 9, 9, A, A))
    A1 = (z["Ap"], z["Aj"], np.ones(len(z["Aj"])))
    out.append(("cage4_ones", True, 9, 9, 9, A1, A1))

    def stencil(tag, name, dims, full):
        rp, col = gallery.poisson_csr(name, *dims)
        val = gallery.fill_values(len(col))
        m = len(rp) - 1
        out.append((tag, full, m, m, m, (rp, col, val), (rp, col, val)))

    stencil("p27_12", "poisson27pt", (12, 12, 12), True)       # EM bin (ub 729) beside the bitonic bins
    stencil("p5_256", "poisson5pt", (256, 256, 1), False)      # reference default -spgemm 1 (main.cu:32-35)
    stencil("p27_51", "poisson27pt", (51, 51, 51), False)      # reference default -spgemm 4 (main.cu:44-47)

    rng = np.random.default_rng(20140519)

    def random_csr(m, n, lens, vals=None):
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        cols = rng.integers(0, n, rows.size)
        key = np.unique(rows * n + cols)
        r = key // n
        rp = np.zeros(m + 1, np.int64)
        np.cumsum(np.bincount(r, minlength=m), out=rp[1:])
        col = (key - r * n).astype(np.int32)
        val = rng.integers(1, 10, len(col)).astype(np.float64) if vals is None else vals(len(col))
        return rp.astype(np.int32), col, val

    # every bin of the reference's table (bhsparse.h:373-406): products per row from 0 to > 512 and rows whose merged
    # list outgrows the 256-entry first EM buffer (rounds 1.. of compute_nnzC_Ct_mergepath, bhsparse_cuda.h:2527ff)
    m = 600
    lensA = rng.integers(0, 40, m)
    lensA[::7] = 0
    A = random_csr(m, 500, lensA)
    B = random_csr(500, 700, rng.integers(0, 40, 500))
    out.append(("rand_bins", True, m, 500, 700, A, B))

    # exact cancellation: structural zeros must stay (membership by key only, bhsparse_cuda.h:608-648, 1311-1397)
    A = random_csr(64, 64, np.full(64, 6), vals=lambda s: rng.choice([-1.0, 1.0], s))
    out.append(("cancel", True, 64, 64, 64, A, A))

    # power-law rows: long EM rows incl. the global-memory merge (bhsparse_cuda.h:2270-2525)
    rp, col = gallery.powerlaw_csr(20000, 20000, 90000, 3000)
    val = gallery.fill_values(len(col))
    out.append(("powerlaw_20k", False, 20000, 20000, 20000, (rp, col, val), (rp, col, val)))
    rp, col = gallery.powerlaw_csr(3000, 3000, 14000, 700)
    val = gallery.fill_values(len(col))
    out.append(("powerlaw_3k", True, 3000, 3000, 3000, (rp, col, val), (rp, col, val)))
    return out


def digest_of(Cp, Cj, Cx):
    t = np.arange(len(Cj), dtype=np.uint64) % np.uint64(8191) + np.uint64(1)
    return {"nnzC": int(len(Cj)), "sum_rowptr": int(Cp.astype(np.uint64).sum()),
            "wsum_col": int((Cj.astype(np.uint64) * t).sum()),
            "sum_val": float(Cx.sum()), "wsum_val": float((Cx * t.astype(np.float64)).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_opencl"))
    ap.add_argument("--only", default="")
    ap.add_argument("--full-size", action="store_true", help="only BASELINE.json's own sizes (digests; merged into the report of --out)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    report = {}
    if args.full_size:
        path = os.path.join(args.out, "ref_opencl_digests.json")
        if os.path.exists(path):
            report = json.load(open(path))
        elif os.path.exists(os.path.join(GOLD, "ref_opencl_digests.json")):
            report = json.load(open(os.path.join(GOLD, "ref_opencl_digests.json")))
        full_size(args, report)
        with open(path, "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        bad = [t for t, e in report.items() if "error" in e or not (e["oracle_rowptr_equal"] and e["oracle_col_equal"])]
        print("cases:", len(report), "reference/oracle disagreements or failures:", bad, flush=True)
        return
    for tag, full, m, k, n, A, B in cases():
        if args.only and tag not in args.only.split(","):
            continue
        Ap, Aj, Ax = A
        Bp, Bj, Bx = B
        entry = {"m": m, "k": k, "n": n, "nnzA": int(len(Aj)), "nnzB": int(len(Bj))}
        try:
            Cp, Cj, Cx, log = run_reference(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
        except Exception as e:  # keep going: the report says which case the reference could not do
            entry["error"] = str(e)[-1500:]
            report[tag] = entry
            print(tag, "REFERENCE FAILED:", entry["error"], flush=True)
            continue
        Cj_s, Cx_s, was_sorted = sort_rows(Cp, Cj, Cx)
        oCp, oCj, oCx = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
        entry.update(digest_of(Cp, Cj_s, Cx_s))
        entry["nnzCt"] = oracle.nnzCt(Ap, Aj, Bp)
        entry["rows_sorted_by_reference"] = was_sorted
        entry["oracle_rowptr_equal"] = bool(np.array_equal(oCp, Cp.astype(np.int64)))
        entry["oracle_col_equal"] = bool(len(oCj) == len(Cj_s) and np.array_equal(oCj, Cj_s))
        if entry["oracle_col_equal"]:
            entry["oracle_val_bit_equal"] = bool(np.array_equal(oCx, Cx_s))
            den = np.maximum(np.abs(oCx), 1e-300)
            entry["oracle_val_max_rel_err"] = float((np.abs(oCx - Cx_s) / den).max()) if len(oCx) else 0.0
        entry["explicit_zeros_in_C"] = int((Cx_s == 0.0).sum())
        entry["reference_stdout_tail"] = log.strip().splitlines()[-8:]
        report[tag] = entry
        print(tag, {k2: v for k2, v in entry.items() if k2 != "reference_stdout_tail"}, flush=True)
        if full:
            np.savez_compressed(os.path.join(args.out, "ref_opencl_%s.npz" % tag),
                                m=m, k=k, n=n,
                                Ap=np.asarray(Ap, np.int32), Aj=np.asarray(Aj, np.int32), Ax=np.asarray(Ax, np.float64),
                                Bp=np.asarray(Bp, np.int32), Bj=np.asarray(Bj, np.int32), Bx=np.asarray(Bx, np.float64),
                                Cp=Cp, Cj=Cj, Cx=Cx, rows_sorted=was_sorted, nnzCt=entry["nnzCt"])
    with open(os.path.join(args.out, "ref_opencl_digests.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    bad = [t for t, e in report.items() if "error" in e or not (e["oracle_rowptr_equal"] and e["oracle_col_equal"])]
    print("cases:", len(report), "reference/oracle disagreements or failures:", bad, flush=True)


if __name__ == "__main__":
    main()
