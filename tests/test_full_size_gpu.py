"""BASELINE.json configs[1] (poisson5pt 1024^2), configs[2] (poisson27pt 128^3) and the power-law stand-in for
configs[3] (webbase-1M; the SuiteSparse file is not in the image) at FULL size.

The oracle needs minutes for these, so parity is established through size-independent properties,
all evaluated on the device:
  * digests of C (nnzC, sum of rowPtr, position-weighted column sum, position-weighted value sum) against
    tests/golden/checksums.json, which scipy.sparse computed for the same inputs (make_golden.py --full);
  * the closed forms of the stencil: nnz(C) = (5N-6)^3 for the 27-point cube, 13N^2 - 20N + 4 for the 5-point
    square (every pair of grid points within two steps);
  * structure: rowPtr monotone from 0 to nnzC, columns strictly increasing inside every row;
  * linearity: C.1 == A.(B.1) and C.w == A.(B.w) for w_j = j + 1 (integer-valued inputs: exact in fp64);
  * idempotence: a second multiply on the same handle gives the same bits.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

CASES = {"p5_1024": ("poisson5pt", (1024, 1024, 1)), "p27_128": ("poisson27pt", (128, 128, 128)),
         "powerlaw_1m": ("powerlaw", (1000005, 3105536, 4700)),   # round 1-3 stand-in for configs[3]: hub-heavy, compression 1.007
         "weblike_1m": ("weblike", (1000005,)),                   # stand-in for configs[3] (webbase-1M, file absent) with its compression: 1.35
         "fem3_40": ("fem3", (40, 40, 40)),                       # poisson27pt (x) ones(3, 3): the big-class kernels
         "rmat_s20": ("rmat", (1 << 20,))}                        # R-MAT graph: 52 k rows of 2 k .. 150 k entries -- the column-window kernels (bhs_row_window.hip.h)


def _spmv(rp, col, val, x):
    import torch
    m = rp.numel() - 1
    rows = torch.repeat_interleave(torch.arange(m, device=rp.device), (rp[1:] - rp[:-1]).long())
    y = torch.zeros(m, dtype=torch.float64, device=rp.device)
    y.index_add_(0, rows, val * x[col.long()])
    return y


# cases the REFERENCE ITSELF multiplied at this size on an MI355X (oracle/make_ref_golden.py --full-size; its OpenCL merge
# truncates rows beyond 25 600 entries, which powerlaw_1m has and weblike_1m does not)
REF_AT_FULL_SIZE = ("p5_1024", "p27_128", "weblike_1m", "fem3_40")


# (tag, library options): the library's own choice of path for every case, and the GENERAL pipeline (no row classes, no
# direct launches) for the two grid inputs whose default is the class path -- the reference's digests pin both
@pytest.mark.parametrize("tag,opts", [("p5_1024", {}), ("p27_128", {}), ("powerlaw_1m", {}), ("weblike_1m", {}), ("fem3_40", {}), ("rmat_s20", {}),
                                      ("p27_128", {"class_path": 0}), ("fem3_40", {"class_path": 0}),
                                      ("p27_128", {"class_path": 0, "wave_first": 0, "lane_first": 0, "direct_bins": 0})],
                         ids=lambda v: v if isinstance(v, str) else ("default" if not v else "-".join("%s%d" % kv for kv in v.items())))
def test_full_size_digests_and_properties(hiplib, tag, opts):
    import torch
    from benchmark_spgemm_using_csr_amd import gallery, facade
    from benchmark_spgemm_using_csr_amd.dist import device_view
    path = os.path.join(GOLDEN, "checksums.json")
    ref = json.load(open(path)).get(tag)
    assert ref is not None, "run tests/golden/make_golden.py --full"
    stencil, dims = CASES[tag]
    dev = torch.device("cuda", 0)
    if stencil == "powerlaw":
        rp, col = gallery.powerlaw_csr(dims[0], dims[0], dims[1], dims[2])
        Bp, Bj = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    elif stencil == "weblike":
        rp, col = gallery.weblike_csr(dims[0])
        Bp, Bj = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    elif stencil == "fem3":
        rp, col = gallery.block_expand_csr(*gallery.poisson_csr("poisson27pt", *dims), 3)
        Bp, Bj = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    elif stencil == "rmat":
        rp, col = gallery.rmat_csr()
        Bp, Bj = torch.from_numpy(np.asarray(rp)).to(dev), torch.from_numpy(np.asarray(col)).to(dev)
    else:
        Bp, Bj = gallery.poisson_csr_torch(stencil, *dims, device=dev)
    Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
    Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
    m = Bp.numel() - 1
    assert m == ref["m"] and Bj.numel() == ref["nnzA"]
    plats = [False] * facade.NUM_PLATFORMS
    plats[facade.BHSPARSE_HIP] = True
    bh = facade.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    for k_, v_ in opts.items():
        assert bh.set_option(k_, v_) == 0
    assert bh.spgemm() == 0
    if "class_path" in opts:                                 # (the general pipeline is what ran)
        assert "numeric_class" not in [s["name"] for s in bh.kernel_stats() if s["launches"] > 0]
    if stencil == "rmat":                                    # (the multiply has enough such rows for the window kernels to be chosen)
        assert "b_windows" in [s["name"] for s in bh.kernel_stats()]
    nnzC = bh.get_nnzC()
    N = dims[0]
    closed = {"poisson27pt": (5 * N - 6) ** 3, "poisson5pt": 13 * N * N - 20 * N + 4}.get(stencil, ref["nnzC"])
    assert nnzC == closed == ref["nnzC"]
    assert bh.nnzCt == ref["nnzCt"]
    pr, pc, pv = bh.get_C_device()
    Cp = device_view(pr, m + 1, torch.int32, dev)
    Cj = device_view(pc, nnzC, torch.int32, dev)
    Cx = device_view(pv, nnzC, torch.float64, dev)

    # structure
    cp64 = Cp.long()
    assert int(cp64[0]) == 0 and int(cp64[-1]) == nnzC and bool((cp64[1:] >= cp64[:-1]).all())
    inc = Cj[1:] > Cj[:-1]
    starts = cp64[1:-1]
    starts = starts[(starts > 0) & (starts < nnzC)]
    inc[starts - 1] = True                                   # a row's first column may be anything
    assert bool(inc.all()), "columns must be strictly increasing inside every row"
    assert int(Cj.min()) >= 0 and int(Cj.max()) < m

    # digests (same definitions as oracle_digest / make_golden.digest; int64 wraps like uint64)
    w = torch.arange(nnzC, device=dev, dtype=torch.int64) % 8191 + 1
    assert int(cp64.sum()) == ref["sum_rowptr"]
    wsum_col = int((Cj.long() * w).sum()) & 0xFFFFFFFFFFFFFFFF
    assert wsum_col == ref["wsum_col"]
    assert float(Cx.sum()) == ref["sum_val"]
    assert float((Cx * w.double()).sum()) == ref["wsum_val"]
    # ... and against the digests of the REFERENCE's own C for this input (tests/golden/ref_opencl_digests.json)
    if tag in REF_AT_FULL_SIZE:
        rref = json.load(open(os.path.join(GOLDEN, "ref_opencl_digests.json"))).get(tag)
        assert rref is not None and "error" not in rref, "run oracle/make_ref_golden.py --full-size on an MI355X box"
        assert rref["rows_sorted_by_reference"] and rref["oracle_digest_equal"]
        assert (nnzC, int(cp64.sum()), wsum_col) == (rref["nnzC"], rref["sum_rowptr"], rref["wsum_col"])
        assert float(Cx.sum()) == rref["sum_val"] and float((Cx * w.double()).sum()) == rref["wsum_val"]
        assert bh.nnzCt == rref["nnzCt"]

    # linearity
    ones = torch.ones(m, dtype=torch.float64, device=dev)
    wcol = torch.arange(1, m + 1, dtype=torch.float64, device=dev)
    for x in (ones, wcol):
        lhs = _spmv(Cp, Cj, Cx, x)
        rhs = _spmv(Ap, Aj, Ax, _spmv(Bp, Bj, Bx, x))
        assert torch.equal(lhs, rhs)

    # idempotence
    keep = (Cp.clone(), Cj.clone(), Cx.clone())
    assert bh.spgemm() == 0
    pr, pc, pv = bh.get_C_device()
    assert torch.equal(keep[0], device_view(pr, m + 1, torch.int32, dev))
    assert torch.equal(keep[1], device_view(pc, nnzC, torch.int32, dev))
    assert torch.equal(keep[2], device_view(pv, nnzC, torch.float64, dev))
    bh.free_mem()
    bh.freePlatform()


def _spmv_chunked(rp, col, val, x, rows_per_chunk=1 << 20):
    """y = M.x in row chunks: keeps the int64 temporaries of a 2-G-entry matrix at a few hundred MB."""
    import torch
    m = rp.numel() - 1
    y = torch.empty(m, dtype=torch.float64, device=rp.device)
    for r0 in range(0, m, rows_per_chunk):
        r1 = min(m, r0 + rows_per_chunk)
        lo, hi = int(rp[r0]), int(rp[r1])
        lens = (rp[r0 + 1:r1 + 1] - rp[r0:r1]).long()
        rows = torch.repeat_interleave(torch.arange(r1 - r0, device=rp.device), lens)
        acc = torch.zeros(r1 - r0, dtype=torch.float64, device=rp.device)
        acc.index_add_(0, rows, val[lo:hi] * x[col[lo:hi].long()])
        y[r0:r1] = acc
    return y


def test_poisson27pt_256_cubed_on_one_gpu(hiplib, oracle):
    """BASELINE.json configs[4]'s matrix (poisson27pt 256^3: m = 16.8 M, 449 M entries, 12.07 G products,
    nnz(C) = 2 067 798 824 -- the int32 edge of the whole design) multiplied on ONE MI355X.  Parity through the
    closed forms, structure, linearity, idempotence, and the oracle on a row block out of the middle of the
    matrix (the oracle takes a row block of A against the full B)."""
    import torch
    from benchmark_spgemm_using_csr_amd import gallery, facade
    from benchmark_spgemm_using_csr_amd.dist import device_view
    N = 256
    dev = torch.device("cuda", 0)
    free_b, _total = torch.cuda.mem_get_info(dev)
    if free_b < 120 * 2 ** 30:
        pytest.skip("needs ~100 GB of free HBM (A, B, C and the checks' temporaries); %.0f GB free" % (free_b / 2 ** 30))
    Bp, Bj = gallery.poisson_csr_torch("poisson27pt", N, N, N, device=dev)
    Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
    Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
    m = Bp.numel() - 1
    assert m == N ** 3 and Bj.numel() == (3 * N - 2) ** 3
    plats = [False] * facade.NUM_PLATFORMS
    plats[facade.BHSPARSE_HIP] = True
    bh = facade.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    assert bh.spgemm() == 0
    nnzC = bh.get_nnzC()
    assert bh.nnzCt == (9 * N - 10) ** 3 == 12072028184
    assert nnzC == (5 * N - 6) ** 3 == 2067798824
    pr, pc, pv = bh.get_C_device()
    Cp = device_view(pr, m + 1, torch.int32, dev)
    Cj = device_view(pc, nnzC, torch.int32, dev)
    Cx = device_view(pv, nnzC, torch.float64, dev)

    # structure, in chunks of rows
    assert int(Cp[0]) == 0 and int(Cp[-1]) == nnzC and bool((Cp[1:] >= Cp[:-1]).all())
    per_row = (Cp[1:] - Cp[:-1])
    assert int(per_row.max()) == 125 and int(per_row.min()) == 27       # interior rows / the 8 corners
    step = 1 << 21
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        lo, hi = int(Cp[r0]), int(Cp[r1])
        cj = Cj[lo:hi]
        inc = cj[1:] > cj[:-1]
        starts = (Cp[r0 + 1:r1].long() - lo)
        inc[starts - 1] = True
        assert bool(inc.all()), "columns must be strictly increasing inside every row (rows %d..%d)" % (r0, r1)
        assert int(cj.min()) >= 0 and int(cj.max()) < m

    # linearity (integer-valued inputs: exact in fp64)
    ones = torch.ones(m, dtype=torch.float64, device=dev)
    wcol = (torch.arange(m, dtype=torch.float64, device=dev) % 1021.0) + 1.0
    for x in (ones, wcol):
        lhs = _spmv_chunked(Cp, Cj, Cx, x)
        rhs = _spmv_chunked(Ap, Aj, Ax, _spmv_chunked(Bp, Bj, Bx, x))
        assert torch.equal(lhs, rhs)

    # the oracle on a block of rows from the middle of the matrix (rows that sit in the EM bin of the reference)
    r0 = (N // 2) * N * N + (N // 2) * N - 1500
    nrows = 3000
    hBp, hBj, hBx = Bp.cpu().numpy(), Bj.cpu().numpy(), Bx.cpu().numpy()
    lo, hi = int(hBp[r0]), int(hBp[r0 + nrows])
    ap = (hBp[r0:r0 + nrows + 1] - lo).astype(np.int32)
    ref = oracle.spgemm(nrows, m, m, ap, hBj[lo:hi], hBx[lo:hi], hBp, hBj, hBx)
    clo, chi = int(Cp[r0]), int(Cp[r0 + nrows])
    got = ((Cp[r0:r0 + nrows + 1].long() - clo).to(torch.int32).cpu().numpy(), Cj[clo:chi].cpu().numpy(),
           Cx[clo:chi].cpu().numpy())
    res = oracle.compare(ref, got, rel_tol=0.0)
    assert res["ok"], res
    del hBp, hBj, hBx

    # idempotence: digests of a second multiply (a full clone of C would double the footprint)
    def digest():
        w = (torch.arange(1 << 22, device=dev, dtype=torch.float64) % 8191.0) + 1.0
        s_col, s_val = 0, 0.0
        for o in range(0, nnzC, 1 << 22):
            e = min(nnzC, o + (1 << 22))
            s_col += int((Cj[o:e].long() * (w[:e - o].long())).sum())
            s_val += float((Cx[o:e] * w[:e - o]).sum())
        return int(Cp.long().sum()), s_col & 0xFFFFFFFFFFFFFFFF, s_val
    d1 = digest()
    assert bh.spgemm() == 0
    pr2, pc2, pv2 = bh.get_C_device()
    assert (pr2, pc2, pv2) == (pr, pc, pv) and bh.get_nnzC() == nnzC
    assert digest() == d1
    bh.free_mem()
    bh.freePlatform()
    del Ap, Aj, Ax, Bp, Bj, Bx, Cp, Cj, Cx
    torch.cuda.empty_cache()            # hand the ~60 GB back before the other modules allocate through hipMalloc


def test_webbase_1m_from_file(hiplib, oracle):
    """BASELINE.json configs[3] on the REAL SuiteSparse file when one is supplied: BHS_WEBBASE_MTX=/path/to/
    webbase-1M.mtx.  The file is not in this image (no network), which is what the skip says."""
    path = os.environ.get("BHS_WEBBASE_MTX", "")
    if not path or not os.path.exists(path):
        pytest.skip("SuiteSparse Williams/webbase-1M is not in the image: set BHS_WEBBASE_MTX=<path to webbase-1M.mtx>"
                    " to run configs[3] on the real file (the seeded power-law stand-in is covered above)")
    import scipy.io
    import scipy.sparse as sp
    from benchmark_spgemm_using_csr_amd.facade import spgemm_csr
    A = sp.csr_matrix(scipy.io.mmread(path))
    A.sort_indices()
    m, n = A.shape
    assert (m, A.nnz) == (1000005, 3105536), "not webbase-1M: %r nnz=%d" % (A.shape, A.nnz)
    rp, col = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    from benchmark_spgemm_using_csr_amd import gallery
    val = gallery.fill_values(len(col))
    Cp, Cj, Cx, info = spgemm_csr(m, n, n, rp, col, val, rp, col, val)
    ref = oracle.spgemm(m, n, n, rp, col, val, rp, col, val)
    assert info["nnzCt"] == oracle.nnzCt(rp, col, rp)
    res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
    assert res["ok"], res
    print("webbase-1M: nnzCt=%d nnzC=%d" % (info["nnzCt"], info["nnzC"]))


def test_nnz_overflow_is_an_error_code(hiplib):
    """nnz(C) beyond int32 (the reference's index type, bhsparse.h:367,431): poisson27pt 260^3 squared has (5 * 260 - 6)^3 =
    2 169 112 376 entries.  The multiply must stop with BHS_ERR_NNZ_OVERFLOW behind its symbolic half -- before any array
    of C is allocated (the inputs are 11.4 GB; C would be 26 GB)."""
    import torch
    from benchmark_spgemm_using_csr_amd import gallery, facade
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40 * (1 << 30):
        pytest.skip("needs 40 GB of free device memory")
    N = 260
    Bp, Bj = gallery.poisson_csr_torch("poisson27pt", N, N, N, device=dev)
    Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
    m = Bp.numel() - 1
    assert (5 * N - 6) ** 3 > 2 ** 31 - 1
    plats = [False] * facade.NUM_PLATFORMS
    plats[facade.BHSPARSE_HIP] = True
    for opts in ({}, {"class_path": 0}):
        bh = facade.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.initData_device(m, m, m, Bj.numel(), Bx, Bp, Bj, Bj.numel(), Bx, Bp, Bj) == 0
        for k_, v_ in opts.items():
            assert bh.set_option(k_, v_) == 0
        assert bh.spgemm() == facade._lib.BHS_ERR_NNZ_OVERFLOW
        assert bh.free_mem() == 0
        assert bh.freePlatform() == 0


@pytest.mark.parametrize("case", ["perturbed_0.1pct", "long_and_hub_rows"])
def test_poisson27pt_128_cubed_with_irregular_rows(hiplib, oracle, case):
    """Round 6, mixed mode of the class path at BASELINE configs[2]'s size: poisson27pt 128^3 with 0.1 % of its rows given an
    extra entry, or with a 300-entry row, a 7000-entry row (its 189 000 products: a hub row, n > 2^20 columns) and an empty
    row.  The class kernels did run, the irregular rows are few, C has CSR's structure, C.x = A.(B.x) exactly, the whole
    of C equals the general pipeline's bit for bit, and a block of rows around an irregular one equals the CPU oracle's."""
    import torch
    from benchmark_spgemm_using_csr_amd import gallery, facade
    from benchmark_spgemm_using_csr_amd.dist import device_view
    dev = torch.device("cuda", 0)
    rp0, col0 = gallery.poisson_csr("poisson27pt", 128, 128, 128)
    m = len(rp0) - 1
    if case.startswith("perturbed"):
        rp, col = gallery.perturb_rows_csr(rp0, col0, m, 0.001, seed=11)
        probe = int(np.flatnonzero(np.diff(rp) != np.diff(rp0))[len(rp) // 4000])
    else:
        rp, col = gallery.perturb_rows_csr(rp0, col0, m, 0.0, long_row=(m // 2 + 5, 300))
        rp, col = gallery.perturb_rows_csr(rp, col, m, 0.0, long_row=(m // 3, 7000))
        lens = np.diff(rp).copy()
        keep = np.ones(len(col), bool); keep[rp[12345]:rp[12346]] = False
        lens[12345] = 0
        col = col[keep]; rp = np.zeros(m + 1, np.int32); rp[1:] = np.cumsum(lens)
        probe = m // 3
    del rp0, col0
    val = gallery.fill_values(len(col))
    Ap, Aj, Ax = (torch.from_numpy(x).to(dev) for x in (rp, col, val))
    plats = [False] * facade.NUM_PLATFORMS
    plats[facade.BHSPARSE_HIP] = True
    outs = []
    for opts in ({}, {"class_path": 0}):
        bh = facade.bhsparse()
        assert bh.initPlatform(plats) == 0
        for k_, v_ in opts.items():
            assert bh.set_option(k_, v_) == 0
        assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Aj.numel(), Ax, Ap, Aj) == 0
        for _ in range(2):
            assert bh.spgemm() == 0
        names = {s["name"] for s in bh.kernel_stats() if s["launches"] > 0}
        if not opts:
            assert "numeric_class" in names and bh.get_info("class_state") == 2
            assert 0 < bh.get_info("mixed_rows") < m // 20
        else:
            assert "numeric_class" not in names
        nnzC = bh.get_nnzC()
        pr, pc, pv = bh.get_C_device()
        Cp = device_view(pr, m + 1, torch.int32, dev).clone()
        Cj = device_view(pc, nnzC, torch.int32, dev).clone()
        Cx = device_view(pv, nnzC, torch.float64, dev).clone()
        outs.append((bh.nnzCt, nnzC, Cp, Cj, Cx))
        assert bh.free_mem() == 0 and bh.freePlatform() == 0
    (ct0, n0, Cp, Cj, Cx), (ct1, n1, Cp1, Cj1, Cx1) = outs
    assert ct0 == ct1 and n0 == n1
    assert torch.equal(Cp, Cp1) and torch.equal(Cj, Cj1) and torch.equal(Cx, Cx1)      # the two pipelines: bit for bit
    cp64 = Cp.long()
    assert int(cp64[0]) == 0 and int(cp64[-1]) == n0 and bool((cp64[1:] >= cp64[:-1]).all())
    inc = Cj[1:] > Cj[:-1]
    starts = cp64[1:-1]; starts = starts[(starts > 0) & (starts < n0)]
    inc[starts - 1] = True
    assert bool(inc.all())
    x = (torch.arange(m, device=dev, dtype=torch.float64) % 7) + 1.0
    lhs = _spmv(Cp, Cj, Cx, x)
    rhs = _spmv(Ap, Aj, Ax, _spmv(Ap, Aj, Ax, x))
    assert torch.equal(lhs, rhs)                                                           # (integer-valued: exact in fp64)
    # a block of rows around an irregular one against the oracle
    r0, r1 = max(0, probe - 40), min(m, probe + 40)
    Ab = ((rp[r0:r1 + 1] - rp[r0]).astype(np.int32), col[rp[r0]:rp[r1]], val[rp[r0]:rp[r1]])
    ref = oracle.spgemm(r1 - r0, m, m, Ab[0], Ab[1], Ab[2], rp, col, val)
    c0, c1 = int(Cp[r0]), int(Cp[r1])
    got = ((Cp[r0:r1 + 1] - c0).cpu().numpy().astype(np.int32), Cj[c0:c1].cpu().numpy(), Cx[c0:c1].cpu().numpy())
    assert oracle.compare(ref, got, rel_tol=0.0)["ok"]
