"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against
the CPU oracle and the committed golden vectors.

Bar (north_star): rowPtr / colInd bit-exact; values within 1e-6 relative
(REL_TOL below); integer-valued inputs must match bit-exactly.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from helpers import check_csr_invariants, poisson_case, random_csr
from benchmark_spgemm_using_csr_amd import facade as bhmod
from benchmark_spgemm_using_csr_amd.facade import spgemm_csr

pytestmark = pytest.mark.gpu
REL_TOL = 1e-6


def _check(oracle, m, k, n, A, B, exact=True, options=None):
    """One multiply against the oracle.  Unless the caller decides about the row-class path (bhs_class.hip.h) itself,
    the multiply runs twice: with the library's defaults -- structured inputs then take the class kernels -- and with
    class_path = 0, the general pipeline whose kernels most tests here are about; the second run's results and
    kernel list are returned, and both must agree with the oracle."""
    Ap, Aj, Ax = A
    Bp, Bj, Bx = B
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
    if options is None or "class_path" not in options:
        Cp, Cj, Cx, info = spgemm_csr(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, options=dict(options or {}, class_path=2))
        assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Bp) and info["nnzC"] == ref[0][-1]
        res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=REL_TOL)
        assert res["ok"], ("class path / defaults", res)
        if exact:
            assert np.array_equal(Cx, ref[2])
        options = dict(options or {}, class_path=0)
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, options=options)
    assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Bp)
    assert info["nnzC"] == ref[0][-1]
    res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=REL_TOL)
    assert res["ok"], res
    if exact:
        assert np.array_equal(Cx, ref[2])
    check_csr_invariants(m, n, Cp, Cj)
    return Cp, Cj, Cx, info


def test_reference_call_sequence_small_test(oracle):
    """test_small_spgemm (SpGEMM_cuda/main.cu:149-246) through the mirrored API."""
    g = load_golden("small_test.npz")
    m, k, n = 4, 6, 4
    platforms = [False] * bhmod.NUM_PLATFORMS
    platforms[bhmod.BHSPARSE_HIP] = True
    csrRowPtrC = np.zeros(m + 1, np.int32)
    bh = bhmod.bhsparse()
    assert bh.initPlatform(platforms) == bhmod.BHSPARSE_SUCCESS
    assert bh.initData(m, k, n, 6, g["Ax"], g["Ap"], g["Aj"], 7, g["Bx"], g["Bp"], g["Bj"], csrRowPtrC) == 0
    assert bh.spgemm() == 0
    nnzC = bh.get_nnzC()
    csrColIndC = np.empty(nnzC, np.int32)
    csrValC = np.empty(nnzC, np.float64)
    assert bh.get_C(csrColIndC, csrValC) == 0
    assert bh.free_mem() == 0
    assert bh.freePlatform() == 0
    assert nnzC == 6 and bh.nnzCt == 7
    assert csrRowPtrC.tolist() == [0, 1, 4, 4, 6]
    assert csrColIndC.tolist() == [0, 0, 1, 3, 1, 3]
    assert csrValC.tolist() == [10, 120, 190, 60, 120, 180]


def test_cuda_and_opencl_flags_alias_the_hip_backend():
    for slot in (bhmod.BHSPARSE_CUDA, bhmod.BHSPARSE_OPENCL):
        p = [False] * bhmod.NUM_PLATFORMS
        p[slot] = True
        bh = bhmod.bhsparse()
        assert bh.initPlatform(p) == 0
        assert bh.freePlatform() == 0
    bh = bhmod.bhsparse()
    assert bh.initPlatform([False] * bhmod.NUM_PLATFORMS) != 0


def test_cage4(oracle):
    g = load_golden("cage4_sq.npz")
    A = (g["Ap"], g["Aj"], g["Ax"])
    Cp, Cj, Cx, info = _check(oracle, 9, 9, 9, A, A, exact=False)
    assert info["nnzCt"] == 269 and info["nnzC"] == 81
    assert np.array_equal(Cj, g["Cj"])
    assert np.allclose(Cx, g["Cx"], rtol=REL_TOL, atol=0)
    ones = (g["Ap"], g["Aj"], np.ones_like(g["Ax"]))
    _, _, C1, _ = _check(oracle, 9, 9, 9, ones, ones)
    assert np.array_equal(C1, g["Cx_ones"])


@pytest.mark.parametrize("name", ["p5_16.npz", "p27_6.npz", "p9_12.npz", "p7_7.npz", "rect_rand.npz"])
def test_golden_fixtures(oracle, name):
    g = load_golden(name)
    Cp, Cj, Cx, info = spgemm_csr(int(g["m"]), int(g["k"]), int(g["n"]), g["Ap"], g["Aj"], g["Ax"],
                                  g["Bp"], g["Bj"], g["Bx"])
    assert np.array_equal(Cp, g["Cp"]) and np.array_equal(Cj, g["Cj"]) and np.array_equal(Cx, g["Cx"])
    assert info["nnzCt"] == int(g["nnzCt"])


REF_OPENCL_CASES = ["small_test", "p5_16", "p27_6", "p9_12", "p7_7", "rect_rand", "cage4", "cage4_ones", "p27_12",
                    "rand_bins", "cancel", "powerlaw_3k"]


@pytest.mark.parametrize("class_path", [0, 2])
@pytest.mark.parametrize("tag", REF_OPENCL_CASES)
def test_hip_equals_reference_opencl_output(tag, class_path):
    """The HIP path against outputs of the REFERENCE ITSELF (its OpenCL branch run on an MI355X,
    oracle/make_ref_golden.py -> tests/golden/ref_opencl_*.npz) -- no oracle in between.  rowPtr / colInd bit-exact,
    values bit-exact for the integer-valued inputs, 1e-6 relative for cage4's reals; every bin of the reference's
    table (rand_bins), its multi-round EM merge (powerlaw_3k) and retained structural zeros (cancel)."""
    g = load_golden("ref_opencl_%s.npz" % tag)
    assert bool(g["rows_sorted"])
    Cp, Cj, Cx, info = spgemm_csr(int(g["m"]), int(g["k"]), int(g["n"]), g["Ap"], g["Aj"], g["Ax"],
                                  g["Bp"], g["Bj"], g["Bx"], options={"class_path": class_path})
    assert info["nnzCt"] == int(g["nnzCt"]) and info["nnzC"] == len(g["Cj"])
    assert np.array_equal(Cp, g["Cp"]) and np.array_equal(Cj, g["Cj"])
    if tag == "cage4":
        assert np.all(np.abs(Cx - g["Cx"]) <= REL_TOL * np.abs(g["Cx"]))
    else:
        assert np.array_equal(Cx, g["Cx"])
    if tag == "cancel":
        assert int((Cx == 0.0).sum()) == 168


@pytest.mark.parametrize("stencil,dims", [("poisson5pt", (256, 256, 1)), ("poisson9pt", (256, 256, 1)),
                                          ("poisson7pt", (51, 51, 51)), ("poisson27pt", (51, 51, 51))])
def test_reference_default_datasets(oracle, stencil, dims):
    """-spgemm 1..4 of the reference driver (main.cu:30-53)."""
    m, rp, col, val = poisson_case(stencil, *dims)
    A = (rp, col, val)
    Cp, Cj, Cx, info = _check(oracle, m, m, m, A, A)
    if stencil in ("poisson5pt", "poisson27pt"):
        tag = "p5_256" if stencil == "poisson5pt" else "p27_51"
        ref = json.load(open(os.path.join(GOLDEN, "checksums.json")))[tag]
        d = oracle.digest(Cp.astype(np.int64), Cj, Cx)
        assert [d[0], d[1], d[2]] == [ref["nnzC"], ref["sum_rowptr"], ref["wsum_col"]]


@pytest.mark.parametrize("ub", [0, 1, 2, 32, 33, 48, 49, 64, 65, 96, 97, 128, 129, 192, 193, 256, 257, 384, 385,
                                512, 513, 768, 769, 1536, 1537, 3072, 3073, 6144, 6145, 24576, 24577, 70000])
def test_bin_edges(oracle, ub):
    """Synthetic rows whose product count sits on every bin edge of the reference
    (bhsparse.h:373-406) and of this implementation's symbolic/numeric bins."""
    rng = np.random.default_rng(ub + 1)
    k = 64
    n = 200000
    # B: k rows; row j has lenB[j] entries.  A row 1 references rows so that sum(len) == ub.
    lens = []
    rest = ub
    while rest > 0:
        L = min(rest, int(rng.integers(1, 2500)))
        lens.append(L)
        rest -= L
    lenB = np.zeros(k, np.int64)
    assert len(lens) <= k
    lenB[:len(lens)] = lens
    Bp = np.zeros(k + 1, np.int64)
    np.cumsum(lenB, out=Bp[1:])
    Bj = np.empty(Bp[-1], np.int32)
    for j in range(k):
        # overlapping column ranges => duplicates across B rows
        base = int(rng.integers(0, 3000))
        Bj[Bp[j]:Bp[j + 1]] = base + np.sort(rng.choice(max(lenB[j] * 2, 1), lenB[j], replace=False))
    Bx = rng.integers(1, 10, Bp[-1]).astype(np.float64)
    # A: row 0 empty, row 1 = the row under test, row 2 a short row, row 3 references an empty B row only
    a1 = np.arange(len(lens), dtype=np.int32)
    rows = [np.empty(0, np.int32), a1, np.array([0], np.int32) if len(lens) else np.empty(0, np.int32),
            np.array([k - 1], np.int32)]
    Ap = np.zeros(5, np.int32)
    Ap[1:] = np.cumsum([len(r) for r in rows])
    Aj = np.concatenate(rows).astype(np.int32)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    _check(oracle, 4, k, n, (Ap, Aj, Ax), (Bp.astype(np.int32), Bj, Bx))


@pytest.mark.parametrize("nA,lenB,dup", [(1, 1, 0), (5, 5, 1), (16, 3, 1), (17, 2, 1), (16, 4, 0), (12, 4, 0), (13, 4, 0),
                                         (3, 40, 1), (2, 33, 0), (16, 9, 1)])
def test_quarter_wave_rows(oracle, nA, lenB, dup):
    """Rows around the limits of the four-rows-per-wavefront kernel (<= 16 A entries, <= 48 products
    symbolic / <= 48 entries numeric, windows of 64 products): many such rows so that full and
    partially filled quartets both occur."""
    rng = np.random.default_rng(nA * 100 + lenB)
    k, n, m = 400, 5000, 203
    Bp = np.arange(k + 1, dtype=np.int64) * lenB
    Bj = np.empty(k * lenB, np.int32)
    for j in range(k):
        lo = (j // 2 if dup else j) * 3 % (n - 4 * lenB)         # dup: neighbouring B rows overlap
        Bj[j * lenB:(j + 1) * lenB] = lo + np.sort(rng.choice(2 * lenB, lenB, replace=False))
    Bx = rng.integers(1, 10, k * lenB).astype(np.float64)
    lens = np.full(m, nA)
    lens[::7] = 0                                                # empty rows in between
    lens[3::11] = max(1, nA - 1)
    Ap = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=Ap[1:])
    Aj = np.concatenate([np.sort(rng.choice(k, L, replace=False)) for L in lens] + [np.empty(0, np.int64)]).astype(np.int32)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    A, B = (Ap.astype(np.int32), Aj, Ax), (Bp.astype(np.int32), Bj, Bx)
    r0 = _check(oracle, m, k, n, A, B, options={"lane_rows": 0})
    _check(oracle, m, k, n, A, B, options={"lane_rows": 0, "no_pack32": 1})
    # the same rows through the lane-per-row kernel (default for nA <= 12; forced otherwise, where rows with
    # more than 12 entries stay with the quarter-wave kernel)
    for mode in (1, 2):
        r1 = _check(oracle, m, k, n, A, B, options={"lane_rows": mode, "lane_numeric": mode - 1})
        assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r1[:3]))


def test_b_addressing_paths(oracle):
    """colIndB / valB loads: 32-bit byte offsets (nnz(B) < 2^29, default) and the general 64-bit path."""
    m, rp, col, val = poisson_case("poisson27pt", 10, 10, 10)
    A = (rp, col, val)
    Cp, Cj, Cx, _ = _check(oracle, m, m, m, A, A)
    Cp2, Cj2, Cx2, _ = _check(oracle, m, m, m, A, A, options={"small_b": 0})
    assert np.array_equal(Cj, Cj2) and np.array_equal(Cx, Cx2)


@pytest.mark.parametrize("case", ["p27", "p5", "p9", "random", "banded_long_rows", "rect_blocks"])
def test_compressed_symbolic_pass(oracle, case):
    """Symbolic pass on the compressed pattern of B ((column >> 5, mask) pairs): forced on, against the oracle
    and against the plain pass; the default (auto) mode must agree as well."""
    rng = np.random.default_rng(77)
    if case == "p27":
        m, rp, col, val = poisson_case("poisson27pt", 11, 9, 10)
        A = B = (rp, col, val); k = n = m
    elif case == "p5":
        m, rp, col, val = poisson_case("poisson5pt", 40, 37)
        A = B = (rp, col, val); k = n = m
    elif case == "p9":
        m, rp, col, val = poisson_case("poisson9pt", 33, 31)
        A = B = (rp, col, val); k = n = m
    elif case == "random":                      # no adjacent columns: compression does not pay, still correct
        m, k, n = 300, 250, 4000
        A = random_csr(m, k, 0.05, rng, empty_rows=(0, 9))
        B = random_csr(k, n, 0.01, rng, empty_rows=(2,))
    elif case == "banded_long_rows":            # A rows with > 64 entries (chunk loop), B rows of 70..90 adjacent columns
        m, k, n = 200, 400, 3000
        A = random_csr(m, k, 0.3, rng)
        lens = rng.integers(70, 91, k)
        starts = rng.integers(0, n - 100, k)
        Bp = np.zeros(k + 1, np.int32); np.cumsum(lens, out=Bp[1:])
        Bj = np.concatenate([np.arange(s0, s0 + l0) for s0, l0 in zip(starts, lens)]).astype(np.int32)
        B = (Bp, Bj, rng.integers(1, 10, len(Bj)).astype(np.float64))
    else:                                       # rectangular, blocks of 4 adjacent columns at random places
        m, k, n = 500, 300, 100000
        A = random_csr(m, k, 0.04, rng)
        heads = [np.sort(rng.choice(n // 4, rng.integers(0, 12), replace=False)) * 4 for _ in range(k)]
        Bj = np.concatenate([np.add.outer(h, np.arange(4)).ravel() for h in heads] + [np.empty(0, np.int64)]).astype(np.int32)
        Bp = np.zeros(k + 1, np.int32); np.cumsum([4 * len(h) for h in heads], out=Bp[1:])
        B = (Bp, Bj, rng.integers(1, 10, len(Bj)).astype(np.float64))
    ref = _check(oracle, m, k, n, A, B, options={"compress_b": 0})
    for mode in (2, 1):
        got = _check(oracle, m, k, n, A, B, options={"compress_b": mode})
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])


@pytest.mark.parametrize("case", ["p5", "p7", "p9", "tiny_random", "tiny_random_short_b", "wide_random", "cancel", "unsorted_b"])
def test_lane_per_row_kernel(oracle, case):
    """k_row_lane (one row per lane, K-way merge of sorted B rows in registers) against the oracle and against
    the table kernels; it must step aside for unsorted B."""
    rng = np.random.default_rng(5)
    kernels_wanted = True
    if case in ("p5", "p7", "p9"):
        dims = {"p5": ("poisson5pt", 37, 41, 1), "p7": ("poisson7pt", 9, 11, 8), "p9": ("poisson9pt", 23, 19, 1)}[case]
        m, rp, col, val = poisson_case(*dims)
        A = B = (rp, col, val); k = n = m
    elif case == "tiny_random":                  # 0..12 entries per A row, B rows of 0..30 entries, empty rows in both
        m, k, n = 777, 300, 2000
        A = random_csr(m, k, 0.02, rng, empty_rows=(0, 5, 776), max_row=12)
        B = random_csr(k, n, 0.008, rng, empty_rows=(1, 2), max_row=30)
    elif case == "tiny_random_short_b":          # ... B rows of at most 12: cheap enough for a lane's walk (round 6: kLaneCost)
        m, k, n = 777, 300, 2000
        A = random_csr(m, k, 0.02, rng, empty_rows=(0, 5, 776), max_row=12)
        B = random_csr(k, n, 0.004, rng, empty_rows=(1, 2), max_row=12)
    elif case == "wide_random":                  # the same with 20 000 columns: 64 lanes would read 64 far-apart B rows
        m, k, n = 3000, 20000, 20000
        A = random_csr(m, k, 0.0003, rng, empty_rows=(0, 5), max_row=12)
        B = random_csr(k, n, 0.0002, rng, empty_rows=(1, 2), max_row=30)
    elif case == "cancel":                       # exact cancellation keeps the structural zero
        m = k = n = 3
        A = (np.array([0, 2, 2, 3], np.int32), np.array([0, 1, 2], np.int32), np.array([1.0, -1.0, 2.0]))
        B = (np.array([0, 2, 4, 5], np.int32), np.array([0, 2, 0, 1, 1], np.int32), np.array([3.0, 1.0, 3.0, 5.0, 7.0]))
    else:
        m, k, n = 200, 100, 500
        A = random_csr(m, k, 0.05, rng, max_row=8)
        Bp, Bj, Bx = random_csr(k, n, 0.02, rng)
        Bj = Bj.copy()
        for j in range(k):
            Bj[Bp[j]:Bp[j + 1]] = Bj[Bp[j]:Bp[j + 1]][::-1]
        B = (Bp, Bj, Bx)
        kernels_wanted = False
    sb = 0 if case == "unsorted_b" else 1          # (left unsorted, B keeps the lane kernel out)
    ref = _check(oracle, m, k, n, A, B, options={"lane_rows": 0})
    for mode, num in ((1, 0), (2, 0), (1, 1), (2, 1)):
        got = _check(oracle, m, k, n, A, B, options={"lane_rows": mode, "lane_numeric": num, "sort_b": sb})
        assert all(np.array_equal(x, y) for x, y in zip(ref[:3], got[:3]))
        names = {s["name"] for s in got[3]["kernels"] if s["launches"]}
        # (round 4: lane_rows = 1 leaves the lane kernels to inputs whose rows of A stay near the diagonal, or whose B is
        # small -- the hint "local_a" of bhs_set_data; the random columns of `wide_random` take them only when
        # lane_rows = 2 insists)
        # (round 6: nor where the B rows are long -- a lane's walk costs ~ nA nB^2, bhsparse_hip.hip kLaneCost: 12 x 30 x 30 of
        # `tiny_random` is beyond it, the wave kernels take every row; its rows of <= 56 entries still leave through the
        # numeric lane kernel where lane_numeric = 1 asks for it)
        wanted = kernels_wanted and (mode == 2 or case not in ("wide_random", "tiny_random"))
        assert ("symbolic_lane" in names) == wanted, names
        assert ("numeric_lane" in names) == ((wanted or (case == "tiny_random" and kernels_wanted)) and num == 1), names
    if case == "cancel":
        assert ref[0].tolist() == [0, 3, 3, 4] and ref[2][0] == 0.0


@pytest.mark.parametrize("lens", [[0, 1, 2, 64, 65, 128, 129, 300, 512, 513, 1024], [1025, 3, 4096, 4097, 0, 9000, 70],
                                  [5] * 3000])
@pytest.mark.parametrize("f32", [False, True])
def test_device_row_sort(oracle, lens, f32):
    """bhs_csr_sort_indices_device against the host csr_sort_indices restatement (ref_spgemm.h:37-62): stable,
    in place, every row-length class (register sort, LDS network, HBM scratch), duplicates kept in input order."""
    import torch
    rng = np.random.default_rng(len(lens))
    lens = np.array(lens)
    rp = np.zeros(len(lens) + 1, np.int32)
    np.cumsum(lens, out=rp[1:])
    n = 20000
    col = np.concatenate([rng.integers(0, n, L) if i % 3 == 0 else rng.permutation(n)[:L] for i, L in enumerate(lens)]
                         + [np.empty(0, np.int64)]).astype(np.int32)
    col[rp[3]:rp[4]] = np.sort(col[rp[3]:rp[4]])                 # one row already in order
    val = np.arange(len(col), dtype=np.float64) + 0.5              # distinct: shows where every entry came from
    ecol, eval_ = col.copy(), val.copy()
    for i in range(len(lens)):
        o = np.argsort(col[rp[i]:rp[i + 1]], kind="stable")
        ecol[rp[i]:rp[i + 1]] = col[rp[i]:rp[i + 1]][o]
        eval_[rp[i]:rp[i + 1]] = val[rp[i]:rp[i + 1]][o]
    oc, ov = col.copy(), val.copy()
    oracle.csr_sort_indices(rp, oc, ov)
    assert np.array_equal(oc, ecol) and np.array_equal(ov, eval_)
    dt = np.float32 if f32 else np.float64
    dev = torch.device("cuda", 0)
    d_rp, d_col = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    d_val = torch.from_numpy(val.astype(dt)).to(dev)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse(value_dtype=dt)
    assert bh.initPlatform(plats) == 0
    assert bh.csr_sort_indices_device(len(lens), d_rp, d_col, d_val) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_col.cpu().numpy(), ecol)
    assert np.array_equal(d_val.cpu().numpy(), eval_.astype(dt))
    assert bh.csr_sort_indices_device(len(lens), d_rp, d_col, d_val) == 0      # idempotent
    assert np.array_equal(d_col.cpu().numpy(), ecol)
    assert bh.freePlatform() == 0


@pytest.mark.parametrize("stencil,dims", [("poisson5pt", (33, 29, 1)), ("poisson7pt", (8, 9, 7)), ("poisson9pt", (21, 18, 1))])
def test_direct_stages_without_queue(oracle, stencil, dims):
    """All rows in the lane bin (symbolic) and in the quad bin (numeric): both stages run without their queue
    (no fill_queues launch); forcing the queues back must give the same C."""
    m, rp, col, val = poisson_case(stencil, *dims)
    A = (rp, col, val)
    r1 = _check(oracle, m, m, m, A, A)
    names = {s["name"] for s in r1[3]["kernels"] if s["launches"]}
    numk = "numeric_quad<64>" if stencil == "poisson9pt" else "numeric_lane"      # lane numeric only while K <= 8
    assert "fill_queues" not in names and "symbolic_lane" in names and numk in names, names
    rows = {s["name"]: (s["rows"], s["products"], s["nnz_out"]) for s in r1[3]["kernels"]}
    assert rows[numk] == (m, r1[3]["nnzCt"], r1[3]["nnzC"])
    r0 = _check(oracle, m, m, m, A, A, options={"direct_bins": 0})
    assert "fill_queues" in {s["name"] for s in r0[3]["kernels"] if s["launches"]}
    assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r1[:3]))
    r2 = _check(oracle, m, m, m, A, A, options={"lane_rows": 0})                  # symbolic and numeric both direct on the quad kernel
    r3 = _check(oracle, m, m, m, A, A, options={"lane_numeric": 1})               # ... and both on the lane kernel
    r4 = _check(oracle, m, m, m, A, A, options={"lane_numeric": 0})
    assert "upper_bound" not in names                                              # lane-first: no upper-bound pass either
    r5 = _check(oracle, m, m, m, A, A, options={"lane_first": 0, "wave_first": 0})
    assert "upper_bound" in {s["name"] for s in r5[3]["kernels"] if s["launches"]}
    r6 = _check(oracle, m, m, m, A, A, options={"lane_first": 0})                  # falls to the wave-first symbolic pass
    n6 = {s["name"] for s in r6[3]["kernels"] if s["launches"]}
    assert "upper_bound" not in n6 and any(x.startswith("symbolic_wave") for x in n6), n6
    for r in (r2, r3, r4, r5, r6):
        assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r[:3]))


def test_wave_first_symbolic_pass(oracle):
    """maxRow(A) x maxRow(B) fits a wave table (poisson27pt: 729 <= 768): no upper-bound pass, no symbolic queue;
    the symbolic kernel supplies ub[] and the product total.  Rows with > 64 entries in A, empty rows and a bound
    that does not fit must all still agree with the plain pipeline."""
    rng = np.random.default_rng(3)
    m, rp, col, val = poisson_case("poisson27pt", 9, 8, 10)
    A = (rp, col, val)
    # (compress_b = 0: on a 9 x 8 x 10 grid a row's 27 columns fall into 3 blocks of 32 -- the compressed symbolic pass, chosen
    # since round 4 where B has <= 25 % as many (block, mask) pairs as entries, would take this input with its upper-bound pass)
    r1 = _check(oracle, m, m, m, A, A, options={"compress_b": 0})
    rc = _check(oracle, m, m, m, A, A)
    assert "compress_b" in {s["name"] for s in rc[3]["kernels"] if s["launches"]}
    assert all(np.array_equal(x, y) for x, y in zip(rc[:3], r1[:3]))
    n1 = {s["name"]: s for s in r1[3]["kernels"] if s["launches"]}
    assert "upper_bound" not in n1 and "symbolic_wave<1024>" in n1, sorted(n1)
    assert n1["symbolic_wave<1024>"]["rows"] == m and n1["symbolic_wave<1024>"]["products"] == r1[3]["nnzCt"]
    r0 = _check(oracle, m, m, m, A, A, options={"wave_first": 0, "compress_b": 0})
    assert "upper_bound" in {s["name"] for s in r0[3]["kernels"] if s["launches"]}
    assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r1[:3]))
    # A rows of up to 100 entries (chunk loop), B rows of <= 20: bound 2000 -> the 4096-slot table
    mm, kk, nn = 300, 400, 3000
    A2 = random_csr(mm, kk, 0.2, rng, empty_rows=(0, 17, 299), max_row=100)
    B2 = random_csr(kk, nn, 0.005, rng, empty_rows=(3,), max_row=20)
    a1 = _check(oracle, mm, kk, nn, A2, B2, options={"compress_b": 0})
    a0 = _check(oracle, mm, kk, nn, A2, B2, options={"wave_first": 0, "compress_b": 0})
    assert "upper_bound" not in {s["name"] for s in a1[3]["kernels"] if s["launches"]}
    assert all(np.array_equal(x, y) for x, y in zip(a0[:3], a1[:3]))
    # one long row of A breaks the bound: plain pipeline
    A3 = random_csr(mm, kk, 0.05, rng)
    A3[1][:] = np.sort(A3[1].reshape(-1)) if False else A3[1]
    lens = np.diff(A3[0]); lens[5] = 390
    rp3 = np.zeros(mm + 1, np.int32); np.cumsum(lens, out=rp3[1:])
    col3 = np.concatenate([np.sort(rng.choice(kk, L, replace=False)) for L in lens]).astype(np.int32)
    A3 = (rp3, col3, rng.integers(1, 10, len(col3)).astype(np.float64))
    b1 = _check(oracle, mm, kk, nn, A3, B2)
    assert "upper_bound" in {s["name"] for s in b1[3]["kernels"] if s["launches"]}


def test_compression_is_automatic_for_heavy_clustered_rows(oracle):
    """3-dof FEM-like pattern (poisson27pt (x) ones(3,3): 81 entries per row, 6561 products, columns in runs of 3):
    the default options compress B's pattern so that the symbolic pass stays in the wave kernels; poisson27pt itself
    (729 products per row) must NOT be compressed."""
    import scipy.sparse as sp
    from benchmark_spgemm_using_csr_amd import gallery
    rp, col = gallery.poisson_csr("poisson27pt", 6, 5, 6)
    P = sp.csr_matrix((np.ones(len(col)), col, rp), shape=(len(rp) - 1,) * 2)
    A3 = sp.kron(P, np.ones((3, 3)), format="csr")
    A3.sort_indices()
    m = A3.shape[0]
    A = (A3.indptr.astype(np.int32), A3.indices.astype(np.int32), gallery.fill_values(A3.nnz))
    r1 = _check(oracle, m, m, m, A, A)
    n1 = {s["name"] for s in r1[3]["kernels"] if s["launches"]}
    assert "compress_b" in n1 and not any(x.startswith("symbolic_wg") for x in n1), n1
    r0 = _check(oracle, m, m, m, A, A, options={"compress_b": 0})
    assert "compress_b" not in {s["name"] for s in r0[3]["kernels"] if s["launches"]}
    assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r1[:3]))
    # (a grid line of 40 nodes: the nine runs of 3 columns of a row sit in nine different blocks of 32 -- a third as many
    # pairs as entries, above the 25 % from which the pass is chosen for rows of fewer than 1536 products.  On a 7 x 7 x 7
    # grid a whole plane's nine columns share a block, and the compressed pass does take it.)
    mm, rp, col, val = poisson_case("poisson27pt", 40, 5, 5)
    r2 = _check(oracle, mm, mm, mm, (rp, col, val), (rp, col, val))
    assert "compress_b" not in {s["name"] for s in r2[3]["kernels"] if s["launches"]}
    mm, rp, col, val = poisson_case("poisson27pt", 7, 7, 7)
    r3 = _check(oracle, mm, mm, mm, (rp, col, val), (rp, col, val))
    assert "compress_b" in {s["name"] for s in r3[3]["kernels"] if s["launches"]}


def test_sort_key_width_paths(oracle):
    """32-bit packed sort keys vs the 64-bit fallback must agree (wave and quarter-wave kernels)."""
    m, rp, col, val = poisson_case("poisson27pt", 14, 14, 14)
    A = (rp, col, val)
    a = _check(oracle, m, m, m, A, A)
    b = _check(oracle, m, m, m, A, A, options={"no_pack32": 1})
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    m, rp, col, val = poisson_case("poisson5pt", 40, 40)
    A = (rp, col, val)
    _check(oracle, m, m, m, A, A, options={"no_pack32": 1})
    _check(oracle, m, m, m, A, A, options={"force_path": 1})     # quad bin off: same rows through the wave kernel


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_rectangular(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    m, k, n = 700 + 13 * seed, 500, 900
    A = random_csr(m, k, 0.02, rng, empty_rows=(0, 5, m - 1))
    B = random_csr(k, n, 0.03, rng, empty_rows=(3, 4))
    _check(oracle, m, k, n, A, B)


def test_float_values_within_tolerance(oracle):
    rng = np.random.default_rng(5)
    A = random_csr(400, 400, 0.05, rng, values="normal")
    B = random_csr(400, 400, 0.05, rng, values="normal")
    _check(oracle, 400, 400, 400, A, B, exact=False)


def test_exact_cancellation_keeps_structural_zero(oracle):
    Ap = np.array([0, 2, 2], np.int32); Aj = np.array([0, 1], np.int32); Ax = np.array([1.0, -1.0])
    Bp = np.array([0, 1, 2], np.int32); Bj = np.array([0, 0], np.int32); Bx = np.array([1.0, 1.0])
    Cp, Cj, Cx, _ = spgemm_csr(2, 2, 2, Ap, Aj, Ax, Bp, Bj, Bx)
    assert Cp.tolist() == [0, 1, 1] and Cj.tolist() == [0] and Cx.tolist() == [0.0]


def test_empty_inputs(oracle):
    z = np.zeros(1, np.int32)
    e_i, e_v = np.empty(0, np.int32), np.empty(0, np.float64)
    # m == 0
    Cp, Cj, Cx, info = spgemm_csr(0, 3, 3, z, e_i, e_v, np.zeros(4, np.int32), e_i, e_v)
    assert Cp.tolist() == [0] and info["nnzC"] == 0
    # all-empty A
    Cp, Cj, Cx, info = spgemm_csr(5, 3, 3, np.zeros(6, np.int32), e_i, e_v,
                                  np.array([0, 1, 2, 3], np.int32), np.array([0, 1, 2], np.int32), np.ones(3))
    assert Cp.tolist() == [0] * 6 and len(Cj) == 0
    # A references only empty rows of B
    Cp, Cj, Cx, info = spgemm_csr(2, 3, 3, np.array([0, 1, 2], np.int32), np.array([1, 1], np.int32), np.ones(2),
                                  np.array([0, 1, 1, 2], np.int32), np.array([0, 2], np.int32), np.ones(2))
    assert Cp.tolist() == [0, 0, 0] and info["nnzCt"] == 0


def test_unsorted_B_rows_still_correct(oracle):
    rng = np.random.default_rng(11)
    A = random_csr(300, 200, 0.05, rng)
    Bp, Bj, Bx = random_csr(200, 5000, 0.02, rng)
    for j in range(200):                     # shuffle inside rows
        p = rng.permutation(Bp[j + 1] - Bp[j])
        Bj[Bp[j]:Bp[j + 1]] = Bj[Bp[j]:Bp[j + 1]][p]
        Bx[Bp[j]:Bp[j + 1]] = Bx[Bp[j]:Bp[j + 1]][p]
    r0 = _check(oracle, 300, 200, 5000, A, (Bp, Bj, Bx), options={"sort_b": 0})     # multiplied as they are
    _check(oracle, 300, 200, 5000, A, (Bp, Bj, Bx), options={"sort_b": 0, "max_table_log2": 6})
    r1 = _check(oracle, 300, 200, 5000, A, (Bp, Bj, Bx))                             # default: sorted at set_data time
    assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r1[:3]))
    # device-pointer API: the caller's arrays are borrowed and must come back untouched
    import torch
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (A[0], A[1], A[2], Bp, Bj, Bx)]
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.initData_device(300, 200, 5000, len(A[1]), t[2], t[0], t[1], len(Bj), t[5], t[3], t[4]) == 0
    assert bh.spgemm() == 0
    assert bh.get_nnzC() == r0[3]["nnzC"] and np.array_equal(bh.get_rowptrC(), r0[0])
    Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
    assert bh.get_C(Cj, Cx) == 0
    assert np.array_equal(Cj, r0[1]) and np.array_equal(Cx, r0[2])
    assert np.array_equal(t[4].cpu().numpy(), Bj) and np.array_equal(t[5].cpu().numpy(), Bx)
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.parametrize("where", ["first_pair", "part_boundary", "last_pair", "sorted"])
def test_long_rows_of_A_and_B_in_the_row_scans(oracle, where):
    """Rows far beyond the lanes-per-row of the per-row scans: k_upper_bound hands A rows with more than 512 entries
    to k_upper_bound_long, k_check_sorted hands B rows with more than 4096 entries to k_check_sorted_long (16
    workgroups per row each).  One swapped pair anywhere in a 40 000-entry row of B must still be seen."""
    rng = np.random.default_rng(21)
    n = 50000
    Ap, Aj = _dense_row_case(n, 6, {7, 30000}, 40000, rng)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    Bj, Bx = Aj.copy(), Ax.copy()
    a0, L = int(Ap[30000]), int(Ap[30001] - Ap[30000])
    assert L == 40000
    if where != "sorted":
        e = a0 + {"first_pair": 0, "part_boundary": L * 5 // 16 - 1, "last_pair": L - 2}[where]
        Bj[e], Bj[e + 1] = Bj[e + 1], Bj[e]
        Bx[e], Bx[e + 1] = Bx[e + 1], Bx[e]
    Cp, Cj, Cx, info = _check(oracle, n, n, n, (Ap, Aj, Ax), (Ap, Bj, Bx))
    assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Ap)
    # multiplied as they are (sort_b = 0): an unsorted B disables the kernels that need ascending rows, which shows
    # in the symbolic kernel mix only if the check saw the swapped pair
    if where != "sorted":
        plats = [False] * bhmod.NUM_PLATFORMS
        plats[bhmod.BHSPARSE_HIP] = True
        bh = bhmod.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.set_option("sort_b", 0) == 0
        c = np.zeros(n + 1, np.int32)
        assert bh.initData(n, n, n, len(Aj), Ax, Ap, Aj, len(Bj), Bx, Ap, Bj, c) == 0
        assert bh.get_info("b_sorted") == 0 and bh.get_info("max_row_b") == 40000
        assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.parametrize("cap", [6, 8, 10])
def test_column_window_path_forced(oracle, cap):
    """Capping the LDS table forces the column-window (long-row) path on ordinary rows:
    the counterpart of the reference's EM overflow/requeue rounds (bhsparse_cuda.h:2527-2780)."""
    m, rp, col, val = poisson_case("poisson27pt", 12, 12, 12)
    A = (rp, col, val)
    _check(oracle, m, m, m, A, A, options={"max_table_log2": cap})
    rng = np.random.default_rng(cap)
    A = random_csr(200, 300, 0.2, rng)
    B = random_csr(300, 40000, 0.01, rng)
    _check(oracle, 200, 300, 40000, A, B, options={"max_table_log2": cap})


def test_hub_row_power_law(oracle):
    """webbase-1M stand-in (config C4, file absent): power-law rows with hub rows far
    beyond any LDS table."""
    from benchmark_spgemm_using_csr_amd import gallery
    m = 30000
    rp, col = gallery.powerlaw_csr(m, m, 100000, 3000, hubs=3)
    val = gallery.fill_values(len(col))
    A = (rp, col, val)
    Cp, Cj, Cx, info = _check(oracle, m, m, m, A, A)
    assert np.diff(Cp).max() > 6144      # at least one row is beyond every LDS table
    names = [k["name"] for k in info["kernels"]]
    assert "numeric_long_rows" in names
    # the three long-row paths give the same bits: bitmap in LDS (default for n <= 2^20), bitmap in HBM
    # (wider matrices), column windows (fallback of both)
    # ... and so do the bins of a stage launched concurrently on side streams or one after another
    for opts in ({"lds_bitmap": 0}, {"spa": 0}, {"concurrent_bins": 1}, {"concurrent_bins": 0}):
        Cp2, Cj2, Cx2, _ = _check(oracle, m, m, m, A, A, options=opts)
        assert np.array_equal(Cj, Cj2) and np.array_equal(Cx, Cx2)
    # float values through the global fp64 atomics stay within tolerance
    valf = np.random.default_rng(1).standard_normal(len(col))
    _check(oracle, m, m, m, (rp, col, valf), (rp, col, valf), exact=False)


def _dense_row_case(n, per_row, dense_rows, dense_nnz, rng):
    """A sparse square matrix with a few rows of dense_nnz entries: in A^2 those rows carry dense_nnz x per_row
    products, and every row that points at one of them inherits its dense_nnz entries."""
    rows = []
    for i in range(n):
        if i in dense_rows:
            rows.append(np.sort(rng.choice(n, dense_nnz, replace=False)))
        else:
            rows.append(np.unique(rng.integers(0, n, int(rng.integers(max(per_row - 3, 1), per_row + 4)))))
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    return rp, col


@pytest.mark.parametrize("case", ["dense_rows", "tiny_items_batches", "wide_segments", "float_values", "f32_build"])
def test_hub_rows_split_across_workgroups(oracle, case):
    """Hub bin (bhs_hub.hip.h): a row with hub_min_products products or more is cut into items that the whole device
    works on, its accumulator a bitmap slot shared by all of its workgroups.  Replaces the reference's multi-round
    global merge (bhsparse_cuda.h:2270-2525, :2527-2780).  Same bits as the one-workgroup-per-row kernels."""
    rng = np.random.default_rng(11)
    value_dtype = np.float32 if case == "f32_build" else np.float64
    if case == "wide_segments":
        # 3 M columns: 16 bitmap segments per row; B rows collide inside a pool, first / last columns present
        n, k = 3000000, 3000
        pool = np.sort(rng.choice(n, 8000, replace=False))
        rowsB = []
        for j in range(k):
            L = int(rng.integers(20, 60))
            rowsB.append(np.sort(rng.choice(pool, L, replace=False) if j < k // 2 else rng.choice(n, L, replace=False)))
        rowsB[k - 1] = np.array([0, 15, 16, 31, 32, n - 2, n - 1])
        Bp = np.zeros(k + 1, np.int32); Bp[1:] = np.cumsum([len(r) for r in rowsB])
        Bj = np.concatenate(rowsB).astype(np.int32)
        rowsA = [np.arange(k), np.sort(rng.choice(k, 1500, replace=False)), np.sort(rng.choice(k, 10, replace=False)),
                 np.empty(0, np.int64), np.sort(rng.choice(k // 2, 900, replace=False)), np.arange(k - 700, k)]
        Ap = np.zeros(len(rowsA) + 1, np.int32); Ap[1:] = np.cumsum([len(r) for r in rowsA])
        Aj = np.concatenate(rowsA).astype(np.int32)
        m = len(rowsA)
        opts = {"hub_min_products": 20000}
    else:
        n = k = m = 40000
        Ap, Aj = _dense_row_case(n, 9, {5, 17000, n - 1}, 16000, rng)
        Bp, Bj = Ap, Aj
        opts = {} if case == "dense_rows" else {"hub_min_products": 3000}
        if case == "tiny_items_batches":
            opts.update({"hub_item_products": 64, "hub_slots": 2})
    if case == "float_values":
        Ax = Bx = rng.standard_normal(len(Aj))
    else:
        Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
        Bx = Ax if Bj is Aj else rng.integers(1, 10, len(Bj)).astype(np.float64)
    A, B = (Ap, Aj, Ax), (Bp, Bj, Bx)
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, Ap, Aj, Ax.astype(value_dtype), Bp, Bj, Bx.astype(value_dtype),
                                  options=opts, value_dtype=value_dtype)
    names = _kernel_names(info)
    assert {"symbolic_hub_rows", "numeric_hub_rows"} <= names, names
    assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Bp)
    assert np.array_equal(Cp, ref[0]) and np.array_equal(Cj, ref[1])
    if case == "float_values":
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=REL_TOL)["ok"]
    else:
        assert np.array_equal(Cx.astype(np.float64), ref[2])             # small integers: exact in either build
    check_csr_invariants(m, n, Cp, Cj)
    hub = [kk for kk in info["kernels"] if kk["name"] == "numeric_hub_rows"][0]
    assert hub["rows"] >= 3 and hub["products"] >= 3 * (20000 if case == "wide_segments" else 3000)
    # the hub bin off: the same rows through the one-workgroup-per-row kernels
    Cp2, Cj2, Cx2, info2 = spgemm_csr(m, k, n, Ap, Aj, Ax.astype(value_dtype), Bp, Bj, Bx.astype(value_dtype),
                                      options={"hub_min_products": 0}, value_dtype=value_dtype)
    assert "numeric_hub_rows" not in _kernel_names(info2)
    assert np.array_equal(Cp, Cp2) and np.array_equal(Cj, Cj2)
    if case != "float_values":
        assert np.array_equal(Cx, Cx2)
    # one atomic per product instead of one per run of neighbouring lanes in the same bitmap word
    Cp3, Cj3, Cx3, info3 = spgemm_csr(m, k, n, Ap, Aj, Ax.astype(value_dtype), Bp, Bj, Bx.astype(value_dtype),
                                      options=dict(opts, hub_aggregate=0), value_dtype=value_dtype)
    assert "numeric_hub_rows" in _kernel_names(info3)
    assert np.array_equal(Cp, Cp3) and np.array_equal(Cj, Cj3)
    if case != "float_values":
        assert np.array_equal(Cx, Cx3)


def test_hub_rows_in_row_ranges(oracle):
    """The numeric half in row ranges with hub rows in several of them (their queue entries carry the range's
    row numbers, their output base comes from rowPtrC)."""
    import ctypes as C
    rng = np.random.default_rng(12)
    n = 30000
    Ap, Aj = _dense_row_case(n, 8, {0, 9000, 20000, n - 1}, 9000, rng)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    ref = oracle.spgemm(n, n, n, Ap, Aj, Ax, Ap, Aj, Ax)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.set_option("hub_min_products", 5000) == 0
    Cp = np.zeros(n + 1, np.int32)
    assert bh.initData(n, n, n, len(Aj), Ax, Ap, Aj, len(Aj), Ax, Ap, Aj, Cp) == 0
    L, h = bh._lib, bh._h
    for nranges in (1, 4):
        ct, cc = C.c_int64(0), C.c_int(0)
        assert L.bhs_spgemm_symbolic(h, C.byref(ct), C.byref(cc)) == 0
        assert cc.value == ref[0][-1]
        cuts = [n * s // nranges for s in range(nranges + 1)]
        for s in reversed(range(nranges)):
            assert L.bhs_spgemm_numeric(h, cuts[s], cuts[s + 1]) == 0
        assert L.bhs_spgemm_finish(h, None) == 0
        Cj = np.empty(cc.value, np.int32); Cx = np.empty(cc.value, np.float64)
        assert bh.get_C(Cj, Cx) == 0
        res = oracle.compare(ref, (bh.get_rowptrC(), Cj, Cx), rel_tol=0.0)
        assert res["ok"], (nranges, res)
    assert "numeric_hub_rows" in {s["name"] for s in bh.kernel_stats()}
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.parametrize("n", [2 ** 20, 2 ** 20 + 1, 3000000, 40000, 200000, 400000, 2 ** 19 + 5])
def test_long_rows_bitmap_accumulators(oracle, n):
    """Rows beyond the LDS hash tables, with and without duplicate columns, either side of the 2^20-column limit
    of the LDS-resident bitmap (wider matrices keep the bitmap in HBM); the narrow ones have the LDS kernel's lanes
    own one or two eight-word groups of the bitmap, or none, instead of four.  Replaces the reference's
    EM_mergepath_global rounds (bhsparse_cuda.h:2270-2525)."""
    rng = np.random.default_rng(n % 1000)
    k = 2000
    pool = np.sort(rng.choice(n, 5000, replace=False))           # half of the B rows collide inside this pool
    rowsB = []
    for j in range(k):
        L = int(rng.integers(20, 60))
        cols = rng.choice(pool, L, replace=False) if j < k // 2 else rng.choice(n, L, replace=False)
        rowsB.append(np.sort(cols))
    rowsB[k - 1] = np.array([0, 15, 16, 31, 32, n - 2, n - 1])   # first / last columns, 16-column group edges
    rowsB[k - 2] = np.array([0, 16, n - 1])
    Bp = np.zeros(k + 1, np.int32); Bp[1:] = np.cumsum([len(r) for r in rowsB])
    Bj = np.concatenate(rowsB).astype(np.int32)
    Bx = rng.integers(1, 10, len(Bj)).astype(np.float64)
    rowsA = [np.sort(rng.choice(k, 1500, replace=False)), np.arange(k), np.sort(rng.choice(k, 10, replace=False)),
             np.empty(0, np.int64), np.sort(rng.choice(k // 2, 600, replace=False)),
             k // 2 + np.sort(rng.choice(k // 2, 300, replace=False)), np.arange(k - 700, k)]
    Ap = np.zeros(len(rowsA) + 1, np.int32); Ap[1:] = np.cumsum([len(r) for r in rowsA])
    Aj = np.concatenate(rowsA).astype(np.int32)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    A, B = (Ap, Aj, Ax), (Bp, Bj, Bx)
    Cp, Cj, Cx, info = _check(oracle, len(rowsA), k, n, A, B)
    assert np.diff(Cp).max() > 6144
    assert "numeric_long_rows" in [kk["name"] for kk in info["kernels"]]
    for opts in ({"lds_bitmap": 0}, {"spa": 0}, {"lds_bitmap_min_log2": 99}, {"window_bitmap": 2}):
        Cp2, Cj2, Cx2, _ = _check(oracle, len(rowsA), k, n, A, B, options=opts)
        assert np.array_equal(Cj, Cj2) and np.array_equal(Cx, Cx2)


# (watchdog: the window kernels hand a ticket from lane 0 to the wave between two divergent regions; when the compiler once
# folded the two, the kernel hung instead of failing -- a hang must fail THIS test, not stall the suite)
@pytest.mark.timeout(120)
@pytest.mark.parametrize("n", [2 ** 20, 300000, 70000])
def test_wave_per_row_column_windows(oracle, n):
    """Rows of a few thousand entries of C, one wave each, column window by column window (bhs_row_window.hip.h; it
    takes the place of the reference's EM_mergepath rounds, bhsparse_cuda.h:1043-1489, for graphs).  B's columns are
    skewed like an R-MAT graph's -- the windows are cut by B's entries, not by columns -- and the rows of A cover: up to
    64 entries (everything in registers), 65 .. 128 (two chunks), more (handed on to k_row_bitmap_lds), a window of
    more than 512 products (pass 2 loads again, several rounds of 384 entries), products crowded into one window
    (handed on), columns hit many times, an empty row of B, the first and the last column."""
    rng = np.random.default_rng(n % 977)
    k = 3000

    def skewed(count):
        c = (n * rng.random(count) ** 3).astype(np.int64)           # a third of the entries in the first 1/27 of the columns
        return np.unique(np.minimum(c, n - 1))

    rowsB = [skewed(int(rng.integers(150, 500))) for _ in range(k)]
    rowsB[5] = np.empty(0, np.int64)
    rowsB[6] = np.array([0, n - 1])
    for j in range(10, 40):                                          # thirty rows of B inside one stretch of 20 000 columns
        rowsB[j] = np.unique(n // 2 + rng.integers(0, 20000, 1500))
    Bp = np.zeros(k + 1, np.int32); Bp[1:] = np.cumsum([len(r) for r in rowsB])
    Bj = np.concatenate(rowsB).astype(np.int32)
    Bx = rng.integers(1, 10, len(Bj)).astype(np.float64)
    free = np.setdiff1d(np.arange(40, k), [5, 6])
    rowsA = [np.sort(rng.choice(free, int(rng.integers(8, 30)), replace=False)) for _ in range(300)]
    rowsA += [np.sort(rng.choice(free, 64, replace=False)), np.sort(rng.choice(free, 65, replace=False)),
              np.sort(rng.choice(free, 128, replace=False)), np.sort(rng.choice(free, 129, replace=False)),
              np.sort(rng.choice(free, 400, replace=False)), np.array([5, 6, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59]),
              np.arange(10, 40), np.array([5]), np.empty(0, np.int64)]
    Ap = np.zeros(len(rowsA) + 1, np.int32); Ap[1:] = np.cumsum([len(r) for r in rowsA])
    Aj = np.concatenate(rowsA).astype(np.int32)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    A, B = (Ap, Aj, Ax), (Bp, Bj, Bx)
    Cp, Cj, Cx, info = _check(oracle, len(rowsA), k, n, A, B, options={"window_bitmap": 2, "class_path": 0})
    names = [kk["name"] for kk in info["kernels"]]
    assert "b_windows" in names and np.diff(Cp).max() > 8192
    Cp0, Cj0, Cx0, info0 = _check(oracle, len(rowsA), k, n, A, B, options={"window_bitmap": 0, "class_path": 0})
    assert "b_windows" not in [kk["name"] for kk in info0["kernels"]]
    assert np.array_equal(Cj, Cj0) and np.array_equal(Cx, Cx0)
    # (by default a multiply with so few such rows leaves them with k_row_bitmap_lds)
    _, _, _, info1 = _check(oracle, len(rowsA), k, n, A, B, options={"class_path": 0})
    assert "b_windows" not in [kk["name"] for kk in info1["kernels"]]
    # the numeric half in row ranges (what the multi-GPU layer does): the index of B is built once per multiply, every
    # range has its own spill lists
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dA, dB = (t(Ap), t(Aj), t(Ax)), (t(Bp), t(Bj), t(Bx))
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.set_option("class_path", 0) == 0 and bh.set_option("window_bitmap", 2) == 0
    m = len(rowsA)
    assert bh.initData_device(m, k, n, len(Aj), dA[2], dA[0], dA[1], len(Bj), dB[2], dB[0], dB[1]) == 0
    L, h = bh._lib, bh._h
    ct, cc = C.c_int64(0), C.c_int(0)
    assert L.bhs_spgemm_symbolic(h, C.byref(ct), C.byref(cc)) == 0 and cc.value == Cp[-1]
    cuts = [0, m // 3, m // 3 + 7, m]
    for s3 in (2, 0, 1):
        assert L.bhs_spgemm_numeric(h, cuts[s3], cuts[s3 + 1]) == 0
    assert L.bhs_spgemm_finish(h, None) == 0
    Cj3 = np.empty(cc.value, np.int32); Cx3 = np.empty(cc.value, np.float64)
    assert bh.get_C(Cj3, Cx3) == 0
    assert np.array_equal(Cj3, Cj) and np.array_equal(Cx3, Cx)
    assert bh.free_mem() == 0 and bh.freePlatform() == 0
    # the value_type float build (sums of small integers: exact in float too)
    Cpf, Cjf, Cxf, infof = spgemm_csr(m, k, n, Ap, Aj, Ax.astype(np.float32), Bp, Bj, Bx.astype(np.float32),
                                      options={"window_bitmap": 2, "class_path": 0}, value_dtype=np.float32)
    assert "b_windows" in [kk["name"] for kk in infof["kernels"]]
    assert np.array_equal(Cpf, Cp) and np.array_equal(Cjf, Cj) and np.array_equal(Cxf.astype(np.float64), Cx)


def test_huge_column_space():
    """n = 2^31 - 1 columns: column indices up to INT_MAX - 1, 64-bit sort keys, bitmap accumulators
    out of range (falls back to column windows).  Checked against a numpy reference built here because
    the oracle's dense marker array would need 16 GB per thread."""
    rng = np.random.default_rng(9)
    n = 2 ** 31 - 1
    k = 300
    lensB = rng.integers(1, 60, k)
    lensB[:3] = 9000                                         # three long B rows -> one C row beyond the LDS tables
    Bp = np.zeros(k + 1, np.int64); np.cumsum(lensB, out=Bp[1:])
    Bj = np.concatenate([np.sort(rng.choice(n // 7, L, replace=False).astype(np.int64) * 7 + (n - 1) % 7)
                         for L in lensB]).astype(np.int32)
    Bj[-1] = n - 1                                           # the largest legal column
    for j in range(k):
        Bj[Bp[j]:Bp[j + 1]].sort()
    Bx = rng.integers(1, 10, len(Bj)).astype(np.float64)
    rowsA = [np.array([0, 1, 2, 5, 9]), np.array([k - 1]), np.empty(0, np.int64), np.arange(3, 40)]
    Ap = np.zeros(len(rowsA) + 1, np.int32); Ap[1:] = np.cumsum([len(r) for r in rowsA])
    Aj = np.concatenate(rowsA).astype(np.int32)
    Ax = rng.integers(1, 10, len(Aj)).astype(np.float64)
    Cp, Cj, Cx, info = spgemm_csr(len(rowsA), k, n, Ap, Aj, Ax, Bp.astype(np.int32), Bj, Bx)
    # reference: per row, accumulate products in a dict-free numpy way
    exp_ptr, exp_col, exp_val = [0], [], []
    for i in range(len(rowsA)):
        cols = np.concatenate([Bj[Bp[j]:Bp[j + 1]] for j in Aj[Ap[i]:Ap[i + 1]]] + [np.empty(0, np.int32)])
        vals = np.concatenate([Ax[Ap[i] + t] * Bx[Bp[j]:Bp[j + 1]] for t, j in enumerate(Aj[Ap[i]:Ap[i + 1]])] +
                              [np.empty(0)])
        u, inv = np.unique(cols, return_inverse=True)
        exp_col.append(u); exp_val.append(np.bincount(inv, weights=vals, minlength=len(u)))
        exp_ptr.append(exp_ptr[-1] + len(u))
    assert Cp.tolist() == exp_ptr
    assert np.array_equal(Cj, np.concatenate(exp_col)) and np.array_equal(Cx, np.concatenate(exp_val))
    assert Cj.max() == n - 1 and np.diff(Cp).max() > 6144


F32_TOL = 2e-5      # float accumulation of <= a few thousand products per entry


@pytest.mark.parametrize("case", ["p27", "p5", "rect", "hub"])
def test_float_value_type_build(oracle, case):
    """libbhsparse_hip_f32.so (value_type float, the reference's other build: README.md:84-86) against the
    double oracle: structure bit-exact, values within F32_TOL relative; integer-valued inputs whose sums stay
    below 2^24 are exact in float too."""
    rng = np.random.default_rng(3)
    if case == "p27":
        m, rp, col, val = poisson_case("poisson27pt", 13, 13, 13); k = n = m
        A = B = (rp, col, val)
    elif case == "p5":
        m, rp, col, val = poisson_case("poisson5pt", 60, 60); k = n = m
        A = B = (rp, col, val)
    elif case == "rect":
        m, k, n = 500, 400, 3000
        A = random_csr(m, k, 0.03, rng, empty_rows=(0, 7), values="normal")
        B = random_csr(k, n, 0.02, rng, values="normal")
    else:
        from benchmark_spgemm_using_csr_amd import gallery
        m = k = n = 20000
        rp, col = gallery.powerlaw_csr(m, m, 70000, 2500, hubs=3)
        A = B = (rp, col, gallery.fill_values(len(col)))
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, A[0], A[1], A[2].astype(np.float32), B[0], B[1], B[2].astype(np.float32),
                                  value_dtype=np.float32)
    assert Cx.dtype == np.float32
    ref = oracle.spgemm(m, k, n, A[0], A[1], A[2].astype(np.float32).astype(np.float64),
                        B[0], B[1], B[2].astype(np.float32).astype(np.float64))
    assert np.array_equal(Cp, ref[0]) and np.array_equal(Cj, ref[1])
    if case in ("p27", "p5", "hub"):
        assert np.array_equal(Cx.astype(np.float64), ref[2])          # small integers: exact
    else:
        # fp64 accumulation of exact products of float inputs, ONE rounding to float at the end: the error of an
        # entry is at most half a float ulp of its own magnitude (2^-24 relative) -- no growth with the number of
        # products and no cancellation term.  (Long-row bins that add in float would not meet this.)
        err = np.abs(Cx.astype(np.float64) - ref[2])
        assert np.all(err <= np.abs(ref[2]) * (2.0 ** -24 * 1.0001) + 1e-300), float(np.max(err / np.maximum(np.abs(ref[2]), 1e-300)))
    check_csr_invariants(m, n, Cp, Cj)


def test_repeated_spgemm_and_data_swap(oracle):
    """One handle, several multiplies and data sets: pooled workspace must not leak state."""
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    for dims in ((20, 20, 20), (9, 9, 9), (30, 30, 30)):
        m, rp, col, val = poisson_case("poisson27pt", *dims)
        Cp = np.zeros(m + 1, np.int32)
        assert bh.initData(m, m, m, len(col), val, rp, col, len(col), val, rp, col, Cp) == 0
        ref = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
        for _ in range(3):
            assert bh.warmup() == 0
        for _ in range(2):
            assert bh.spgemm() == 0
            nnzC = bh.get_nnzC()
            Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
            assert bh.get_C(Cj, Cx) == 0
            assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
        assert bh.free_mem() == 0
    assert bh.get_nnzC() == 0
    assert bh.freePlatform() == 0


@pytest.mark.gpu
def test_two_handles_back_to_back_on_the_general_pipeline(oracle):
    """the one-pass scan of rowPtrC (k_scan_onepass) takes a tile word whose epoch matches for published.  A
    new handle starts at the same epoch as the one before it and may be handed the very memory that handle's scan left
    its words in: create, multiply, destroy, create, multiply on DIFFERENT matrices of more than 8192 rows (several scan
    tiles), general pipeline, and compare rowPtrC / the whole product."""
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    rng = np.random.default_rng(77)
    cases = []
    for m, per in ((40000, 6), (52000, 9), (40000, 4)):
        c = np.sort((np.arange(m)[:, None] + rng.integers(-3000, 3000, (m, per))) % m, axis=1)
        keep = np.ones(c.shape, bool)
        keep[:, 1:] = c[:, 1:] != c[:, :-1]
        rp = np.zeros(m + 1, np.int32)
        np.cumsum(keep.sum(axis=1), out=rp[1:])
        col = c[keep].astype(np.int32)
        val = rng.integers(1, 10, len(col)).astype(np.float64)
        cases.append((m, rp, col, val))
    for rep in range(2):
        for m, rp, col, val in cases:
            bh = bhmod.bhsparse()
            assert bh.initPlatform(plats) == 0
            assert bh.set_option("class_path", 0) == 0
            Cp = np.zeros(m + 1, np.int32)
            assert bh.initData(m, m, m, len(col), val, rp, col, len(col), val, rp, col, Cp) == 0
            assert bh.spgemm() == 0
            nnzC = bh.get_nnzC()
            Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
            assert bh.get_C(Cj, Cx) == 0
            ref = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
            assert np.array_equal(ref[0], Cp)
            assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=1e-12)["ok"]
            assert bh.free_mem() == 0
            assert bh.freePlatform() == 0


@pytest.mark.parametrize("kind", ["banded", "blocks", "grid"])
def test_rows_accumulated_over_their_column_span(oracle, kind):
    """bhs_row_span.hip.h (round 5; off by default -- it lost to the hash kernels, profiles/r05_experiments.md): option
    "span_path" = 1 before the hand-over lets its scans choose the span kernels where every row of C is narrow.  Same C as
    the oracle's, the kernels did run (bhs_get_info "span_words"), and the same handle without the option agrees."""
    rng = np.random.default_rng(5)
    if kind == "banded":
        m = 6000
        bw = rng.integers(2, 20, m)
        rows = [np.arange(max(0, i - bw[i]), min(m, i + bw[i] + 1)) for i in range(m)]
    elif kind == "blocks":
        m, rows, i = 0, [], 0
        sizes = rng.integers(3, 30, 400)
        m = int(sizes.sum())
        for sz in sizes:
            for _ in range(sz):
                rows.append(np.arange(i, i + sz))
            i += sz
    else:
        n = 70
        m = n * n
        rows = []
        for y in range(n):
            for x in range(n):
                c = [yy * n + xx for yy in (y - 1, y, y + 1) for xx in (x - 2, x - 1, x, x + 1, x + 2) if 0 <= yy < n and 0 <= xx < n]
                rows.append(np.array(c))
    rp = np.zeros(m + 1, np.int32)
    rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    val = rng.integers(1, 10, len(col)).astype(np.float64)
    ref = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    if bh.set_option("span_path", 1) != 0:              # (round 6: the kernel lost and is not in the product library)
        assert bh.freePlatform() == 0
        pytest.skip("bhs_row_span.hip.h is compiled into lab builds only: tools/lab_tests.sh")
    for span in (1, 0):
        assert bh.set_option("span_path", span) == 0
        assert bh.set_option("class_path", 0) == 0
        Cp = np.zeros(m + 1, np.int32)
        assert bh.initData(m, m, m, len(col), val, rp, col, len(col), val, rp, col, Cp) == 0
        assert bh.spgemm() == 0
        assert (bh.get_info("span_words") > 0) == (span == 1)
        nnzC = bh.get_nnzC()
        Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
        assert bh.get_C(Cj, Cx) == 0
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
        assert bh.free_mem() == 0
    assert bh.freePlatform() == 0


@pytest.mark.parametrize("stencil,dims", [("poisson5pt", (61, 47, 1)), ("poisson5pt", (9, 200, 1))])
def test_rows_of_at_most_32_products_in_registers(oracle, stencil, dims):
    """bhs_row_tiny.hip.h (round 5; off by default -- slower than the lane-per-row merge, profiles/r05_experiments.md):
    option "tiny_rows" = 1 takes rows of <= 5 x 5 products through a register sorting network.  Same C as the oracle's."""
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    probe = bhmod.bhsparse()
    assert probe.initPlatform(plats) == 0
    lab = probe.set_option("tiny_rows", 1) == 0
    assert probe.freePlatform() == 0
    if not lab:                                          # (round 6: the kernel lost and is not in the product library)
        pytest.skip("bhs_row_tiny.hip.h is compiled into lab builds only: tools/lab_tests.sh")
    m, rp, col, val = poisson_case(stencil, *dims)
    ref = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
    for tiny in (1, 0):
        Cp, Cj, Cx, info = spgemm_csr(m, m, m, rp, col, val, rp, col, val, options={"tiny_rows": tiny, "class_path": 0})
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
        assert "numeric_lane" in _kernel_names(info)


def test_errors_are_codes_not_exceptions():
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.spgemm() != 0                       # before initPlatform
    assert bh.initPlatform(plats) == 0
    assert bh.spgemm() == bhmod._lib.BHS_ERR_NOT_READY     # before initData
    assert bh.get_C(np.empty(1, np.int32), np.empty(1)) == bhmod._lib.BHS_ERR_NOT_READY
    bad = np.zeros(3, np.int64)
    assert bh.initData(2, 2, 2, 0, np.empty(0), bad, np.empty(0, np.int32), 0, np.empty(0),
                       np.zeros(3, np.int32), np.empty(0, np.int32), None) == bhmod._lib.BHS_ERR_INVALID_ARG
    assert bh.freePlatform() == 0


def test_empty_multiply_after_nonempty_on_one_handle():
    """the empty-product early exit must not hand out the PREVIOUS multiply's staged rowPtrC.
    One handle: a non-empty multiply through the host-pointer API (rowPtrC staged in pinned memory), then an
    empty one with more rows; the C-ABI's rowPtrC_out must come back all zero."""
    m1, rp, col, val = poisson_case("poisson27pt", 8, 8, 8)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    Cp1 = np.zeros(m1 + 1, np.int32)
    assert bh.initData(m1, m1, m1, len(col), val, rp, col, len(col), val, rp, col, Cp1) == 0
    assert bh.spgemm() == 0 and Cp1[-1] == bh.get_nnzC() > 0
    # empty A with MORE rows than before (the stale staging buffer would also be too short), B unchanged
    m2 = 3 * m1
    rp0 = np.zeros(m2 + 1, np.int32)
    Cp2 = np.full(m2 + 1, -7, np.int32)
    assert bh.initData(m2, m1, m1, 0, np.empty(0), rp0, np.empty(0, np.int32), len(col), val, rp, col, Cp2) == 0
    assert bh.spgemm() == 0
    assert bh.get_nnzC() == 0 and bh.nnzCt == 0
    assert not Cp2.any()
    # and back to a non-empty product on the same handle
    assert bh.initData(m1, m1, m1, len(col), val, rp, col, len(col), val, rp, col, Cp1) == 0
    Cp1[:] = -1
    assert bh.spgemm() == 0 and Cp1[0] == 0 and Cp1[-1] == bh.get_nnzC() > 0
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


def test_empty_multiply_after_nonempty_device_inputs():
    """Same hazard through bhs_set_data_device + bhs_spgemm(rowPtrC_out) called directly on the C-ABI."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    m1, rp, col, val = poisson_case("poisson5pt", 30, 30)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Bp, Bj, Bx = t(rp), t(col), t(val)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.initData_device(m1, m1, m1, len(col), Bx, Bp, Bj, len(col), Bx, Bp, Bj) == 0
    out1 = np.full(m1 + 1, -1, np.int32)
    nnzCt, nnzC = C.c_int64(0), C.c_int(0)
    assert bh._lib.bhs_spgemm(bh._h, out1.ctypes.data_as(C.c_void_p), C.byref(nnzCt), C.byref(nnzC), None) == 0
    assert out1[0] == 0 and out1[-1] == nnzC.value > 0
    m2 = 5 * m1
    Ap0 = torch.zeros(m2 + 1, dtype=torch.int32, device=dev)
    e_i, e_v = torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.float64, device=dev)
    assert bh.initData_device(m2, m1, m1, 0, e_v, Ap0, e_i, len(col), Bx, Bp, Bj) == 0
    out2 = np.full(m2 + 1, -9, np.int32)
    assert bh._lib.bhs_spgemm(bh._h, out2.ctypes.data_as(C.c_void_p), C.byref(nnzCt), C.byref(nnzC), None) == 0
    assert nnzC.value == 0 and nnzCt.value == 0 and not out2.any()
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.parametrize("stencil,dims", [("poisson5pt", (40, 40, 1)), ("poisson27pt", (12, 12, 12))])
def test_direct_launch_bounds_are_verified_on_the_device(oracle, stencil, dims):
    """Lane-first / wave-first launches are chosen from the row bounds seen at bhs_set_data time; the multiply
    verifies them itself.  Borrowed device arrays are changed AFTER bhs_set_data_device so that one row of A is
    far longer than the longest row seen then: the kernels must refute the speculation and the multiply must
    still be right (general pipeline), on this call and the next.  The row-class path takes its hints from
    bhs_set_data time too and classifies every row on the device: with it the multiply must be right as well, on
    the class kernels or back on the general pipeline (a row longer than the classifier was sized for finds no class)."""
    import torch
    dev = torch.device("cuda", 0)
    m, rp, col, val = poisson_case(stencil, *dims)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for class_path in (0, 1):
        Bp, Bj, Bx = t(rp), t(col), t(val)
        Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
        plats = [False] * bhmod.NUM_PLATFORMS
        plats[bhmod.BHSPARSE_HIP] = True
        bh = bhmod.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.set_option("class_path", 2 * class_path) == 0
        assert bh.initData_device(m, m, m, len(col), Ax, Ap, Aj, len(col), Bx, Bp, Bj) == 0
        assert bh.spgemm() == 0
        first = {s["name"] for s in bh.kernel_stats()}
        assert "upper_bound" not in first            # the direct path (or the class path) ran
        assert ("numeric_class" in first) == bool(class_path)
        # row 5 of A swallows rows 5..11 (same arrays, same nnz): 7 rows' worth of entries in one row
        rp2 = rp.copy()
        rp2[6:12] = rp[12]
        Ap.copy_(t(rp2))
        torch.cuda.synchronize()
        ref = oracle.spgemm(m, m, m, rp2, col, val, rp, col, val)
        for _ in range(2):
            assert bh.spgemm() == 0
            names = {s["name"] for s in bh.kernel_stats()}
            if class_path:
                # a row beyond what the classifier was sized for from the longest row seen at bhs_set_data time (round 3:
                # entries per lane follow that hint) finds no class: round 6's mixed mode keeps the class kernels for the other
                # rows and sends this one (and the empty rows behind it) through the general pipeline's; until round 5 the whole
                # multiply went there
                assert "numeric_class" in names or "upper_bound" in names
            else:
                assert "upper_bound" in names and "numeric_class" not in names   # general pipeline
            Cp = bh.get_rowptrC()
            nnzC = bh.get_nnzC()
            Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
            assert bh.get_C(Cj, Cx) == 0
            assert bh.nnzCt == oracle.nnzCt(rp2, col, rp)
            assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
        assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.gpu
def test_speculative_lane_launch_is_verified_on_the_device(oracle):
    """Round 6: a lane-first multiply (poisson5pt: every row through k_row_lane) launches its numeric kernel on the last
    multiply's nnz(C) too (k_lane_spec_check).  New values: the launch stands.  A column index of A moved so that a row of C
    gains an entry: nnz(C) differs, the launch is refuted, the multiply runs again and is right.  A row of A grown beyond the
    lane kernel's heads: refuted by the kernel's own check, right again."""
    import torch
    dev = torch.device("cuda", 0)
    m, rp, col, val = poisson_case("poisson5pt", 64, 50, 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Bp, Bj, Bx = t(rp), t(col), t(val)
    Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.set_option("class_path", 0) == 0
    assert bh.initData_device(m, m, m, len(col), Ax, Ap, Aj, len(col), Bx, Bp, Bj) == 0

    def check(arp, acol, aval):
        ref = oracle.spgemm(m, m, m, arp, acol, aval, rp, col, val)
        assert bh.spgemm() == 0
        Cp = bh.get_rowptrC()
        nnzC = bh.get_nnzC()
        Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
        assert bh.get_C(Cj, Cx) == 0
        assert bh.nnzCt == oracle.nnzCt(arp, acol, rp)
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
        return {s_["name"] for s_ in bh.kernel_stats() if s_["launches"]}

    names = check(rp, col, val)
    assert "numeric_lane" in names and "upper_bound" not in names and bh.get_info("spec_launches") == 0
    check(rp, col, val); check(rp, col, val)
    assert bh.get_info("spec_launches") == 2 and bh.get_info("spec_refuted") == 0
    val2 = val.copy(); val2[::5] += 2.0
    Ax.copy_(t(val2)); torch.cuda.synchronize()
    check(rp, col, val2)
    assert bh.get_info("spec_launches") == 3 and bh.get_info("spec_refuted") == 0
    r = m // 2 + 3                                       # an interior row's first column (r - 64) moves one to the left: its row of C changes shape
    col2 = col.copy(); col2[rp[r]] -= 1
    Aj.copy_(t(col2)); torch.cuda.synchronize()
    nnz_before = bh.get_nnzC()
    check(rp, col2, val2)
    assert bh.get_info("spec_launches") == 4 and bh.get_info("spec_refuted") == (1 if bh.get_nnzC() != nnz_before else 0)
    refuted = bh.get_info("spec_refuted")
    check(rp, col2, val2)
    assert bh.get_info("spec_refuted") == refuted
    # row 5 of A swallows rows 5..8 (20 entries > the lane kernel's heads): the kernels' own check refutes whatever was assumed
    rp2 = rp.copy(); rp2[6:9] = rp[9]
    Ap.copy_(t(rp2)); torch.cuda.synchronize()
    names = check(rp2, col2, val2)
    assert "upper_bound" in names
    check(rp2, col2, val2)
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.gpu
def test_speculative_numeric_launch_is_verified_on_the_device(oracle):
    """From a data set's second multiply on, the class path launches its numeric kernel on the figures of the multiply before
    -- how many classes, their longest lists, the LDS they need, nnzC -- without reading this multiply's back first
    (k_class_spec_check decides on the device; `spec_launches` / `spec_refuted`).  Values changed in the borrowed arrays: the
    launch stands and the product has the new values.  One column index of A changed (a new relative pattern: one class more):
    the launch is refuted, the multiply runs again the slow way and is right; the multiply after that speculates again."""
    import torch
    dev = torch.device("cuda", 0)
    m, rp, col, val = poisson_case("poisson27pt", 24, 24, 24)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Bp, Bj, Bx = t(rp), t(col), t(val)
    Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.set_option("class_path", 2) == 0
    assert bh.initData_device(m, m, m, len(col), Ax, Ap, Aj, len(col), Bx, Bp, Bj) == 0

    def check(acol, aval):
        ref = oracle.spgemm(m, m, m, rp, acol, aval, rp, col, val)
        assert bh.spgemm() == 0
        assert "numeric_class" in {s["name"] for s in bh.kernel_stats()}
        Cp = bh.get_rowptrC()
        nnzC = bh.get_nnzC()
        Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
        assert bh.get_C(Cj, Cx) == 0
        assert bh.nnzCt == oracle.nnzCt(rp, acol, rp)
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]

    check(col, val)
    assert bh.get_info("spec_launches") == 0
    check(col, val); check(col, val)
    assert bh.get_info("spec_launches") == 2 and bh.get_info("spec_refuted") == 0
    val2 = val.copy(); val2[::7] += 3.0
    Ax.copy_(t(val2)); torch.cuda.synchronize()
    check(col, val2)
    assert bh.get_info("spec_launches") == 3 and bh.get_info("spec_refuted") == 0
    # an interior row's first column moves one to the left (still ascending, still inside the matrix)
    r = m // 2
    col2 = col.copy()
    assert col2[rp[r]] > 0 and (rp[r] == rp[r - 1] or True)
    col2[rp[r]] -= 1
    Aj.copy_(t(col2)); torch.cuda.synchronize()
    check(col2, val2)
    assert bh.get_info("spec_launches") == 4 and bh.get_info("spec_refuted") == 1
    check(col2, val2)
    assert bh.get_info("spec_launches") == 5 and bh.get_info("spec_refuted") == 1
    assert bh.set_option("spec_numeric", 0) == 0
    check(col2, val2); check(col2, val2)
    assert bh.get_info("spec_launches") == 5
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 1, 3])
@pytest.mark.parametrize("kind", ["grid", "band32_holes"])
def test_lane_per_row_classifier(oracle, kind, shift):
    """k_class_tile (bhs_class_tile.hip.h) brings 63 rows' column indices in as 16-byte loads from the 16-byte boundary at
    or below the first of them: the borrowed colInd arrays are handed over `shift` ints off such a boundary, A's ending on
    the allocation's last byte.  Rows of exactly 32 entries (the most a class row of this kernel has), empty rows in between,
    more rows than one piece; against the oracle and against the lane-group classifier (class_tile = 0)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    if kind == "grid":
        m, rp, col, val = poisson_case("poisson27pt", 19, 18, 17)
    else:
        m = 9000
        rp, col, val = _toeplitz(m, m, tuple(range(-40, 88, 4)), rng, holes=(0, 1, 500, 501, 502, 4000, m - 1))
        assert np.diff(rp).max() == 32
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def shifted(a):                                   # the array at `shift` elements into a buffer that ends with it
        buf = torch.empty(len(a) + shift, dtype=torch.int32, device=dev)
        buf[shift:] = t(a)
        return buf[shift:]
    Bp, Bj, Bx = t(rp), shifted(col), t(val)
    Ap, Aj, Ax = t(rp), shifted(col), t(val)
    ref = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    for tile in (1, 0):
        bh = bhmod.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.set_option("class_path", 2) == 0 and bh.set_option("class_tile", tile) == 0
        assert bh.initData_device(m, m, m, len(col), Ax, Ap, Aj, len(col), Bx, Bp, Bj) == 0
        for _ in range(2):
            assert bh.spgemm() == 0
            assert "numeric_class" in {s["name"] for s in bh.kernel_stats()}
            Cp = bh.get_rowptrC()
            nnzC = bh.get_nnzC()
            Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
            assert bh.get_C(Cj, Cx) == 0
            assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
        assert bh.free_mem() == 0 and bh.freePlatform() == 0


def _toeplitz(m, n, offsets, rng, holes=()):
    """m x n matrix whose row i has the columns i + o for o in offsets (those inside the matrix): every interior row
    has the same relative pattern.  `holes`: rows left empty."""
    rows = []
    for i in range(m):
        c = np.array([i + o for o in offsets if 0 <= i + o < n], np.int64)
        rows.append(np.empty(0, np.int64) if i in holes else c)
    rp = np.zeros(m + 1, np.int32)
    rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32) if rp[-1] else np.empty(0, np.int32)
    return rp, col, rng.integers(1, 10, len(col)).astype(np.float64)


def _kron_ones(rp, col, dof):
    """(stencil matrix) (x) ones(dof, dof): the pattern of a grid with dof unknowns per node, every coupling a full block."""
    import scipy.sparse as sp
    P = sp.csr_matrix((np.ones(len(col)), col, rp), shape=(len(rp) - 1,) * 2)
    A = sp.kron(P, np.ones((dof, dof)), format="csr")
    A.sort_indices()
    return A.shape[0], A.indptr.astype(np.int32), A.indices.astype(np.int32)


def test_row_classes_with_a_column_twice_in_rows_of_b(oracle):
    """k_class_patterns ranks a class's products inside their entry of C by the A entry they come from (round 5: a mask per
    entry instead of a second sort) -- on the premise that a row of B holds a column once.  A row of B that holds one
    twice (the reference adds such duplicates up like any others) must send the class back to the sort: same C as the
    oracle's, on the class kernels."""
    rng = np.random.default_rng(12)
    m = k = n = 3000
    A = _toeplitz(m, k, (-30, -1, 0, 1, 30), rng)
    B = _toeplitz(k, n, (-30, -2, 0, 0, 2, 2, 30), rng)           # columns j and j + 2 twice in every interior row
    ref = oracle.spgemm(m, k, n, *A, *B)
    for opts in ({"class_path": 2, "sort_b": 0}, {"class_path": 2}, {"class_path": 0}):
        Cp, Cj, Cx, info = spgemm_csr(m, k, n, *A, *B, options=opts)
        assert info["nnzC"] == ref[0][-1]
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"], opts
        if opts.get("class_path") == 2 and "sort_b" in opts:
            assert "numeric_class" in _kernel_names(info)


@pytest.mark.parametrize("case", ["p27", "p5", "p7", "p9", "rect_toeplitz", "holes", "float_values", "f32_build",
                                  "unsorted_b", "row_block", "fem_3dof", "fem_4dof", "fem_3dof_f32", "long_b_rows",
                                  "fem_3dof_row_block", "slab_of_512_values", "five_steps_five_pairs"])
def test_row_class_path(oracle, case):
    """Row classes (bhs_class.hip.h): inputs whose rows repeat one another's relative pattern take the class kernels --
    classify_rows / class_patterns / numeric_class instead of upper bound, symbolic and numeric bins -- and give the
    oracle's C: rowPtr and colInd bit-exact, values exact for integer-valued inputs."""
    rng = np.random.default_rng(31)
    value_dtype = np.float32 if case == "f32_build" else np.float64
    opts = {"class_path": 2}                  # (whatever the average number of products per row)
    if case in ("p27", "float_values", "f32_build"):
        m, rp, col, val = poisson_case("poisson27pt", 13, 12, 11); k = n = m
        A = B = (rp, col, val)
    elif case in ("p5", "p7", "p9"):
        dims = {"p5": (70, 61, 1), "p7": (17, 16, 15), "p9": (45, 52, 1)}[case]
        m, rp, col, val = poisson_case({"p5": "poisson5pt", "p7": "poisson7pt", "p9": "poisson9pt"}[case], *dims); k = n = m
        A = B = (rp, col, val)
    elif case == "row_block":
        # A = rows [r0, r1) of a stencil matrix (what a rank of the multi-GPU layer multiplies): m << k, so only the
        # rows of B between A's smallest and largest column are classified
        k, rp, col, val = poisson_case("poisson27pt", 14, 13, 12); n = k
        r0, r1 = 700, 1100
        m = r1 - r0
        A = ((rp[r0:r1 + 1] - rp[r0]).astype(np.int32), col[rp[r0]:rp[r1]], val[rp[r0]:rp[r1]])
        B = (rp, col, val)
    elif case in ("fem_3dof", "fem_3dof_f32", "fem_4dof", "fem_3dof_row_block"):
        # several unknowns per node: the classes are beyond the register kernels' tables (81 entries per row, 6561
        # products and 375 entries per row of C; 4 unknowns on poisson9pt: 36, 1296, 100) -- bhs_class_big.hip.h
        if case == "fem_4dof":
            _, rp0, col0, _ = poisson_case("poisson9pt", 23, 19, 1)
            m, rp, col = _kron_ones(rp0, col0, 4)
        else:
            _, rp0, col0, _ = poisson_case("poisson27pt", 7, 6, 5)
            m, rp, col = _kron_ones(rp0, col0, 3)
        k = n = m
        val = rng.integers(1, 10, len(col)).astype(np.float64)
        A = B = (rp, col, val)
        if case == "fem_3dof_row_block":
            r0, r1 = 100, 431
            m = r1 - r0
            A = ((rp[r0:r1 + 1] - rp[r0]).astype(np.int32), col[rp[r0]:rp[r1]], val[rp[r0]:rp[r1]])
        if case == "fem_3dof_f32":
            value_dtype = np.float32
    elif case == "long_b_rows":
        m, k, n = 2000, 2500, 4000          # 3 entries per row of A, 100 per row of B: few products, but a B entry's number needs 7 bits
        A = _toeplitz(m, k, (-7, 0, 300), rng)
        B = _toeplitz(k, n, tuple(range(-50, 50)), rng)
    elif case == "slab_of_512_values":
        # 32 B rows of 15 entries: a slab of exactly 4 x 64 16-byte units -- the last one, number 255, was taken for "no unit"
        # and never loaded (the mixed-mode soak's seed 361, round 6); 480 products, 358 entries: k_class_ring<16, 8>
        m, k, n = 9000, 9200, 9100
        # (offsets in fours: every cut-off pattern along the borders has >= 4 rows, no class is too small to keep)
        offa = np.unique(rng.integers(-375, 376, 60) * 4)[:32]; offb = np.unique(rng.integers(-375, 376, 30) * 4)[:15]
        assert len(offa) == 32 and len(offb) == 15 and len(np.unique(offa[:, None] + offb[None, :])) > 256
        A = _toeplitz(m, k, tuple(int(o) for o in offa), rng)
        B = _toeplitz(k, n, tuple(int(o) for o in offb), rng)
    elif case == "five_steps_five_pairs":
        # 9 x 31 entries: 279 products, 263 entries a row -- k_class_ring<16, 8>, whose 16-byte stores of two values had the
        # next pair's entry number written over their first word two wait states too early (the soak's seed 327)
        m, k, n = 20000, 20500, 21000
        offa = np.unique(rng.integers(-625, 626, 12) * 4)[:9]; offb = np.unique(rng.integers(-625, 626, 50) * 4)[:31]
        assert len(offa) == 9 and len(offb) == 31 and len(np.unique(offa[:, None] + offb[None, :])) > 256
        A = _toeplitz(m, k, tuple(int(o) for o in offa), rng)
        B = _toeplitz(k, n, tuple(int(o) for o in offb), rng)
    elif case == "rect_toeplitz":
        m, k, n = 3000, 3500, 5000          # relative columns far from 0, rows cut off at every border
        A = _toeplitz(m, k, (-40, -3, 0, 1, 2, 500, 501, 3400), rng)
        B = _toeplitz(k, n, (-700, -1, 0, 1, 5, 6, 7, 1499), rng)
    elif case == "holes":
        m = k = n = 4000
        A = _toeplitz(m, k, (-64, -1, 0, 1, 64), rng, holes={0, 17, 2000, 3999})
        B = _toeplitz(k, n, (-64, -2, 0, 2, 64), rng, holes={1, 18, 1999})
    else:                                    # rows of B stored in descending order: the class pattern is sorted anyway
        m = k = n = 2500
        A = _toeplitz(m, k, (-50, -1, 0, 1, 50), rng)
        Bp, Bj, Bx = _toeplitz(k, n, (-50, -1, 0, 1, 50), rng)
        for j in range(k):
            Bj[Bp[j]:Bp[j + 1]] = Bj[Bp[j]:Bp[j + 1]][::-1]
            Bx[Bp[j]:Bp[j + 1]] = Bx[Bp[j]:Bp[j + 1]][::-1]
        B = (Bp, Bj, Bx)
        opts = {"sort_b": 0, "class_path": 2}
    Ax, Bx = A[2], B[2]
    if case == "float_values":
        Ax = rng.standard_normal(len(Ax)); Bx = Ax if B is A else rng.standard_normal(len(Bx))
    ref = oracle.spgemm(m, k, n, A[0], A[1], Ax, B[0], B[1], Bx)
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, A[0], A[1], Ax.astype(value_dtype), B[0], B[1], Bx.astype(value_dtype),
                                  options=opts, value_dtype=value_dtype)
    names = _kernel_names(info)
    assert {"classify_rows", "class_patterns", "numeric_class"} <= names and "upper_bound" not in names, names
    assert not any(nm.startswith("symbolic_") or nm.startswith("numeric_w") for nm in names), names
    assert info["nnzCt"] == oracle.nnzCt(A[0], A[1], B[0]) and info["nnzC"] == ref[0][-1]
    assert np.array_equal(Cp, ref[0]) and np.array_equal(Cj, ref[1])
    if case == "float_values":
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=REL_TOL)["ok"]
    else:
        assert np.array_equal(Cx.astype(np.float64), ref[2])
    check_csr_invariants(m, n, Cp, Cj)
    numc = [kk for kk in info["kernels"] if kk["name"] == "numeric_class"][0]
    assert numc["rows"] == m and numc["products"] == info["nnzCt"] and numc["nnz_out"] == info["nnzC"]
    if case in ("p27", "p5", "p9"):
        # left to itself (class_path = 1) the library takes the class kernels where they pay: rows with a few hundred
        # products (poisson27pt: 729; from 64 on) and at least 1e7 products in all -- not the small stencils, whose whole general
        # pipeline is cheaper than classifying, and not small matrices
        _, _, _, info1 = spgemm_csr(m, k, n, A[0], A[1], Ax, B[0], B[1], Bx)
        assert "numeric_class" not in _kernel_names(info1)
        if case == "p27":
            m2, rp2, col2, val2 = poisson_case("poisson27pt", 72, 72, 72)
            Cp2, Cj2, Cx2, info2 = spgemm_csr(m2, m2, m2, rp2, col2, val2, rp2, col2, val2)
            assert "numeric_class" in _kernel_names(info2)
            assert info2["nnzCt"] == oracle.nnzCt(rp2, col2, rp2)
            Cp3, Cj3, Cx3, info3 = spgemm_csr(m2, m2, m2, rp2, col2, val2, rp2, col2, val2, options={"class_path": 0})
            assert np.array_equal(Cp2, Cp3) and np.array_equal(Cj2, Cj3) and np.array_equal(Cx2, Cx3)


@pytest.mark.parametrize("seed", list(range(12)))
def test_row_class_path_randomized(oracle, seed):
    """Randomised structured inputs against the oracle with the class kernels forced on: random offset lists (a few to
    64 entries per row), rectangular shapes, rows cut off at the borders, a sprinkling of rows with extra or missing
    entries (more classes, some met once), empty rows, B rows stored in random order, values of either sign."""
    _randomized_class_case(oracle, seed, big=False)


@pytest.mark.parametrize("seed", [327, 361, 304, 352, 392, 1305, 3366, 4332, 6302, 6342, 6355])
def test_mixed_mode_draws_of_the_soak(oracle, seed):
    """A handful of the soak's draws in the default suite (the full soaks: BHS_SOAK=1): the two that found round 6's faults
    (327: k_class_ring<16, 8> clean; 361: a slab of 512 values, 4932 irregular rows), clean and mixed ones, rows of up to 64
    entries, the float build, block-structured ones -- three multiplies of one handle each, the last in row ranges."""
    _randomized_mixed_case(oracle, seed)


@pytest.mark.parametrize("seed", [504, 508, 515, 524, 543, 1506, 5564])
def test_general_pipeline_draws_of_the_soak(oracle, seed):
    """... and of the general pipeline's: long rows through the bitmaps in LDS and in HBM, hub rows, column windows, every
    size of hash kernel; with the defaults, with one switch thrown, and in row ranges."""
    _general_soak_case(oracle, seed)


if os.environ.get("BHS_SOAK") == "1":             # (BHS_SOAK=1 python -m pytest tests -m gpu -k soak -n 4: 1.5 minutes)
    @pytest.mark.parametrize("seed", list(range(100, 220)))
    def test_row_class_path_soak(oracle, seed):
        """The same draw with up to 40 000 rows of at most 32 entries: many steps and pieces of the lane-per-row classifier
        (bhs_class_tile.hip.h), heads in every lane, last steps of every length.  Round 5: 120 of 120 bit-exact."""
        _randomized_class_case(oracle, seed, big=True)


    @pytest.mark.parametrize("seed", list(range(300 + int(os.environ.get("BHS_SOAK_BASE", "0")), 396 + int(os.environ.get("BHS_SOAK_BASE", "0")))))
    def test_row_class_path_mixed_soak(oracle, seed):
        """Mixed mode (bhs_class_mix.hip.h) on random structured products of 8 000 - 60 000 rows: a share of the rows of A
        and of B with an entry more or less, a few rows of 130 - 700 entries (longer than the ring kernel's chunk of A's
        values), a run of neighbouring irregular rows, empty rows; two multiplies of one handle (the second one launched on
        what the first one found), both against the oracle bit for bit."""
        _randomized_mixed_case(oracle, seed)


    @pytest.mark.parametrize("seed", list(range(500 + int(os.environ.get("BHS_SOAK_BASE", "0")), 596 + int(os.environ.get("BHS_SOAK_BASE", "0")))))
    def test_general_pipeline_soak(oracle, seed):
        """The general pipeline (upper bound, bins, symbolic and numeric kernels of every size, bitmaps in LDS and in HBM,
        column windows, hub rows) on random inputs of every shape the draw below knows -- row lengths constant, uniform,
        power-law or with a few very long rows; columns uniform, banded or from a small pool; 500 to 60 000 rows, up to
        3 M columns -- against the oracle bit for bit, with the defaults and with one switch thrown."""
        _general_soak_case(oracle, seed)


def _general_soak_case(oracle, seed):
    if True:
        (m, k, n, how), A, B = _general_soak_inputs(seed)
        ref = oracle.spgemm(m, k, n, *A, *B)
        rng = np.random.default_rng(seed)
        alt = [{"concurrent_bins": 0}, {"lds_bitmap": 0}, {"spa": 0}, {"window_bitmap": 2}, {"spec_numeric": 0}, {"lane_from_counts": 0},
               {"max_table_log2": 8}, {"no_pack32": 1}, {"hub_min_products": 200000}][int(rng.integers(0, 9))]
        # (every fifth draw on the float build: sums of small integers below 2^24 are exact there too)
        vd = np.float32 if seed % 5 == 2 and float(np.abs(ref[2]).max(initial=0.0)) < 2.0 ** 22 else np.float64
        how += ", float" if vd == np.float32 else ""
        for opts in ({}, alt):
            Cp, Cj, Cx, info = spgemm_csr(m, k, n, A[0], A[1], A[2].astype(vd), B[0], B[1], B[2].astype(vd), options=opts, value_dtype=vd)
            assert info["nnzCt"] == oracle.nnzCt(A[0], A[1], B[0]) and info["nnzC"] == ref[0][-1], (seed, how, opts)
            res = oracle.compare(ref, (Cp, Cj, Cx.astype(np.float64)), rel_tol=0.0)
            assert res["ok"], (seed, how, opts, res, sorted(_kernel_names(info)))
        # ... and on one handle: a multiply, then one in row ranges
        plats = [False] * bhmod.NUM_PLATFORMS
        plats[bhmod.BHSPARSE_HIP] = True
        bh = bhmod.bhsparse()
        assert bh.initPlatform(plats) == 0
        Cp = np.zeros(m + 1, np.int32)
        assert bh.initData(m, k, n, len(A[1]), A[2], A[0], A[1], len(B[1]), B[2], B[0], B[1], Cp) == 0
        assert bh.spgemm() == 0
        cuts = _multiply_in_row_ranges(bh, m, rng)
        Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
        assert bh.get_C(Cj, Cx) == 0
        res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
        assert res["ok"], (seed, how, "row ranges", cuts, res)
        assert bh.free_mem() == 0 and bh.freePlatform() == 0
        print("general soak, seed %d: %d x %d x %d, %s, %d products -> %d entries; %s" % (
            seed, m, k, n, how, info["nnzCt"], info["nnzC"], " ".join(sorted(_kernel_names(info)))))


@pytest.mark.parametrize("m,L", [(4000, 4), (16000, 4), (32000, 8), (34000, 8), (70000, 8), (33000, 3), (66000, 16)])
def test_row_class_path_with_about_as_many_classes_as_table_slots(oracle, m, L):
    """Runs of L rows with a pattern of their own (two of A's offsets change from run to run): 1000 to 8750 classes of L rows
    each against a table of 4096 slots -- below it the class kernels with as many pattern workgroups, at and beyond it
    whatever the library decides (the rows that find no slot are irregular rows, or the class path gives way), and runs of 3
    rows are too short to keep a class.  Whatever runs, C is the oracle's."""
    rng = np.random.default_rng(m + L)
    k = n = m + 500
    blk = np.arange(m) // L
    offs = np.stack([np.full(m, -40), np.full(m, -1), np.zeros(m, np.int64), blk % 97 + 2, 100 + blk // 97], axis=1)
    cols = np.arange(m)[:, None] + offs
    ok = (cols >= 0) & (cols < k)
    Ap = np.zeros(m + 1, np.int32); Ap[1:] = np.cumsum(ok.sum(axis=1))
    Aj = cols[ok].astype(np.int32)
    Ax = rng.integers(-9, 10, len(Aj)).astype(np.float64)
    B = _toeplitz(k, n, (-3, -1, 0, 2, 40), rng)
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, *B)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0 and bh.set_option("class_path", 2) == 0
    Cp = np.zeros(m + 1, np.int32)
    assert bh.initData(m, k, n, len(Aj), Ax, Ap, Aj, len(B[1]), B[2], B[0], B[1], Cp) == 0
    for it in range(2):
        assert bh.spgemm() == 0
        Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
        assert bh.get_C(Cj, Cx) == 0
        res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
        names = sorted(s_["name"] for s_ in bh.kernel_stats() if s_["launches"])
        assert res["ok"], (m, L, it, res, names)
    print("many classes: %d rows in runs of %d -> class_state %d, %d irregular rows, %s" % (
        m, L, bh.get_info("class_state"), bh.get_info("mixed_rows"), "numeric_class" if "numeric_class" in names else "general pipeline"))
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


def _class_boundary_cases():
    cases = []
    for na in (1, 2, 3, 16, 31, 32, 33, 63, 64):
        for nb in (1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 100, 128, 256):
            if na * nb > 1024: continue
            for kind in ("band", "random", "two_bands", "band_x_random"):
                cases.append((na, nb, kind))
    return cases


@pytest.mark.parametrize("na,nb,kind", [(1, 1, "band"), (64, 16, "band"), (2, 256, "band"), (32, 32, "band"), (33, 31, "two_bands"),
                                        (16, 64, "two_bands"), (63, 16, "two_bands"), (3, 100, "random"), (32, 16, "random")])
def test_row_class_path_at_its_limits(oracle, na, nb, kind):
    """Nine of the sweep below in the default suite."""
    _class_boundary_case(oracle, na, nb, kind)


if os.environ.get("BHS_SOAK") == "1":
    @pytest.mark.parametrize("na,nb,kind", _class_boundary_cases())
    def test_row_class_path_at_its_limits_soak(oracle, na, nb, kind):
        """Row lengths at the edges of the class tables (1, 2, 16 +- 1, 32 +- 1, 64 entries of A; up to 256 of B; up to 1024
        products), as dense bands (one chain of consecutive columns: the longest rings), random offsets (chains of one: the
        widest slabs), two bands, and a band times random offsets."""
        _class_boundary_case(oracle, na, nb, kind)


def _class_boundary_case(oracle, na, nb, kind):
    rng = np.random.default_rng(na * 1000 + nb)

    def offsets(cnt, how):
        if how == "band": return np.arange(cnt) - cnt // 2
        if how == "two_bands": return np.concatenate((np.arange(cnt // 2) - 900, np.arange(cnt - cnt // 2) + 400))
        o = np.unique(rng.integers(-300, 301, 4 * cnt) * 4)
        return np.sort(rng.choice(o, cnt, replace=False))
    offa = offsets(na, "band" if kind == "band_x_random" else kind)
    offb = offsets(nb, "random" if kind == "band_x_random" else kind)
    m, k, n = 7000, 7100, 7200
    A = _toeplitz(m, k, tuple(int(o) for o in offa), rng)
    B = _toeplitz(k, n, tuple(int(o) for o in offb), rng)
    ref = oracle.spgemm(m, k, n, *A, *B)
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, *A, *B, options={"class_path": 2})
    assert info["nnzCt"] == oracle.nnzCt(A[0], A[1], B[0]) and info["nnzC"] == ref[0][-1]
    res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
    assert res["ok"], (na, nb, kind, res, sorted(_kernel_names(info)))
    print("class limits: %d x %d entries, %s -> %d entries of C a row at most; class_state %d, %d irregular rows, %s" % (
        na, nb, kind, int(np.diff(ref[0]).max()), info["class_state"], info["mixed_rows"],
        "numeric_class" if "numeric_class" in _kernel_names(info) else "general pipeline"))


def _multiply_in_row_ranges(bh, m, rng):
    """bhs_spgemm_symbolic, bhs_spgemm_numeric on 1 .. 5 random row ranges (some empty), bhs_spgemm_finish: what the multi-GPU
    layer runs on one handle."""
    assert bh.spgemm_symbolic() == 0
    cuts = np.concatenate(([0], np.sort(rng.integers(0, m + 1, int(rng.integers(0, 5)))), [m]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert bh.spgemm_numeric(int(a), int(b)) == 0
    assert bh.spgemm_finish() == 0
    return [int(c) for c in cuts]


def _general_soak_inputs(seed):
    rng = np.random.default_rng(9000 + seed)

    def lengths(rows, cols, kind, mean):
        if kind == "const": L = np.full(rows, mean)
        elif kind == "uniform": L = rng.integers(0, 2 * mean + 1, rows)
        elif kind == "powerlaw": L = np.minimum((rng.pareto(1.3, rows) * mean / 3).astype(np.int64), 6000)
        else:                                                        # short rows and a few very long ones
            L = rng.integers(0, mean + 1, rows)
            idx = rng.integers(0, rows, int(rng.integers(1, 12)))
            L[idx] = rng.integers(300, 8000, len(idx))
        L = np.minimum(L, cols)
        L[rng.random(rows) < 0.01] = 0
        return L.astype(np.int64)

    def matrix(rows, cols, kindL, mean, kindC):
        L = lengths(rows, cols, kindL, mean)
        r = np.repeat(np.arange(rows, dtype=np.int64), L)
        if kindC == "uniform": c = rng.integers(0, cols, len(r))
        elif kindC == "banded":
            w = int(rng.choice([50, 2000, 40000]))
            c = np.clip(r * cols // rows + (rng.standard_normal(len(r)) * w).astype(np.int64), 0, cols - 1)
        else:
            pool = rng.integers(0, cols, max(8, min(cols, int(rng.choice([64, 3000, 50000])))))
            c = pool[rng.integers(0, len(pool), len(r))]
        key = np.unique(r * cols + c)                                # (sorted rows without duplicates; a row may come out shorter)
        r, c = key // cols, key % cols
        rp = np.zeros(rows + 1, np.int32)
        rp[1:] = np.cumsum(np.bincount(r, minlength=rows))
        return rp, c.astype(np.int32), rng.integers(-9, 10, len(c)).astype(np.float64)

    m = int(np.exp(rng.uniform(np.log(500), np.log(60000)))); k = int(np.exp(rng.uniform(np.log(500), np.log(60000))))
    n = int(np.exp(rng.uniform(np.log(500), np.log(3.0e6))))
    kA, kB = (str(rng.choice(["const", "uniform", "powerlaw", "long"])) for _ in range(2))
    cA, cB = (str(rng.choice(["uniform", "banded", "pool"])) for _ in range(2))
    meanA, meanB = int(rng.choice([1, 3, 8, 20, 60])), int(rng.choice([1, 3, 8, 20, 60, 200]))
    A = matrix(m, k, kA, meanA, cA)
    B = matrix(k, n, kB, meanB, cB)
    prod = np.add.reduceat(np.append(np.diff(B[0])[A[1]].astype(np.int64), 0), np.minimum(A[0][:-1], len(A[1]))) if len(A[1]) else np.zeros(m, np.int64)
    prod[np.diff(A[0]) == 0] = 0
    total = int(prod.sum())
    if total > 30000000:                                            # (the oracle in a second or two: the first rows that hold 3e7 products)
        m = max(1, int(np.searchsorted(np.cumsum(prod), 30000000)))
        A = (A[0][:m + 1].copy(), A[1][:A[0][m]].copy(), A[2][:A[0][m]].copy())
    if seed % 4 == 1:                                                # rows of B in random stored order (sorted at set_data time by default)
        Bp, Bj, Bx = B
        key = np.repeat(np.arange(k), np.diff(Bp)) + rng.random(len(Bj))
        o = np.argsort(key, kind="stable")
        B = (Bp, Bj[o], Bx[o])
    how = "A %s/%s %d, B %s/%s %d" % (kA, cA, meanA, kB, cB, meanB)
    return (m, k, n, how), A, B


def _mixed_soak_inputs(seed):
    rng = np.random.default_rng(7000 + seed)
    # (row i of A has its entries at i * sa + offsets, row j of B at j * sb + offsets: the rows of C repeat a handful of patterns)
    # (a step other than 1: no two rows with the same columns relative to the row -- the class path gives way)
    sa, sb = ([(1, 1)] * 6 + [(1, 2), (2, 1)])[int(rng.integers(0, 8))]
    m = int(rng.integers(8000, 60000)); k = m * sa + int(rng.integers(-3000, 3000)); n = k * sb + int(rng.integers(-3000, 3000))
    na = int(rng.choice([5, 9, 17, 27, 32])); nb = int(rng.choice([7, 9, 15, 27, 32]))
    if 3300 <= seed < 3600:                                      # (BHS_SOAK_BASE=3000 .. 3200: rows of up to 64 entries, up to 1024 products a row)
        na = int(rng.choice([3, 12, 31, 33, 48, 64])); nb = int(rng.choice([2, 8, 16, 31, 48, 64, 100]))
        while na * nb > 1024: nb = max(2, nb // 2)
        m = min(m, 24000000 // (na * nb)); k = m * sa + int(rng.integers(-3000, 3000)); n = k * sb + int(rng.integers(-3000, 3000))
    noise = float(rng.choice([0.0, 0.0005, 0.005, 0.03]))
    d = 1
    if 6300 <= seed < 6600:                                      # (BHS_SOAK_BASE=6000 .. 6200: d unknowns per node, every coupling a full d x d block -- the big-class kernel)
        d = int(rng.choice([2, 3, 4])); sa = sb = 1
        na = int(rng.choice([3, 5, 7, 9])); nb = int(rng.choice([3, 5, 7, 9]))
        m = int(rng.integers(3000, 16000)) * d; k = m + d * int(rng.integers(-500, 500)); n = k + d * int(rng.integers(-500, 500))

    def structured(rows, cols, step, cnt, longs):
        offs = np.unique(rng.integers(-min(cols // d // 3, 3000), min(cols // d // 3, 3000) + 1, cnt))
        base = (np.arange(rows // d, dtype=np.int64) * step)[:, None] + offs[None, :]
        ok = (base >= 0) & (base < cols // d)
        if d == 1:
            out = [base[i][ok[i]] for i in range(rows)]
        else:                                                    # (node i's columns, d scalar columns each, for each of its d rows)
            out = []
            for i in range(rows // d):
                c = (base[i][ok[i]][:, None] * d + np.arange(d)[None, :]).ravel()
                out.extend([c] * d)
        odd = np.flatnonzero(rng.random(rows) < noise)
        if rng.random() < 0.3:                                   # a run of neighbouring irregular rows
            r0 = int(rng.integers(0, rows - 200)); odd = np.union1d(odd, np.arange(r0, r0 + int(rng.integers(2, 200))))
        for i in odd:
            u = rng.random()
            c = out[i]
            if u < 0.4 and len(c) > 1: c = np.delete(c, rng.integers(0, len(c)))
            elif u < 0.9: c = np.unique(np.append(c, rng.integers(0, cols, int(rng.integers(1, 4)))))
            else: c = c[:0]
            out[i] = c
        for _ in range(longs):
            i = int(rng.choice([0, rows - 1, rng.integers(0, rows)]))
            out[i] = np.unique(rng.integers(0, cols, int(rng.integers(130, 700))))
        rp = np.zeros(rows + 1, np.int32)
        rp[1:] = np.cumsum([len(r) for r in out])
        return rp, np.concatenate(out).astype(np.int32)

    Ap, Aj = structured(m, k, sa, na, int(rng.integers(0, 4)))
    Bp, Bj = structured(k, n, sb, nb, int(rng.integers(0, 2)) if rng.random() < 0.3 else 0)
    Ax = rng.integers(-9, 10, len(Aj)).astype(np.float64)
    Bx = rng.integers(-9, 10, len(Bj)).astype(np.float64)
    return (m, k, n, sa, sb, na, nb, noise), (Ap, Aj, Ax), (Bp, Bj, Bx)


def _randomized_mixed_case(oracle, seed):
    (m, k, n, sa, sb, na, nb, noise), (Ap, Aj, Ax), (Bp, Bj, Bx) = _mixed_soak_inputs(seed)
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
    vd = np.float32 if seed % 5 == 2 else np.float64              # (the float build: sums of a few dozen small integers are exact there too)
    Ax, Bx = Ax.astype(vd), Bx.astype(vd)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse(value_dtype=vd)
    assert bh.initPlatform(plats) == 0
    assert bh.set_option("class_path", 2) == 0
    Cp = np.zeros(m + 1, np.int32)
    assert bh.initData(m, k, n, len(Aj), Ax, Ap, Aj, len(Bj), Bx, Bp, Bj, Cp) == 0
    rng = np.random.default_rng(seed)
    for it in range(3):                                          # (the third one in row ranges)
        cuts = None
        if it < 2: assert bh.spgemm() == 0
        else: cuts = _multiply_in_row_ranges(bh, m, rng)
        Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), vd)
        assert bh.get_C(Cj, Cx) == 0
        assert bh.nnzCt == oracle.nnzCt(Ap, Aj, Bp)
        res = oracle.compare(ref, (Cp, Cj, Cx.astype(np.float64)), rel_tol=0.0)
        names = sorted(s_["name"] for s_ in bh.kernel_stats() if s_["launches"])
        assert res["ok"], (seed, it, cuts, res, names)
    print("mixed soak%s, seed %d: %d x %d x %d (steps %d, %d), %d / %d entries per row or node, noise %.4f -> class_state %d, %d irregular rows, %s" % (
        " (float)" if vd == np.float32 else "", seed, m, k, n, sa, sb, na, nb, noise, bh.get_info("class_state"), bh.get_info("mixed_rows"),
        "numeric_class" if "numeric_class" in names else "general pipeline"))
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


def _randomized_class_case(oracle, seed, big):
    rng = np.random.default_rng(1000 + seed)
    if big:
        m = int(rng.integers(2000, 40000)); k = int(rng.integers(2000, 40000)); n = int(rng.integers(2000, 50000))
        na = int(rng.choice([1, 2, 5, 9, 17, 27, 31, 32])); nb = int(rng.choice([1, 3, 7, 15, 16, 27, 32]))
    else:
        m = int(rng.integers(200, 3000)); k = int(rng.integers(200, 3000)); n = int(rng.integers(200, 4000))
        na = int(rng.choice([1, 2, 5, 9, 17, 31, 64])); nb = int(rng.choice([1, 3, 7, 15, 16, 33]))

    def structured(rows, cols, cnt, noise):
        offs = np.unique(rng.integers(-cols // 3, cols // 3 + 1, cnt))
        out = []
        for i in range(rows):
            # (row i around column i: the rows repeat one another's columns relative to the row, classes form.  Every fourth
            # seed keeps the draw of round 5 -- row i around column i * cols / rows: no two rows alike unless the matrix is
            # square, the class path gives way -- which is all that round's "120 of 120" had run)
            c = (i * cols // rows if seed % 4 == 3 else i) + offs
            c = c[(c >= 0) & (c < cols)]
            u = rng.random()
            if u < noise / 2 and len(c) > 1:
                c = np.delete(c, rng.integers(0, len(c)))
            elif u < noise:
                c = np.unique(np.append(c, rng.integers(0, cols)))
            elif u < noise * 1.2:
                c = c[:0]
            out.append(np.unique(c)[:64])
        rp = np.zeros(rows + 1, np.int32)
        rp[1:] = np.cumsum([len(r) for r in out])
        col = np.concatenate(out).astype(np.int32) if rp[-1] else np.empty(0, np.int32)
        return rp, col

    Ap, Aj = structured(m, k, na, 0.02)
    Bp, Bj = structured(k, n, nb, 0.02)
    if len(Aj) == 0 or len(Bj) == 0:
        pytest.skip("degenerate draw")
    Ax = rng.integers(-9, 10, len(Aj)).astype(np.float64)
    Bx = rng.integers(-9, 10, len(Bj)).astype(np.float64)
    opts = {"class_path": 2}
    if seed % 3 == 0:                              # rows of B in random stored order, multiplied as they are
        for j in range(k):
            pm = rng.permutation(Bp[j + 1] - Bp[j])
            Bj[Bp[j]:Bp[j + 1]] = Bj[Bp[j]:Bp[j + 1]][pm]
            Bx[Bp[j]:Bp[j + 1]] = Bx[Bp[j]:Bp[j + 1]][pm]
        opts["sort_b"] = 0
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, options=opts)
    assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Bp) and info["nnzC"] == ref[0][-1]
    res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
    assert res["ok"], (seed, res, sorted(_kernel_names(info)))
    check_csr_invariants(m, n, Cp, Cj)
    print("class draw, seed %d: %d x %d x %d, %d / %d entries per row -> class_state %d, %d irregular rows, %s" % (
        seed, m, k, n, na, nb, info["class_state"], info["mixed_rows"], "numeric_class" if "numeric_class" in _kernel_names(info) else "general pipeline"))


@pytest.mark.parametrize("seed", list(range(20)))
def test_row_class_path_big_classes_randomized(oracle, seed):
    """Classes beyond the register kernels' tables (bhs_class_big.hip.h) on randomised block-structured inputs: a random
    stencil (x) ones(d, d) for d = 2 .. 4 -- rows of up to ~200 entries, thousands of products -- with some rows thinned
    (groups of entries that do not qualify, rows that repeat nobody), some rows emptied, rectangular shapes, either sign."""
    import scipy.sparse as sp
    rng = np.random.default_rng(4000 + seed)
    d = int(rng.choice([2, 3, 4]))
    nodes_m, nodes_k, nodes_n = (int(rng.integers(150, 500)) for _ in range(3))
    na = int(rng.choice([5, 9, 13, 21])); nb = int(rng.choice([5, 9, 13, 21]))
    while (na * d) * (nb * d) > 8000:                      # (a class may have 8192 products)
        na -= 1
    spread = int(rng.choice([8, 12, 30]))                  # (offsets within +- spread nodes: the products overlap)

    def blocks(rows, cols, cnt, noise):
        offs = np.unique(rng.integers(-spread, spread + 1, cnt))
        data, ri, ci = [], [], []
        for i in range(rows):
            c = i * cols // rows + offs
            c = c[(c >= 0) & (c < cols)]
            ri += [i] * len(c); ci += list(c)
        P = sp.csr_matrix((np.ones(len(ri)), (ri, ci)), shape=(rows, cols))
        M = sp.kron(P, np.ones((d, d)), format="lil")
        for i in rng.integers(0, rows * d, max(1, int(noise * rows * d))):     # thinned / emptied scalar rows
            cols_i = M.rows[i]
            if len(cols_i) > 1:
                keep = rng.random(len(cols_i)) < (0.0 if rng.random() < 0.2 else 0.7)
                M.rows[i] = [c for c, kp in zip(cols_i, keep) if kp]
                M.data[i] = [1.0] * len(M.rows[i])
        M = M.tocsr(); M.sort_indices()
        return M.indptr.astype(np.int32), M.indices.astype(np.int32)

    Ap, Aj = blocks(nodes_m, nodes_k, na, 0.02)
    Bp, Bj = blocks(nodes_k, nodes_n, nb, 0.02)
    m, k, n = nodes_m * d, nodes_k * d, nodes_n * d
    Ax = rng.integers(-9, 10, len(Aj)).astype(np.float64)
    Bx = rng.integers(-9, 10, len(Bj)).astype(np.float64)
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, options={"class_path": 2})
    assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Bp) and info["nnzC"] == ref[0][-1]
    res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
    assert res["ok"], (seed, d, na, nb, res, sorted(_kernel_names(info)))
    check_csr_invariants(m, n, Cp, Cj)
    print("big classes, seed %d: %d unknowns per node, %d x %d entries per row -> %s" % (
        seed, d, na * d, nb * d, "numeric_class" if "numeric_class" in _kernel_names(info) else "general pipeline"))


def _mixed_run(oracle, m, k, n, A, B, options=None, multiplies=2, expect_rows=None, value_dtype=np.float64):
    """One data set on the class path with irregular rows (mixed mode, bhs_class_mix.hip.h): every multiply gives the
    oracle's C bit for bit, numeric_class runs, and the irregular rows are few.  Returns the last multiply's kernel names
    and its number of irregular rows."""
    ref = oracle.spgemm(m, k, n, *A, *B)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse(value_dtype=value_dtype)
    assert bh.initPlatform(plats) == 0
    for key, val in dict({"class_path": 2}, **(options or {})).items():
        assert bh.set_option(key, val) == 0
    Cp = np.zeros(m + 1, np.int32)
    A = (A[0], A[1], np.ascontiguousarray(A[2], value_dtype)); B = (B[0], B[1], np.ascontiguousarray(B[2], value_dtype))
    assert bh.initData(m, k, n, len(A[1]), A[2], A[0], A[1], len(B[1]), B[2], B[0], B[1], Cp) == 0
    names, rows = set(), 0
    for it in range(multiplies):
        assert bh.spgemm() == 0
        names = {s_["name"] for s_ in bh.kernel_stats() if s_["launches"]}
        rows = bh.get_info("mixed_rows")
        Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), value_dtype)
        assert bh.get_C(Cj, Cx) == 0
        assert bh.nnzCt == oracle.nnzCt(A[0], A[1], B[0])
        res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0 if value_dtype == np.float64 else REL_TOL)
        assert res["ok"], (it, res)
        check_csr_invariants(m, n, Cp, Cj)
        assert "numeric_class" in names, (it, names)
    if expect_rows is not None:
        assert expect_rows[0] <= rows <= expect_rows[1], rows
        # (a handful of odd rows may each find room in the class table: then there is nothing irregular about them)
        assert bh.get_info("class_state") == (2 if rows > 0 else 1)
    assert bh.free_mem() == 0 and bh.freePlatform() == 0
    return names, rows


@pytest.mark.parametrize("case", ["p27_extra_entries", "p27_one_long_row", "p27_hub_row", "p9_extra_entries", "tridiagonal_long_row",
                                  "rect_toeplitz_extra", "p27_empty_rows", "p27_f32", "p27_long_row_first_and_last",
                                  "p27_one_percent", "p27_row_block", "fem3_extra_entries", "fem3_long_row"])
def test_row_class_path_mixed_mode(oracle, case):
    """Round 6: a structured matrix with a few irregular rows stays on the class kernels; the irregular rows -- and the rows
    of A that point at an irregular row of B -- go through the general pipeline's kernels inside the same multiply, one
    scan gives rowPtrC (SpGEMM_cuda/bhsparse.h:483-586: the reference bins every row for itself)."""
    from benchmark_spgemm_using_csr_amd import gallery
    rng = np.random.default_rng(61)
    vd = np.float32 if case == "p27_f32" else np.float64
    if case.startswith("p27"):
        nx, ny, nz = 24, 22, 20
        rp, col = gallery.poisson_csr("poisson27pt", nx, ny, nz)
        m = k = n = len(rp) - 1
        if case in ("p27_extra_entries", "p27_f32", "p27_row_block"):
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.005 if case == "p27_row_block" else 0.002, seed=5)
            expect = (20, 1200)
        elif case == "p27_one_percent":
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.01, seed=6)
            expect = (100, int(0.3 * m))
        elif case == "p27_one_long_row":
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.0, long_row=(m // 2 + 7, 300))
            expect = (1, 700)
        elif case == "p27_hub_row":
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.0, long_row=(m // 3, 6000))      # 6000 x 27 products: hub / long-row bins
            expect = (1, 8000)
        elif case == "p27_long_row_first_and_last":
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.0, long_row=(0, 200))
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.0, long_row=(m - 1, 150))
            expect = (2, 900)
        else:                                                    # rows without entries: no class, no products
            keep = np.ones(len(col), bool)
            for r in (0, 100, 101, 5000, m - 1):
                keep[rp[r]:rp[r + 1]] = False
            lens = np.diff(rp).copy(); lens[[0, 100, 101, 5000, m - 1]] = 0
            rp = np.zeros(m + 1, np.int32); rp[1:] = np.cumsum(lens)
            col = col[keep]
            expect = (0, 400)                                    # (an empty row is a class like any other; the rows that point at one find room in the table)
        A = B = (rp, col, rng.integers(1, 10, len(col)).astype(np.float64))
        if case == "p27_row_block":
            # a rank's share of the multi-GPU job: rows [r0, r1) of the perturbed matrix against the whole of it -- only the rows of
            # B that the block points at are classified, counted and pruned
            r0, r1 = 3000, 6500
            m = r1 - r0
            A = ((rp[r0:r1 + 1] - rp[r0]).astype(np.int32), col[rp[r0]:rp[r1]], B[2][rp[r0]:rp[r1]])
            expect = (50, 1100)
    elif case.startswith("fem3"):
        # several unknowns per node (81 entries a row, 6561 products: the big-class kernels, bhs_class_big.hip.h) with irregular rows:
        # k_class_numeric_big passes them by as the ring kernel does
        rp, col = gallery.block_expand_csr(*gallery.poisson_csr("poisson27pt", 14, 13, 12), 3)
        m = k = n = len(rp) - 1
        if case == "fem3_extra_entries":
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.004, seed=21)
            expect = (20, int(0.3 * m))
        else:
            rp, col = gallery.perturb_rows_csr(rp, col, n, 0.0, long_row=(m // 2 + 1, 500))
            expect = (1, 2500)
        A = B = (rp, col, rng.integers(1, 10, len(col)).astype(np.float64))
    elif case == "p9_extra_entries":
        rp, col = gallery.poisson_csr("poisson9pt", 120, 90, 1)
        m = k = n = len(rp) - 1
        rp, col = gallery.perturb_rows_csr(rp, col, n, 0.003, seed=9)
        A = B = (rp, col, rng.integers(1, 10, len(col)).astype(np.float64))
        expect = (10, 600)
    elif case == "tridiagonal_long_row":
        m = k = n = 20000
        rp, col, val = _toeplitz(m, k, (-1, 0, 1), rng)
        rp, col = gallery.perturb_rows_csr(rp, col, n, 0.0, long_row=(700, 300))
        A = B = (rp, col, rng.integers(1, 10, len(col)).astype(np.float64))
        expect = (1, 400)
    else:
        m, k, n = 9000, 9500, 10000
        A = _toeplitz(m, k, (-3, -1, 0, 2, 40), rng)
        B = _toeplitz(k, n, (-2, 0, 1, 7), rng)
        rpa, cola = gallery.perturb_rows_csr(A[0], A[1], k, 0.002, seed=3)
        rpb, colb = gallery.perturb_rows_csr(B[0], B[1], n, 0.002, seed=4)
        A = (rpa, cola, rng.integers(1, 10, len(cola)).astype(np.float64))
        B = (rpb, colb, rng.integers(1, 10, len(colb)).astype(np.float64))
        expect = (0, 400)
    _mixed_run(oracle, m, k, n, A, B, expect_rows=expect, value_dtype=vd)


def test_mixed_mode_in_row_ranges_and_back_to_clean(oracle):
    """The multiply in two halves with the numeric half in row ranges (what the multi-GPU layer runs) on a data set with
    irregular rows; then the arrays change under the handle (borrowed pointers): a clean data set runs the clean flow again,
    and mixed mode switched off sends the perturbed one to the general pipeline as until round 5."""
    from benchmark_spgemm_using_csr_amd import gallery
    rng = np.random.default_rng(62)
    rp0, col0 = gallery.poisson_csr("poisson27pt", 20, 18, 16)
    m = k = n = len(rp0) - 1
    rp, col = gallery.perturb_rows_csr(rp0, col0, n, 0.003, seed=8, long_row=(m // 2, 400))
    val = rng.integers(1, 10, len(col)).astype(np.float64)
    ref = oracle.spgemm(m, k, n, rp, col, val, rp, col, val)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0 and bh.set_option("class_path", 2) == 0
    Cp = np.zeros(m + 1, np.int32)
    assert bh.initData(m, k, n, len(col), val, rp, col, len(col), val, rp, col, Cp) == 0
    assert bh.spgemm() == 0                                      # (finds the irregular rows; the data set is "mixed" from here on)
    for nranges in (1, 3, 5):
        assert bh.spgemm_symbolic() == 0
        nnzCt, nnzC = bh.nnzCt, bh.nnzC
        assert nnzCt == oracle.nnzCt(rp, col, rp) and nnzC == ref[0][-1]
        cuts = np.linspace(0, m, nranges + 1).astype(int)
        for a, b in zip(cuts[:-1], cuts[1:]):
            assert bh.spgemm_numeric(int(a), int(b)) == 0
        assert bh.spgemm_finish() == 0
        assert bh.get_info("mixed_rows") > 0
        Cj = np.empty(nnzC, np.int32); Cx = np.empty(nnzC, np.float64)
        assert bh.get_C(Cj, Cx) == 0
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"], nranges
    assert bh.free_mem() == 0
    # mixed mode off: the general pipeline, for good
    assert bh.set_option("class_mixed", 0) == 0
    assert bh.initData(m, k, n, len(col), val, rp, col, len(col), val, rp, col, Cp) == 0
    for it in range(2):
        assert bh.spgemm() == 0
        names = {s_["name"] for s_ in bh.kernel_stats() if s_["launches"]}
        assert "numeric_class" not in names                      # (a 400-entry row: with mixed mode off the hand-over hint keeps the class path away)
        Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
        assert bh.get_C(Cj, Cx) == 0
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.parametrize("case", ["random_short_rows", "too_many_products", "one_long_row", "too_many_entries"])
def test_row_class_path_gives_way_to_the_general_pipeline(oracle, case):
    """Inputs the class tables cannot take: more classes than table slots (unstructured rows), a class with more than
    8192 products or 512 entries per row, a row with more than 256 entries.  The multiply starts over on the general
    pipeline (same call, right answer) and the data set stays there: the next multiply does not classify again."""
    rng = np.random.default_rng(32)
    if case == "random_short_rows":
        m, k, n = 12000, 12000, 12000
        A = random_csr(m, k, 8.0 / k, rng)
        B = random_csr(k, n, 8.0 / n, rng)
        tried = True
    elif case == "too_many_products":
        m = k = n = 1500                       # 96 x 96 = 9216 products per row (191 distinct columns)
        offs = tuple(range(-48, 48))
        A = _toeplitz(m, k, offs, rng)
        B = _toeplitz(k, n, offs, rng)
        tried = True
    elif case == "too_many_entries":
        m = k = n = 3000                       # 24 x 30 = 720 products, 24 * 30 distinct columns > 512
        A = _toeplitz(m, k, tuple(100 * o for o in range(-12, 12)), rng)
        B = _toeplitz(k, n, tuple(range(-15, 15)), rng)
        tried = True
    else:
        m = k = n = 2000
        rp, col, val = _toeplitz(m, k, (-1, 0, 1), rng)
        rows = [col[rp[i]:rp[i + 1]] for i in range(m)]
        rows[700] = np.arange(600, 900)        # 300 entries: until round 5 the hint from bhs_set_data kept the class path away
        rp = np.zeros(m + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
        col = np.concatenate(rows).astype(np.int32)
        A = B = (rp, col, rng.integers(1, 10, len(col)).astype(np.float64))
        # round 6: ONE odd row no longer moves the other rows off the class kernels (mixed mode: numeric_class DID run); with
        # mixed mode off it still does, and that is what the loop below checks
        names, rows_irregular = _mixed_run(oracle, m, k, n, A, B, expect_rows=(1, 400))
        assert "classify_rows" in names and "numeric_class" in names
        tried = False
    ref = oracle.spgemm(m, k, n, *A, *B)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    Cp = np.zeros(m + 1, np.int32)
    assert bh.set_option("class_path", 2) == 0
    if case == "one_long_row":
        assert bh.set_option("class_mixed", 0) == 0
    assert bh.initData(m, k, n, len(A[1]), A[2], A[0], A[1], len(B[1]), B[2], B[0], B[1], Cp) == 0
    for it in range(2):
        assert bh.spgemm() == 0
        names = {s["name"] for s in bh.kernel_stats() if s["launches"]}
        assert "numeric_class" not in names
        assert ("classify_rows" in names) == (tried and it == 0), (it, names)
        Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
        assert bh.get_C(Cj, Cx) == 0
        assert bh.nnzCt == oracle.nnzCt(A[0], A[1], B[0])
        assert oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)["ok"]
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


def _banded(m, offsets, rng, drop=0.0):
    """Rows i -> columns {i + o} (clipped), optionally thinned."""
    rows = []
    for i in range(m):
        c = np.array(sorted({i + o for o in offsets if 0 <= i + o < m}), np.int32)
        if drop:
            c = c[rng.random(len(c)) >= drop]
        rows.append(c)
    rp = np.zeros(m + 1, np.int32)
    rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32) if rp[-1] else np.empty(0, np.int32)
    return rp, col


@pytest.mark.parametrize("case", ["p27_slab", "band_runs", "band_thin", "two_level"])
def test_banded_matrices(oracle, case):
    """Clustered columns: thin grids, runs of entries around far-apart diagonals, thinned bands, rows of A with more
    than 64 entries -- through the default path and through the general pipeline alone: the oracle's C, and identical
    bits from both."""
    rng = np.random.default_rng(17)
    if case == "p27_slab":
        m, rp, col, val = poisson_case("poisson27pt", 40, 40, 3)
    elif case == "band_runs":                                   # runs of 4 around far-apart diagonals
        m = 6000
        offs = [d + t for d in (-1900, -611, -30, 0, 33, 700, 2500) for t in range(4)]
        rp, col = _banded(m, offs, rng)
        val = rng.integers(1, 10, len(col)).astype(np.float64)
    elif case == "band_thin":
        m = 5000
        rp, col = _banded(m, list(range(-40, 41, 3)) + [400, 401, 402], rng, drop=0.3)
        val = rng.integers(1, 10, len(col)).astype(np.float64)
    else:                                                       # > 64-entry rows of A, clustered columns
        m = 3000
        rp, col = _banded(m, list(range(-35, 36)), rng, drop=0.1)
        val = rng.integers(1, 10, len(col)).astype(np.float64)
    A = (rp, col, val)
    Cp, Cj, Cx, info = _check(oracle, m, m, m, A, A)
    Cp2, Cj2, Cx2, info2 = _check(oracle, m, m, m, A, A, options={"class_path": 0, "wave_first": 0, "lane_first": 0, "direct_bins": 0})
    assert np.array_equal(Cp, Cp2) and np.array_equal(Cj, Cj2) and np.array_equal(Cx, Cx2)


def _kernel_names(info):
    return {s["name"] for s in info["kernels"]}


def _phase_cases():
    from benchmark_spgemm_using_csr_amd import gallery
    rng = np.random.default_rng(5)
    yield "p27", poisson_case("poisson27pt", 15, 14, 13)[1:], None
    yield "p5", poisson_case("poisson5pt", 70, 60)[1:], None
    rp, col = gallery.powerlaw_csr(30000, 30000, 110000, 2500, hubs=3)
    yield "powerlaw", (rp, col, gallery.fill_values(len(col))), None
    A = random_csr(700, 500, 0.03, rng, empty_rows=(0, 3, 699))
    B = random_csr(500, 2500, 0.02, rng)
    yield "rect", A, B


@pytest.mark.parametrize("path", ["general", "classes"])
@pytest.mark.parametrize("nranges", [1, 3, 7])
def test_symbolic_numeric_halves_in_row_ranges(oracle, nranges, path):
    """bhs_spgemm_symbolic / bhs_spgemm_numeric(row range) / bhs_spgemm_finish == bhs_spgemm, for every kernel family
    (direct lane / wave launches and binned queues), with the ranges issued in any order, into the library's own C
    arrays and into caller-owned ones (bhs_set_output_device)."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    for tag, A, B in _phase_cases():
        B = A if B is None else B
        Ap, Aj, Ax = A
        Bp, Bj, Bx = B
        m, k = len(Ap) - 1, len(Bp) - 1
        n = int(max(Bj.max() + 1, k)) if len(Bj) else k
        ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        dA, dB = (t(Ap), t(Aj), t(Ax)), (t(Bp), t(Bj), t(Bx))
        plats = [False] * bhmod.NUM_PLATFORMS
        plats[bhmod.BHSPARSE_HIP] = True
        bh = bhmod.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.set_option("class_path", 2 if path == "classes" else 0) == 0     # (2: row classes whenever the rows classify: p27, p5)
        assert bh.initData_device(m, k, n, len(Aj), dA[2], dA[0], dA[1], len(Bj), dB[2], dB[0], dB[1]) == 0
        L, h = bh._lib, bh._h
        for external in (False, True):
            ct, cc = C.c_int64(0), C.c_int(0)
            assert L.bhs_spgemm_symbolic(h, C.byref(ct), C.byref(cc)) == 0, tag
            assert ct.value == oracle.nnzCt(Ap, Aj, Bp) and cc.value == ref[0][-1]
            if external:
                oj = torch.full((cc.value + 5,), -1, dtype=torch.int32, device=dev)
                ox = torch.full((cc.value + 5,), -1.0, dtype=torch.float64, device=dev)
                assert L.bhs_set_output_device(h, C.c_void_p(oj.data_ptr()), C.c_void_p(ox.data_ptr()), cc.value + 5) == 0
            cuts = [m * s // nranges for s in range(nranges + 1)]
            order = list(range(nranges))[::-1] if external else list(range(nranges))      # any order
            for s in order:
                assert L.bhs_spgemm_numeric(h, cuts[s], cuts[s + 1]) == 0, (tag, s)
            assert L.bhs_spgemm_numeric(h, 5, 4) == bhmod._lib.BHS_ERR_INVALID_ARG
            assert L.bhs_spgemm_finish(h, None) == 0
            assert L.bhs_spgemm_numeric(h, 0, m) == bhmod._lib.BHS_ERR_NOT_READY      # no multiply open any more
            Cp = bh.get_rowptrC()
            if external:
                Cj, Cx = oj[:cc.value].cpu().numpy(), ox[:cc.value].cpu().numpy()
                assert bool((oj[cc.value:] == -1).all())                                # nothing written past the end
                assert L.bhs_set_output_device(h, None, None, 0) == 0
            else:
                Cj = np.empty(cc.value, np.int32); Cx = np.empty(cc.value, np.float64)
                assert bh.get_C(Cj, Cx) == 0
            res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)
            assert res["ok"], (tag, external, res)
        # and an ordinary multiply on the same handle afterwards
        assert bh.spgemm() == 0 and bh.get_nnzC() == ref[0][-1]
        if path == "classes" and tag in ("p27", "p5"):
            assert "numeric_class" in {s["name"] for s in bh.kernel_stats() if s["launches"]}, tag
        assert bh.free_mem() == 0 and bh.freePlatform() == 0


@pytest.mark.parametrize("sub_blocks", [1, 4])
@pytest.mark.parametrize("kind", ["powerlaw", "stencil_classes"])
def test_native_allgatherv_world_size_1(oracle, sub_blocks, kind):
    """libbhsparse_dist.so on one GPU (world_size 1: communicator, size exchange, in-place output, row-pointer rebase,
    numeric half in row ranges; no peers to send to): the assembled CSR equals the oracle's, twice in a row.  Once on
    the general pipeline (power-law rows) and once on the row-class kernels (what poisson27pt 256^3 takes per rank)."""
    import torch
    from benchmark_spgemm_using_csr_amd import dist as bdist, gallery
    dev = torch.device("cuda", 0)
    if kind == "powerlaw":
        rp, col = gallery.powerlaw_csr(20000, 20000, 80000, 2000, hubs=3)
        val = gallery.fill_values(len(col))
    else:
        _, rp, col, val = poisson_case("poisson27pt", 17, 16, 15)
    m = len(rp) - 1
    ref = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dp, dj, dx = t(rp), t(col), t(val)
    plats = [False] * bhmod.NUM_PLATFORMS
    plats[bhmod.BHSPARSE_HIP] = True
    bh = bhmod.bhsparse()
    assert bh.initPlatform(plats) == 0
    assert bh.set_option("class_path", 2 if kind == "stencil_classes" else 0) == 0
    assert bh.initData_device(m, m, m, len(col), dx, dp, dj, len(col), dx, dp, dj) == 0
    nd = bdist.NativeDist(bh, world=1, rank=0)
    cap = int(ref[0][-1]) + 17
    frp = torch.empty(m + 1, dtype=torch.int32, device=dev)
    fc = torch.empty(cap, dtype=torch.int32, device=dev)
    fv = torch.empty(cap, dtype=torch.float64, device=dev)
    for _ in range(2):
        frp.fill_(-1); fc.fill_(-1); fv.fill_(-1.0)
        ct, cc = nd.spgemm_allgatherv(m, m, frp, fc, fv, sub_blocks=sub_blocks)
        torch.cuda.synchronize()
        assert ct == oracle.nnzCt(rp, col, rp) and cc == ref[0][-1]
        res = oracle.compare(ref, (frp.cpu().numpy(), fc[:cc].cpu().numpy(), fv[:cc].cpu().numpy()), rel_tol=0.0)
        assert res["ok"], res
    # too small a destination is an error code, not a crash
    small_c = torch.empty(10, dtype=torch.int32, device=dev)
    small_v = torch.empty(10, dtype=torch.float64, device=dev)
    with pytest.raises(RuntimeError):
        nd.spgemm_allgatherv(m, m, frp, small_c, small_v, sub_blocks=sub_blocks)
    assert bh.spgemm() == 0                       # the handle is usable afterwards ...
    assert ("numeric_class" in {s["name"] for s in bh.kernel_stats() if s["launches"]}) == (kind == "stencil_classes")
    # ... and writes into its OWN arrays again: the gather's destination is unbound on every exit of the call
    frp.fill_(-7); fc.fill_(-7); fv.fill_(-7.0)
    assert bh.spgemm() == 0
    torch.cuda.synchronize()
    assert int(fc.max()) == -7 and float(fv.max()) == -7.0
    Cj = np.empty(bh.get_nnzC(), np.int32); Cx = np.empty(bh.get_nnzC(), np.float64)
    assert bh.get_C(Cj, Cx) == 0
    assert oracle.compare(ref, (bh.get_rowptrC(), Cj, Cx), rel_tol=0.0)["ok"]
    assert nd.nranks() == 1                       # what RCCL itself counts (bhs_dist_nranks)
    nd.close()
    assert bh.free_mem() == 0 and bh.freePlatform() == 0


def test_columns_rebuilt_from_row_classes(hiplib):
    """bhs_get_class_tables_device / bhs_expand_class_columns_device (the values-only all-gatherv of the multi-GPU layer
    rebuilds the other ranks' column indices with them): after a multiply by row classes the column indices rebuilt from
    (class of every row, class tables, rowPtrC) equal colIndC -- for the whole matrix, and for a row block of A rebuilt
    at another place as a peer would (local row numbers); a multiply on the general pipeline reports `usable` = 0."""
    import torch
    from benchmark_spgemm_using_csr_amd import gallery, facade
    from benchmark_spgemm_using_csr_amd.dist import device_view
    dev = torch.device("cuda", 0)
    Bp, Bj = gallery.poisson_csr_torch("poisson27pt", 14, 13, 12, device=dev)
    Bx = gallery.fill_values_torch(Bj.numel(), device=dev)
    k = Bp.numel() - 1
    plats = [False] * facade.NUM_PLATFORMS
    plats[facade.BHSPARSE_HIP] = True
    for r0, r1 in ((0, k), (700, 1500)):
        lo, hi = int(Bp[r0]), int(Bp[r1])
        Ap = (Bp[r0:r1 + 1] - Bp[r0]).contiguous()
        Aj, Ax = Bj[lo:hi].clone(), Bx[lo:hi].clone()
        m = r1 - r0
        bh = facade.bhsparse()
        assert bh.initPlatform(plats) == 0
        assert bh.set_option("class_path", 2) == 0
        assert bh.initData_device(m, k, k, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
        assert bh.spgemm() == 0
        assert any(s["name"] == "numeric_class" and s["ms"] > 0 for s in bh.kernel_stats())
        cC, cInfo, cRel, slots, stride, usable = bh.class_tables_device()
        assert usable and slots > 0 and stride > 0
        pr, pc, _ = bh.get_C_device()
        nnzC = bh.get_nnzC()
        Cj = device_view(pc, nnzC, torch.int32, dev)
        # a peer's copies of the tables and classes, and its own output array
        info2 = device_view(cInfo, slots * 4, torch.int32, dev).clone()
        rel2 = device_view(cRel, slots * stride, torch.int32, dev).clone()
        cls2 = device_view(cC, m, torch.int32, dev).clone()
        rp2 = device_view(pr, m + 1, torch.int32, dev).clone()
        out = torch.full((nnzC,), -7, dtype=torch.int32, device=dev)
        assert bh.expand_class_columns_device(m, 0, cls2.data_ptr(), info2.data_ptr(), rel2.data_ptr(), stride, rp2.data_ptr(), out.data_ptr()) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, Cj)
        # ... and the general pipeline has no tables to offer
        assert bh.set_option("class_path", 0) == 0
        assert bh.spgemm() == 0
        assert bh.class_tables_device()[5] is False
        bh.free_mem()
        bh.freePlatform()
