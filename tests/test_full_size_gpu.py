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
         "powerlaw_1m": ("powerlaw", (1000005, 3105536, 4700))}   # stand-in for configs[3] (webbase-1M, file absent)


def _spmv(rp, col, val, x):
    import torch
    m = rp.numel() - 1
    rows = torch.repeat_interleave(torch.arange(m, device=rp.device), (rp[1:] - rp[:-1]).long())
    y = torch.zeros(m, dtype=torch.float64, device=rp.device)
    y.index_add_(0, rows, val * x[col.long()])
    return y


@pytest.mark.parametrize("tag", ["p5_1024", "p27_128", "powerlaw_1m"])
def test_full_size_digests_and_properties(hiplib, tag):
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
    assert bh.spgemm() == 0
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
