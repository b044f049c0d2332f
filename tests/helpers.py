"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from benchmark_spgemm_using_csr_amd import gallery


def poisson_case(name, nx, ny, nz=1):
    rp, col = gallery.poisson_csr(name, nx, ny, nz)
    val = gallery.fill_values(len(col))
    m = len(rp) - 1
    return m, rp, col, val


def random_csr(m, n, density, rng, empty_rows=(), values="int", max_row=None):
    """Random CSR with sorted duplicate-free rows."""
    lens = rng.binomial(n, density, size=m)
    if max_row is not None:
        lens = np.minimum(lens, max_row)
    for r in empty_rows:
        if r < m:
            lens[r] = 0
    rp = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    cols = np.empty(rp[-1], np.int32)
    for i in range(m):
        cols[rp[i]:rp[i + 1]] = np.sort(rng.choice(n, lens[i], replace=False))
    if values == "int":
        val = rng.integers(1, 10, rp[-1]).astype(np.float64)
    elif values == "signed":
        val = rng.integers(-4, 5, rp[-1]).astype(np.float64)
        val[val == 0] = 1.0
    else:
        val = rng.standard_normal(rp[-1])
    return rp.astype(np.int32), cols, val


def check_csr_invariants(m, n, Cp, Cj):
    """Output postconditions of SURVEY.md §8b: rowPtr monotone from 0, rows strictly ascending."""
    assert Cp[0] == 0 and np.all(np.diff(Cp) >= 0)
    if len(Cj):
        assert Cj.min() >= 0 and Cj.max() < n
        d = np.diff(Cj.astype(np.int64))
        starts = np.zeros(len(Cj), bool)
        starts[Cp[1:-1][Cp[1:-1] < len(Cj)]] = True
        assert np.all((d > 0) | starts[1:]), "rows must be strictly ascending by column"
