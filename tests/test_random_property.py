"""Property tests on random CSR pairs (hypothesis): the oracle against scipy on the CPU, the HIP path
against the oracle on the GPU.  Shapes, densities and row-length skew are drawn so that every
accumulator family (quarter-wave, wave, workgroup, bitmap accumulator) is reachable."""
import numpy as np
import pytest
import scipy.sparse as sp
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import check_csr_invariants


def _random_pair(seed, m, k, n, dA, dB, skew, signed):
    rng = np.random.default_rng(seed)

    def mat(rows, cols, dens, hub):
        lens = rng.binomial(cols, min(1.0, dens), size=rows)
        if hub and rows:
            idx = rng.choice(rows, size=max(1, rows // 50), replace=False)
            lens[idx] = np.minimum(cols, (lens[idx] + 1) * hub)
        lens[rng.random(rows) < 0.1] = 0                       # some empty rows
        rp = np.zeros(rows + 1, np.int64)
        np.cumsum(lens, out=rp[1:])
        cj = np.concatenate([np.sort(rng.choice(cols, L, replace=False)) for L in lens] +
                            [np.empty(0, np.int64)]).astype(np.int32)
        v = rng.integers(1, 10, len(cj)).astype(np.float64)
        if signed:
            v *= rng.choice([-1.0, 1.0], len(cj))
        return rp.astype(np.int32), cj, v
    return mat(m, k, dA, skew), mat(k, n, dB, skew)


CASE = dict(seed=st.integers(0, 2 ** 31 - 1), m=st.integers(1, 300), k=st.integers(1, 300),
            n=st.integers(1, 4000), dA=st.floats(0.002, 0.3), dB=st.floats(0.002, 0.3),
            skew=st.sampled_from([0, 0, 8, 40]))


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(**CASE)
def test_oracle_matches_scipy_on_positive_values(oracle, seed, m, k, n, dA, dB, skew):
    (Ap, Aj, Ax), (Bp, Bj, Bx) = _random_pair(seed, m, k, n, dA, dB, skew, signed=False)
    Cp, Cj, Cx = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, nthreads=2)
    C = (sp.csr_matrix((Ax, Aj, Ap), shape=(m, k)) @ sp.csr_matrix((Bx, Bj, Bp), shape=(k, n))).tocsr()
    C.sort_indices()
    assert np.array_equal(C.indptr, Cp) and np.array_equal(C.indices, Cj) and np.array_equal(C.data, Cx)
    assert oracle.nnzCt(Ap, Aj, Bp) == int(sum(Bp[j + 1] - Bp[j] for j in Aj))


# option sets that route rows through the alternative kernels / pipeline shapes (include/bhsparse_hip.h)
OPTION_SETS = [{}, {}, {"lane_rows": 2, "lane_numeric": 1}, {"lane_rows": 0}, {"compress_b": 2},
               {"wave_first": 0, "lane_first": 0}, {"direct_bins": 0}, {"concurrent_bins": 1},
               {"lane_rows": 2, "lane_numeric": 0, "small_b": 0}, {"sort_b": 0}]


@pytest.mark.gpu
@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(signed=st.booleans(), opts=st.sampled_from(OPTION_SETS), shuffle_b=st.booleans(), tiny=st.booleans(), **CASE)
def test_hip_matches_oracle(oracle, seed, m, k, n, dA, dB, skew, signed, opts, shuffle_b, tiny):
    from benchmark_spgemm_using_csr_amd.facade import spgemm_csr
    if tiny:                                                    # rows of a few entries: lane-first / wave-first territory
        dA, dB, skew = min(dA, 6.0 / max(k, 1)), min(dB, 8.0 / max(n, 1)), 0
    (Ap, Aj, Ax), (Bp, Bj, Bx) = _random_pair(seed, m, k, n, dA, dB, skew, signed)
    if shuffle_b:                                               # unsorted rows of B (sorted by the library, or not: sort_b)
        rng = np.random.default_rng(seed ^ 0x5bd1)
        for j in range(k):
            p = rng.permutation(Bp[j + 1] - Bp[j])
            Bj[Bp[j]:Bp[j + 1]] = Bj[Bp[j]:Bp[j + 1]][p]
            Bx[Bp[j]:Bp[j + 1]] = Bx[Bp[j]:Bp[j + 1]][p]
    Cp, Cj, Cx, info = spgemm_csr(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx, options=opts)
    ref = oracle.spgemm(m, k, n, Ap, Aj, Ax, Bp, Bj, Bx)
    res = oracle.compare(ref, (Cp, Cj, Cx), rel_tol=0.0)       # integer-valued inputs: bit-exact, zeros kept
    assert res["ok"], res
    assert info["nnzCt"] == oracle.nnzCt(Ap, Aj, Bp)
    check_csr_invariants(m, n, Cp, Cj)
