"""Generates the golden fixtures under tests/golden/ (run in the dev container;
needs scipy).  Fixtures are DATA: inputs + expected CSR outputs.

  small_test.npz   inputs of the reference's built-in test (SpGEMM_cuda/main.cu:153-205)
                   + expected C (hand-derived in SURVEY.md §4, re-derived with scipy here)
  cage4.mtx        the reference's sample matrix (SpGEMM_cuda/cage4.mtx), data file
  cage4_sq.npz     C = cage4^2 with the file's values and with all-ones values
  p5_16.npz        poisson5pt 16x16,  C = A^2, values 1..9 (gallery.fill_values)
  p27_6.npz        poisson27pt 6^3,   C = A^2
  rect_rand.npz    random rectangular A (37x53) * B (53x29) with empty rows/cols, positive values
  checksums.json   digests (nnzCt, nnzC, sum rowPtr, weighted col sum, sum val) of larger
                   cases computed with scipy: p5 256^2, p27 51^3 (reference defaults, main.cu:32-51)

scipy.sparse is a valid second oracle here because every value is > 0, so no
exact cancellation occurs (scipy drops numerical zeros; the reference keeps them).
"""
import json
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import importlib.util
_spec = importlib.util.spec_from_file_location(
    "gallery", os.path.join(HERE, "..", "..", "benchmark_spgemm_using_csr_amd", "gallery.py"))
gallery = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gallery)


def product(A, B):
    C = (A @ B).tocsr()
    C.sort_indices()
    nnzCt = int(sum(B.indptr[j + 1] - B.indptr[j] for j in A.indices))
    return C, nnzCt


def save(name, A, B, C, nnzCt):
    np.savez_compressed(os.path.join(HERE, name),
                        m=A.shape[0], k=A.shape[1], n=B.shape[1],
                        Ap=A.indptr.astype(np.int32), Aj=A.indices.astype(np.int32), Ax=A.data.astype(np.float64),
                        Bp=B.indptr.astype(np.int32), Bj=B.indices.astype(np.int32), Bx=B.data.astype(np.float64),
                        Cp=C.indptr.astype(np.int64), Cj=C.indices.astype(np.int32), Cx=C.data.astype(np.float64),
                        nnzCt=nnzCt)


def digest(C, nnzCt):
    t = np.arange(C.nnz, dtype=np.uint64)
    return {"nnzCt": int(nnzCt), "nnzC": int(C.nnz),
            "sum_rowptr": int(C.indptr.astype(np.uint64).sum()),
            "wsum_col": int((C.indices.astype(np.uint64) * (t % np.uint64(8191) + np.uint64(1))).sum()),
            "sum_val": float(C.data.sum())}


def main():
    # reference built-in test, main.cu:153-205
    A = sp.csr_matrix((np.arange(1, 7) * 10.0, [0, 1, 2, 3, 3, 1], [0, 1, 4, 5, 6]), shape=(4, 6))
    B = sp.csr_matrix((np.arange(1, 8) * 1.0, [0, 1, 3, 0, 1, 1, 3], [0, 1, 3, 5, 5, 5, 7]), shape=(6, 4))
    C, ct = product(A, B)
    assert C.indptr.tolist() == [0, 1, 4, 4, 6] and C.indices.tolist() == [0, 0, 1, 3, 1, 3]
    assert C.data.tolist() == [10, 120, 190, 60, 120, 180] and ct == 7
    save("small_test.npz", A, B, C, ct)

    M = scipy.io.mmread(os.path.join(HERE, "cage4.mtx")).tocsr()
    M.sort_indices()
    C, ct = product(M, M)
    assert ct == 269 and C.nnz == 81
    ones = sp.csr_matrix((np.ones(M.nnz), M.indices, M.indptr), shape=M.shape)
    C1, _ = product(ones, ones)
    assert C1.data[:9].tolist() == [5, 3, 2, 3, 3, 3, 3, 3, 2] and C1.data.sum() == 269
    np.savez_compressed(os.path.join(HERE, "cage4_sq.npz"), m=9, Ap=M.indptr.astype(np.int32),
                        Aj=M.indices.astype(np.int32), Ax=M.data, Cp=C.indptr.astype(np.int64),
                        Cj=C.indices.astype(np.int32), Cx=C.data, Cx_ones=C1.data, nnzCt=ct)

    sums = {}
    for tag, name, dims, full in (("p5_16", "poisson5pt", (16, 16, 1), True),
                                  ("p27_6", "poisson27pt", (6, 6, 6), True),
                                  ("p9_12", "poisson9pt", (12, 12, 1), True),
                                  ("p7_7", "poisson7pt", (7, 7, 7), True),
                                  ("p5_256", "poisson5pt", (256, 256, 1), False),
                                  ("p27_51", "poisson27pt", (51, 51, 51), False)):
        rp, col = gallery.poisson_csr(name, *dims)
        val = gallery.fill_values(len(col))
        m = len(rp) - 1
        A = sp.csr_matrix((val, col, rp), shape=(m, m))
        C, ct = product(A, A)
        if full:
            save(tag + ".npz", A, A, C, ct)
        sums[tag] = digest(C, ct)
        sums[tag].update({"stencil": name, "dims": list(dims), "m": m, "nnzA": int(A.nnz)})

    rng = np.random.default_rng(7)
    A = sp.random(37, 53, density=0.12, random_state=rng, format="csr", data_rvs=lambda s: rng.integers(1, 10, s).astype(float))
    B = sp.random(53, 29, density=0.15, random_state=rng, format="csr", data_rvs=lambda s: rng.integers(1, 10, s).astype(float))
    A = A.tolil(); A[5, :] = 0; A[20, :] = 0; A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    B = B.tolil(); B[7, :] = 0; B[:, 3] = 0; B = B.tocsr(); B.eliminate_zeros(); B.sort_indices()
    C, ct = product(A, B)
    save("rect_rand.npz", A, B, C, ct)
    sums["rect_rand"] = digest(C, ct)

    with open(os.path.join(HERE, "checksums.json"), "w") as f:
        json.dump(sums, f, indent=1, sort_keys=True)
    print(json.dumps(sums, indent=1, sort_keys=True))


def full_size():
    """BASELINE.json configs[1] and configs[2] at full size (poisson5pt 1024^2, poisson27pt 128^3): digests only,
    merged into checksums.json (scipy needs ~2 minutes and ~12 GB for the 27-point case)."""
    path = os.path.join(HERE, "checksums.json")
    sums = json.load(open(path))
    for tag, name, dims in (("p5_1024", "poisson5pt", (1024, 1024, 1)), ("p27_128", "poisson27pt", (128, 128, 128)),
                            ("powerlaw_1m", "powerlaw", (1000005, 3105536, 4700)),
                            ("weblike_1m", "weblike", (1000005,)), ("fem3_40", "fem3", (40, 40, 40)),
                            ("rmat_s20", "rmat", (1 << 20,))):
        if tag in sums and "--force" not in sys.argv:
            continue
        if name == "powerlaw":        # stand-in for configs[3] (webbase-1M: the SuiteSparse file is not in the image)
            rp, col = gallery.powerlaw_csr(dims[0], dims[0], dims[1], dims[2])
        elif name == "weblike":       # the stand-in with webbase-1M's compression (nnzCt / nnzC = 1.35; published 1.36)
            rp, col = gallery.weblike_csr(dims[0])
        elif name == "rmat":          # R-MAT graph, 2^20 rows: the rows the column-window kernels are for
            rp, col = gallery.rmat_csr()
        elif name == "fem3":          # poisson27pt (x) ones(3, 3): 3 unknowns per node
            rp, col = gallery.block_expand_csr(*gallery.poisson_csr("poisson27pt", *dims), 3)
        else:
            rp, col = gallery.poisson_csr(name, *dims)
        val = gallery.fill_values(len(col))
        m = len(rp) - 1
        A = sp.csr_matrix((val, col, rp), shape=(m, m))
        C, _ = product_no_ct(A, A)
        ones = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
        ct = int((ones @ np.diff(A.indptr).astype(np.float64)).sum())       # sum over A entries of the B row length
        d = digest(C, ct)
        t = np.arange(C.nnz, dtype=np.uint64) % np.uint64(8191) + np.uint64(1)
        d["wsum_val"] = float((C.data * t.astype(np.float64)).sum())      # integers < 2^53: exact in any order
        d.update({"stencil": name, "dims": list(dims), "m": m, "nnzA": int(A.nnz)})
        sums[tag] = d
        print(tag, d, flush=True)
    with open(path, "w") as f:
        json.dump(sums, f, indent=1, sort_keys=True)


def product_no_ct(A, B):
    C = (A @ B).tocsr()
    C.sort_indices()
    return C, None


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--full":
        full_size()
    else:
        main()
