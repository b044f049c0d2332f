"""world_size-2 gloo tests (CPU) of the multi-GPU layer: row blocks of A, B replicated,
all-gatherv of the per-rank CSR blocks == single-shot product."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dims, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from benchmark_spgemm_using_csr_amd import gallery, dist as bdist
        from oracle import oracle
        nx, ny, nz = dims
        m = nx * ny * nz
        Bp, Bj = gallery.poisson_csr("poisson27pt", nx, ny, nz)
        Bx = gallery.fill_values(len(Bj))
        r0, r1 = bdist.row_block(m, rank, world)
        # the rank's row block of A (rebased row pointer), same values as rows [r0,r1) of B
        lo, hi = Bp[r0], Bp[r1]
        Ap = (Bp[r0:r1 + 1] - lo).astype(np.int32)
        Aj, Ax = Bj[lo:hi], Bx[lo:hi]
        # per-rank product: the CPU oracle stands in for the HIP path here (no GPU in this test)
        Cp, Cj, Cx = oracle.spgemm(r1 - r0, m, m, Ap, Aj, Ax, Bp, Bj, Bx, nthreads=2)
        rp, cc, vv, info = bdist.allgatherv_csr(m, torch.from_numpy(Cp.astype(np.int32)), torch.from_numpy(Cj),
                                                torch.from_numpy(Cx))
        assert info["rows"] == [bdist.row_block(m, r, world)[1] - bdist.row_block(m, r, world)[0] for r in range(world)]
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rp=rp.numpy(), cc=cc.numpy(), vv=vv.numpy())
        # second call reusing the output buffers
        rp2, cc2, vv2, _ = bdist.allgatherv_csr(m, torch.from_numpy(Cp.astype(np.int32)), torch.from_numpy(Cj),
                                                torch.from_numpy(Cx), out=(rp, cc, vv))
        assert torch.equal(rp2, rp) and torch.equal(cc2, cc) and torch.equal(vv2, vv)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims", [(2, (7, 6, 5)), (3, (5, 5, 7))])
def test_allgatherv_rowblocks_equal_single_shot(oracle, tmp_path, world, dims):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dims, str(tmp_path)), nprocs=world, join=True)
    from benchmark_spgemm_using_csr_amd import gallery
    nx, ny, nz = dims
    m = nx * ny * nz
    Bp, Bj = gallery.poisson_csr("poisson27pt", nx, ny, nz)
    Bx = gallery.fill_values(len(Bj))
    Cp, Cj, Cx = oracle.spgemm(m, m, m, Bp, Bj, Bx, Bp, Bj, Bx)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["rp"], Cp.astype(np.int32))
        assert np.array_equal(z["cc"], Cj) and np.array_equal(z["vv"], Cx)


def test_row_block_partition():
    from benchmark_spgemm_using_csr_amd.dist import row_block
    for m, w in ((10, 3), (16777216, 8), (5, 8), (0, 2)):
        blocks = [row_block(m, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == m
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1
