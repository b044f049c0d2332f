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


def test_balanced_row_blocks_power_law_and_stencil():
    """Work-balanced partition: equal products per block within one row's worth, equal rows (+-1 % at
    the grid boundary) for a stencil, and the C helper of libbhsparse_dist.so applies the same rule."""
    import ctypes as C
    from benchmark_spgemm_using_csr_amd import gallery, _lib
    from benchmark_spgemm_using_csr_amd.dist import row_blocks_balanced
    rp, col = gallery.powerlaw_csr(20000, 20000, 90000, 2500, hubs=4)
    lens = np.diff(rp).astype(np.int64)
    ub = np.add.reduceat(lens[col], rp[:-1][np.diff(rp) > 0]) if len(col) else np.zeros(0)
    work = np.full(len(rp) - 1, 8, np.int64)
    work[np.diff(rp) > 0] += ub
    for world in (2, 3, 8):
        st = row_blocks_balanced(rp, col, rp, world)
        assert st[0] == 0 and st[-1] == len(rp) - 1 and all(a <= b for a, b in zip(st, st[1:]))
        per = [work[a:b].sum() for a, b in zip(st, st[1:])]
        assert max(per) - min(per) <= 2 * work.max() + 8, per
        rows = [b - a for a, b in zip(st, st[1:])]
        assert max(rows) > 3 * min(rows)                  # a power-law matrix: equal work is far from equal rows
    # the C entry point of the multi-GPU library, same rule
    so = os.path.join(_lib.CSRC, "libbhsparse_dist.so")
    if os.path.exists(so):
        L = C.CDLL(so)
        out = (C.c_int * 9)()
        i32 = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(C.c_void_p)
        rp32, col32 = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(col, np.int32)
        assert L.bhs_dist_partition_rows(len(rp) - 1, i32(rp32), i32(col32), i32(rp32), 8, out) == 0
        assert list(out) == row_blocks_balanced(rp, col, rp, 8)
    # stencil: the equal-rows split up to boundary effects
    rp, col = gallery.poisson_csr("poisson27pt", 16, 16, 16)
    st = row_blocks_balanced(rp, col, rp, 4)
    assert all(abs((b - a) - 1024) <= 102 for a, b in zip(st, st[1:])), st      # boundary planes carry fewer products


def _worker_uneven(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from benchmark_spgemm_using_csr_amd import gallery, dist as bdist
        from oracle import oracle
        m = 3000
        rp, col = gallery.powerlaw_csr(m, m, 12000, 700, hubs=2)
        val = gallery.fill_values(len(col))
        st = bdist.row_blocks_balanced(rp, col, rp, world)
        r0, r1 = st[rank], st[rank + 1]
        lo, hi = rp[r0], rp[r1]
        Ap = (rp[r0:r1 + 1] - lo).astype(np.int32)
        Cp, Cj, Cx = oracle.spgemm(r1 - r0, m, m, Ap, col[lo:hi], val[lo:hi], rp, col, val, nthreads=2)
        out = bdist.allgatherv_csr(m, torch.from_numpy(Cp.astype(np.int32)), torch.from_numpy(Cj), torch.from_numpy(Cx))
        np.savez(os.path.join(out_dir, "u%d.npz" % rank), rp=out[0].numpy(), cc=out[1].numpy(), vv=out[2].numpy(),
                 rows=np.array(out[3]["rows"]))
    finally:
        dist.destroy_process_group()


def test_allgatherv_uneven_power_law_blocks_world3(oracle, tmp_path):
    """world_size 3, work-balanced (hence very uneven) row blocks of a power-law matrix: the assembled CSR equals the
    single-shot product on every rank."""
    port = _free_port()
    mp.spawn(_worker_uneven, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    from benchmark_spgemm_using_csr_amd import gallery
    m = 3000
    rp, col = gallery.powerlaw_csr(m, m, 12000, 700, hubs=2)
    val = gallery.fill_values(len(col))
    Cp, Cj, Cx = oracle.spgemm(m, m, m, rp, col, val, rp, col, val)
    for r in range(3):
        z = np.load(os.path.join(str(tmp_path), "u%d.npz" % r))
        assert np.array_equal(z["rp"], Cp.astype(np.int32)) and np.array_equal(z["cc"], Cj) and np.array_equal(z["vv"], Cx)
        assert z["rows"].sum() == m and z["rows"].max() > 2 * z["rows"].min()


@pytest.mark.parametrize("world,S", [(2, 1), (2, 4), (3, 3), (5, 4), (8, 4), (8, 16)])
def test_native_allgatherv_plans_match_across_ranks(world, S):
    """The native all-gatherv's point-to-point plan (bhs_dist_plan: what bhs_dist_spgemm_allgatherv issues inside its
    ncclGroupStart / End pairs) replayed for ALL ranks of a job on the CPU: between any two ranks the k-th send meets the
    k-th receive with the same array, size and group; every rank's receives plus its own block tile the assembled
    arrays exactly once; empty blocks and empty sub-blocks are skipped on both sides."""
    import ctypes as C
    from benchmark_spgemm_using_csr_amd import _lib
    so = os.path.join(_lib.CSRC, "libbhsparse_dist.so")
    if not os.path.exists(so):
        pytest.skip("libbhsparse_dist.so not built")
    L = C.CDLL(so)
    rng = np.random.default_rng(100 * world + S)
    for trial in range(6):
        rows = rng.integers(0, 50, world).astype(np.int64)
        if trial == 0:
            rows[rng.integers(0, world)] = 0                       # a rank without rows
        cuts = np.zeros((world, S + 1), np.int64)
        for r in range(world):
            per = rng.integers(0, 40, S) * (rows[r] > 0)
            if trial == 1:
                per[rng.integers(0, S)] = 0                        # an empty sub-block
            cuts[r, 1:] = np.cumsum(per)
        row_off = np.concatenate([[0], np.cumsum(rows)])
        nnz_off = np.concatenate([[0], np.cumsum(cuts[:, -1])])
        plans = []
        for r in range(world):
            buf = (C.c_int64 * (6 * 6 * S * world))()
            n = L.bhs_dist_plan(world, r, S, rows.ctypes.data_as(C.c_void_p), cuts.ctypes.data_as(C.c_void_p), buf,
                                6 * S * world)
            assert 0 <= n <= 6 * S * world
            plans.append(np.array(buf[:6 * n], np.int64).reshape(n, 6))
        for x in range(world):
            for y in range(world):
                if x == y:
                    continue
                sends = [tuple(op[2:]) for op in plans[x] if op[0] == 0 and op[1] == y]
                recvs = [tuple(op[2:]) for op in plans[y] if op[0] == 1 and op[1] == x]
                assert sends == recvs, (x, y, sends, recvs)        # same arrays, offsets, counts, groups, same order
        for r in range(world):
            for array, total, own in ((0, nnz_off[-1], (nnz_off[r], nnz_off[r + 1])),
                                      (1, nnz_off[-1], (nnz_off[r], nnz_off[r + 1])),
                                      (2, row_off[-1], (row_off[r], row_off[r + 1]))):
                cover = np.zeros(int(total), np.int32)
                cover[own[0]:own[1]] += 1
                for op in plans[r]:
                    if op[0] == 1 and op[2] == array:
                        cover[op[3]:op[3] + op[4]] += 1
                assert (cover == 1).all(), (r, array)
            groups = plans[r][:, 5]
            assert (np.diff(groups) >= 0).all()                    # issued group by group


@pytest.mark.parametrize("world,S", [(2, 1), (3, 3), (8, 4)])
def test_native_allgatherv_values_only_plans(world, S):
    """The plan of the values-only mode (include/bhsparse_dist.h, option "values_only"), replayed for all ranks: no
    column-index operation at all; values, row pointers and row classes tile their arrays exactly once per rank; every
    rank receives the class tables of every other rank that has rows, once, at that rank's place; sends meet receives
    in order.  (The columns themselves are rebuilt by bhs_expand_class_columns_device: tests/test_parity_gpu.py.)"""
    import ctypes as C
    from benchmark_spgemm_using_csr_amd import _lib
    so = os.path.join(_lib.CSRC, "libbhsparse_dist.so")
    if not os.path.exists(so):
        pytest.skip("libbhsparse_dist.so not built")
    L = C.CDLL(so)
    L.bhs_dist_plan_values_only.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    T = 4096 * 4 + 4096 * 512
    rng = np.random.default_rng(7 * world + S)
    for trial in range(4):
        rows = rng.integers(1, 50, world).astype(np.int64)
        if trial == 0:
            rows[rng.integers(0, world)] = 0
        cuts = np.zeros((world, S + 1), np.int64)
        for r in range(world):
            cuts[r, 1:] = np.cumsum(rng.integers(0, 40, S) * (rows[r] > 0))
        row_off = np.concatenate([[0], np.cumsum(rows)])
        nnz_off = np.concatenate([[0], np.cumsum(cuts[:, -1])])
        plans = []
        for r in range(world):
            cap = 10 * S * world
            buf = (C.c_int64 * (6 * cap))()
            n = L.bhs_dist_plan_values_only(world, r, S, rows.ctypes.data_as(C.c_void_p), cuts.ctypes.data_as(C.c_void_p), T, buf, cap)
            assert 0 <= n <= cap
            plans.append(np.array(buf[:6 * n], np.int64).reshape(n, 6))
            assert not (plans[-1][:, 2] == 0).any()                  # no column indices on the wire
        for x in range(world):
            for y in range(world):
                if x != y:
                    sends = [tuple(op[2:]) for op in plans[x] if op[0] == 0 and op[1] == y]
                    recvs = [tuple(op[2:]) for op in plans[y] if op[0] == 1 and op[1] == x]
                    assert sends == recvs, (x, y)
        for r in range(world):
            for array, total, own in ((1, nnz_off[-1], (nnz_off[r], nnz_off[r + 1])), (2, row_off[-1], (row_off[r], row_off[r + 1])),
                                      (3, row_off[-1], (row_off[r], row_off[r + 1]))):
                cover = np.zeros(int(total), np.int32)
                cover[own[0]:own[1]] += 1
                for op in plans[r]:
                    if op[0] == 1 and op[2] == array:
                        cover[op[3]:op[3] + op[4]] += 1
                assert (cover == 1).all(), (r, array)
            got = sorted(int(op[3]) // T for op in plans[r] if op[0] == 1 and op[2] == 4)
            assert got == [q for q in range(world) if q != r and rows[q] > 0]
            assert all(op[4] == T and op[3] % T == 0 for op in plans[r] if op[2] == 4)
