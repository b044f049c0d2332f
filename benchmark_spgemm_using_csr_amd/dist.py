"""Multi-GPU layer: rows of A (and therefore of C) shard across the GPUs of one
node, B is replicated, every rank runs the single-GPU pipeline on its row
block, and one all-gatherv over RCCL/xGMI assembles the full CSR of C on every
rank (north_star; the reference is single-device: device 0 is hard-coded at
bhsparse_cuda.h:100-101).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm; "gloo"
on CPU tensors for the world_size-2 tests).  The only data-path collective is
the all-gatherv: message sizes differ per rank (nnz of each row block), so it
is issued as one group of point-to-point transfers (ncclGroupStart / ncclSend /
ncclRecv / ncclGroupEnd under torch's batch_isend_irecv): every GPU sends its
block straight to each of its 7 peers over the fully connected xGMI links,
after a 16-byte all_gather of the per-rank sizes.
"""
import torch
import torch.distributed as dist


def row_block(m, rank, world):
    """Contiguous row range [r0, r1) of rank `rank`: equal row counts (for the
    stencil matrices equal rows == equal work)."""
    base, rem = divmod(m, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def row_blocks_balanced(rowptr_a, col_a, rowptr_b, world):
    """Contiguous row blocks balanced by WORK: starts[r]..starts[r+1] are rank r's rows, chosen on
    the prefix of the per-row product counts (the upper bound compute_nnzCt produces, bhsparse_cuda.h:210-237) plus a
    constant per row.  Equal rows for stencils, very unequal rows for power-law matrices.  Accepts numpy arrays or
    torch tensors (any device); returns a list of world + 1 ints.  Same rule as bhs_dist_partition_rows."""
    t = torch.as_tensor
    ap, aj, bp = t(rowptr_a).long(), t(col_a).long(), t(rowptr_b).long()
    m = ap.numel() - 1
    if m <= 0:
        return [0] * (world + 1)
    lens_b = bp[1:] - bp[:-1]
    w = torch.full((m,), 8, dtype=torch.int64, device=ap.device)
    if aj.numel():
        rows = torch.repeat_interleave(torch.arange(m, device=ap.device), ap[1:] - ap[:-1])
        w.index_add_(0, rows, lens_b[aj])
    pre = torch.cat([torch.zeros(1, dtype=torch.int64, device=ap.device), torch.cumsum(w, 0)])
    total = int(pre[-1])
    starts = [0]
    for r in range(1, world):
        target = total // world * r + (total % world) * r // world
        lo = int(torch.searchsorted(pre, torch.tensor([target], dtype=torch.int64, device=ap.device))[0])
        starts.append(min(max(lo, starts[-1]), m))
    starts.append(m)
    return starts


class CollectiveError(RuntimeError):
    """A failure of the library's all-gatherv that EVERY rank of the call returns from (so every rank may react to it
    in the same step); anything else raised on the way is local to one rank and must not be answered with a collective."""


class NativeDist(object):
    """ctypes binding of libbhsparse_dist.so (include/bhsparse_dist.h): the multiply of this rank's row block and the
    RCCL all-gatherv of C issued by the library itself (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on its own
    stream, overlapped with the numeric kernels of the next row range).  `bh` is the rank's facade.bhsparse with
    initData_device already done; the communicator is bootstrapped through torch.distributed (only to hand rank 0's
    unique id to the other ranks) or alone when world == 1."""

    def __init__(self, bh, world=1, rank=0, group=None):
        import ctypes as C
        import os
        from . import _lib
        path = os.path.join(_lib.CSRC, "libbhsparse_dist.so")
        if not os.path.exists(path):
            raise ImportError("%s is not built: make -C %s" % (path, _lib.CSRC))
        self._C = C
        self._L = C.CDLL(path)
        L = self._L
        vp, i64 = C.c_void_p, C.c_int64
        L.bhs_dist_unique_id.argtypes = [C.c_char_p]
        L.bhs_dist_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_char_p]
        L.bhs_dist_destroy.argtypes = [vp]
        L.bhs_dist_spgemm_allgatherv.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, C.POINTER(i64),
                                                 C.POINTER(i64), C.POINTER(C.c_double)]
        L.bhs_dist_nranks.argtypes = [vp, C.POINTER(C.c_int)]
        L.bhs_dist_nranks.restype = C.c_int
        L.bhs_dist_last_link_floor_ms.argtypes = [vp]
        L.bhs_dist_last_link_floor_ms.restype = C.c_double
        L.bhs_dist_set_option.argtypes = [vp, C.c_char_p, i64]
        L.bhs_dist_last_values_only.argtypes = [vp]
        idbuf = C.create_string_buffer(128)
        if rank == 0:
            err = L.bhs_dist_unique_id(idbuf)
            if err:
                raise RuntimeError("bhs_dist_unique_id: %d" % err)
        if world > 1:
            dev = torch.device("cuda", torch.cuda.current_device())
            tid = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).to(dev)
            dist.broadcast(tid, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
            idbuf = C.create_string_buffer(bytes(tid.cpu().numpy().tobytes()), 128)
        h = vp()
        err = L.bhs_dist_create(C.byref(h), bh._h, world, rank, idbuf)
        if err:
            raise RuntimeError("bhs_dist_create: %d" % err)
        self._d, self.world, self.rank = h, world, rank
        self.ms = (0.0, 0.0, 0.0)

    def spgemm_allgatherv(self, m_local, m_total, rowptr, col, val, sub_blocks=4):
        """rowptr int32[m_total+1], col int32[cap], val float64[cap]: device tensors that receive the assembled C.
        Returns (nnzCt_total, nnzC_total)."""
        C = self._C
        ct, cc = C.c_int64(0), C.c_int64(0)
        ms = (C.c_double * 3)()
        err = self._L.bhs_dist_spgemm_allgatherv(self._d, m_local, m_total, sub_blocks, C.c_void_p(rowptr.data_ptr()),
                                                 C.c_void_p(col.data_ptr()), C.c_void_p(val.data_ptr()),
                                                 int(col.numel()), C.byref(ct), C.byref(cc), ms)
        if err:
            from . import _lib
            msg = "bhs_dist_spgemm_allgatherv: %d (%s)" % (err, _lib.strerror(err))
            # BHS_ERR_INVALID_ARG is returned before the first agreement (this rank only); every other code is
            # reached by all ranks of the call together (include/bhsparse_dist.h)
            raise (RuntimeError if err == _lib.BHS_ERR_INVALID_ARG else CollectiveError)(msg)
        self.ms = tuple(ms)
        return int(ct.value), int(cc.value)

    def set_option(self, key, value):
        """bhs_dist_set_option: "values_only" 1 -- where every rank's multiply went by row classes, the other ranks'
        column indices are rebuilt from their classes instead of received (8 instead of 12 bytes per entry and link)."""
        return self._L.bhs_dist_set_option(self._d, key.encode(), int(value))

    def values_only_used(self):
        return bool(self._L.bhs_dist_last_values_only(self._d))

    def nranks(self):
        """Ranks RCCL counts in the communicator (ncclCommCount)."""
        n = self._C.c_int(0)
        err = self._L.bhs_dist_nranks(self._d, self._C.byref(n))
        return int(n.value) if err == 0 else -1

    def link_floor_ms(self):
        return float(self._L.bhs_dist_last_link_floor_ms(self._d))

    def close(self):
        if self._d:
            self._L.bhs_dist_destroy(self._d)
            self._d = None


def _all_gatherv(parts, group):
    """parts = [(full, local, sizes, offsets), ...]: for every part,
    full[offsets[r]:offsets[r]+sizes[r]] <- rank r's `local`, on every rank.

    Issued as ONE group of point-to-point transfers (every rank sends its blocks straight to every
    peer and receives every peer's blocks in place): on RCCL this is ncclGroupStart / ncclSend /
    ncclRecv / ncclGroupEnd, i.e. 7 simultaneous direct xGMI transfers per GPU on a fully connected
    node, with no staging copy and no padding for the uneven block sizes.  gloo runs the same ops."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    for full, local, sizes, offsets in parts:
        if sizes[rank]:
            full[offsets[rank]:offsets[rank] + sizes[rank]].copy_(local[:sizes[rank]])
    if world == 1:
        return
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    ops = []
    for step in range(1, world):                       # staggered peers: rank r talks to r+step / r-step
        dst = (rank + step) % world
        src = (rank - step) % world
        for full, local, sizes, offsets in parts:      # same part order on both ends of every pair
            if sizes[rank]:
                ops.append(dist.P2POp(dist.isend, local[:sizes[rank]], peer(dst), group=group))
            if sizes[src]:
                ops.append(dist.P2POp(dist.irecv, full[offsets[src]:offsets[src] + sizes[src]], peer(src),
                                      group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def allgatherv_csr(m_total, local_rowptr, local_col, local_val, group=None, out=None):
    """Assemble the global CSR of C from per-rank row blocks.

    local_rowptr: int32[m_local+1] starting at 0; local_col int32[nnz_local];
    local_val float64[nnz_local] (device tensors for RCCL, CPU tensors for gloo).
    Returns (rowptr int32[m_total+1], col, val, sizes) identical on every rank.
    `out` may carry preallocated (rowptr, col, val) buffers to reuse across steps.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_col.device
    m_local = local_rowptr.numel() - 1
    nnz_local = int(local_col.numel())
    # 1. sizes: (rows, nnz) of every rank — int64, offsets stay 64-bit until the final fix-up
    mine = torch.tensor([m_local, nnz_local], dtype=torch.int64, device=dev)
    allsz = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allsz, mine, group=group)
    allsz = allsz.view(world, 2).cpu()
    rows = [int(x) for x in allsz[:, 0]]
    nnzs = [int(x) for x in allsz[:, 1]]
    assert sum(rows) == m_total, (rows, m_total)
    nnz_total = sum(nnzs)
    if nnz_total >= 2 ** 31:
        raise OverflowError("nnz(C) = %d does not fit the int32 index_type of the bhsparse API" % nnz_total)
    row_off = [0] * world
    nnz_off = [0] * world
    for r in range(1, world):
        row_off[r] = row_off[r - 1] + rows[r - 1]
        nnz_off[r] = nnz_off[r - 1] + nnzs[r - 1]
    if out is not None and out[1].numel() >= nnz_total and out[0].numel() == m_total + 1:
        rowptr, col, val = out[0], out[1][:nnz_total], out[2][:nnz_total]
    else:
        rowptr = torch.empty(m_total + 1, dtype=torch.int32, device=dev)
        col = torch.empty(nnz_total, dtype=torch.int32, device=dev)
        val = torch.empty(nnz_total, dtype=torch.float64, device=dev)
    # 2. the all-gatherv proper (values, columns), plus row pointers rebased by the rank's nnz offset
    rebased = (local_rowptr[:m_local] + nnz_off[rank]).to(torch.int32)
    _all_gatherv([(val, local_val[:nnz_local], nnzs, nnz_off), (col, local_col[:nnz_local], nnzs, nnz_off),
                  (rowptr, rebased, rows, row_off)], group)
    rowptr[m_total] = nnz_total
    # On RCCL, Work.wait() only orders torch's current stream behind the transfers (and the local copy is an
    # ordinary asynchronous kernel): the HOST goes on.  The blocks being sent live in the library's C buffers,
    # which the next bhs_spgemm -- on the library's own stream -- overwrites, so everything queued here must be
    # over before this function returns.
    if col.is_cuda:
        torch.cuda.current_stream(dev).synchronize()
    return rowptr, col, val, {"rows": rows, "nnz": nnzs}


class _DevArray(object):
    """Zero-copy view of library-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_view(ptr, n, dtype, device):
    """torch tensor aliasing `n` elements of `dtype` at device address `ptr`."""
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    typestr = {torch.int32: "<i4", torch.float64: "<f8", torch.int64: "<i8"}[dtype]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)
