"""Input generators for the SpGEMM benchmark: Poisson stencil matrices and the
deterministic value fill.

Replaces the reference's use of cusp::gallery::poisson{5,9,7,27}pt
(SpGEMM_cuda/main.cu:30-53) and its `rand()%9+1` value overwrite
(main.cu:79-94; made deterministic here).  Only the *pattern*
of the gallery matrices matters because the driver overwrites the values.

Grid index = x + nx*(y + ny*z); rows are column-sorted.  Two back-ends with the
same arithmetic: numpy (host; tests) and torch (device; bench at full size).
"""
import numpy as np

SEED = 20140519

STENCILS = {
    # name: (ndim, offsets predicate)
    "poisson5pt": 2, "poisson9pt": 2, "poisson7pt": 3, "poisson27pt": 3,
}


def stencil_offsets(name):
    """Offsets (dx,dy,dz) in ascending linear-index order."""
    offs = []
    if name == "poisson5pt":
        offs = [(0, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0)]
    elif name == "poisson9pt":
        offs = [(dx, dy, 0) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    elif name == "poisson7pt":
        offs = [(0, 0, -1), (0, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    elif name == "poisson27pt":
        offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    else:
        raise ValueError("unknown stencil %r" % (name,))
    return offs


def poisson_closed_form(name, nx, ny, nz=1):
    """(m, nnzA) without building the matrix."""
    m = nx * ny * nz
    if name == "poisson5pt":
        return m, 5 * m - 2 * nx - 2 * ny
    if name == "poisson9pt":
        return m, (3 * nx - 2) * (3 * ny - 2)
    if name == "poisson7pt":
        return m, 7 * m - 2 * (nx * ny + ny * nz + nx * nz)
    if name == "poisson27pt":
        return m, (3 * nx - 2) * (3 * ny - 2) * (3 * nz - 2)
    raise ValueError(name)


def poisson_csr(name, nx, ny, nz=1, row_begin=0, row_end=None):
    """numpy CSR pattern (rowptr int32[m_local+1], col int32[nnz]) of rows
    [row_begin,row_end) of the stencil matrix (values are filled separately)."""
    offs = stencil_offsets(name)
    if STENCILS[name] == 2:
        nz = 1
    m = nx * ny * nz
    row_end = m if row_end is None else row_end
    rows = np.arange(row_begin, row_end, dtype=np.int64)
    x = rows % nx
    y = (rows // nx) % ny
    z = rows // (nx * ny)
    nloc = rows.size
    cols = np.empty((nloc, len(offs)), np.int32)
    mask = np.empty((nloc, len(offs)), bool)
    for t, (dx, dy, dz) in enumerate(offs):
        ok = ((x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) &
              (z + dz >= 0) & (z + dz < nz))
        mask[:, t] = ok
        cols[:, t] = (rows + dx + nx * (dy + ny * dz)).astype(np.int32)
    rowptr = np.zeros(nloc + 1, np.int64)
    np.cumsum(mask.sum(axis=1), out=rowptr[1:])
    assert rowptr[-1] < 2 ** 31
    return rowptr.astype(np.int32), np.ascontiguousarray(cols[mask])


def fill_values(nnz, seed=SEED, offset=0):
    """Integer-valued fp64 entries in 1..9: 1 + (lcg(seed, i) % 9), stateless per
    index so that any shard can generate its own slice.  Exact in fp64 under any
    summation order."""
    i = np.arange(offset, offset + nnz, dtype=np.uint64)
    x = (i + np.uint64(seed)) * np.uint64(6364136223846793005) + np.uint64(1442695040888963407)
    return (1 + ((x >> np.uint64(33)) % np.uint64(9))).astype(np.float64)


# ---------------------------------------------------------------- torch (device)
def poisson_csr_torch(name, nx, ny, nz=1, row_begin=0, row_end=None, device="cuda", chunk=1 << 22):
    """Same pattern as poisson_csr, generated on `device` in row chunks."""
    import torch
    offs = stencil_offsets(name)
    if STENCILS[name] == 2:
        nz = 1
    m = nx * ny * nz
    row_end = m if row_end is None else row_end
    counts, colparts = [], []
    for r0 in range(row_begin, row_end, chunk):
        r1 = min(row_end, r0 + chunk)
        rows = torch.arange(r0, r1, dtype=torch.int64, device=device)
        x = rows % nx
        y = (rows // nx) % ny
        z = rows // (nx * ny)
        cols = torch.empty((r1 - r0, len(offs)), dtype=torch.int32, device=device)
        mask = torch.empty((r1 - r0, len(offs)), dtype=torch.bool, device=device)
        for t, (dx, dy, dz) in enumerate(offs):
            mask[:, t] = ((x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) &
                          (z + dz >= 0) & (z + dz < nz))
            cols[:, t] = (rows + (dx + nx * (dy + ny * dz))).to(torch.int32)
        counts.append(mask.sum(dim=1))
        colparts.append(cols[mask])
        del rows, x, y, z, cols, mask
    cnt = torch.cat(counts) if counts else torch.zeros(0, dtype=torch.int64, device=device)
    rowptr = torch.zeros(row_end - row_begin + 1, dtype=torch.int64, device=device)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    col = torch.cat(colparts) if colparts else torch.zeros(0, dtype=torch.int32, device=device)
    return rowptr.to(torch.int32), col.contiguous()


def fill_values_torch(nnz, seed=SEED, offset=0, device="cuda"):
    """Bit-identical to fill_values (int64 wrap-around == uint64 arithmetic)."""
    import torch
    i = torch.arange(offset, offset + nnz, dtype=torch.int64, device=device)
    a = 6364136223846793005
    c = 1442695040888963407
    x = (i + seed) * a + c                      # wraps mod 2^64
    hi = (x >> 33) & 0x7FFFFFFF                 # logical shift of the uint64 pattern
    return (1 + hi % 9).to(torch.float64)


def block_expand_csr(rp, col, dof):
    """(rp, col) (x) ones(dof, dof): the pattern of a grid with `dof` unknowns per node, every coupling a full block
    (the 3-dof FEM stand-in of DESIGN.md: poisson27pt, 40^3 nodes, dof 3).  Rows and columns of node i are
    i * dof .. i * dof + dof - 1; columns ascending."""
    rp = np.asarray(rp, np.int64)
    col = np.asarray(col, np.int64)
    nnz_node = np.diff(rp)
    row0 = np.repeat(col * dof, dof) + np.tile(np.arange(dof, dtype=np.int64), len(col))   # a node's first row, all nodes
    new_len = np.repeat(nnz_node * dof, dof)                       # entries of every scalar row
    new_rp = np.zeros(len(new_len) + 1, np.int64)
    np.cumsum(new_len, out=new_rp[1:])
    seg_start = np.repeat(rp[:-1] * dof, dof)                      # where that row's copy of row0 begins
    idx = np.arange(new_rp[-1], dtype=np.int64) - np.repeat(new_rp[:-1], new_len) + np.repeat(seg_start, new_len)
    return new_rp.astype(np.int32), row0[idx].astype(np.int32)


def dense_rows_csr(n, per_row=8, dense=4, dense_nnz=200000, seed=SEED):
    """Seeded sparse square pattern (about per_row uniformly random entries per row) with `dense` rows of dense_nnz
    entries spread evenly over the matrix: in A^2 each of them carries about dense_nnz x per_row products, and every
    row that points at one inherits its dense_nnz entries -- the hub-row case the reference hands to its
    multi-round global merge (bhsparse_cuda.h:2270-2525).  Rows sorted and duplicate-free."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(max(per_row - 3, 1), per_row + 4, n).astype(np.int64)
    hub = (np.arange(dense, dtype=np.int64) * n) // max(dense, 1) + n // (2 * max(dense, 1)) if dense else np.empty(0, np.int64)
    lens[hub] = min(n, dense_nnz)
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    cols = rng.integers(0, n, rows.size)
    key = np.unique(rows * n + cols)
    r = key // n
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=rowptr[1:])
    return rowptr.astype(np.int32), (key - r * n).astype(np.int32)


def powerlaw_csr(m, n, nnz_target, max_row, seed=SEED, alpha=1.8, hubs=8, colpow=1.6):
    """Seeded power-law CSR pattern: stand-in for SuiteSparse webbase-1M when the
    file is absent.  Row lengths ~ Zipf scaled to about
    nnz_target, `hubs` rows of length ~max_row, longest rows first; columns are
    skewed toward low indices (u**colpow), i.e. toward the long rows, so that
    popular columns are also long B rows as on a web graph.  Rows sorted and
    duplicate-free.  With (m=n=1000005, nnz_target=3105536, max_row=4700) the
    defaults give nnz=3.01 M, 70.1 M products and 69.6 M entries in A^2 (webbase-1M:
    3.11 M, ~69.5 M, ~51.1 M), largest row of A^2 102 k entries."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.zipf(alpha, m), max_row).astype(np.float64)
    lens = np.minimum(max_row, np.round(lens * (nnz_target / max(1.0, lens.sum())))).astype(np.int64)
    if hubs and m:
        lens[rng.choice(m, size=min(hubs, m), replace=False)] = min(n, max_row)
    lens = lens[np.argsort(-lens, kind="stable")]
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    cols = np.minimum(n - 1, (rng.random(rows.size) ** colpow * n).astype(np.int64))
    key = np.unique(rows * n + cols)                     # sorted, duplicate-free
    r = key // n
    rowptr = np.zeros(m + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=m), out=rowptr[1:])
    return rowptr.astype(np.int32), (key - r * n).astype(np.int32)


def weblike_csr(m=1000005, max_row=4700, seed=SEED, host_alpha=1.5, max_host=3000, p_nav=0.6, nav_max=5,
                sect_alpha=1.4, sect_max=120, sect_min=5, spec_mean=0.45, local_pow=4.0,
                ndir=300, ntopics=50, dir_pow=1.2, p_portal=0.0055, portal_max=8, topic_pow=2.0, topic_width=3.0):
    """Seeded web-graph-like CSR pattern: the stand-in for SuiteSparse webbase-1M that is
    webbase-like where it counts for A^2 -- duplicate accumulation.  Pages live in hosts (contiguous index blocks,
    Zipf sizes); every page of a host carries the host's navigation template (links to its first few pages), those
    section pages list 5..120 pages of their host skewed towards its front (so the sections a page reaches overlap),
    ordinary pages add a few host-local links; `ndir` directory pages (rows of up to max_row entries) list pages of
    their topic's column block, the directories of one topic overlap, and portal pages link to several directories
    of one topic (long rows of A^2 WITH duplicates).  Rows sorted and duplicate-free, no empty rows.  Defaults:
    m = 1 000 005, nnz = 3 113 694, longest row 4729, 70.84 M products, 52.44 M entries in A^2 (compression 1.351;
    webbase-1M: 3 105 536 / 4700 / ~69.5 M / ~51.1 M = 1.36), 2 200 rows of A^2 with more than 3072 products."""
    rng = np.random.default_rng(seed)
    sizes = []; tot = 0
    while tot < m:
        s = np.minimum(rng.zipf(host_alpha, 200000), max_host); sizes.append(s); tot += int(s.sum())
    sizes = np.concatenate(sizes); cs = np.cumsum(sizes)
    nh = int(np.searchsorted(cs, m)) + 1
    sizes = sizes[:nh].astype(np.int64); sizes[-1] -= cs[nh - 1] - m
    start = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    host_of = np.repeat(np.arange(nh, dtype=np.int64), sizes)
    hs, hz = start[host_of], sizes[host_of]
    page_in_host = np.arange(m, dtype=np.int64) - hs
    nav_t = np.where((rng.random(nh) < p_nav) & (sizes > 2), np.minimum(rng.integers(1, nav_max + 1, nh), sizes - 1), 0)
    is_nav = page_in_host < nav_t[host_of]
    lens = np.round((np.minimum(rng.zipf(2.0, m), 40) - 1) * spec_mean).astype(np.int64)
    lens[(lens == 0) & (nav_t[host_of] == 0)] = 1
    sect = np.minimum(sect_max, sect_min + rng.zipf(sect_alpha, m)).astype(np.int64)
    lens[is_nav] = np.minimum(hz[is_nav], sect[is_nav])
    # directories in topics
    dirs = rng.choice(m, size=ndir, replace=False)
    dlen = np.maximum(40, (max_row / (1 + (np.arange(ndir) % ntopics) // 2) ** dir_pow)).astype(np.int64)
    topic = np.arange(ndir) % ntopics
    twidth = np.zeros(ntopics, np.int64); np.maximum.at(twidth, topic, (dlen * topic_width).astype(np.int64))
    tstart = (rng.random(ntopics) * (m - twidth)).astype(np.int64)
    is_dir = np.zeros(m, bool); is_dir[dirs] = True
    dir_id = np.full(m, -1, np.int64); dir_id[dirs] = np.arange(ndir)
    lens[dirs] = (-dlen * topic_width * np.log(1.0 - 1.0 / topic_width)).astype(np.int64)                   # draws; duplicates collapse
    is_portal = (rng.random(m) < p_portal) & ~is_dir & ~is_nav
    lens[is_portal] = rng.integers(2, portal_max + 1, int(is_portal.sum()))
    ptopic = np.minimum(ntopics - 1, (rng.random(m) ** topic_pow * ntopics).astype(np.int64))
    nav_cnt = nav_t[host_of]
    nav_rows = np.repeat(np.arange(m, dtype=np.int64), nav_cnt)
    nav_off = np.arange(nav_rows.size, dtype=np.int64) - np.repeat(np.cumsum(nav_cnt) - nav_cnt, nav_cnt)
    nav_cols = hs[nav_rows] + nav_off
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    u = rng.random(rows.size)
    rh, rz = hs[rows], hz[rows]
    cols = rh + np.minimum(rz - 1, (u ** local_pow * rz).astype(np.int64))
    # portal links: directories of the row's topic
    pr = is_portal[rows]
    per_topic = ndir // ntopics
    cols = np.where(pr, dirs[np.minimum(ndir - 1, ptopic[rows] + ntopics * (u * per_topic).astype(np.int64))], cols)
    dr = is_dir[rows]
    d = dir_id[rows]
    cols = np.where(dr, tstart[topic[np.maximum(d, 0)]] + (u * twidth[topic[np.maximum(d, 0)]]).astype(np.int64), cols)
    key = np.unique(np.concatenate((rows * m + cols, nav_rows * m + nav_cols)))
    r = key // m
    rowptr = np.zeros(m + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=m), out=rowptr[1:])
    return rowptr.astype(np.int32), (key - r * m).astype(np.int32)


# ---------------------------------------------------------------- the results table's inputs (tools/suite_table.py)
# No SuiteSparse file is in the image (no network): structurally distinct seeded stand-ins of about a million rows.
def _csr_from_pairs(m, n, rows, cols):
    key = np.unique(rows.astype(np.int64) * n + cols.astype(np.int64))
    r = key // n
    rowptr = np.zeros(m + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=m), out=rowptr[1:])
    return rowptr.astype(np.int32), (key - r * n).astype(np.int32)


def rmat_csr(scale=20, edge_factor=2, a=0.57, b=0.19, c=0.19, seed=SEED):
    """R-MAT / Kronecker graph (Graph500 parameters), 2^scale rows, ~edge_factor entries per row before duplicates
    collapse; vertices are not permuted: the hubs sit at the low indices."""
    rng = np.random.default_rng(seed)
    ne = edge_factor << scale
    rows = np.zeros(ne, np.int64)
    cols = np.zeros(ne, np.int64)
    for _ in range(scale):
        u = rng.random(ne)
        rbit = u >= a + b
        cbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
        rows = (rows << 1) | rbit
        cols = (cols << 1) | cbit
    n = 1 << scale
    return _csr_from_pairs(n, n, rows, cols)


def banded_csr(n=1 << 20, lo=3, hi=24, seed=SEED):
    """Banded matrix with an irregular bandwidth: row i holds every column within h(i) of the diagonal, h a random
    walk between lo and hi (1-D finite elements of varying order / a reordered FEM band)."""
    rng = np.random.default_rng(seed)
    h = np.clip(np.cumsum(rng.integers(-1, 2, n)) % (2 * (hi - lo)), 0, None)
    h = lo + np.where(h > hi - lo, 2 * (hi - lo) - h, h)
    left = np.maximum(0, np.arange(n) - h)
    right = np.minimum(n - 1, np.arange(n) + h)
    lens = right - left + 1
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    cols = np.arange(rowptr[-1], dtype=np.int64) - np.repeat(rowptr[:-1], lens) + np.repeat(left, lens)
    return rowptr.astype(np.int32), cols.astype(np.int32)


def mesh2d_csr(npoints=1 << 20, seed=SEED):
    """Adjacency (plus diagonal) of the Delaunay triangulation of random points in the unit square, points numbered
    along a 1024 x 1024 grid of cells: an unstructured 2-D mesh with a locality-preserving numbering (~7 per row)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.random((npoints, 2))
    cell = (np.minimum(1023, (pts[:, 1] * 1024).astype(np.int64)) << 10) | np.minimum(1023, (pts[:, 0] * 1024).astype(np.int64))
    pts = pts[np.argsort(cell, kind="stable")]
    tri = Delaunay(pts).simplices.astype(np.int64)
    e = np.concatenate((tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]))
    d = np.arange(npoints, dtype=np.int64)
    rows = np.concatenate((e[:, 0], e[:, 1], d))
    cols = np.concatenate((e[:, 1], e[:, 0], d))
    return _csr_from_pairs(npoints, npoints, rows, cols)


def blockdiag_csr(n=1 << 20, bmin=4, bmax=32, seed=SEED):
    """Block-diagonal with dense blocks of bmin..bmax rows (independent subsystems / supernodes)."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(bmin, bmax + 1, 3 * n // (bmin + bmax) + 16)
    cs = np.cumsum(sizes)
    nb = int(np.searchsorted(cs, n)) + 1
    sizes = sizes[:nb].astype(np.int64)
    sizes[-1] -= cs[nb - 1] - n
    start = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    blk = np.repeat(np.arange(nb), sizes)
    lens = sizes[blk]
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    cols = np.arange(rowptr[-1], dtype=np.int64) - np.repeat(rowptr[:-1], lens) + np.repeat(start[blk], lens)
    return rowptr.astype(np.int32), cols.astype(np.int32)


def uniform_csr(n=1 << 20, per_row=8, seed=SEED):
    """Uniformly random columns, per_row draws per row (Erdos-Renyi: no structure at all)."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n, dtype=np.int64), per_row)
    return _csr_from_pairs(n, n, rows, rng.integers(0, n, rows.size))


def roadlike_csr(nx=1024, ny=1024, keep=0.62, seed=SEED):
    """Road-network-like: the 4-neighbour grid graph with 38 % of its edges removed, symmetric, plus the diagonal
    (average 3.5 entries per row, long paths, no hubs)."""
    rng = np.random.default_rng(seed)
    idx = np.arange(nx * ny, dtype=np.int64).reshape(ny, nx)
    e = np.concatenate((np.stack((idx[:, :-1].ravel(), idx[:, 1:].ravel()), 1), np.stack((idx[:-1, :].ravel(), idx[1:, :].ravel()), 1)))
    e = e[rng.random(len(e)) < keep]
    d = np.arange(nx * ny, dtype=np.int64)
    return _csr_from_pairs(nx * ny, nx * ny, np.concatenate((e[:, 0], e[:, 1], d)), np.concatenate((e[:, 1], e[:, 0], d)))


def perturb_rows_csr(rp, col, ncols, fraction=0.001, seed=SEED, long_row=None):
    """A structured matrix with irregular rows (round 6, mixed mode of the class path): `fraction` of the rows get ONE extra
    entry at a random column they do not hold yet; long_row = (row, entries): that row is replaced by `entries` consecutive
    columns around its diagonal.  Rows stay sorted and duplicate-free (the input's rows must be).  Returns (rowPtr int32,
    colInd int32).  (Splices into the sorted arrays: two seconds for poisson27pt 128^3.)"""
    rp = np.asarray(rp, np.int64).copy()
    col = np.asarray(col, np.int32)
    m = len(rp) - 1
    rng = np.random.default_rng(seed)
    nper = int(round(m * fraction))
    rows = np.sort(rng.choice(m, nper, replace=False)) if nper > 0 else np.zeros(0, np.int64)
    newc = rng.integers(0, ncols, nper)
    if long_row is not None:
        r, L = long_row
        c0 = max(0, min(ncols - L, r - L // 2))
        col = np.concatenate([col[:rp[r]], np.arange(c0, c0 + L, dtype=np.int32), col[rp[r + 1]:]])
        rp[r + 1:] += L - (rp[r + 1] - rp[r])
        keep = rows != r
        rows, newc = rows[keep], newc[keep]
    # where each new entry goes in its row (binary search inside the row), and whether the row holds the column already
    pos = np.empty(len(rows), np.int64)
    fresh = np.ones(len(rows), bool)
    for i, (r, c) in enumerate(zip(rows, newc)):
        seg = col[rp[r]:rp[r + 1]]
        k = int(np.searchsorted(seg, c))
        pos[i] = rp[r] + k
        fresh[i] = not (k < len(seg) and seg[k] == c)
    rows, newc, pos = rows[fresh], newc[fresh], pos[fresh]
    col2 = np.insert(col, pos, newc.astype(np.int32))
    rp2 = rp + np.searchsorted(rows, np.arange(m + 1), side="left")          # (entries inserted into the rows before row i)
    return rp2.astype(np.int32), col2
