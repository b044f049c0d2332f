import sys, os, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from benchmark_spgemm_using_csr_amd import gallery, facade
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dof = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rp, col = gallery.poisson_csr("poisson27pt", N, N, N)
P = sp.csr_matrix((np.ones(len(col)), col, rp), shape=(len(rp) - 1,) * 2)
A = sp.kron(P, np.ones((dof, dof)), format="csr"); A.sort_indices()
m = A.shape[0]
val = gallery.fill_values(A.nnz)
dev = torch.device("cuda", 0)
Bp, Bj, Bx = (torch.from_numpy(x).to(dev) for x in (A.indptr.astype(np.int32), A.indices.astype(np.int32), val))
Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
for opts in [dict(kv.split("=") for kv in o.split(",") if kv) for o in os.environ.get("FEM_OPTS", ";").split(";")]:
    plats = [False] * 9; plats[3] = True
    bh = facade.bhsparse(); assert bh.initPlatform(plats) == 0
    for k_, v_ in opts.items(): assert bh.set_option(k_, int(v_)) == 0
    assert bh.initData_device(m, m, m, Aj.numel(), Ax, Ap, Aj, Bj.numel(), Bx, Bp, Bj) == 0
    for _ in range(2): assert bh.spgemm() == 0
    acc = {}; st = np.zeros(4); n = 5
    for _ in range(n):
        assert bh.spgemm() == 0
        st += np.array(bh.stage_ms) / n
        for s in bh.kernel_stats(): acc[s["name"]] = acc.get(s["name"], 0) + s["ms"] / n
    print(opts, "m=%d nnzA=%d nnzCt=%d nnzC=%d" % (m, Aj.numel(), bh.nnzCt, bh.nnzC), "stages", np.round(st, 3), "total %.3f ms %.1f GFLOPs" % (st.sum(), 2 * bh.nnzCt / st.sum() / 1e6))
    print("   ", {k: round(v, 3) for k, v in acc.items() if v > 0.01})
    bh.free_mem(); bh.freePlatform()
