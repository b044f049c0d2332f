#!/usr/bin/env python3
"""Benchmark of the SpGEMM hot path (C = A^2, fp64, CSR) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one bhsparse::spgemm() over the synthetic matrix, inputs already resident in HBM: for the stencil
workloads here that is the row-class pipeline (classify the rows of B and A -> class patterns -> scan -> numeric,
all of it inside the step, nothing kept between steps; `general_path` in the output line is the same multiply on the
general pipeline: upper bound + binning -> symbolic -> scan -> numeric, every per-dataset shortcut off); at N > 1 each
rank multiplies its row block of A by the replicated B and the step ends with the
RCCL all-gatherv that assembles the full CSR of C on every rank.

Workloads (BASELINE.json `configs`; values = gallery.fill_values, seed 20140519):
  p27_weak (default) poisson27pt, 128^3 rows per GPU: 128^3 at N=1 (configs[2]),
                     128x128x256 at N=2, 128x256x256 at N=4, 256^3 at N=8 (configs[4])
  p27_128 / p27_160 / p27_256 / p5_1024   fixed-size variants (strong scaling at N > 1)
Prints ONE JSON line on rank 0.  At N=1 its `cpu_baseline` object holds the CPU oracle's figure on a bounded sample
(kind "port") and, when oracle/_ref was built, `reference_opencl_same_gpu`: the reference's own OpenCL implementation
run once on the same GPU and matrix, as timed by its own spgemm() timer.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def workload_dims(name, world):
    if name == "p27_weak":
        dims = {1: (128, 128, 128), 2: (128, 128, 256), 4: (128, 256, 256), 8: (256, 256, 256)}.get(world)
        if dims is None:
            dims = (128, 128, 128 * world)
        return "poisson27pt", dims, "weak"
    table = {"p27_128": ("poisson27pt", (128, 128, 128)), "p27_160": ("poisson27pt", (160, 160, 160)),
             "p27_256": ("poisson27pt", (256, 256, 256)), "p27_51": ("poisson27pt", (51, 51, 51)), "p27_72": ("poisson27pt", (72, 72, 72)),
             "p5_1024": ("poisson5pt", (1024, 1024, 1)), "p5_256": ("poisson5pt", (256, 256, 1)),
             "p9_1024": ("poisson9pt", (1024, 1024, 1))}
    st, dims = table[name]
    return st, dims, "strong"


def gpu_state():
    """Clocks, power and temperature of GPU 0 as rocm-smi reports them (None where it cannot be asked): recorded before
    and after the timed loop so that a run-to-run difference can be told from a different state of the GPU."""
    import subprocess
    try:
        # (a clean environment: under rocprofv3 the profiler's preloaded library would initialise the GPU in the child
        # before rocm-smi's `#!/usr/bin/env python3` hop, and the GPU boxes refuse an exec after that)
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
        p = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20, env=env)
        card = next(iter(json.loads(p.stdout).values()))
        out = {}
        for key, val in card.items():
            kl = key.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or "temperature" in kl or "performance level" in kl:
                out[key] = val
        return out or None
    except Exception as e:
        return {"error": str(e)[-120:]}


def stream_copy_GBs(dev, gib=2, reps=8):
    """Bytes read + written per second by a plain device-to-device copy of `gib` GiB (torch's copy kernel): what THIS box's
    memory system streams.  The boxes of the pool run the memory-bound class kernels of one build in 1.45 to 1.69 ms while
    the VALU-bound general pipeline takes the same time on all of them (profiles/r04_box_probe.txt); this figure goes
    beside the headline so that a slow line can be told from a slow build."""
    import torch
    try:
        n = gib << 27                                      # float64 elements
        a = torch.empty(n, dtype=torch.float64, device=dev)
        b = torch.empty_like(a)
        a.fill_(1.0)
        b.copy_(a)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        del a, b
        torch.cuda.empty_cache()
        return round(2.0 * n * 8 / (ms * 1e-3) / 1e9, 1)
    except Exception as e:                                 # a diagnostic must not take the bench line down
        return {"error": str(e)[-120:]}


def device_digest(bh, m, dev):
    """The sums of oracle/make_ref_golden.digest_of over the library's C, evaluated on the device."""
    import torch
    from benchmark_spgemm_using_csr_amd.dist import device_view
    pr, pc, pv = bh.get_C_device()
    n = bh.nnzC
    Cp = device_view(pr, m + 1, torch.int32, dev)
    Cj = device_view(pc, n, torch.int32, dev)
    Cx = device_view(pv, n, torch.float64, dev)
    w = torch.arange(n, device=dev, dtype=torch.int64) % 8191 + 1
    d = {"nnzC": int(n), "sum_rowptr": int(Cp.long().sum()), "wsum_col": int((Cj.long() * w).sum()) & 0xFFFFFFFFFFFFFFFF,
         "sum_val": float(Cx.sum()), "wsum_val": float((Cx * w.double()).sum())}
    del w
    return d


def reference_opencl_leg(m, rp, col, val, nnzCt, mine=None):
    """One C = A^2 of the same matrix through oracle/_ref/ref_opencl_spgemm (the reference's SpGEMM_opencl, unmodified);
    `mine`: digests of this library's C for the same input -- compared with the digests of the reference's C."""
    import re
    import subprocess
    import tempfile
    import numpy as np
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    exe = os.path.join(ref_dir, "ref_opencl_spgemm")
    if not os.path.exists(exe):
        return None
    if nnzCt >= 2 ** 31:
        return {"skipped": "the reference counts products in int32 (bhsparse.h:367): %d overflow it" % nnzCt}
    try:
        with tempfile.TemporaryDirectory() as td:
            fin = os.path.join(td, "in.bin")
            with open(fin, "wb") as f:
                np.array([m, m, m, len(col), len(col)], np.int32).tofile(f)
                for a, dt in ((rp, np.int32), (col, np.int32), (rp, np.int32), (col, np.int32), (val, np.float64), (val, np.float64)):
                    np.ascontiguousarray(a, dt).tofile(f)
            env = dict(os.environ, AMD_OCL_BUILD_OPTIONS_APPEND="-Dinline=static")     # see oracle/make_ref_golden.py
            p = subprocess.run([exe, fin, "=" if mine else "-"], cwd=ref_dir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=240, env=env)
        mt = re.search(r"SpGEMM time: ([0-9.eE+-]+) ms\. Gflops = ([0-9.eE+-]+)", p.stdout)
        mn = re.search(r"-> nnzC=(\d+)", p.stdout)
        if p.returncode != 0 or not mt:
            return {"error": "rc %d" % p.returncode, "stdout_tail": p.stdout.strip().splitlines()[-4:]}
        res = {"ms": float(mt.group(1)), "gflops": float(mt.group(2)), "nnzC": int(mn.group(1)) if mn else None,
               "what": "SpGEMM_opencl's own timer around its spgemm() (stages 1-4, incl. its host statistics and buffer allocation)"}
        md = re.search(r"ref_opencl_digest: nnzC=(\d+) sum_rowptr=(\d+) wsum_col=(\d+) sum_val=(\S+) wsum_val=(\S+) rows_sorted=(\d)", p.stdout)
        if mine and md:
            theirs = {"nnzC": int(md.group(1)), "sum_rowptr": int(md.group(2)), "wsum_col": int(md.group(3)),
                      "sum_val": float(md.group(4)), "wsum_val": float(md.group(5))}
            res["digest"] = theirs
            res["rows_sorted_by_reference"] = bool(int(md.group(6)))
            res["hip_digest_equals_reference"] = bool(theirs == mine)
        return res
    except Exception as e:                                  # a baseline that cannot run must not take the bench line down
        return {"error": str(e)[-300:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="p27_weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-reference", action="store_true", help="N=1: skip the run of the reference's own OpenCL build (oracle/_ref) on this GPU")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gatherv (compute-only)")
    ap.add_argument("--no-extra", action="store_true", help="N=1: skip the short runs of the other BASELINE configs")
    ap.add_argument("--no-general", action="store_true", help="N=1: skip the second headline (general pipeline)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from benchmark_spgemm_using_csr_amd import gallery, facade, dist as bdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_GATHER=1: run the RCCL all-gatherv path even with one rank (self-test of the N > 1 code)
    force_gather = os.environ.get("BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    stencil, dims, scaling = workload_dims(args.workload, world)
    nx, ny, nz = dims
    m = nx * ny * nz
    # ---- synthetic inputs, generated on the device (B = A, separate buffers: main.cpp:225-232)
    Bp, Bj = gallery.poisson_csr_torch(stencil, nx, ny, nz, device=dev)
    nnzB = int(Bj.numel())
    Bx = gallery.fill_values_torch(nnzB, device=dev)
    # row blocks balanced by work (prefix of the per-row product counts); one block at N = 1
    starts = bdist.row_blocks_balanced(Bp, Bj, Bp, world) if world > 1 else [0, m]
    r0, r1 = starts[rank], starts[rank + 1]
    if world == 1:
        Ap, Aj, Ax = Bp.clone(), Bj.clone(), Bx.clone()
    else:
        # rank-local row block of A: same pattern/values as rows [r0,r1) of B
        lo, hi = int(Bp[r0].item()), int(Bp[r1].item())
        Ap = (Bp[r0:r1 + 1] - lo).contiguous()
        Aj = Bj[lo:hi].clone()
        Ax = Bx[lo:hi].clone()
    nnzA = int(Aj.numel())
    torch.cuda.synchronize()

    plats = [False] * facade.NUM_PLATFORMS
    plats[facade.BHSPARSE_HIP] = True
    bh = facade.bhsparse()
    err = bh.initPlatform(plats, device=local_rank)
    assert err == 0, facade._lib.strerror(err)
    # BENCH_LIB_OPTS=key=value,..: library options for a MEASUREMENT run (tools/prof.sh of the general pipeline); the line says so
    lib_opts = os.environ.get("BENCH_LIB_OPTS", "")
    for kv in lib_opts.split(","):
        if kv:
            k_, v_ = kv.split("=")
            assert bh.set_option(k_, int(v_)) == 0, kv
    t_setup = time.perf_counter()
    err = bh.initData_device(r1 - r0, m, m, nnzA, Ax, Ap, Aj, nnzB, Bx, Bp, Bj)
    setup_ms = (time.perf_counter() - t_setup) * 1e3     # bhs_set_data_device: row-length and sortedness scans (host-synchronous)
    assert err == 0, facade._lib.strerror(err)
    t_setup = time.perf_counter()                         # again, with the scan kernels' code objects loaded
    err = bh.initData_device(r1 - r0, m, m, nnzA, Ax, Ap, Aj, nnzB, Bx, Bp, Bj)
    setup_ms_warm = (time.perf_counter() - t_setup) * 1e3
    assert err == 0, facade._lib.strerror(err)

    gather_out = [None]
    # N > 1: the library's own all-gatherv (libbhsparse_dist.so: ncclSend / ncclRecv groups on its stream, row ranges
    # of the numeric half overlapped with the transfers).  BENCH_GATHER=torch keeps the torch.distributed version.
    native = None
    use_native = os.environ.get("BENCH_GATHER", "native") != "torch"
    sub_blocks = int(os.environ.get("BENCH_SUB_BLOCKS", "4"))
    native_fallback = [None]              # why the run left the native all-gatherv for the torch.distributed one (None: it did not)
    if (world > 1 or force_gather) and not args.no_gather and use_native:
        # every rank must end up on the same path: a rank whose communicator cannot be made takes all of them to the
        # torch.distributed all-gatherv (the N > 1 native path has never met hardware in development)
        why = None
        try:
            if os.environ.get("BENCH_NATIVE_FAIL") == "init":       # (test hook)
                raise RuntimeError("injected")
            native = bdist.NativeDist(bh, world=world, rank=rank)
            # BENCH_VALUES_ONLY=1: column indices of the other ranks' blocks rebuilt from their row classes instead of
            # received (include/bhsparse_dist.h, option "values_only"): 8 instead of 12 bytes per entry on every link
            # (opt-in: the mode has never run between two devices, and a column check that fails at N > 1 has no fallback)
            if os.environ.get("BENCH_VALUES_ONLY", "0") == "1":
                assert native.set_option("values_only", 1) == 0
        except Exception as e:
            why = "init: %s" % e
        if world > 1:
            okf = torch.tensor([0 if why else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0 and why is None:
                why = "init failed on another rank"
        if why is not None:
            if native is not None:
                native.close()
            native = None
            native_fallback[0] = why
    gather_ms = [0.0, 0.0, 0.0]
    native_totals = [0, 0]

    def leave_native(e):
        # the library's failures are collective (every rank of the call gets an error, include/bhsparse_dist.h), so
        # every rank comes here in the same step and goes on with the torch.distributed all-gatherv
        nonlocal native
        native_fallback[0] = "step: %s" % e
        try:
            native.close()
        except Exception:
            pass
        native = None
        gather_out[0] = None
        assert bh.set_output_device(None, None, 0) == 0

    def step():
        if native is not None:
            try:
                return native_step()
            except bdist.CollectiveError as e:          # every rank is here in the same step; a local failure (spgemm,
                leave_native(e)                         # out of memory, bad argument) propagates: its peers would wait in RCCL
        return torch_step()

    def native_step():
        if True:
            if gather_out[0] is None:
                # capacity: this rank's share times the world (weak scaling: equal shares) plus slack, grown on demand
                e = bh.spgemm()
                assert e == 0, facade._lib.strerror(e)
                # capacity of the assembled C: the exact total, the SAME on every rank (a rank that fell short alone
                # would leave the collective while its peers wait in it)
                tot = torch.tensor([bh.nnzC], dtype=torch.int64, device=dev)
                if world > 1:
                    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
                capn = int(tot.item()) + 1024
                gather_out[0] = (torch.empty(m + 1, dtype=torch.int32, device=dev),
                                 torch.empty(capn, dtype=torch.int32, device=dev),
                                 torch.empty(capn, dtype=torch.float64, device=dev))
            rp, cc, vv = gather_out[0]
            tq = time.perf_counter()
            if os.environ.get("BENCH_NATIVE_FAIL") == "step":       # (test hook: injected on every rank)
                raise bdist.CollectiveError("injected")
            ct, cn = native.spgemm_allgatherv(r1 - r0, m, rp, cc, vv, sub_blocks=sub_blocks)
            bh.time_ms = (time.perf_counter() - tq) * 1e3 - native.ms[2]      # multiply + overlapped part
            for i in range(3):
                gather_ms[i] += native.ms[i]
            native_totals[0], native_totals[1] = ct, cn
            if world == 1:
                bh.nnzCt, bh.nnzC = ct, cn
            return rp, cc[:cn], vv[:cn]

    def torch_step():
        e = bh.spgemm()
        if e != 0:
            raise RuntimeError("spgemm: " + facade._lib.strerror(e))
        if (world > 1 or force_gather) and not args.no_gather:
            pr, pc, pv = bh.get_C_device()
            nnz = bh.nnzC
            lr = bdist.device_view(pr, r1 - r0 + 1, torch.int32, dev)
            lc = bdist.device_view(pc, nnz, torch.int32, dev)
            lv = bdist.device_view(pv, nnz, torch.float64, dev)
            rp, cc, vv, _ = bdist.allgatherv_csr(m, lr, lc, lv, out=gather_out[0])
            gather_out[0] = (rp, cc, vv) if gather_out[0] is None or gather_out[0][1].numel() < cc.numel() else gather_out[0]
            return rp, cc, vv
        return None

    def barrier():
        if world > 1 or force_gather:
            dist.barrier()
        torch.cuda.synchronize()

    # Inside the timed region only the numeric kernels carry hipEvent pairs (kernel_stats = 2: the roofline's launch durations are
    # measured live over the timed steps, as the contract asks, at a fifth of the events -- pairs around every family are 14 events
    # a multiply, ~30 us of a 1.6 ms step); the per-family breakdown is taken over three more multiplies BEHIND the timed region.
    assert bh.set_option("kernel_stats", 2) == 0
    for _ in range(args.warmup):
        step()
    barrier()
    state_before = gpu_state() if rank == 0 else None
    kstats = {}
    stage = np.zeros(4)
    t_compute = 0.0
    step_ms = []
    # (the per-kernel timers of every step are fetched inside the loop -- one C call into an array set up beforehand -- and read
    # after it: the bookkeeping between two multiplies is this script's, not the multiply's, and took 40-50 us of Python a step)
    from benchmark_spgemm_using_csr_amd import _lib as _rawlib
    raw_stats = [(_rawlib.KernelStat * 64)() for _ in range(args.steps)]
    raw_n = [0] * args.steps
    stage_l, tcomp_l = [None] * args.steps, [0.0] * args.steps
    t0 = time.perf_counter()
    for i_ in range(args.steps):
        tc = time.perf_counter()
        full = step()
        step_ms.append((time.perf_counter() - tc) * 1e3)
        raw_n[i_] = bh.kernel_stats_raw(raw_stats[i_])
        stage_l[i_] = bh.stage_ms
        tcomp_l[i_] = bh.time_ms
    barrier()
    elapsed = time.perf_counter() - t0
    kfam = {}
    assert bh.set_option("kernel_stats", 1) == 0
    for _ in range(3):                                        # (every rank alike: the steps of an N > 1 job are collective)
        step()
        for s in bh.kernel_stats():
            if s["launches"] > 0:
                kfam[s["name"]] = kfam.get(s["name"], 0.0) + s["ms"] / 3
    barrier()
    assert bh.set_option("kernel_stats", 2) == 0
    for i_ in range(args.steps):
        for s in bh.decode_kernel_stats(raw_stats[i_], raw_n[i_]):
            d = kstats.setdefault(s["name"], {"ms": 0.0, "launches": 0, "rows": 0, "products": 0, "nnz_out": 0,
                                              "nnzA_rows": 0, "steps": 0})
            d["ms"] += s["ms"]; d["launches"] += s["launches"]; d["steps"] += 1
            for kk in ("rows", "products", "nnz_out", "nnzA_rows"):
                d[kk] = s[kk]
        stage += np.array(stage_l[i_])
        t_compute += tcomp_l[i_]
    state_after = gpu_state() if rank == 0 else None
    # ... and once UNDER LOAD (outside the timed region): rocm-smi is asked while this rank keeps multiplying for about a
    # second -- boxes of the pool differ by up to 15 % on the same build, and the clocks an idle GPU reports say nothing
    state_load = None
    if rank == 0 and world == 1 and not args.no_general:
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(s=gpu_state()))
        t_l = time.perf_counter()
        n_l = 0
        while time.perf_counter() - t_l < 0.4:            # (clocks ramp up first)
            assert bh.spgemm() == 0
            n_l += 1
        th.start()
        while th.is_alive() and time.perf_counter() - t_l < 20.0:
            assert bh.spgemm() == 0
            n_l += 1
        th.join()
        state_load = box.get("s")
        if isinstance(state_load, dict):
            state_load["multiplies_meanwhile"] = n_l
            state_load["stream_copy_GBs"] = stream_copy_GBs(dev)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if native is not None:
            nnzCt_total, nnzC_total = native_totals          # the library's size exchange already summed them
        else:
            cnt = torch.tensor([bh.nnzCt, bh.nnzC], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            nnzCt_total, nnzC_total = int(cnt[0].item()), int(cnt[1].item())
    else:
        nnzCt_total, nnzC_total = bh.nnzCt, bh.nnzC
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnzCt_total / (ms_per_step * 1e6)

    # ---- correctness of the assembled result at N > 1 (cheap digest check on rank 0)
    if full is not None and rank == 0:
        rp, cc, vv = full
        assert int(rp[-1].item()) == nnzC_total and int(rp[0].item()) == 0
        assert bool((rp[1:] >= rp[:-1]).all())
        if stencil == "poisson27pt":                     # closed form of the 27-point stencil's square
            assert nnzC_total == (5 * dims[0] - 6) * (5 * dims[1] - 6) * (5 * dims[2] - 6), (nnzC_total, dims)
        if world > 1:
            # the assembled columns (in the values-only mode the other ranks' are REBUILT here from their row classes):
            # inside the matrix and strictly ascending in every row, checked in chunks of 2^26 entries
            n_all = int(cc.numel())
            starts = rp[1:-1].long()
            for c0 in range(0, n_all, 1 << 26):
                c1 = min(n_all, c0 + (1 << 26) + 1)
                seg = cc[c0:c1]
                assert int(seg.min()) >= 0 and int(seg.max()) < m
                inc = seg[1:] > seg[:-1]
                st = starts[(starts > c0) & (starts < c1)] - c0 - 1          # a row's first column may be anything
                inc[st] = True
                assert bool(inc.all()), "assembled colIndC: a row is not strictly ascending"
                del seg, inc, st
        if world == 1:      # forced self-test: the gathered copy must equal the local result bit for bit
            if native is not None:                # ... of an ordinary multiply into the library's own arrays
                rp, cc, vv = rp.clone(), cc.clone(), vv.clone()
                assert bh.set_output_device(None, None, 0) == 0
                assert bh.spgemm() == 0
            pr, pc, pv = bh.get_C_device()
            assert torch.equal(rp, bdist.device_view(pr, m + 1, torch.int32, dev))
            assert torch.equal(cc, bdist.device_view(pc, bh.nnzC, torch.int32, dev))
            assert torch.equal(vv, bdist.device_view(pv, bh.nnzC, torch.float64, dev))

    if rank != 0:
        if native is not None:
            native.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (device time from hipEvents on the library's stream)
    kname = max(kstats, key=lambda k2: kstats[k2]["ms"]) if kstats else None
    roof = None
    if kname:
        ks = kstats[kname]
        avg_ms = ks["ms"] / max(1, ks["launches"])
        rows_k, nnzA_k, out_k = ks["rows"], ks["nnzA_rows"], ks["nnz_out"]
        if kname == "numeric_class":
            # row classes (bhs_class.hip.h): A rows (col+val) + rowPtrA, the class of every row, rowPtrB and B's VALUES
            # once (the kernel never reads colIndB), C rows written once + rowPtrC read
            alg = 12 * nnzA_k + 4 * rows_k + 4 * rows_k + 8 * nnzB + 4 * (m + 1) + 12 * out_k + 4 * rows_k
        elif kname.startswith("numeric"):
            # A rows (col+val) + their rowPtr pairs, B once (col+val+rowPtr), C rows written once + rowPtrC read
            alg = 12 * nnzA_k + 8 * rows_k + 12 * nnzB + 4 * (m + 1) + 12 * out_k + 4 * rows_k
        elif kname.startswith("symbolic"):
            alg = 4 * nnzA_k + 8 * rows_k + 4 * nnzB + 4 * (m + 1) + 4 * rows_k
        else:
            alg = 4 * nnzA + 8 * (r1 - r0) + 4 * (m + 1)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes (tools/prof.sh -> tools/hbm_traffic.py): used only when the
        # file was produced by THIS build of the device sources, otherwise null (a stale constant is worse than none)
        traffic, traffic_build = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic_build = tj.get("build")
                if traffic_build == facade._lib.source_digest():
                    traffic = tj.get(args.workload if world == 1 else "", {}).get(kname)
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "alg_bytes_per_launch": int(alg), "avg_launch_ms": round(avg_ms, 5),
                "measured_hbm_GBs": round(traffic / (avg_ms * 1e-3) / 1e9, 2) if traffic else None,
                "launches_timed": ks["launches"], "build": facade._lib.source_digest(),
                "traffic_from_build": traffic_build}
    # whole-pipeline compulsory-bytes model: read A, read B, write C once
    bytes_alg_total = (4 * (r1 - r0 + 1) + 12 * nnzA) + (4 * (m + 1) + 12 * nnzB) + (4 * (r1 - r0 + 1) + 12 * bh.nnzC)
    pipeline_frac = bytes_alg_total / (np.sum(stage) / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS

    # ---- CPU baseline: the oracle (kind "port") on a bounded row-block sample, all host cores
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        hBp, hBj, hBx = Bp.cpu().numpy(), Bj.cpu().numpy(), Bx.cpu().numpy()
        cores = oracle.max_threads()

        def run_block(nrows):
            bp = hBp[:nrows + 1]
            nn = int(bp[-1])
            tA = time.perf_counter()
            ref = oracle.spgemm(nrows, m, m, bp, hBj[:nn], hBx[:nn], hBp, hBj, hBx)
            dt = time.perf_counter() - tA
            return ref, dt, oracle.nnzCt(bp, hBj[:nn], hBp)
        probe = min(m, 1 << 16)
        _, dtp, ctp = run_block(probe)           # calibration (also pages the oracle + OpenMP team in)
        _, dtp, ctp = run_block(probe)
        rate = ctp / max(dtp, 1e-6)              # products / s with all cores on a small block
        want = int(min(m, max(probe, args.cpu_seconds * rate / max(1.0, ctp / probe))))
        ref, dts, cts = run_block(want)
        # single-thread figure on a bounded block (about cpu_seconds/5 of work)
        one = int(min(want, max(1024, args.cpu_seconds / 5 * (rate / cores) / max(1.0, ctp / probe))))
        bp1 = hBp[:one + 1]
        t1 = time.perf_counter()
        oracle.spgemm(one, m, m, bp1, hBj[:int(bp1[-1])], hBx[:int(bp1[-1])], hBp, hBj, hBx, nthreads=1)
        dt1 = time.perf_counter() - t1
        ct1 = oracle.nnzCt(bp1, hBj[:int(bp1[-1])], hBp)
        # parity of the GPU result on the same rows, in the same run
        got_rp = bh.get_rowptrC()[:want + 1]
        nn = int(got_rp[-1])
        pr, pc, pv = bh.get_C_device()
        gc = bdist.device_view(pc, bh.nnzC, torch.int32, dev)[:nn].cpu().numpy()
        gv = bdist.device_view(pv, bh.nnzC, torch.float64, dev)[:nn].cpu().numpy()
        chk = oracle.compare(ref, (got_rp, gc, gv), rel_tol=1e-6)
        cpu = {"value": round(2.0 * cts / dts / 1e9, 4), "unit": "GFLOP/s", "cores": cores, "kind": "port",
               "sample": "oracle (Gustavson+sort, OpenMP) on rows [0,%d) of A x full B: %d products in %.2f s"
                         % (want, cts, dts),
               "gpu_matches_oracle_on_sample": bool(chk["ok"]),
               "value_1thread": round(2.0 * ct1 / dt1 / 1e9, 4),
               "sample_1thread": "rows [0,%d): %d products in %.2f s" % (one, ct1, dt1)}
        if not chk["ok"]:
            cpu["mismatch"] = chk
        # Same baseline leg, second figure: the REFERENCE ITSELF -- its OpenCL branch, built unmodified into oracle/_ref
        # (oracle/Makefile) -- on this same GPU and the same matrix, timed by its own spgemm() timer ("SpGEMM time").
        # Its own process, after the timed steps above; skipped when the binary is absent.  Reported beside the CPU
        # figure, never as `value`.
        if not args.no_reference:
            cpu["reference_opencl_same_gpu"] = reference_opencl_leg(m, hBp, hBj, hBx, bh.nnzCt, mine=device_digest(bh, m, dev))

    # ---- second headline: the same multiply with every launch shortcut that rests on per-dataset row bounds
    # switched off (no row classes, no lane-first / wave-first / numeric-first: upper-bound pass, host round trip and
    # queues as for an arbitrary matrix).  The shortcuts are verified on the device inside the timed multiply (a refuted bound
    # re-runs the general pipeline), but their choice comes from bhs_set_data's scans -- `setup_ms` -- so both
    # figures are printed.
    general = None
    if world == 1 and not args.no_general:
        headline_digest = device_digest(bh, m, dev)
        for key in ("class_path", "wave_first", "lane_first", "direct_bins"):
            assert bh.set_option(key, 0) == 0
        for _ in range(2):
            assert bh.spgemm() == 0
        torch.cuda.synchronize()
        tg = []
        for _ in range(max(3, args.steps // 2)):
            tq = time.perf_counter()
            assert bh.spgemm() == 0
            tg.append((time.perf_counter() - tq) * 1e3)
        general = {"options": "class_path=0 wave_first=0 lane_first=0 direct_bins=0", "ms_median": round(float(np.median(tg)), 4),
                   "ms_min": round(float(np.min(tg)), 4), "gflops_median": round(2.0 * bh.nnzCt / (float(np.median(tg)) * 1e6), 2),
                   # the general pipeline's C against the headline's (row classes), array by array as digests
                   "digest_equals_headline": device_digest(bh, m, dev) == headline_digest,
                   "kernels_ms": {s["name"]: round(s["ms"], 4) for s in bh.kernel_stats() if s["launches"] > 0}}
        for key in ("class_path", "wave_first", "lane_first", "direct_bins"):
            assert bh.set_option(key, 1) == 0

    # ---- what a caller pays who multiplies each data set ONCE (bhs_set_data_device's scans -- longest rows, period hint,
    # sortedness of B: the choices of the direct launches and of the class path rest on them -- plus the multiply), and
    # the spread of the steady-state figure over fresh handles in this same process
    incl_setup, fresh = None, None
    if world == 1 and not args.no_general:
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            tq = time.perf_counter()
            assert bh.initData_device(r1 - r0, m, m, nnzA, Ax, Ap, Aj, nnzB, Bx, Bp, Bj) == 0
            assert bh.spgemm() == 0
            ts.append((time.perf_counter() - tq) * 1e3)
        incl_setup = {"ms_median": round(float(np.median(ts)), 4), "ms_min": round(float(np.min(ts)), 4),
                      "what": "bhs_set_data_device + bhs_spgemm on a warm handle, same arrays"}
        meds, firsts = [], []
        for _ in range(3):
            b2 = facade.bhsparse()
            assert b2.initPlatform(plats, device=local_rank) == 0
            assert b2.set_option("kernel_stats", 0) == 0
            torch.cuda.synchronize()
            tq = time.perf_counter()
            assert b2.initData_device(r1 - r0, m, m, nnzA, Ax, Ap, Aj, nnzB, Bx, Bp, Bj) == 0
            assert b2.spgemm() == 0
            firsts.append(round((time.perf_counter() - tq) * 1e3, 4))       # incl. the handle's workspace allocations
            for _ in range(3):
                assert b2.spgemm() == 0
            tt = []
            for _ in range(max(10, args.steps)):
                tq = time.perf_counter()
                assert b2.spgemm() == 0
                tt.append((time.perf_counter() - tq) * 1e3)
            meds.append(round(float(np.median(tt)), 4))
            b2.free_mem()
            b2.freePlatform()
        fresh = {"handles": len(meds), "ms_median_each": meds, "ms_min": min(meds), "ms_median": float(np.median(meds)),
                 "ms_max": max(meds), "first_multiply_ms_each": firsts,
                 "what": "new handle each: set_data + first multiply (allocations), 3 warm-ups, median of the timed multiplies; kernel_stats off"}

    # ---- the other single-GPU configurations of BASELINE.json, short runs, reported beside the headline
    extra = None
    if world == 1 and not args.no_extra and args.workload == "p27_weak":
        extra = {}
        bh.free_mem()
        del Ap, Aj, Ax, Bp, Bj, Bx
        torch.cuda.empty_cache()
        clean_ms = ms_per_step
        for wname in ("p5_1024", "p9_1024", "p27_160", "p27_128", "p27_128_perturbed", "p27_128_perturbed_1pct", "p27_128_one_long_row", "fem3_40",
                      "fem3_40_one_long_row",
                      "weblike_1m", "rmat_s20", "p27_256_block_1of8", "p27_256"):
            k2 = None                                   # (rows of B, where A is not the whole of it)
            if wname.startswith("p27_128_"):
                # Round 6 (mixed mode of the class path): the headline matrix with irregular rows -- 0.1 % / 1 % of its rows given
                # one extra random entry (B = A: every row of A that points at such a row of B is irregular too, 2.6 % / 23.5 %
                # of the rows), one row replaced by 300 consecutive entries.  vs_clean: this multiply / the headline's.
                st2, d2 = "poisson27pt + irregular rows (gallery.perturb_rows_csr)", (128, 128, 128)
                rp0, col0 = gallery.poisson_csr("poisson27pt", 128, 128, 128)
                mm = len(rp0) - 1
                if wname == "p27_128_one_long_row":
                    rpw, colw = gallery.perturb_rows_csr(rp0, col0, mm, 0.0, long_row=(mm // 2 + 5, 300))
                else:
                    rpw, colw = gallery.perturb_rows_csr(rp0, col0, mm, 0.01 if wname.endswith("1pct") else 0.001, seed=11)
                bp2, bj2 = torch.from_numpy(rpw).to(dev), torch.from_numpy(colw).to(dev)
                del rp0, col0, rpw, colw
            elif wname == "p27_256_block_1of8":
                # BASELINE configs[4] seen from one rank (N = 1 anchor): the fourth of eight row blocks of
                # poisson27pt 256^3 as A, the whole matrix as B -- no gather; what a GPU of the 8-GPU job multiplies per step
                st2, d2 = "poisson27pt, rows [3/8, 4/8) of A against the whole of B", (256, 256, 256)
                bp2, bj2 = gallery.poisson_csr_torch("poisson27pt", 256, 256, 256, device=dev)
                k2 = int(bp2.numel()) - 1
            elif wname == "rmat_s20":     # a social-network graph (R-MAT, 2^20 rows): rows of C of thousands of entries by the ten thousand (bhs_row_window.hip.h)
                st2, d2 = "R-MAT graph (gallery.rmat_csr)", (1 << 20,)
                rpw, colw = gallery.rmat_csr()
                bp2, bj2 = torch.from_numpy(np.asarray(rpw)).to(dev), torch.from_numpy(np.asarray(colw)).to(dev)
            elif wname == "weblike_1m":   # stand-in for configs[3] (SuiteSparse webbase-1M is not in the image): same size, compression 1.35
                st2, d2 = "web-like power-law (gallery.weblike_csr)", (1000005,)
                rpw, colw = gallery.weblike_csr()
                bp2, bj2 = torch.from_numpy(rpw).to(dev), torch.from_numpy(colw).to(dev)
            elif wname in ("fem3_40", "fem3_40_one_long_row"):      # 3 unknowns per node on poisson27pt 40^3, every coupling a full block
                st2, d2 = "poisson27pt (x) ones(3,3)", (40, 40, 40)
                rp0, col0 = gallery.poisson_csr("poisson27pt", 40, 40, 40)
                rp3, col3 = gallery.block_expand_csr(rp0, col0, 3)
                if wname.endswith("one_long_row"):     # (round 6: mixed mode with the big-class kernel; vs_clean against fem3_40)
                    st2 = "poisson27pt (x) ones(3,3) + one 500-entry row"
                    rp3, col3 = gallery.perturb_rows_csr(rp3, col3, len(rp3) - 1, 0.0, long_row=((len(rp3) - 1) // 2 + 1, 500))
                bp2, bj2 = torch.from_numpy(rp3).to(dev), torch.from_numpy(col3).to(dev)
            else:
                st2, d2, _ = workload_dims(wname, 1)
                bp2, bj2 = gallery.poisson_csr_torch(st2, *d2, device=dev)
            bx2 = gallery.fill_values_torch(int(bj2.numel()), device=dev)
            if k2 is not None:
                q0, q1 = 3 * (k2 // 8), 4 * (k2 // 8)
                e0, e1 = int(bp2[q0]), int(bp2[q1])
                ap2, aj2, ax2 = (bp2[q0:q1 + 1] - e0).to(torch.int32).contiguous(), bj2[e0:e1].contiguous(), bx2[e0:e1].contiguous()
                m2 = q1 - q0
            else:
                ap2, aj2, ax2 = bp2.clone(), bj2.clone(), bx2.clone()
                m2 = k2 = int(bp2.numel()) - 1
            assert bh.initData_device(m2, k2, k2, int(aj2.numel()), ax2, ap2, aj2, int(bj2.numel()), bx2, bp2, bj2) == 0
            assert bh.set_option("kernel_stats", 0) == 0      # (library default; the headline keeps them for the roofline)
            for _ in range(3):
                assert bh.spgemm() == 0
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(10):
                assert bh.spgemm() == 0
            msq = (time.perf_counter() - tq) / 10 * 1e3
            nb_touched = int(bj2.numel())
            if wname == "p27_256_block_1of8":        # (the rows of B this block's columns reach: the block and a grid plane + line + 1 either side)
                halo = 256 * 256 + 256 + 1
                nb_touched = int(bp2[min(k2, q1 + halo)]) - int(bp2[max(0, q0 - halo)])
            balg = (4 * (m2 + 1) + 12 * int(aj2.numel())) + (4 * (min(k2, m2 + 2 * (256 * 256 + 257)) + 1) + 12 * nb_touched) + 4 * (m2 + 1) + 12 * bh.nnzC
            extra[wname] = {"workload": "%s %s C=A^2" % (st2, "x".join(map(str, d2))), "ms_per_step": round(msq, 4),
                            "gflops": round(2.0 * bh.nnzCt / (msq * 1e6), 2), "nnzCt": bh.nnzCt, "nnzC": bh.nnzC,
                            "pipeline_frac_of_hbm_peak": round(balg / (msq * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            # the kernel families of this configuration (hipEvent pairs around each, three more multiplies outside the figure above)
            assert bh.set_option("kernel_stats", 1) == 0
            kacc = {}
            for _ in range(3):
                assert bh.spgemm() == 0
                for s_ in bh.kernel_stats():
                    if s_["launches"] > 0:
                        kacc[s_["name"]] = kacc.get(s_["name"], 0.0) + s_["ms"] / 3
            extra[wname]["kernels_ms_per_step"] = {k_: round(v_, 4) for k_, v_ in sorted(kacc.items()) if v_ >= 0.0005}
            if wname == "fem3_40":
                fem_ms = msq
            if wname == "fem3_40_one_long_row":
                extra[wname]["vs_clean"] = round(msq / fem_ms, 4)
                extra[wname]["irregular_rows"] = int(bh.get_info("mixed_rows"))
                extra[wname]["class_state"] = int(bh.get_info("class_state"))
            if wname == "p27_128":
                clean_ms = msq                                      # (the headline's matrix timed the way these short runs are: kernel_stats off)
            if wname.startswith("p27_128_"):
                extra[wname]["vs_clean"] = round(msq / clean_ms, 4)
                extra[wname]["irregular_rows"] = int(bh.get_info("mixed_rows"))
                extra[wname]["class_state"] = int(bh.get_info("class_state"))
            bh.free_mem()
            del bp2, bj2, bx2, ap2, aj2, ax2
            torch.cuda.empty_cache()

    out = {
        "metric": "spgemm_gflops (2*nnz_intermediate/t, C=A^2, fp64 CSR)",
        "value": round(gflops, 3), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s %s C=A^2 (%s)" % (stencil, "x".join(map(str, dims)), args.workload),
                   "m": m, "nnzA_total": nnzB, "nnzCt": nnzCt_total, "nnzC": nnzC_total,
                   "parallelism": "rowblock%d+allgatherv" % world if world > 1 else "single",
                   "values": "1+lcg%9 seed 20140519", "gather_in_step": bool((world > 1 or force_gather) and not args.no_gather),
                   "gather": ("native ncclSend/ncclRecv groups, %d row ranges overlapped" % sub_blocks) if native is not None
                             else ("torch batch_isend_irecv" if (world > 1 or force_gather) and not args.no_gather else None),
                   "row_blocks": "balanced by products" if world > 1 else "single",
                   "library_options": lib_opts or None},
        "ms_min": round(float(np.min(step_ms)), 4), "ms_median": round(float(np.median(step_ms)), 4),
        "setup_ms": round(setup_ms, 4), "setup_ms_warm": round(setup_ms_warm, 4),
        "ms_per_step_incl_setup": incl_setup, "fresh_handles": fresh, "gpu_state": {"before": state_before, "after": state_after, "under_load": state_load},
        "general_path": general,
        "nnzC_per_s": round(nnzC_total / (ms_per_step * 1e-3), 1),
        "device_ms_per_step": round(float(np.sum(stage)) / args.steps, 4),
        "stage_ms": [round(float(x) / args.steps, 4) for x in stage],
        "host_ms_per_step_spgemm": round(t_compute / args.steps, 4),
        "gather_ms_per_step": round(ms_per_step - t_compute / args.steps, 4) if (world > 1 or force_gather) else 0.0,
        "gather_link_floor_ms": round(native.link_floor_ms(), 4) if native is not None else None,
        "nranks_seen": native.nranks() if native is not None else None,
        "gather_values_only": native.values_only_used() if native is not None else None,
        "native_fallback": native_fallback[0],
        "native_ms_per_step": [round(x / args.steps, 4) for x in gather_ms] if native is not None else None,
        "compute_only_gflops": round(2.0 * nnzCt_total / (t_compute / args.steps * 1e6), 3),
        # which figure scales with N: `value` is end to end -- every rank ends up with the WHOLE of C, over xGMI links that
        # move a block slower than a GPU multiplies it (gather_link_floor_ms) --, `compute_only_gflops` is the multiplies alone
        "scaling_note": ("north_star's >= 6x at 8 GPUs can hold for compute_only_gflops only: value includes the all-gatherv, "
                         "bounded below by gather_link_floor_ms per step") if world > 1 else None,
        "pipeline_compulsory_bytes": int(bytes_alg_total),
        "pipeline_frac_of_hbm_peak": round(float(pipeline_frac), 5),
        # (numeric kernels: the timed steps' own events; the other families: three multiplies in front of the timed region)
        "kernels_ms_per_step": dict({k2: round(v2, 4) for k2, v2 in sorted(kfam.items())},
                                    **{k2: round(v["ms"] / max(1, v["steps"]), 4) for k2, v in sorted(kstats.items()) if v["ms"] > 0}),
        "kernel_events_in_timed_region": "numeric kernels only (kernel_stats = 2); every family over 3 untimed multiplies behind it",
        "roofline": roof, "cpu_baseline": cpu, "additional_configs": extra,
    }
    print(json.dumps(out))
    if native is not None:
        native.close()
    if world > 1 or force_gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
