"""The RCCL all-gatherv path on real hardware (it had never executed on a GPU).

This module sorts first on purpose: the launcher below is started before anything in this pytest process has
touched the GPU, as one fresh child process tree (`python -m torch.distributed.run`, one rank), backend "nccl"
(== RCCL), world_size 1.  The rank runs bench.py with BENCH_FORCE_GATHER=1: every step is spgemm() followed by
dist.allgatherv_csr() on the library's device buffers (size exchange with all_gather_into_tensor on RCCL, local
block copied in place), and bench.py asserts that the gathered CSR equals the local result bit for bit.  A second
run drives the C-ABI entry bhs_allgatherv_csr (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd issued by the
library on its own stream) through tests/driver/spgemm -gpus 1.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench_rank(workload, extra_env=None):
    env = dict(os.environ, BENCH_FORCE_GATHER="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--workload", workload, "--no-cpu-baseline", "--no-extra"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("gather", ["native", "torch"])
@pytest.mark.parametrize("workload", ["p27_51", "p5_256"])
def test_rccl_allgatherv_world_size_1(workload, gather):
    out = _bench_rank(workload, {"BENCH_GATHER": gather})
    assert out["config"]["gather_in_step"] is True and out["n_gpus"] == 1
    assert out["config"]["gather"].startswith(gather)
    assert out["gather_ms_per_step"] > 0.0
    if workload == "p27_51":
        assert out["config"]["nnzCt"] == (9 * 51 - 10) ** 3 and out["config"]["nnzC"] == (5 * 51 - 6) ** 3
    if gather == "native":
        # the values-only mode (bench.py asks for it): taken where the multiply went by row classes (poisson27pt 51^3 is
        # below the class path's work threshold: ordinary mode), and the gathered copy equals the local result either way
        assert out["gather_values_only"] in (True, False)


def test_rccl_allgatherv_values_only_world_size_1():
    """The same through the values-only mode on an input that takes the row classes (poisson27pt 72^3): the library
    decides for it from the sizes exchange, copies its classes and tables into place, and the assembled C still equals
    the local result (world 1: no peers whose columns would be rebuilt -- that kernel is covered by
    test_columns_rebuilt_from_row_classes, the plan by tests/test_dist_cpu.py)."""
    out = _bench_rank("p27_72", {"BENCH_GATHER": "native", "BENCH_VALUES_ONLY": "1"})
    assert out["config"]["gather"].startswith("native") and out["gather_values_only"] is True
    assert out["config"]["nnzC"] == (5 * 72 - 6) ** 3
    out = _bench_rank("p27_72", {"BENCH_GATHER": "native", "BENCH_VALUES_ONLY": "0"})
    assert out["gather_values_only"] is False and out["config"]["nnzC"] == (5 * 72 - 6) ** 3
