"""The RCCL ("nccl") branches of scorp_amd.parallel and of the loops built on it, EXECUTED on the one GPU a test box has:
a fresh child process forms a process group of one rank with device_id=cuda:0 and runs every helper with
SINGLE_RANK_COLLECTIVES - broadcast, all_gather_into_tensor, reduce_scatter_tensor, all_reduce on device tensors - then
one data-parallel training run (dense and visibility-sparse average, statistics all-reduce in front of a densify step),
the sharded rotation sweep and the object-sharded refinement.  The exchanges are trivial, the code path is the one the
8-GPU run takes (tests/test_parallel_cpu.py covers the arithmetic on two gloo ranks)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_branches_run_on_a_single_rank_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               SCORP_SINGLE_RANK_COLLECTIVES="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "children", "rccl_single_rank.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=420)
    assert p.returncode == 0 and "RCCL_SINGLE_RANK_OK" in p.stdout, f"rc {p.returncode}\n{p.stdout[-3000:]}\n{p.stderr[-6000:]}"
