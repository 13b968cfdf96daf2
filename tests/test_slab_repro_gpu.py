"""Reproducibility of the window (p2p) slab loop under load: 4 ranks share the one GPU of the box on bench.py's 4-GPU
weak-scaling problem (130 x 512 x 256 cells per rank, fp32 state, the middle ranks exchange with both neighbours).  Each
loop runs twice from the same start: the histories must agree bit for bit within a loop, and to summation-order level
between the window loop and the collective loop.  (This is the test that exposed a missing wait state after the
hand-written 128-bit window store: its payload was occasionally overwritten before the store had read it.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_window_loop_is_reproducible_under_load(dtype):
    env = dict(os.environ, MFS_BENCH_SHARED_GPU="1", MFS_P2P_TIMEOUT_MS="20000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "tools", "slab_repro.py"), dtype, "5"]
    p = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    assert p.returncode == 0, p.stdout[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["world"] == 4 and out["grid"] == [512, 512, 256]
    assert out["rccl_vs_rccl"] == 0.0 and out["p2p_vs_p2p"] == 0.0, out
    assert out["p2p_vs_rccl"] < 1e-12, out


def test_slab_viscosity_solve_is_reproducible_under_load():
    """the whole slab viscosity solve (3 ranks sharing the GPU, 96^3, fp32 state) twice per transport: bit-identical within
    a transport, summation-order level between the window and the collective transport"""
    env = dict(os.environ, MFS_BENCH_SHARED_GPU="1", MFS_P2P_TIMEOUT_MS="20000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "tools", "vslab_repro.py"), "96", "f32"]
    p = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    assert p.returncode == 0, p.stdout[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["p2p_identical"] and out["rccl_identical"], out
    assert len(set(out["iterations"])) == 1 and out["p2p_vs_rccl_first_entries"] < 1e-12, out
