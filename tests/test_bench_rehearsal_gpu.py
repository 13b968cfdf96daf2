"""bench.py's N > 1 flow (window self-test, p2p-vs-collective cross-check, transport calibration, timed loop, one
JSON line from rank 0) rehearsed on ONE MI355X: two ranks share cuda:0 and bootstrap over gloo
(MFS_BENCH_SHARED_GPU=1).  The driver's own N = 2/4/8 runs are the measurement; this checks the code path."""
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


@pytest.mark.parametrize("transport,extra", [("auto", {}), ("rccl", {}),
                                             # the production-size flow builds the solve's sparse lists in the slab loops (round 3):
                                             # forced onto this small grid, so that the cross-checks against the dense phase loop run with them
                                             ("auto", {"MFS_SPARSE_MIN": "1"})], ids=["auto", "rccl", "auto-sparse-lists"])
def test_bench_two_ranks_on_one_gpu(transport, extra):
    env = dict(os.environ, MFS_BENCH_SHARED_GPU="1", MFS_P2P_TIMEOUT_MS="5000", **extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--edge", "48",
           "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--transport", transport]
    p = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]          # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["grid"] == [96, 48, 48] and out["config"]["decomposition"] == "x-slabs x2"
    assert "rehearsal" in out and out["roofline"]["bound"] == "hbm"
    if transport == "auto":
        ti = out["transport_info"]
        assert ti["p2p_selftest"] == "ok" and ti["p2p_crosscheck"] == "ok", ti
        assert ti["p2p_vs_rccl_history_dev"] < 1e-5
        assert out["config"]["transport"] in ("p2p", "rccl")
    else:
        assert out["config"]["transport"] == "rccl"


def test_bench_downgrades_when_the_window_selftest_fails():
    """what the driver's N > 1 run does if the HIP-IPC windows do not work on the real node -- here the
    self-test is MADE to fail (MFS_P2P_SELFTEST_FAIL=1): the run must fall back to the collective loop, SAY so in
    transport_info, and still print exactly one JSON line."""
    env = dict(os.environ, MFS_BENCH_SHARED_GPU="1", MFS_P2P_TIMEOUT_MS="5000", MFS_P2P_SELFTEST_FAIL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--edge", "48",
           "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--transport", "auto"]
    p = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]
    out = json.loads(lines[0])
    ti = out["transport_info"]
    assert out["config"]["transport"] == "rccl", out["config"]
    assert ti["p2p_selftest"] != "ok" and "injected" in ti["p2p_selftest"], ti
    assert "p2p_crosscheck" not in ti                      # never trusted, never cross-checked
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["steps"] == 20
