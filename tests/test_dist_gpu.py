"""The multi-GPU driver path on ONE MI355X: a 1-rank RCCL process group, the
phase-by-phase Python loop of mfs.dist.SlabCG (split interior/edge applies,
all-reduces on the device-resident scalars) against the native single-launch loop
(mfs_pcg3d_solve).  world_size>1 over gloo is covered on the CPU (test_dist_cpu)."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import golden
from mfs.dist import SlabCG, SlabPartition
from mfs.pcg import PcgEngine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pg():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["p3d_e_allfluid_12", "p3d_d_20"])
@pytest.mark.parametrize("overlap", [True, False])
def test_phase_loop_matches_native_loop(pg, name, overlap):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
    res = []
    for mode in ("native", "phases"):
        eng = PcgEngine(gres, torch.float64, DEV)
        eng.setup(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
        b, x, d, r, q = T(g["b"]), *(torch.zeros(gres, dtype=torch.float64, device=DEV) for _ in range(4))
        eng.bind(b, x, d, r, q)
        if mode == "native":
            eng.set_resident(False)     # the launch-per-phase loop (the resident small-grid loop has tests/test_resident_gpu.py)
            ok, it = eng.solve(float(g["tol"]), int(np.prod(gres)), 16)
            assert ok
        else:
            cg = SlabCG(eng, SlabPartition(gres[0], 1, 0), d, pg, overlap=overlap, force_multi=True)
            cg.begin(float(g["tol"]))
            for _ in range(400):
                cg.iterate(8)
                if eng.poll()["done"]:
                    break
            assert eng.poll()["done"]
        res.append((eng.poll(), eng.history(), x.cpu().numpy()))
    (sa, ha, xa), (sb, hb, xb) = res
    n = min(21, len(ha), len(hb))
    np.testing.assert_allclose(hb[:n], ha[:n], rtol=1e-11)
    np.testing.assert_allclose(ha[:n], g["history"][:n], rtol=1e-9)
    if "allfluid" in name:
        assert sa["iterations"] == sb["iterations"] == int(g["iters"])
        np.testing.assert_allclose(hb, ha, rtol=1e-10)
        np.testing.assert_allclose(xb, xa, rtol=0, atol=1e-12 * np.abs(xa).max())
    else:
        assert abs(sa["iterations"] - sb["iterations"]) <= max(2, sa["iterations"] // 10)
        np.testing.assert_allclose(xb, xa, rtol=0, atol=1e-4 * np.abs(xa).max())
    # extra iterations after convergence are device-side no-ops: state frozen
    assert sb["delta"] == hb[-1] and sb["delta"] < float(g["tol"]) ** 2
