"""The OPT-IN Jacobi-preconditioned pressure loop (mfs_pcg3d_set_jacobi; an extra of this build -- the reference's
CG is unpreconditioned, so there is no reference output to pin it to): the HIP loop against the oracle's
restatement of the same preconditioned iteration on the goldens' inputs, and against the plain solve's solution."""
import numpy as np
import pytest
import torch

from conftest import golden
from mfs.pcg import PcgEngine
from oracle import mfs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), device=DEV).to(dt)  # noqa: E731


@pytest.mark.parametrize("name", ["p3d_d_20", "p3d_b_10x12x14_sv", "p3d_a_12"])
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_jacobi_loop_matches_its_oracle(name, dt):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    tol = float(g["tol"])
    diag = O.pressure_diag3d(gres, g["wx"], g["wy"], g["wz"], g["lphi"])
    x, d, r, q = (np.zeros(gres) for _ in range(4))
    hist = []
    ap = lambda V, Q: O.pressure_apply3d(gres, V[0], Q[0], g["wx"], g["wy"], g["wz"], g["lphi"])  # noqa: E731
    it_ref, _ = O.cg_jacobi(ap, diag, g["b"], x, d, r, q, tol, np.prod(gres), hist)
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
    eng.set_jacobi(True)
    bx = [T(g["b"], dt)] + [torch.zeros(gres, dtype=dt, device=DEV) for _ in range(4)]
    eng.bind(*bx)
    ok, it = eng.solve(tol, int(np.prod(gres)), 16)
    assert ok
    h = eng.history()
    n = min(17, len(h), len(hist))
    np.testing.assert_allclose(h[:n], hist[:n], rtol=1e-9 if dt == torch.float64 else 1e-5)
    assert abs(it - it_ref) <= max(2, it_ref // 10)
    xs = bx[1].cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(xs, g["x"], rtol=0, atol=(1e-4 if dt == torch.float64 else 1e-3) * np.abs(g["x"]).max())
    if "allfluid" not in name:
        assert it < int(g["iters"])          # what the preconditioner is for
    # switching it off again restores the reference's loop
    eng.set_jacobi(False)
    ok, it2 = eng.solve(tol, int(np.prod(gres)), 16)
    assert ok
    if dt == torch.float64:
        assert abs(it2 - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
        np.testing.assert_allclose(eng.history()[:11], g["history"][:11], rtol=1e-9)
