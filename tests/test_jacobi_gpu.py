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


def _jac_run(gres, dt, fuse, defer_x, iters, sc):
    import solver.PressureCGSolver3D as P
    import solver.SolidFraction3D as S
    wx = torch.zeros((gres[0] + 1, gres[1], gres[2]), dtype=dt, device=DEV)
    wy = torch.zeros((gres[0], gres[1] + 1, gres[2]), dtype=dt, device=DEV)
    wz = torch.zeros((gres[0], gres[1], gres[2] + 1), dtype=dt, device=DEV)
    S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
    b, x, d, r, q = (torch.zeros(gres, dtype=dt, device=DEV) for _ in range(5))
    P.initialize_solver(sc["cell_size"], gres, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(sc["lphi"], wx, wy, wz)
    eng.set_jacobi(True)
    eng.set_fuse(fuse)
    if defer_x is not None:
        eng.set_defer_x(defer_x)
    eng.bind(b, x, d, r, q)
    info = eng.loop_info()
    eng.begin(0.0)
    eng.iterate(iters)
    eng.finish()
    torch.cuda.synchronize()
    assert eng.poll()["iterations"] == iters
    return dict(info=info, hist=eng.history()[: 2 * iters + 1], x=x, d=d, r=r)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("gres", [(40, 36, 32), (24, 70, 16), (64, 64, 64), (9, 20, 128)], ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("defer_x", [False, True], ids=["x_in_update", "x_deferred"])
def test_fused_jacobi_loop_matches_three_launch_loop(gres, dt, defer_x):
    """The fused Jacobi loop (z = r / diag stored by the r update, d = z + beta d formed inside the stencil launch) against
    the three-launch loop that forms z where it is consumed: the same iteration, so fp64 state agrees to rounding of the
    dot products' grouping; fp32 state additionally rounds the stored z to fp32 (1e-7 relative per entry)."""
    from mfs import scenes
    sc = scenes.pressure_scene_3d(gres, seed=4, device=DEV)
    iters = 12
    a = _jac_run(gres, dt, True, defer_x, iters, sc)
    b = _jac_run(gres, dt, False, None, iters, sc)
    assert a["info"]["fused_direction_update"] and a["info"]["jacobi"] and bool(a["info"]["deferred_x_update"]) == defer_x, a["info"]
    assert not b["info"]["fused_direction_update"]
    tol = 1e-11 if dt == torch.float64 else 2e-5
    np.testing.assert_allclose(a["hist"], b["hist"], rtol=tol)
    for k in ("x", "r", "d"):
        ref = b[k].double()
        assert float((a[k].double() - ref).abs().max()) <= tol * float(ref.abs().max()), k
