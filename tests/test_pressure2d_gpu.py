"""BASELINE config 1 (PressureCGSolver2D 64^2 synthetic RHS) on the GPU against the
golden vectors; tolerances as in test_pressure_gpu.py except the history window:
the 2D operator (true linear edge fractions) amplifies rounding faster, measured 2.1e-9
within the first 10 iterations on MI355X -> window 10 iterations at 1e-8."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def close(a, b, rtol, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b, np.float64)
    np.testing.assert_allclose(a.astype(np.float64), b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-300), err_msg=what)


@pytest.mark.parametrize("name", golden_names("p2d_"))
def test_pressure2d_vs_golden(name):
    import solver.CGSolverBuffer as B
    import solver.PressureCGSolver2D as P
    import solver.SolidFraction2D as S
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    Nx, Ny = gres
    wx = torch.full((Nx + 1, Ny), -3.0, dtype=torch.float64, device=DEV)
    wy = torch.full((Nx, Ny + 1), -3.0, dtype=torch.float64, device=DEV)
    S.compute_solid_frac(gres, T(g["sphi"]), wx, wy)
    # faces no cell writes stay untouched (x = Nx, y = Ny rows partly; reference writes x<Nx-1,y<Ny-1 cells only)
    ref_wx, ref_wy = np.full((Nx + 1, Ny), -3.0), np.full((Nx, Ny + 1), -3.0)
    ref_wx[:Nx, :Ny - 1] = g["wx"][:Nx, :Ny - 1]
    ref_wy[:Nx - 1, :Ny] = g["wy"][:Nx - 1, :Ny]
    close(wx, ref_wx, 1e-15, "wx")
    close(wy, ref_wy, 1e-15, "wy")

    b = torch.zeros(gres, dtype=torch.float64, device=DEV)
    P.initialize_solver(g["bound_size"] / g["gres"], gres, T(g["in_vx"]), T(g["in_vy"]), T(g["sphi"]), T(g["sv"]),
                        T(g["lphi"]), b, T(g["wx"]), T(g["wy"]))
    close(b, g["b"], 1e-12, "rhs")
    q = torch.zeros(gres, dtype=torch.float64, device=DEV)
    P.matvecmul(gres, T(g["b"]), q, T(g["wx"]), T(g["wy"]), T(g["lphi"]))
    close(q, g["q1"], 1e-12, "apply")

    buf = B.CGSolverBuffer(gres, precision="fp64", device=DEV)
    s = P.PressureCGSolver2D(buf, gres, g["bound_size"])
    vx, vy = T(g["in_vx"]), T(g["in_vy"])
    s.solve(vx, vy, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), tol=float(g["tol"]))
    h, hg = s.history, g["history"]
    n = min(21, len(h), len(hg))
    np.testing.assert_allclose(h[:n], hg[:n], rtol=1e-8)
    assert s.converged and abs(s.iterations - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
    close(s.x, g["x"], 1e-4, "x")
    close(vx, g["out_vx"], 1e-4, "vx")
    close(vy, g["out_vy"], 1e-4, "vy")


def test_pressure2d_does_not_raise_when_not_converged():
    import solver.CGSolverBuffer as B
    import solver.PressureCGSolver2D as P
    g = golden("p2d_b_24x20_sv")
    gres = tuple(int(v) for v in g["gres"])
    buf = B.CGSolverBuffer(gres, device=DEV)
    s = P.PressureCGSolver2D(buf, gres, g["bound_size"])
    s.max_iter = 3
    vx, vy = T(g["in_vx"]), T(g["in_vy"])
    s.solve(vx, vy, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), tol=float(g["tol"]))     # reference: no raise (Q3)
    assert s.iterations == 3 and not s.converged
    np.testing.assert_allclose(s.history, g["history"][:7], rtol=1e-9)
    assert not torch.equal(vx, T(g["in_vx"]))          # the partial pressure was applied
