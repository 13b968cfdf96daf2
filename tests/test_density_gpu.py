"""DensityCGSolver3D on the MI355X against the goldens produced by executing the
reference's solver/DensityCGSolver3D.py (tests/golden/make_goldens.py d3d_*) and against the oracle.
Tolerances: per-kernel 1e-12 (fp64; the particle splat adds with fp atomics in arbitrary order, so it is
compared at 1e-11 of the array maximum); CG history over the leading window 1e-9; converged fields 1e-6."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names, require_default_engine
from mfs.pcg import PcgEngine
from oracle import mfs_oracle as O
import solver.DensityCGSolver3D as D
from solver.CGSolverBuffer import CGSolverBuffer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), device=DEV) if dt is None else torch.as_tensor(np.ascontiguousarray(a), device=DEV).to(dt)  # noqa: E731
N = lambda t: t.detach().cpu().numpy()  # noqa: E731


@pytest.mark.parametrize("name", golden_names("d3d_"))
def test_module_functions(name):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    cs = np.asarray(g["bound_size"], np.float64) / np.asarray(gres, np.float64)
    gm, gvol = torch.zeros(gres, dtype=torch.float64, device=DEV), torch.zeros(gres, dtype=torch.float64, device=DEV)
    D.initialize_density(g["bound_min"], cs, gres, T(g["px"]), T(g["pm"]), float(g["pvol"]), gm, gvol)
    np.testing.assert_allclose(N(gm), g["gm"], rtol=0, atol=1e-11 * np.abs(g["gm"]).max())
    np.testing.assert_allclose(N(gvol), g["gvol_raw"], rtol=0, atol=1e-11 * np.abs(g["gvol_raw"]).max())
    wx, wy, wz, lphi, sphi = T(g["wx"]), T(g["wy"]), T(g["wz"]), T(g["lphi"]), T(g["sphi"])
    gv = T(g["gvol_raw"])
    D.fix_volume(cs, gres, T(g["lvol"]), gv, sphi, lphi, wx, wy, wz)
    np.testing.assert_allclose(N(gv), g["gvol"], rtol=1e-13, atol=0)
    b = torch.zeros(gres, dtype=torch.float64, device=DEV)
    D.initialize_solver(float(g["rho0"]), float(g["dt"]), gres, cs, T(g["gm"]), T(g["gvol"]), lphi, wx, wy, wz, b)
    np.testing.assert_allclose(N(b), g["b"], rtol=1e-12, atol=1e-12 * np.abs(g["b"]).max())
    qr = torch.full(gres, 7.0, dtype=torch.float64, device=DEV)
    D.matvecmul(gres, T(g["rv"]), qr, wx, wy, wz, lphi)
    np.testing.assert_allclose(N(qr), g["qr"], rtol=1e-12, atol=1e-12)
    dx, dy, dz = (torch.zeros(s, dtype=torch.float64, device=DEV) for s in (g["dx"].shape, g["dy"].shape, g["dz"].shape))
    D.compute_displacement(gres, float(g["dt"]), cs, dx, dy, dz, T(g["x"]), lphi)
    for a, k in ((dx, "dx"), (dy, "dy"), (dz, "dz")):
        np.testing.assert_allclose(N(a), g[k], rtol=1e-12, atol=1e-12 * np.abs(g[k]).max())
    # gather: the three axis passes in the reference's order, against the oracle on the same arrays
    px = T(g["px"])
    ref = g["px"].copy()
    for d, bias, ax in ((g["dx"], (0, .5, .5), 0), (g["dy"], (.5, 0, .5), 1), (g["dz"], (.5, .5, 0), 2)):
        D.apply_displacement(px, T(d), g["bound_min"], cs, bias, ax)
        O.density_advect3d(ref, d, g["bound_min"], cs, bias, ax)
    if ref.dtype == np.float64:
        np.testing.assert_allclose(N(px), ref, rtol=0, atol=1e-15)
    else:
        np.testing.assert_array_equal(N(px), ref)
    np.testing.assert_allclose(N(px), g["out_px"], rtol=0, atol=1e-6 * np.abs(g["out_px"] - g["px"]).max() + 1e-7)


@pytest.mark.parametrize("name", golden_names("d3d_"))
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_engine_operator_matches_stateless_kernel(name, dt):
    """the per-iteration kernel (coefficient form, compressed access, asymmetric -z tap) == matvecmul_kernel"""
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    wx, wy, wz, lphi = T(g["wx"]), T(g["wy"]), T(g["wz"]), T(g["lphi"])
    v = T(g["rv"], dt)
    want = torch.full(gres, 7.0, dtype=dt, device=DEV)
    D.matvecmul(gres, v, want, wx, wy, wz, lphi)
    eng = PcgEngine(gres, dt, DEV)
    eng.setup_density(lphi, wx, wy, wz)
    outs = []
    for comp in (1, 0):
        eng.set_compress(comp)
        got = torch.full(gres, 7.0, dtype=dt, device=DEV)
        eng.apply(v, got)
        outs.append(got)
        if dt == torch.float64:
            np.testing.assert_allclose(N(got), N(want), rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(N(got), g["qr"], rtol=1e-12, atol=1e-12)
        else:
            np.testing.assert_allclose(N(got), N(want), rtol=0, atol=2e-6 * float(want.abs().max()))
    assert torch.equal(outs[0], outs[1])          # compressed == dense coefficient access, bit for bit


@pytest.mark.parametrize("name", golden_names("d3d_"))
def test_class_solve_matches_reference(name):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    buf = CGSolverBuffer(gres, precision="fp64", device=DEV)
    s = D.DensityCGSolver3D(buf, gres, g["bound_min"], g["bound_size"])
    px = T(g["px"])
    s.solve(float(g["rho0"]), float(g["dt"]), px, T(g["pm"]), float(g["pvol"]), None, None, None, T(g["sphi"]), T(g["sv"]),
            T(g["lphi"]), T(g["lvol"]), tol=float(g["tol"]))
    h = s.history
    n = min(21, len(h), len(g["history"]))
    np.testing.assert_allclose(h[:n], g["history"][:n], rtol=1e-9)
    assert abs(s.iterations - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
    assert s.delta < float(g["tol"]) ** 2
    np.testing.assert_allclose(N(s.wx), g["wx"], rtol=0, atol=0)
    np.testing.assert_allclose(N(s.m), g["out_gm"], rtol=0, atol=1e-11 * np.abs(g["out_gm"]).max())
    np.testing.assert_allclose(N(s.vol), g["out_gvol"], rtol=0, atol=1e-11 * np.abs(g["out_gvol"]).max())
    np.testing.assert_allclose(N(s.x), g["x"], rtol=0, atol=1e-6 * np.abs(g["x"]).max())
    for a, k in ((s.dx, "dx"), (s.dy, "dy"), (s.dz, "dz")):
        np.testing.assert_allclose(N(a), g[k], rtol=0, atol=1e-6 * np.abs(g[k]).max())
    np.testing.assert_allclose(N(px), g["out_px"], rtol=0, atol=1e-6 * np.abs(g["out_px"] - g["px"]).max() + 1e-7)


@pytest.mark.parametrize("name", ["d3d_c_20", "d3d_a_12"])
@pytest.mark.parametrize("jacobi", [False, True], ids=["reference_cg", "jacobi"])
def test_deferred_x_update_with_the_density_operator(name, jacobi):
    """the x update deferred into the next stencil launch (its default beyond the Infinity Cache, forced here) with the
    density operator's asymmetric -z tap: the same history, solution and vectors, bit for bit, as with the update in the vector
    phase -- on the engine, from the golden's stored right-hand side (two `solve`s of the class differ in the last bits of
    their RHS: the particle splat adds with fp atomics in arbitrary order); launch-per-phase loops (the resident small-grid
    loop carries x in registers anyway)"""
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    outs = []
    for defer in (True, False):
        eng = PcgEngine(gres, torch.float64, DEV)
        eng.setup_density(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
        eng.set_resident(False)
        eng.set_defer_x(defer)
        eng.set_jacobi(jacobi)
        b = T(g["b"]).clone()
        x, d, r, q = (torch.zeros(gres, dtype=torch.float64, device=DEV) for _ in range(4))
        eng.bind(b, x, d, r, q)
        info = eng.loop_info()
        assert info["deferred_x_update"] == defer and info["jacobi"] == jacobi and not info["resident"], info
        ok, it = eng.solve(float(g["tol"]), int(np.prod(gres)), 8)
        assert ok
        outs.append((it, eng.history(), x.clone(), d.clone(), r.clone()))
    assert outs[0][0] == outs[1][0]
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for a, bb in zip(outs[0][2:], outs[1][2:]):
        assert torch.equal(a, bb)
    if not jacobi:
        np.testing.assert_allclose(N(outs[0][2]), g["x"], rtol=0, atol=1e-6 * np.abs(g["x"]).max())


def test_density_weights_feed_the_pressure_solve():
    """the notebook's hand-over (ipynb:4590 -> :4648): PressureSolver.solve(..., wx=DensitySolver.wx, ...)"""
    from solver.PressureCGSolver3D import PressureCGSolver3D
    g, p = golden("d3d_a_12"), golden("p3d_a_12")
    gres = tuple(int(v) for v in g["gres"])
    buf = CGSolverBuffer(gres, precision="fp64", device=DEV)
    ds = D.DensityCGSolver3D(buf, gres, g["bound_min"], g["bound_size"])
    ds.solve(float(g["rho0"]), float(g["dt"]), T(g["px"]), T(g["pm"]), float(g["pvol"]), None, None, None, T(g["sphi"]),
             T(g["sv"]), T(g["lphi"]), T(g["lvol"]))
    ps = PressureCGSolver3D(buf, gres, p["bound_size"])
    vx, vy, vz = T(p["in_vx"]), T(p["in_vy"]), T(p["in_vz"])
    ps.solve(vx, vy, vz, T(p["sphi"]), T(p["sv"]), T(p["lphi"]), wx=ds.wx, wy=ds.wy, wz=ds.wz)
    assert ps.iterations > 0 and ps.delta < 1e-6


@pytest.mark.parametrize("name,world,transport", [("d3d_a_12", 2, "p2p"), ("d3d_a_12", 3, "p2p"), ("d3d_b_10x12x14_f32", 2, "p2p"),
                                                  ("d3d_a_12", 2, "rccl"), ("d3d_c_20", 2, "p2p+defer_x")])
def test_slab_density_solver_matches_reference(name, world, transport, tmp_path):
    """SlabDensityCGSolver3D (replicated particles, CG loop slab-decomposed over `world` processes sharing the GPU):
    history, solution, displacements and moved particles against the goldens of the reference's own solve."""
    require_default_engine("test_slab_density_solver_matches_reference")
    from test_p2p_gpu import _run_ranks
    g = golden(name)
    extra = {}
    if transport.endswith("+defer_x"):       # the window slab loop with the x update deferred into the edge / interior launches
        transport, extra = "p2p", {"MFS_DEFER_X": "1"}
    res = _run_ranks(name, world, tmp_path, "f64", P2P_TEST_MODE="density", P2P_TEST_TRANSPORT=transport, **extra)
    for r in res:
        assert str(r["transport"]) == transport
        h = r["hist"]
        n = min(21, len(h), len(g["history"]))
        np.testing.assert_allclose(h[:n], g["history"][:n], rtol=1e-9)
        assert abs(int(r["iters"]) - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
        assert float(r["delta"]) < float(g["tol"]) ** 2
        assert not r["lq"][0].any() and not r["lq"][-1].any() and not r["lr"][0].any() and not r["lr"][-1].any()
        np.testing.assert_allclose(r["x"], g["x"], rtol=0, atol=1e-6 * np.abs(g["x"]).max())
        for k in ("dx", "dy", "dz"):
            np.testing.assert_allclose(r[k], g[k], rtol=0, atol=1e-6 * np.abs(g[k]).max())
        np.testing.assert_allclose(r["px"], g["out_px"], rtol=0, atol=1e-6 * np.abs(g["out_px"] - g["px"]).max() + 1e-7)
    for r in res[1:]:
        np.testing.assert_array_equal(r["hist"], res[0]["hist"])
        np.testing.assert_array_equal(r["x"], res[0]["x"])
