"""Parity of the HIP viscosity path (through the C ABI) against the golden vectors
and the oracle.  GPU only.  Tolerances as in test_pressure_gpu.py: per-kernel
1e-12 (fp64); CG history over the leading window (fp64 10 it @1e-9, fp32 8 it
@1e-5); converged fields 1e-4 of the field maximum."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names
from mfs import scenes
from oracle import mfs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=DEV)
    return t if dtype is None else t.to(dtype)


def close(a, b, rtol, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    np.testing.assert_allclose(a.astype(np.float64), b, rtol=rtol, atol=rtol * scale, err_msg=what)


def hist_window(h, hg, iters, rtol):
    n = min(2 * iters + 1, len(h), len(hg))
    np.testing.assert_allclose(np.asarray(h)[:n], np.asarray(hg)[:n], rtol=rtol)


@pytest.fixture(scope="module")
def V():
    import solver.ViscosityCGSolver3D as V
    return V


def _params(g):
    cell_vol = float(np.prod(g["bound_size"] / g["gres"]))
    return float(g["dt"]) / cell_vol / float(g["rho"]), float(g["mu"]), g["lvol"] / (cell_vol * 0.125)


@pytest.mark.parametrize("name", golden_names("v3d_"))
def test_module_functions_vs_golden(V, name):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    scale, mu, vol = _params(g)
    ex, ey, ez = (T(g[k], torch.float64) for k in ("in_vx", "in_vy", "in_vz"))
    V.extrapolate(gres, 3, ex, ey, ez, T(g["sphi"]))
    close(ex, g["ex"], 1e-13)
    close(ey, g["ey"], 1e-13)
    close(ez, g["ez"], 1e-13)
    b = [torch.full(g[k].shape, 5.0, dtype=torch.float64, device=DEV) for k in ("bx", "by", "bz")]
    V.initialize_solver(gres, scale, mu, T(g["ex"]), T(g["ey"]), T(g["ez"]), T(g["sphi"]), T(g["sv"]), T(vol), *b)
    for t, k in zip(b, ("bx", "by", "bz")):
        ref = g[k].copy()
        # the reference leaves array-boundary faces untouched (golden arrays started at 0, ours at 5)
        m = np.ones(ref.shape, bool)
        m[1:-1, 1:-1, 1:-1] = False
        ref[m] = 5.0
        close(t, ref, 1e-12, k)
    q = [torch.full(g[k].shape, 7.0, dtype=torch.float64, device=DEV) for k in ("qx", "qy", "qz")]
    V.matvecmul(gres, scale, mu, T(g["ex"]), T(g["ey"]), T(g["ez"]), *q, T(g["sphi"]), T(vol))
    for t, k in zip(q, ("qx", "qy", "qz")):
        close(t, g[k], 1e-12, k)


@pytest.mark.parametrize("name", golden_names("v3d_"))
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_engine_apply_matches_reference_operator(name, dt):
    from mfs.vcg import VcgEngine
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    scale, mu, vol = _params(g)
    eng = VcgEngine(gres, dt, DEV)
    eng.setup(scale, mu, T(g["sphi"]), T(vol))
    v, vv = eng.new_vector()
    o, ov = eng.new_vector()
    for t, k in zip(vv, ("ex", "ey", "ez")):
        t.copy_(T(g[k]))
    o.fill_(7.0)
    eng.apply(v, o)
    ref = [np.full(g[k].shape, 7.0) for k in ("qx", "qy", "qz")]
    O.visc_apply3d(gres, scale, mu, *[t.cpu().numpy().astype(np.float64) for t in vv], *ref, g["sphi"], vol)
    for t, r in zip(ov, ref):
        close(t, r, 1e-12 if dt == torch.float64 else 3e-7, "engine apply")


@pytest.mark.parametrize("name", golden_names("v3d_"))
@pytest.mark.parametrize("prec", ["fp64", "fp32"])
def test_solve_vs_golden(V, name, prec):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    s = V.ViscosityCGSolver3D(gres, g["bound_size"], precision=prec, device=DEV, check_every=5)
    vx, vy, vz = T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])
    s.solve(float(g["dt"]), float(g["mu"]), float(g["rho"]), vx, vy, vz, T(g["sphi"]), T(g["sv"]), T(g["lphi"]),
            T(g["lvol"]), tol=float(g["tol"]))
    it = int(g["iters"])
    # mu = 1 cases are well conditioned (the oracle's history is insensitive to rounding: 1e-15 under a
    # reversed summation order); the mu = 50 case is rounding-chaotic like the pressure pool scenes
    # (its oracle history moves by 0.7 under the same perturbation) -> leading window + field check
    # at the accuracy the stopping rule itself defines, plus the reference's stopping rule re-evaluated
    # with the ORACLE operator on the GPU's solution.
    stable = "mu50" not in name
    if prec == "fp64":
        hist_window(s.history, g["history"], 10, 1e-9)
        if stable:
            assert s.iterations == it
            np.testing.assert_allclose(s.history, g["history"], rtol=1e-9)
        else:
            assert abs(s.iterations - it) <= max(2, it // 10)
    else:
        hist_window(s.history, g["history"], 8, 1e-5)
        assert 0.8 * it - 2 <= s.iterations <= 1.5 * it + 2
    assert s.delta < float(g["tol"]) ** 2 and s.delta == s.history[-1]
    vscale = max(np.abs(g[k]).max() for k in ("x_x", "x_y", "x_z"))
    ftol = (1e-10 if prec == "fp64" else 2e-5) if stable else 1e-3
    for a, k in ((s.x_x, "x_x"), (s.x_y, "x_y"), (s.x_z, "x_z"), (vx, "out_vx"), (vy, "out_vy"), (vz, "out_vz")):
        np.testing.assert_allclose(a.cpu().numpy().astype(np.float64), g[k], rtol=0, atol=ftol * vscale, err_msg=k)
    if prec == "fp64":   # true residual of the GPU solution under the oracle's operator meets the stopping rule
        scale, mu, vol = _params(g)
        q = [np.zeros_like(g[k]) for k in ("bx", "by", "bz")]
        O.visc_apply3d(gres, scale, mu, *[t.cpu().numpy() for t in (s.x_x, s.x_y, s.x_z)], *q, g["sphi"], vol)
        true_delta = sum(float(((g[k] - qq) ** 2).sum()) for k, qq in zip(("bx", "by", "bz"), q))
        assert true_delta < 1.01 * float(g["tol"]) ** 2, true_delta
    for a, k in ((vx, "out_vx"), (vy, "out_vy"), (vz, "out_vz")):
        assert a.dtype == torch.as_tensor(g[k]).dtype
    close(s.vol, g["lvol"] / (float(np.prod(g["bound_size"] / g["gres"])) * 0.125), 1e-15, "vol")


def test_failed_to_converge_raises(V):
    g = golden("v3d_c_16_mu50")
    gres = tuple(int(v) for v in g["gres"])
    s = V.ViscosityCGSolver3D(gres, g["bound_size"], device=DEV)
    s.max_iter = 4
    with pytest.raises(ValueError, match="Failed to converge!"):
        s.solve(float(g["dt"]), float(g["mu"]), float(g["rho"]), T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"]),
                T(g["sphi"]), T(g["sv"]), T(g["lphi"]), T(g["lvol"]))
    assert s.iterations == 4
    np.testing.assert_allclose(s.history, g["history"][:9], rtol=1e-9)


def test_config3_128_properties():
    """BASELINE config 3 size: symmetry / positivity of the operator, A.0 = 0, determinism, and
    agreement of the compact per-iteration kernel with the doubled-grid operator."""
    import solver.ViscosityCGSolver3D as V
    from mfs.vcg import VcgEngine
    N = 128
    gres = (N, N, N)
    sc = scenes.viscosity_scene_3d(gres, seed=3, device=DEV)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / N))
    scale = sc["dt"] / cell_vol / sc["rho"]
    vol = sc["lvol"] / (cell_vol * 0.125)
    eng = VcgEngine(gres, torch.float64, DEV)
    eng.setup(scale, 50.0, sc["sphi"], vol)
    gen = torch.Generator(device=DEV).manual_seed(0)
    u, uv = eng.new_vector()
    w, wv = eng.new_vector()
    Au, Auv = eng.new_vector()
    Aw, _ = eng.new_vector()
    valid = [sc["sphi"][0::2, 1::2, 1::2] >= 0, sc["sphi"][1::2, 0::2, 1::2] >= 0, sc["sphi"][1::2, 1::2, 0::2] >= 0]
    for vec in (uv, wv):
        for t, m in zip(vec, valid):
            t[1:-1, 1:-1, 1:-1] = torch.randn(t[1:-1, 1:-1, 1:-1].shape, generator=gen, device=DEV, dtype=torch.float64)
            t.mul_(m)                     # DOFs live on non-solid interior faces only
    eng.apply(u, Au)
    eng.apply(w, Aw)
    uAw, wAu, uAu = (u * Aw).sum().item(), (w * Au).sum().item(), (u * Au).sum().item()
    assert abs(uAw - wAu) <= 1e-11 * max(abs(uAw), abs(uAu))
    assert uAu > 0
    ref = [torch.zeros_like(t) for t in Auv]
    V.matvecmul(gres, scale, 50.0, *uv, *ref, sc["sphi"], vol)
    for a, b in zip(Auv, ref):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12 * b.abs().max().item())
    Au2, _ = eng.new_vector()
    eng.apply(u, Au2)
    assert torch.equal(Au, Au2)
    z, _ = eng.new_vector()
    Az, _ = eng.new_vector()
    Az.fill_(3.0)
    eng.apply(z, Az)
    assert ((Az == 0) | (Az == 3)).all()
