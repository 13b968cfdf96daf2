"""Failure detection of the device-resident CG loops, and the converging-iteration hazard of the
x-updating direction kernel.  GPU only.

Reference behaviour being matched ("failure detection"):
  * d.q == 0  ->  `alpha = self.delta / cp.sum(d*q).item()` raises ZeroDivisionError
    (solver/PressureCGSolver3D.py:211, solver/ViscosityCGSolver3D.py:594);
  * NaN in the inputs -> `nan < tol**2` is never true, the loop runs max_iter = prod(gres) times and
    raises ValueError("Failed to converge!") (:222-223).  The device loop stops at the first non-finite
    dot product instead (MFS_E_NONFINITE) and raises an exception that is BOTH a FloatingPointError and a
    ValueError with the reference's message -- within one `check_every`, not after 16.8 M iterations.
"""
import os
import time

import numpy as np
import pytest
import torch

from mfs import _lib, scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _pressure(gres, precision, seed=2):
    from solver.CGSolverBuffer import CGSolverBuffer
    from solver.PressureCGSolver3D import PressureCGSolver3D
    sc = scenes.pressure_scene_3d(gres, seed=seed)
    buf = CGSolverBuffer(gres, precision=precision, device=DEV)
    return sc, PressureCGSolver3D(buf, gres, sc["bound_size"])


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("gres", [(24, 20, 16), (64, 64, 64)])
def test_pressure_nan_input_stops_at_once(gres, precision):
    sc, s = _pressure(gres, precision)
    vx = sc["vx"].copy()
    vx[gres[0] // 2, gres[1] // 3, gres[2] // 2] = np.nan           # a fluid face of the pool scene
    t0 = time.perf_counter()
    with pytest.raises(FloatingPointError) as ei:
        s.solve(T(vx), T(sc["vy"]), T(sc["vz"]), T(sc["sphi"]), T(sc["sv"]), T(sc["lphi"]))
    assert time.perf_counter() - t0 < 5.0                            # max_iter = prod(gres): would be minutes
    assert isinstance(ei.value, ValueError) and "Failed to converge!" in str(ei.value)
    assert isinstance(ei.value, _lib.MfsNonFinite)
    assert s._engine.poll_raw()["iterations"] <= s.check_every


def test_pressure_zero_dq_raises_zero_division():
    """zero RHS with tol = 0: delta0 = 0 is not < 0, the loop is entered, d = 0, d.q = 0 -> the reference divides by
    a Python float 0.0"""
    gres = (16, 16, 16)
    sc, s = _pressure(gres, "fp64")
    z = lambda a: T(np.zeros_like(a))  # noqa: E731
    with pytest.raises(ZeroDivisionError):
        s.solve(z(sc["vx"]), z(sc["vy"]), z(sc["vz"]), T(sc["sphi"]), z(sc["sv"]), T(sc["lphi"]), tol=0.0)
    assert s._engine.poll_raw()["iterations"] == 1


def test_pressure_all_solid_scene_returns():
    """no fluid cell at all: b = 0, delta0 = 0 < tol^2 -> the reference skips its loop; so do we (no error)."""
    gres = (16, 12, 8)
    sc, s = _pressure(gres, "fp64")
    sphi = -np.ones_like(sc["sphi"])
    lphi = np.ones_like(sc["lphi"])
    vx, vy, vz = T(sc["vx"]), T(sc["vy"]), T(sc["vz"])
    s.solve(vx, vy, vz, T(sphi), T(sc["sv"]), T(lphi))
    assert s.iterations == 0 and s.delta == 0.0


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_viscosity_nan_input_stops_at_once(precision):
    from solver.ViscosityCGSolver3D import ViscosityCGSolver3D
    gres = (24, 24, 24)
    sc = scenes.viscosity_scene_3d(gres, seed=3)
    s = ViscosityCGSolver3D(gres, sc["bound_size"], precision=precision, device=DEV)
    vx = sc["vx"].copy()
    vx[12, 14, 12] = np.nan                                           # inside the fluid block
    t0 = time.perf_counter()
    with pytest.raises(FloatingPointError) as ei:
        s.solve(sc["dt"], sc["mu"], sc["rho"], T(vx), T(sc["vy"]), T(sc["vz"]), T(sc["sphi"]), T(sc["sv"]),
                T(sc["lphi"]), T(sc["lvol"]))
    assert time.perf_counter() - t0 < 5.0
    assert isinstance(ei.value, ValueError)


def test_pressure2d_nan_input_stops():
    """the 2D solver returns silently on non-convergence (PressureCGSolver2D.py:165-177) -- but a NaN must not make
    it spin for prod(gres) iterations either"""
    from solver.CGSolverBuffer import CGSolverBuffer
    from solver.PressureCGSolver2D import PressureCGSolver2D
    gres = (32, 32)
    sc = scenes.pressure_scene_2d(gres, 1)
    buf = CGSolverBuffer(gres, precision="fp64", device=DEV)
    s = PressureCGSolver2D(buf, gres, sc["bound_size"])
    vx = sc["vx"].copy()
    vx[16, 16] = np.nan
    with pytest.raises(FloatingPointError):
        s.solve(T(vx), T(sc["vy"]), T(sc["sphi"]), T(sc["sv"]), T(sc["lphi"]))


def _visc_solve(gres, sc, env):
    """one fp64-state viscosity solve with the engine created under `env` (the knobs are read at creation)"""
    from solver.ViscosityCGSolver3D import ViscosityCGSolver3D
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = ViscosityCGSolver3D(gres, sc["bound_size"], precision="fp64", device=DEV)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    s.solve(sc["dt"], sc["mu"], sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"])
    torch.cuda.synchronize()
    return s, (vx, vy, vz)


def test_final_x_update_survives_late_blocks():
    """in the default viscosity loop `x += alpha d` rides in the direction-update kernel, whose
    bookkeeping thread raises `done` on the converging iteration IN THE SAME LAUNCH; a block that started after that
    store used to return at its top and skip its stripes of the last update.  With 32 blocks per CU the grid (8 k
    blocks at 128^3 fp64) is four times what is resident, so most blocks start late.  The converged x must equal,
    bit for bit, that of the loop whose x update runs in its own phase (split_x = 0) on the same grid."""
    gres = (128, 128, 128)
    sc = scenes.viscosity_scene_3d(gres, seed=3, device=DEV)
    for bpc in ("32", "8"):
        ref, vref = _visc_solve(gres, sc, {"MFS_VISC_SPLIT_X": "0", "MFS_VEC_BLOCKS_PER_CU": bpc})
        new, vnew = _visc_solve(gres, sc, {"MFS_VISC_SPLIT_X": "1", "MFS_VEC_BLOCKS_PER_CU": bpc})
        assert ref.iterations == new.iterations and ref.iterations > 3, (ref.iterations, new.iterations)
        np.testing.assert_array_equal(ref.history, new.history)
        for c in "xyz":
            assert torch.equal(getattr(ref, "x_" + c), getattr(new, "x_" + c)), (bpc, c)
        for va, vb in zip(vref, vnew):
            assert torch.equal(va, vb), bpc
        del ref, new, vref, vnew
        torch.cuda.empty_cache()
