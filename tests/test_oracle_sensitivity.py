"""Evidence for the windowed CG-history tolerance used by the GPU parity tests.

Plain CG on the reference's ghost-fluid pressure operator is chaotic in rounding:
changing NOTHING but the summation order of the two dot products (a 1-ulp-level
perturbation) makes the oracle's residual history leave its own baseline by more
than 1e-3 within the solve, while the leading iterations and the converged field
stay put.  Entry-wise history parity is therefore only meaningful over a leading
window; the converged solution is compared as a field.  CPU only."""
import functools

import numpy as np

from mfs import scenes
from oracle import mfs_oracle as O


def _revdot(A, B):
    return float(sum(np.sum((a * b).ravel()[::-1]) for a, b in zip(A, B)))


def _solve(gres, sc, perturbed):
    s = O.PressureCGSolver3D(gres, sc["bound_size"])
    v = [sc["vx"].copy(), sc["vy"].copy(), sc["vz"].copy()]
    orig = O.cg
    if perturbed:
        O.cg = functools.partial(orig, dot=_revdot)
    try:
        s.solve(*v, sc["sphi"], sc["sv"], sc["lphi"])
    finally:
        O.cg = orig
    return np.array(s.history), s.x, v


def test_oracle_history_is_rounding_chaotic_but_window_and_field_are_stable():
    gres = (16, 16, 16)
    sc = scenes.pressure_scene_3d(gres, seed=0)
    h0, x0, v0 = _solve(gres, sc, False)
    h1, x1, v1 = _solve(gres, sc, True)
    n = min(len(h0), len(h1))
    rel = np.abs(h0[:n] - h1[:n]) / np.abs(h0[:n])
    assert rel[:21].max() < 1e-9            # leading 10 iterations: stable
    assert rel.max() > 1e-3                 # later: an ulp-level change is amplified to O(1e-3..1)
    assert abs(len(h0) - len(h1)) <= 0.1 * len(h0)
    assert np.abs(x0 - x1).max() <= 1e-4 * np.abs(x0).max()
    for a, b in zip(v0, v1):
        assert np.abs(a.astype(np.float64) - b).max() <= 1e-4 * np.abs(a).max()


def test_all_fluid_history_is_stable():
    gres = (12, 12, 12)
    sc = scenes.pressure_scene_3d(gres, seed=8, vel_dtype=np.float64, all_fluid=True)
    h0, x0, _ = _solve(gres, sc, False)
    h1, x1, _ = _solve(gres, sc, True)
    assert len(h0) == len(h1)
    np.testing.assert_allclose(h1, h0, rtol=1e-11)
    np.testing.assert_allclose(x1, x0, rtol=0, atol=1e-13 * np.abs(x0).max())
