"""The oracle's C restatement (oracle/mfs_oracle_c.c) against the golden vectors
and the numpy oracle.  CPU only."""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import cbaseline as CB
from oracle import mfs_oracle as O


@pytest.mark.parametrize("name", golden_names("p3d_"))
def test_c_apply_and_cg_vs_golden(name):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    out = np.full(gres, 7.0)
    CB.apply(gres, np.ascontiguousarray(g["rv"]), out, g["wx"], g["wy"], g["wz"], g["lphi"])
    np.testing.assert_allclose(out, g["qr"], rtol=1e-13, atol=1e-13 * np.abs(g["qr"]).max())
    res = CB.cg(gres, g["b"], g["lphi"], g["wx"], g["wy"], g["wz"], float(g["tol"]), int(np.prod(gres)), 4096)
    assert res["converged"]
    h, hg = res["history"], g["history"]
    n = min(21, len(h), len(hg))
    np.testing.assert_allclose(h[:n], hg[:n], rtol=1e-10)           # leading window (rounding-chaotic later)
    if "allfluid" in name:
        assert res["iterations"] == int(g["iters"])
        np.testing.assert_allclose(h, hg, rtol=1e-9)
        np.testing.assert_allclose(res["x"], g["x"], rtol=0, atol=1e-12 * np.abs(g["x"]).max())
    else:
        assert abs(res["iterations"] - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
        np.testing.assert_allclose(res["x"], g["x"], rtol=0, atol=1e-4 * np.abs(g["x"]).max())


def test_c_cg_fixed_iterations_matches_numpy_oracle():
    g = golden("p3d_a_12")
    gres = tuple(int(v) for v in g["gres"])
    res = CB.cg(gres, g["b"], g["lphi"], g["wx"], g["wy"], g["wz"], 0.0, 8, 64)
    x, d, r, q = (np.zeros(gres) for _ in range(4))
    hist = []
    ap = lambda V, Q: O.pressure_apply3d(gres, V[0], Q[0], g["wx"], g["wy"], g["wz"], g["lphi"])  # noqa: E731
    O.cg(ap, g["b"], x, d, r, q, 0.0, 8, hist, raise_on_fail=False)
    assert res["iterations"] == 8 and not res["converged"]
    np.testing.assert_allclose(res["history"], np.array(hist), rtol=1e-11)
    np.testing.assert_allclose(res["x"], x, rtol=0, atol=1e-12 * np.abs(x).max())
    assert CB.threads() >= 1


@pytest.mark.parametrize("name", golden_names("v3d_"))
def test_c_viscosity_apply_and_cg_vs_golden(name):
    """the C restatement of the viscosity operator and CG loop against the executed-reference goldens"""
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    cell_vol = float(np.prod(g["bound_size"] / g["gres"]))
    scale, mu = float(g["dt"]) / cell_vol / float(g["rho"]), float(g["mu"])
    vol = g["lvol"] / (cell_vol * 0.125)
    q = [np.full(g[k].shape, 7.0) for k in ("qx", "qy", "qz")]
    CB.visc_apply(gres, scale, mu, g["ex"], g["ey"], g["ez"], *q, g["sphi"], vol)
    for got, k in zip(q, ("qx", "qy", "qz")):
        np.testing.assert_allclose(got, g[k], rtol=1e-13, atol=1e-13 * np.abs(g[k]).max())   # incl. the untouched 7s
    b = np.concatenate([g[k].ravel() for k in ("bx", "by", "bz")])
    x0 = np.concatenate([g[k].ravel() for k in ("ex", "ey", "ez")])
    res = CB.visc_cg(gres, scale, mu, b, x0, g["sphi"], vol, float(g["tol"]), int(np.prod(gres)), 4096)
    assert res["converged"]
    h, hg = res["history"], g["history"]
    n = min(21, len(h), len(hg))
    np.testing.assert_allclose(h[:n], hg[:n], rtol=1e-10)
    assert abs(res["iterations"] - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
    xg = np.concatenate([g[k].ravel() for k in ("x_x", "x_y", "x_z")])
    # converged field: 1e-4 of the field maximum, as everywhere (the history is rounding-chaotic past its leading window);
    # the ill-conditioned mu = 50 scene: 1e-3 -- its converged field moves by 3.6e-4 of its maximum when only the rounding of
    # this very oracle changes (tests/golden/envelope_v3d_c_16_mu50.npz: x_dev_max).  Round 2 asserted 1e-4 there and passed
    # or failed with the number of OpenMP threads.
    np.testing.assert_allclose(res["x"], xg, rtol=0, atol=(1e-3 if "mu50" in name else 1e-4) * np.abs(xg).max())
