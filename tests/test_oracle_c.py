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
