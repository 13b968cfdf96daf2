"""The density-solver restatement (oracle/mfs_oracle.py, solver/DensityCGSolver3D.py) against goldens
produced by executing the reference's own source (tests/golden/make_goldens.py, d3d_*)."""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import mfs_oracle as O


@pytest.mark.parametrize("name", golden_names("d3d_"))
def test_density_functions_and_class(name):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    cs = np.asarray(g["bound_size"], np.float64) / np.asarray(gres, np.float64)
    gm, gvol = np.zeros(gres), np.zeros(gres)
    O.density_splat3d(g["bound_min"], cs, gres, g["px"], g["pm"], float(g["pvol"]), gm, gvol)
    np.testing.assert_allclose(gm, g["gm"], rtol=1e-12, atol=1e-15 * np.abs(g["gm"]).max())
    np.testing.assert_allclose(gvol, g["gvol_raw"], rtol=1e-12, atol=1e-15 * np.abs(g["gvol_raw"]).max())
    gv = g["gvol_raw"].copy()
    O.density_fix_volume3d(cs, gres, g["lvol"], gv, g["sphi"], g["lphi"], g["wx"], g["wy"], g["wz"])
    np.testing.assert_allclose(gv, g["gvol"], rtol=1e-13, atol=0)
    b = np.zeros(gres)
    O.density_rhs3d(float(g["rho0"]), float(g["dt"]), gres, cs, g["gm"], g["gvol"], g["lphi"], g["wx"], g["wy"], g["wz"], b)
    np.testing.assert_allclose(b, g["b"], rtol=1e-12, atol=1e-12 * np.abs(g["b"]).max())
    qr = np.full(gres, 7.0)
    O.density_apply3d(gres, g["rv"], qr, g["wx"], g["wy"], g["wz"], g["lphi"])
    np.testing.assert_allclose(qr, g["qr"], rtol=1e-13, atol=1e-13)
    assert (qr[0] == 7).all() and (qr[:, :, -1] == 7).all()

    s = O.DensityCGSolver3D(gres, g["bound_min"], g["bound_size"])
    px = g["px"].copy()
    s.solve(float(g["rho0"]), float(g["dt"]), px, g["pm"], float(g["pvol"]), None, None, None, g["sphi"], g["sv"],
            g["lphi"], g["lvol"], tol=float(g["tol"]))
    h = np.array(s.history)
    n = min(21, len(h), len(g["history"]))
    np.testing.assert_allclose(h[:n], g["history"][:n], rtol=1e-10)
    assert abs(s.iterations - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
    np.testing.assert_allclose(s.x, g["x"], rtol=0, atol=1e-6 * np.abs(g["x"]).max())
    for a, k in ((s.dx, "dx"), (s.dy, "dy"), (s.dz, "dz")):
        np.testing.assert_allclose(a, g[k], rtol=0, atol=1e-6 * np.abs(g[k]).max())
    np.testing.assert_allclose(px, g["out_px"], rtol=0, atol=1e-6 * np.abs(g["out_px"] - g["px"]).max() + 1e-7)


def test_density_operator_is_not_the_pressure_operator():
    """the two quirks that make it a different matrix: unit diagonal weights and the -z tap's weight"""
    g = golden(golden_names("d3d_")[0])
    gres = tuple(int(v) for v in g["gres"])
    a, b = np.zeros(gres), np.zeros(gres)
    O.density_apply3d(gres, g["rv"], a, g["wx"], g["wy"], g["wz"], g["lphi"])
    O.pressure_apply3d(gres, g["rv"], b, g["wx"], g["wy"], g["wz"], g["lphi"])
    assert np.abs(a - b).max() > 1e-3
