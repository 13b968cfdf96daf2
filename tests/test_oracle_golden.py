"""The numpy oracle (oracle/mfs_oracle.py) against the golden vectors produced by
executing the reference's own source (tests/golden/make_goldens.py).  CPU only.

Tolerances: the oracle and the golden run perform the same fp64 operations in
the same per-cell order; only numpy's pairwise `sum` vs the golden run's
`np.sum` (identical here) could differ, so element-wise results are required
to agree to 1e-13 relative and CG histories to 1e-10 relative.
"""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import mfs_oracle as O

RT = 1e-13


def close(a, b, rtol=RT, atol=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=max(atol, rtol * scale))


def test_fraction_tables():
    g = golden("fractions")
    v = g["vals"]
    A, B = np.meshgrid(v, v, indexing="ij")
    np.testing.assert_array_equal(O.edge_in_fraction(A, B), g["edge"])
    A, B, C = np.meshgrid(v, v, v, indexing="ij")
    np.testing.assert_array_equal(O.tri_in_fraction(A, B, C), g["tri"])
    assert set(np.unique(g["tri"])) <= {0.0, 1.0}
    q = g["quads"]
    np.testing.assert_array_equal(O.face_in_fraction(q[:, 0], q[:, 1], q[:, 2], q[:, 3]), g["face"])


@pytest.mark.parametrize("name", golden_names("p3d_"))
def test_pressure3d_pieces(name):
    g = golden(name)
    gres = tuple(int(x) for x in g["gres"])
    Nx, Ny, Nz = gres
    wx, wy, wz = np.zeros((Nx + 1, Ny, Nz)), np.zeros((Nx, Ny + 1, Nz)), np.zeros((Nx, Ny, Nz + 1))
    O.compute_solid_frac3d(gres, g["sphi"], wx, wy, wz)
    np.testing.assert_array_equal(wx, g["wx"])
    np.testing.assert_array_equal(wy, g["wy"])
    np.testing.assert_array_equal(wz, g["wz"])
    assert not wx[Nx].any() and not wy[:, Ny].any() and not wz[:, :, Nz].any()   # Q8
    cs = g["bound_size"] / g["gres"]
    b = np.zeros(gres)
    O.pressure_rhs3d(cs, gres, g["in_vx"], g["in_vy"], g["in_vz"], g["sphi"], g["sv"], g["lphi"], b, wx, wy, wz)
    close(b, g["b"])
    q1 = np.zeros(gres)
    O.pressure_apply3d(gres, g["b"], q1, wx, wy, wz, g["lphi"])
    close(q1, g["q1"])
    qr = np.full(gres, 7.0)
    O.pressure_apply3d(gres, g["rv"], qr, wx, wy, wz, g["lphi"])
    close(qr, g["qr"])
    assert (qr[0] == 7).all() and (qr[:, -1] == 7).all() and (qr[:, :, 0] == 7).all()   # Q6


@pytest.mark.parametrize("name", golden_names("p3d_"))
def test_pressure3d_solve(name):
    g = golden(name)
    gres = tuple(int(x) for x in g["gres"])
    s = O.PressureCGSolver3D(gres, g["bound_size"])
    vx, vy, vz = g["in_vx"].copy(), g["in_vy"].copy(), g["in_vz"].copy()
    s.solve(vx, vy, vz, g["sphi"], g["sv"], g["lphi"], tol=float(g["tol"]))
    assert s.iterations == int(g["iters"])
    close(np.array(s.history), g["history"], rtol=1e-10)
    close(s.x, g["x"], rtol=1e-10)
    for a, b in ((vx, g["out_vx"]), (vy, g["out_vy"]), (vz, g["out_vz"])):
        assert a.dtype == b.dtype
        close(a, b, rtol=1e-6 if a.dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("name", golden_names("p2d_"))
def test_pressure2d(name):
    g = golden(name)
    gres = tuple(int(x) for x in g["gres"])
    Nx, Ny = gres
    wx, wy = np.zeros((Nx + 1, Ny)), np.zeros((Nx, Ny + 1))
    O.compute_solid_frac2d(gres, g["sphi"], wx, wy)
    np.testing.assert_array_equal(wx, g["wx"])
    np.testing.assert_array_equal(wy, g["wy"])
    b = np.zeros(gres)
    O.pressure_rhs2d(g["bound_size"] / g["gres"], gres, g["in_vx"], g["in_vy"], g["sphi"], g["sv"],
                     g["lphi"], b, wx, wy)
    close(b, g["b"])
    q1 = np.zeros(gres)
    O.pressure_apply2d(gres, g["b"], q1, wx, wy, g["lphi"])
    close(q1, g["q1"])
    s = O.PressureCGSolver2D(gres, g["bound_size"])
    vx, vy = g["in_vx"].copy(), g["in_vy"].copy()
    s.solve(vx, vy, g["sphi"], g["sv"], g["lphi"], tol=float(g["tol"]))
    assert s.iterations == int(g["iters"])
    close(np.array(s.history), g["history"], rtol=1e-9)
    close(s.x, g["x"], rtol=1e-9)
    close(vx, g["out_vx"], rtol=1e-9)
    close(vy, g["out_vy"], rtol=1e-9)


@pytest.mark.parametrize("name", golden_names("v3d_"))
def test_viscosity3d(name):
    g = golden(name)
    gres = tuple(int(x) for x in g["gres"])
    cell_vol = float(np.prod(g["bound_size"] / g["gres"]))
    scale = float(g["dt"]) / cell_vol / float(g["rho"])
    mu = float(g["mu"])
    vol = g["lvol"] / (cell_vol * 0.125)
    ex, ey, ez = (g[k].astype(np.float64) for k in ("in_vx", "in_vy", "in_vz"))
    O.visc_extrapolate3d(gres, 3, ex, ey, ez, g["sphi"])
    close(ex, g["ex"]); close(ey, g["ey"]); close(ez, g["ez"])
    bx, by, bz = (np.zeros_like(g[k]) for k in ("bx", "by", "bz"))
    O.visc_rhs3d(gres, scale, mu, g["ex"], g["ey"], g["ez"], g["sphi"], g["sv"], vol, bx, by, bz)
    close(bx, g["bx"]); close(by, g["by"]); close(bz, g["bz"])
    qx, qy, qz = (np.full_like(g[k], 7.0) for k in ("qx", "qy", "qz"))
    O.visc_apply3d(gres, scale, mu, g["ex"], g["ey"], g["ez"], qx, qy, qz, g["sphi"], vol)
    close(qx, g["qx"]); close(qy, g["qy"]); close(qz, g["qz"])

    s = O.ViscosityCGSolver3D(gres, g["bound_size"])
    vx, vy, vz = g["in_vx"].copy(), g["in_vy"].copy(), g["in_vz"].copy()
    s.solve(float(g["dt"]), mu, float(g["rho"]), vx, vy, vz, g["sphi"], g["sv"], g["lphi"], g["lvol"],
            tol=float(g["tol"]))
    assert s.iterations == int(g["iters"])
    close(np.array(s.history), g["history"], rtol=1e-9)
    close(s.x_x, g["x_x"], rtol=1e-9); close(s.x_y, g["x_y"], rtol=1e-9); close(s.x_z, g["x_z"], rtol=1e-9)
    for a, b in ((vx, g["out_vx"]), (vy, g["out_vy"]), (vz, g["out_vz"])):
        assert a.dtype == b.dtype
        close(a, b, rtol=1e-6 if a.dtype == np.float32 else 1e-9)
