"""Structural properties of the operators as the reference defines them, checked on the oracle (CPU) over
seeded random scenes -- SURVEY.md section 4 / Appendix A: the pressure and viscosity operators are symmetric
and positive on their active DOFs (so plain CG is the right solver), boundary entries are never written, A 0 = 0;
the density solver's operator is NOT symmetric (unit diagonal weights, -z tap weighted by wz[z+1],
solver/DensityCGSolver3D.py:184) -- the reference runs CG on it anyway, and so does the drop-in."""
import numpy as np
import pytest

from mfs import scenes
from oracle import mfs_oracle as O


def _pressure_setup(gres, seed):
    sc = scenes.pressure_scene_3d(gres, seed, noise=0.3)
    Nx, Ny, Nz = gres
    wx, wy, wz = np.zeros((Nx + 1, Ny, Nz)), np.zeros((Nx, Ny + 1, Nz)), np.zeros((Nx, Ny, Nz + 1))
    O.compute_solid_frac3d(gres, sc["sphi"], wx, wy, wz)
    rng = np.random.default_rng(seed)
    lphi = sc["lphi"] + 0.02 * rng.standard_normal(gres)          # ragged free surface -> ghost-fluid terms everywhere
    return wx, wy, wz, lphi, rng


@pytest.mark.parametrize("gres,seed", [((9, 10, 11), 1), ((12, 8, 10), 2)])
def test_pressure_operator_symmetric_positive(gres, seed):
    wx, wy, wz, lphi, rng = _pressure_setup(gres, seed)
    A = lambda v: (lambda o: (O.pressure_apply3d(gres, v, o, wx, wy, wz, lphi), o)[1])(np.full(gres, 7.0))  # noqa: E731
    u, v = rng.standard_normal(gres), rng.standard_normal(gres)
    au, av = A(u), A(v)
    inner = (slice(1, -1),) * 3
    assert (au[0] == 7).all() and (au[:, -1] == 7).all() and (au[:, :, 0] == 7).all()   # boundary cells untouched
    np.testing.assert_array_equal(A(np.zeros(gres))[inner], 0.0)
    # symmetry / positivity on the interior fluid cells (the operator's range); boundary values of u do enter A u
    # in the reference (it reads v[nb] for boundary neighbours), so test with operands that vanish there, like CG's
    for w in (u, v):
        w[0] = w[-1] = 0; w[:, 0] = w[:, -1] = 0; w[:, :, 0] = w[:, :, -1] = 0
        w[lphi >= 0] = 0
    au, av = A(u), A(v)
    assert abs((au[inner] * v[inner]).sum() - (u[inner] * av[inner]).sum()) < 1e-10 * abs((au[inner] * v[inner]).sum() + 1)
    assert (au[inner] * u[inner]).sum() > 0


def test_density_operator_is_not_symmetric():
    gres = (9, 10, 11)
    wx, wy, wz, lphi, rng = _pressure_setup(gres, 3)
    A = lambda v: (lambda o: (O.density_apply3d(gres, v, o, wx, wy, wz, lphi), o)[1])(np.zeros(gres))  # noqa: E731
    u, v = rng.standard_normal(gres), rng.standard_normal(gres)
    for w in (u, v):
        w[0] = w[-1] = 0; w[:, 0] = w[:, -1] = 0; w[:, :, 0] = w[:, :, -1] = 0
        w[lphi >= 0] = 0
    a, b = (A(u) * v).sum(), (u * A(v)).sum()
    assert abs(a - b) > 1e-6 * abs(a)


@pytest.mark.parametrize("gres,seed", [((8, 9, 10), 4)])
def test_viscosity_operator_symmetric_positive(gres, seed):
    sc = scenes.viscosity_scene_3d(gres, seed=seed)
    rng = np.random.default_rng(seed)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale, mu = sc["dt"] / cell_vol / sc["rho"], 50.0
    vol = sc["lvol"] / (cell_vol * 0.125) + 0.05 * rng.uniform(size=sc["lvol"].shape)
    shapes = [tuple(np.array(gres) + np.eye(3, dtype=int)[a]) for a in range(3)]

    def field():
        f = [rng.standard_normal(s) for s in shapes]
        for a, t in enumerate(f):                      # zero on array-boundary and solid faces, like CG's operand
            t[0] = t[-1] = 0; t[:, 0] = t[:, -1] = 0; t[:, :, 0] = t[:, :, -1] = 0
        ok = [sc["sphi"][0::2, 1::2, 1::2] >= 0, sc["sphi"][1::2, 0::2, 1::2] >= 0, sc["sphi"][1::2, 1::2, 0::2] >= 0]
        return [t * m for t, m in zip(f, ok)]

    def A(f):
        out = [np.zeros(s) for s in shapes]
        O.visc_apply3d(gres, scale, mu, f[0], f[1], f[2], out[0], out[1], out[2], sc["sphi"], vol)
        return out
    u, v = field(), field()
    au, av = A(u), A(v)
    dot = lambda p, q: sum((a * b).sum() for a, b in zip(p, q))  # noqa: E731
    assert abs(dot(au, v) - dot(u, av)) < 1e-9 * (abs(dot(au, v)) + 1)
    assert dot(au, u) > 0


def _golden_visc(name):
    import os
    with np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    cell_vol = float(np.prod(np.array(g["bound_size"], dtype=np.float64) / np.array(gres, dtype=np.float64)))
    scale = float(g["dt"]) / cell_vol / float(g["rho"])
    vol = np.asarray(g["lvol"], np.float64) / (cell_vol * 0.125)
    return g, gres, scale, float(g["mu"]), vol


def test_visc_diag_is_the_diagonal_of_the_executed_reference_operator():
    """visc_diag3d (operand of the opt-in Jacobi loops) against the operator restatement the goldens pin: (A e_i)_i on random faces"""
    g, gres, scale, mu, vol = _golden_visc("v3d_a_12")
    D = O.visc_diag3d(gres, scale, mu, g["sphi"], vol)
    rng = np.random.default_rng(0)
    shp = [d.shape for d in D]
    checked = 0
    for _ in range(60):
        c = int(rng.integers(0, 3))
        idx = tuple(int(rng.integers(1, s - 1)) for s in shp[c])
        E = [np.zeros(s) for s in shp]
        E[c][idx] = 1.0
        Q = [np.zeros(s) for s in shp]
        O.visc_apply3d(gres, scale, mu, *E, *Q, g["sphi"], vol)
        assert Q[c][idx] == pytest.approx(D[c][idx], rel=1e-14, abs=0.0)
        checked += D[c][idx] != 0.0
    assert checked > 10


def test_visc_cg_jacobi_solves_the_reference_system():
    """the preconditioned restatement converges to the solution of the SAME system as the reference's loop (both iterated to 1e-12)"""
    g, gres, scale, mu, vol = _golden_visc("v3d_a_12")
    B = [np.array(g[k], dtype=np.float64) for k in ("bx", "by", "bz")]
    Xj = [np.array(g[k], dtype=np.float64) for k in ("ex", "ey", "ez")]
    itj, _ = O.visc_cg_jacobi(gres, scale, mu, B, Xj, g["sphi"], vol, 1e-12, 10000)
    Xp = [np.array(g[k], dtype=np.float64) for k in ("ex", "ey", "ez")]
    Q, R, Dv = ([np.zeros_like(b) for b in B] for _ in range(3))

    def ap(V, Qo):
        for q in Qo:
            q[...] = 0.0
        O.visc_apply3d(gres, scale, mu, *V, *Qo, g["sphi"], vol)
    # plain CG on the three-component system (the reference's loop, :575-612), to the same tolerance
    ap(Xp, Q)
    R = [b - q for b, q in zip(B, Q)]
    Dv = [r.copy() for r in R]
    delta = sum(float(np.sum(r * r)) for r in R)
    itp = 0
    while delta >= 1e-24 and itp < 10000:
        itp += 1
        ap(Dv, Q)
        alpha = delta / sum(float(np.sum(d * q)) for d, q in zip(Dv, Q))
        for x, d in zip(Xp, Dv):
            x += alpha * d
        for r, q in zip(R, Q):
            r -= alpha * q
        dn = sum(float(np.sum(r * r)) for r in R)
        Dv = [r + (dn / delta) * d for r, d in zip(R, Dv)]
        delta = dn
    assert itj < itp
    nrm = max(np.abs(x).max() for x in Xp)
    for a, b in zip(Xj, Xp):
        assert np.abs(a - b).max() <= 1e-9 * nrm
