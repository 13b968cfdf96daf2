#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN SOURCE (container only).

How the oracle is pinned: the reference ships no tests,
fixtures or golden vectors for this path, and its solver files import `cupy`
and `numba.cuda`, neither of which is installable here.  This script puts the
container-only plumbing in tests/golden/refshim/ (numpy as the array container,
a sequential per-thread launcher) ahead of /root/reference on sys.path and then
imports `solver.PressureCGSolver3D`, `solver.ViscosityCGSolver3D`,
`solver.PressureCGSolver2D`, `solver.SolidFraction{2D,3D}`,
`solver.CGSolverBuffer`, `solver.DensityCGSolver3D` UNMODIFIED and calls their public functions / classes.
Every arithmetic statement that produces a fixture is the reference's.

What this does NOT pin: cupy/numba-CUDA execution itself (FMA contraction and
`cp.sum` reduction order differ from CPython/numpy at the 1e-16 level).

Needs /root/reference; never runs on the GPU box.  The fixtures it writes are
data (inputs + expected outputs); no reference source text is stored.

Usage:  python tests/golden/make_goldens.py [case-prefix ...]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MFS_REFERENCE", "/root/reference")

sys.dont_write_bytecode = True          # the reference mount is read-only
# order matters: the plumbing first, then the REFERENCE's `solver` package (it must shadow this repo's
# own drop-in package of the same name), then this repo's package for `mfs.scenes` only
sys.path.insert(0, os.path.join(REPO, "python-fluid-simulation_amd"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "refshim"))

import numpy as np  # noqa: E402
import cupy as cp   # noqa: E402  (tests/golden/refshim/cupy.py)

from mfs import scenes  # noqa: E402

import solver.CGSolverBuffer as RB          # noqa: E402  (reference modules)
import solver.PressureCGSolver2D as RP2     # noqa: E402
import solver.PressureCGSolver3D as RP3     # noqa: E402
import solver.SolidFraction2D as RS2        # noqa: E402
import solver.SolidFraction3D as RS3        # noqa: E402
import solver.SolidFractionCommon as RSC    # noqa: E402
import solver.ViscosityCGSolver3D as RV3    # noqa: E402
import solver.DensityCGSolver3D as RD3      # noqa: E402

assert RP3.__file__.startswith(REF), RP3.__file__


class _SumLogger:
    """Proxy for a reference module's global `cp`: logs every cp.sum() result.

    pressure log  = [delta0, dq1, delta1, dq2, delta2, ...]
    viscosity log = [rx,ry,rz (delta0), dqx,dqy,dqz, rx,ry,rz, ...]
    (SURVEY.md 8(c) 'Capturing the residual history')."""

    def __init__(self, real):
        self._real = real
        self.log = []

    def __getattr__(self, name):
        return getattr(self._real, name)

    def sum(self, a, *args, **kw):
        v = self._real.sum(a, *args, **kw)
        self.log.append(float(np.asarray(v)))
        return v


def C(a, dtype=None):
    return cp.array(np.array(a, dtype=dtype, copy=True))


def gen_pressure3d(name, gres, seed, vel_dtype, solid_velocity, tol=1e-3, all_fluid=False):
    sc = scenes.pressure_scene_3d(gres, seed, vel_dtype=vel_dtype, solid_velocity=solid_velocity,
                                  all_fluid=all_fluid)
    g = C(gres, np.int64)
    bsz = C(sc["bound_size"], np.float64)
    sphi, sv, lphi = C(sc["sphi"]), C(sc["sv"]), C(sc["lphi"])

    # module-level functions, called directly
    wx = cp.zeros((gres[0] + 1, gres[1], gres[2]))
    wy = cp.zeros((gres[0], gres[1] + 1, gres[2]))
    wz = cp.zeros((gres[0], gres[1], gres[2] + 1))
    RS3.compute_solid_frac(g, sphi, wx, wy, wz)
    cell_size = bsz / g
    b = cp.zeros(gres)
    RP3.initialize_solver(cell_size, g, C(sc["vx"]), C(sc["vy"]), C(sc["vz"]), sphi, sv, lphi, b, wx, wy, wz)
    q1 = cp.zeros(gres)
    RP3.matvecmul(g, b, q1, wx, wy, wz, lphi)
    # matvec of a random vector including non-zero boundary cells
    rv = np.random.default_rng(seed + 100).standard_normal(gres)
    qr = cp.array(np.full(gres, 7.0))          # sentinel: boundary cells must stay 7
    RP3.matvecmul(g, C(rv), qr, wx, wy, wz, lphi)

    # the class, end to end, with the residual history logged
    buf = RB.CGSolverBuffer(g)
    slv = RP3.PressureCGSolver3D(buf, g, bsz)
    vx, vy, vz = C(sc["vx"]), C(sc["vy"]), C(sc["vz"])
    logger = _SumLogger(cp)
    RP3.cp = logger
    t0 = time.time()
    try:
        slv.solve(vx, vy, vz, sphi, sv, lphi, tol=tol)
    finally:
        RP3.cp = cp
    hist = np.array(logger.log)
    iters = (len(hist) - 1) // 2
    print(f"  {name}: gres={gres} iters={iters} delta0={hist[0]:.4e} delta_end={hist[-1]:.4e}"
          f" ({time.time() - t0:.1f}s)")
    assert np.array_equal(np.asarray(slv.wx), np.asarray(wx))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        kind="pressure3d", gres=np.array(gres), bound_size=np.array(sc["bound_size"]), tol=tol,
        seed=seed, solid_velocity=solid_velocity, all_fluid=all_fluid,
        in_vx=sc["vx"], in_vy=sc["vy"], in_vz=sc["vz"], sphi=sc["sphi"], sv=sc["sv"], lphi=sc["lphi"],
        wx=np.asarray(wx), wy=np.asarray(wy), wz=np.asarray(wz), b=np.asarray(b), q1=np.asarray(q1),
        rv=rv, qr=np.asarray(qr),
        history=hist, iters=iters, x=np.asarray(slv.x),
        out_vx=np.asarray(vx), out_vy=np.asarray(vy), out_vz=np.asarray(vz),
        alpha=slv.alpha, beta=slv.beta, delta=slv.delta)


def gen_pressure2d(name, gres, seed, solid_velocity, tol=1e-6):
    sc = scenes.pressure_scene_2d(gres, seed, solid_velocity=solid_velocity)
    g = C(gres, np.int64)
    bsz = C(sc["bound_size"], np.float64)
    sphi, sv, lphi = C(sc["sphi"]), C(sc["sv"]), C(sc["lphi"])
    wx = cp.zeros((gres[0] + 1, gres[1]))
    wy = cp.zeros((gres[0], gres[1] + 1))
    RS2.compute_solid_frac(g, sphi, wx, wy)
    b = cp.zeros(gres)
    RP2.initialize_solver(bsz / g, g, C(sc["vx"]), C(sc["vy"]), sphi, sv, lphi, b, wx, wy)
    q1 = cp.zeros(gres)
    RP2.matvecmul(g, b, q1, wx, wy, lphi)

    buf = RB.CGSolverBuffer(g)
    slv = RP2.PressureCGSolver2D(buf, g, bsz)
    vx, vy = C(sc["vx"]), C(sc["vy"])
    logger = _SumLogger(cp)
    RP2.cp = logger
    t0 = time.time()
    try:
        slv.solve(vx, vy, sphi, sv, lphi, tol=tol)
    finally:
        RP2.cp = cp
    hist = np.array(logger.log)
    iters = (len(hist) - 1) // 2
    print(f"  {name}: gres={gres} iters={iters} delta0={hist[0]:.4e} delta_end={hist[-1]:.4e}"
          f" ({time.time() - t0:.1f}s)")
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        kind="pressure2d", gres=np.array(gres), bound_size=np.array(sc["bound_size"]), tol=tol,
        seed=seed, solid_velocity=solid_velocity,
        in_vx=sc["vx"], in_vy=sc["vy"], sphi=sc["sphi"], sv=sc["sv"], lphi=sc["lphi"],
        wx=np.asarray(wx), wy=np.asarray(wy), b=np.asarray(b), q1=np.asarray(q1),
        history=hist, iters=iters, x=np.asarray(slv.x),
        out_vx=np.asarray(vx), out_vy=np.asarray(vy))


def gen_viscosity3d(name, gres, seed, vel_dtype, tol=1e-3, mu=None):
    sc = scenes.viscosity_scene_3d(gres, seed, vel_dtype=vel_dtype)
    if mu is not None:
        sc["mu"] = mu
    g = C(gres, np.int64)
    bsz = C(sc["bound_size"], np.float64)
    sphi, sv, lphi, lvol = C(sc["sphi"]), C(sc["sv"]), C(sc["lphi"]), C(sc["lvol"])
    fx = (gres[0] + 1, gres[1], gres[2])
    fy = (gres[0], gres[1] + 1, gres[2])
    fz = (gres[0], gres[1], gres[2] + 1)

    # module-level functions, called directly on fp64 copies (as solve() does)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale = sc["dt"] / cell_vol / sc["rho"]
    vol = cp.array(sc["lvol"] / (cell_vol * 0.125))
    ex, ey, ez = C(sc["vx"], np.float64), C(sc["vy"], np.float64), C(sc["vz"], np.float64)
    RV3.extrapolate(g, 3, ex, ey, ez, sphi)
    bx, by, bz = cp.zeros(fx), cp.zeros(fy), cp.zeros(fz)
    RV3.initialize_solver(g, scale, sc["mu"], ex, ey, ez, sphi, sv, vol, bx, by, bz)
    qx, qy, qz = cp.array(np.full(fx, 7.0)), cp.array(np.full(fy, 7.0)), cp.array(np.full(fz, 7.0))
    RV3.matvecmul(g, scale, sc["mu"], ex, ey, ez, qx, qy, qz, sphi, vol)

    slv = RV3.ViscosityCGSolver3D(g, bsz)
    vx, vy, vz = C(sc["vx"]), C(sc["vy"]), C(sc["vz"])
    logger = _SumLogger(cp)
    RV3.cp = logger
    t0 = time.time()
    try:
        slv.solve(sc["dt"], sc["mu"], sc["rho"], vx, vy, vz, sphi, sv, lphi, lvol, tol=tol)
    finally:
        RV3.cp = cp
    log = np.array(logger.log).reshape(-1, 3).sum(axis=1)   # triples -> scalars, same order as :585,592,604
    iters = (len(log) - 1) // 2
    print(f"  {name}: gres={gres} iters={iters} delta0={log[0]:.4e} delta_end={log[-1]:.4e}"
          f" ({time.time() - t0:.1f}s)")
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        kind="viscosity3d", gres=np.array(gres), bound_size=np.array(sc["bound_size"]), tol=tol,
        seed=seed, dt=sc["dt"], mu=sc["mu"], rho=sc["rho"],
        in_vx=sc["vx"], in_vy=sc["vy"], in_vz=sc["vz"], sphi=sc["sphi"], sv=sc["sv"], lphi=sc["lphi"],
        lvol=sc["lvol"],
        ex=np.asarray(ex), ey=np.asarray(ey), ez=np.asarray(ez),
        bx=np.asarray(bx), by=np.asarray(by), bz=np.asarray(bz),
        qx=np.asarray(qx), qy=np.asarray(qy), qz=np.asarray(qz),
        history=log, history_raw=np.array(logger.log), iters=iters,
        x_x=np.asarray(slv.x_x), x_y=np.asarray(slv.x_y), x_z=np.asarray(slv.x_z),
        out_vx=np.asarray(vx), out_vy=np.asarray(vy), out_vz=np.asarray(vz))


def gen_density3d(name, gres, seed, per_cell=4, tol=1e-3, px_dtype=np.float64):
    """solver/DensityCGSolver3D.py: the module functions one by one, then the class end to end."""
    sc = scenes.density_scene_3d(gres, seed, per_cell=per_cell, px_dtype=px_dtype)
    g = C(gres, np.int64)
    bmin, bsz = C(sc["bound_min"], np.float64), C(sc["bound_size"], np.float64)
    sphi, sv, lphi, lvol = C(sc["sphi"]), C(sc["sv"]), C(sc["lphi"]), C(sc["lvol"])
    cell_size = bsz / g
    wx = cp.zeros((gres[0] + 1, gres[1], gres[2]))
    wy = cp.zeros((gres[0], gres[1] + 1, gres[2]))
    wz = cp.zeros((gres[0], gres[1], gres[2] + 1))
    RS3.compute_solid_frac(g, sphi, wx, wy, wz)
    gm, gvol = cp.zeros(gres), cp.zeros(gres)
    RD3.initialize_density(bmin, cell_size, g, C(sc["px"]), C(sc["pm"]), sc["pvol"], gm, gvol, sphi, lphi)
    gvol_raw = np.array(gvol)
    RD3.fix_volume(cell_size, g, lvol, gvol, sphi, lphi, wx, wy, wz)
    b = cp.zeros(gres)
    RD3.initialize_solver(sc["rho0"], sc["dt"], g, cell_size, gm, gvol, lphi, wx, wy, wz, b)
    q1 = cp.zeros(gres)
    RD3.matvecmul(g, b, q1, wx, wy, wz, lphi)
    rv = np.random.default_rng(seed + 100).standard_normal(gres)
    qr = cp.array(np.full(gres, 7.0))          # sentinel: boundary cells must stay 7
    RD3.matvecmul(g, C(rv), qr, wx, wy, wz, lphi)

    buf = RB.CGSolverBuffer(g)
    slv = RD3.DensityCGSolver3D(buf, g, bmin, bsz)
    px = C(sc["px"])
    logger = _SumLogger(cp)
    RD3.cp = logger
    t0 = time.time()
    try:
        slv.solve(sc["rho0"], sc["dt"], px, C(sc["pm"]), sc["pvol"], None, None, None, sphi, sv, lphi, lvol, tol=tol)
    finally:
        RD3.cp = cp
    hist = np.array(logger.log)
    iters = (len(hist) - 1) // 2
    print(f"  {name}: gres={gres} particles={len(sc['px'])} iters={iters} delta0={hist[0]:.4e} delta_end={hist[-1]:.4e}"
          f" ({time.time() - t0:.1f}s)")
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        kind="density3d", gres=np.array(gres), bound_min=np.array(sc["bound_min"]), bound_size=np.array(sc["bound_size"]),
        tol=tol, seed=seed, rho0=sc["rho0"], dt=sc["dt"], pvol=sc["pvol"],
        px=sc["px"], pm=sc["pm"], sphi=sc["sphi"], sv=sc["sv"], lphi=sc["lphi"], lvol=sc["lvol"],
        wx=np.asarray(wx), wy=np.asarray(wy), wz=np.asarray(wz), gm=np.asarray(gm), gvol_raw=gvol_raw,
        gvol=np.asarray(gvol), b=np.asarray(b), q1=np.asarray(q1), rv=rv, qr=np.asarray(qr),
        history=hist, iters=iters, x=np.asarray(slv.x), dx=np.asarray(slv.dx), dy=np.asarray(slv.dy),
        dz=np.asarray(slv.dz), out_px=np.asarray(px), out_gm=np.asarray(slv.m), out_gvol=np.asarray(slv.vol),
        alpha=slv.alpha, beta=slv.beta, delta=slv.delta)


def gen_fraction_tables(name):
    """Known-answer tables of the three device functions in
    solver/SolidFractionCommon.py, evaluated on a grid of sign patterns."""
    vals = np.array([-1.5, -0.4, -0.1, 0.0, 0.2, 0.7, 2.0])
    e = np.array([[float(RSC.edge_in_fraction(a, b)) for b in vals] for a in vals])
    t = np.array([[[float(RSC.tri_in_fraction(a, b, c)) for c in vals] for b in vals] for a in vals])
    rng = np.random.default_rng(11)
    quads = rng.uniform(-1, 1, size=(400, 4))
    f = np.array([float(RSC.face_in_fraction(*q)) for q in quads])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="fractions", vals=vals, edge=e,
                        tri=t, quads=quads, face=f)
    print(f"  {name}: edge {e.shape} tri {t.shape} face {f.shape}; tri values {np.unique(t)}")


CASES = [
    ("fractions", lambda n: gen_fraction_tables(n)),
    ("p3d_a_12", lambda n: gen_pressure3d(n, (12, 12, 12), 0, np.float32, False)),
    ("p3d_b_10x12x14_sv", lambda n: gen_pressure3d(n, (10, 12, 14), 5, np.float64, True)),
    ("p3d_c_16x12x8_sv", lambda n: gen_pressure3d(n, (16, 12, 8), 6, np.float32, True)),
    ("p3d_d_20", lambda n: gen_pressure3d(n, (20, 20, 20), 7, np.float32, False)),
    ("p3d_e_allfluid_12", lambda n: gen_pressure3d(n, (12, 12, 12), 8, np.float64, False, all_fluid=True)),
    ("p2d_a_64", lambda n: gen_pressure2d(n, (64, 64), 1, False)),
    ("p2d_b_24x20_sv", lambda n: gen_pressure2d(n, (24, 20), 2, True)),
    ("v3d_a_12", lambda n: gen_viscosity3d(n, (12, 12, 12), 3, np.float32)),
    ("v3d_b_10x12x14", lambda n: gen_viscosity3d(n, (10, 12, 14), 4, np.float64)),
    ("v3d_c_16_mu50", lambda n: gen_viscosity3d(n, (16, 16, 16), 5, np.float32, mu=50.0)),
    ("d3d_a_12", lambda n: gen_density3d(n, (12, 12, 12), 9)),
    ("d3d_b_10x12x14_f32", lambda n: gen_density3d(n, (10, 12, 14), 10, per_cell=3, px_dtype=np.float32)),
    # round 2: cases that span more than one tile of the march kernels ("N = 32-48 are practical")
    ("p3d_f_40x36x32_sv", lambda n: gen_pressure3d(n, (40, 36, 32), 12, np.float32, True)),
    ("v3d_d_24", lambda n: gen_viscosity3d(n, (24, 24, 24), 13, np.float32)),
    ("v3d_e_20x24x36_mu20", lambda n: gen_viscosity3d(n, (20, 24, 36), 14, np.float64, mu=20.0)),
    ("d3d_c_20", lambda n: gen_density3d(n, (20, 20, 20), 15)),
]

if __name__ == "__main__":
    want = sys.argv[1:]
    for cname, fn in CASES:
        if want and not any(cname.startswith(w) for w in want):
            continue
        fn(cname)
