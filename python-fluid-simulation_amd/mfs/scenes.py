"""Seeded synthetic inputs for the pressure / viscosity CG solvers.

Everything here *generates inputs*; nothing here is solver arithmetic.  The
array conventions are the reference's: C-order, axis
order [x, y, z]; cell arrays (Nx,Ny,Nz); face arrays vx (Nx+1,Ny,Nz),
vy (Nx,Ny+1,Nz), vz (Nx,Ny,Nz+1); doubled grid sphi / lvol
(2Nx+1,2Ny+1,2Nz+1) with (2x+1,2y+1,2z+1) the cell centre; sv (...,3).
`sphi < 0` is inside solid, `lphi < 0` is inside fluid
(reference: solver/PressureCGSolver3D.py:137,183, solver/SolidFraction3D.py:12-18).

Generators are written once against a tiny array-namespace adapter so that the
same formulas run in numpy (goldens, CPU tests, the oracle) and in torch on the
device (bench-size slabs that should never be built on the host).
"""
from __future__ import annotations

import math

import numpy as np


class _NP:
    float64 = np.float64

    @staticmethod
    def arange(a, b):
        return np.arange(a, b, dtype=np.float64)

    minimum = staticmethod(np.minimum)
    maximum = staticmethod(np.maximum)
    sqrt = staticmethod(np.sqrt)
    sin = staticmethod(np.sin)
    cos = staticmethod(np.cos)
    abs = staticmethod(np.abs)

    @staticmethod
    def clip(a, lo, hi):
        return np.clip(a, lo, hi)

    @staticmethod
    def full(shape, v):
        return np.full(shape, v, dtype=np.float64)

    @staticmethod
    def stack_last(arrs):
        return np.stack(arrs, axis=-1)


class _Torch:
    def __init__(self, device):
        import torch
        self.t = torch
        self.device = device
        self.float64 = torch.float64
        self.minimum = torch.minimum
        self.maximum = torch.maximum
        self.sqrt = torch.sqrt
        self.sin = torch.sin
        self.cos = torch.cos
        self.abs = torch.abs

    def arange(self, a, b):
        return self.t.arange(a, b, dtype=self.t.float64, device=self.device)

    def clip(self, a, lo, hi):
        return self.t.clamp(a, lo, hi)

    def full(self, shape, v):
        return self.t.full(shape, v, dtype=self.t.float64, device=self.device)

    def stack_last(self, arrs):
        return self.t.stack(arrs, dim=-1)


def _xp(device):
    return _NP() if device is None else _Torch(device)


def _axes(xp, lo, hi, step, origin=0.0):
    """coordinate = origin + index*step for index in [lo, hi)."""
    return origin + xp.arange(lo, hi) * step


def _b3(ax, ay, az):
    return ax[:, None, None], ay[None, :, None], az[None, None, :]


# ----------------------------------------------------------------------------
# solid level set shared by the 3D scenes: closed box walls + a sphere obstacle
# ----------------------------------------------------------------------------
def _sphi_box_sphere(xp, X, Y, Z, size, wall, sph_c, sph_r):
    """>0 in the open domain, <0 in walls (thickness `wall`) and in the sphere."""
    box = xp.minimum(xp.minimum(xp.minimum(X, size[0] - X), xp.minimum(Y, size[1] - Y)),
                     xp.minimum(Z, size[2] - Z)) - wall
    sph = xp.sqrt((X - sph_c[0]) ** 2 + (Y - sph_c[1]) ** 2 + (Z - sph_c[2]) ** 2) - sph_r
    return xp.minimum(box + 0 * sph, sph + 0 * box)


def pressure_scene_3d(gres, seed=0, *, bound_size=(1.0, 1.0, 1.0), vel_dtype=np.float32,
                      solid_velocity=False, all_fluid=False, x_range=None, device=None,
                      noise=1e-2):
    """C2 of SURVEY.md 8(d): pool with a flat-ish free surface, box walls, sphere.

    x_range=(a, b) builds only cell planes [a, b) of the global grid (and the
    matching face / doubled-grid planes): the slab a rank owns plus its ghosts.
    Returns a dict of arrays (numpy when device is None, torch otherwise).
    """
    xp = _xp(device)
    Nx, Ny, Nz = (int(g) for g in gres)
    a, b = (0, Nx) if x_range is None else x_range
    size = tuple(float(s) for s in bound_size)
    cs = (size[0] / Nx, size[1] / Ny, size[2] / Nz)
    wall = 1.5 * min(cs)
    sph_c = (0.5 * size[0], 0.3 * size[1], 0.5 * size[2])
    sph_r = 0.12 * min(size)

    # doubled grid nodes of the slab: indices [2a, 2b] inclusive
    dx = _axes(xp, 2 * a, 2 * b + 1, 0.5 * cs[0])
    dy = _axes(xp, 0, 2 * Ny + 1, 0.5 * cs[1])
    dz = _axes(xp, 0, 2 * Nz + 1, 0.5 * cs[2])
    X, Y, Z = _b3(dx, dy, dz)
    if all_fluid:
        # open interior everywhere except the outermost half-cell shell
        sphi = xp.minimum(xp.minimum(xp.minimum(X, size[0] - X), xp.minimum(Y, size[1] - Y)),
                          xp.minimum(Z, size[2] - Z)) + 0.25 * min(cs)
    else:
        sphi = _sphi_box_sphere(xp, X, Y, Z, size, wall, sph_c, sph_r)

    if solid_velocity:
        svx = 0.3 * xp.sin(2.0 * Y) + 0 * X + 0 * Z
        svy = -0.2 * xp.cos(3.0 * Z) + 0 * X + 0 * Y
        svz = 0.1 * xp.sin(X + Y) + 0 * Z
    else:
        svx = svy = svz = 0 * (X + Y + Z)
    sv = xp.stack_last([svx, svy, svz])

    # cell centres
    cx = _axes(xp, a, b, cs[0], 0.5 * cs[0])
    cy = _axes(xp, 0, Ny, cs[1], 0.5 * cs[1])
    cz = _axes(xp, 0, Nz, cs[2], 0.5 * cs[2])
    CX, CY, CZ = _b3(cx, cy, cz)
    if all_fluid:
        lphi = xp.full((b - a, Ny, Nz), -1.0)
    else:
        lphi = (CY - 0.62 * size[1]) + 0.03 * size[1] * xp.sin(2 * math.pi * CX / size[0]) \
            * xp.cos(2 * math.pi * CZ / size[2])

    # staggered velocity = gradient of a smooth potential (divergent) + noise
    two_pi = 2 * math.pi

    def pot_grad(PX, PY, PZ, axis):
        sx, sy, sz = (xp.sin(two_pi * PX / size[0]), xp.sin(two_pi * PY / size[1]),
                      xp.sin(two_pi * PZ / size[2]))
        cxx, cyy, czz = (xp.cos(two_pi * PX / size[0]), xp.cos(two_pi * PY / size[1]),
                         xp.cos(two_pi * PZ / size[2]))
        if axis == 0:
            return two_pi / size[0] * cxx * sy * sz * cs[0]
        if axis == 1:
            return two_pi / size[1] * sx * cyy * sz * cs[1]
        return two_pi / size[2] * sx * sy * czz * cs[2]

    fx = _axes(xp, a, b + 1, cs[0])
    fy = _axes(xp, 0, Ny + 1, cs[1])
    fz = _axes(xp, 0, Nz + 1, cs[2])
    vx = pot_grad(*_b3(fx, cy, cz), 0)
    vy = pot_grad(*_b3(cx, fy, cz), 1)
    vz = pot_grad(*_b3(cx, cy, fz), 2)

    out = dict(gres=(Nx, Ny, Nz), x_range=(a, b), bound_size=size, cell_size=cs,
               sphi=sphi, sv=sv, lphi=lphi)
    if device is None:
        rng = np.random.default_rng(seed)
        # noise drawn for the full grid so that a slab is a slice of the global field
        nvx = rng.standard_normal((Nx + 1, Ny, Nz))[a:b + 1]
        nvy = rng.standard_normal((Nx, Ny + 1, Nz))[a:b]
        nvz = rng.standard_normal((Nx, Ny, Nz + 1))[a:b]
        out["vx"] = (vx + noise * nvx).astype(vel_dtype)
        out["vy"] = (vy + noise * nvy).astype(vel_dtype)
        out["vz"] = (vz + noise * nvz).astype(vel_dtype)
    else:
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(seed * 7919 + a)
        tdt = {np.float32: torch.float32, np.float64: torch.float64}.get(vel_dtype, vel_dtype)
        for name, arr in (("vx", vx), ("vy", vy), ("vz", vz)):
            n = torch.randn(arr.shape, generator=g, device=device, dtype=torch.float64)
            out[name] = (arr + noise * n).to(tdt)
    return out


# ----------------------------------------------------------------------------
# 2D pressure scene (config 1 of BASELINE.json; reference solver/PressureCGSolver2D.py)
# ----------------------------------------------------------------------------
def pressure_scene_2d(gres, seed=1, *, bound_size=(1.0, 1.0), vel_dtype=np.float64,
                      solid_velocity=False):
    Nx, Ny = (int(g) for g in gres)
    size = tuple(float(s) for s in bound_size)
    cs = (size[0] / Nx, size[1] / Ny)
    dx = np.arange(2 * Nx + 1) * 0.5 * cs[0]
    dy = np.arange(2 * Ny + 1) * 0.5 * cs[1]
    X, Y = dx[:, None], dy[None, :]
    sphi = np.minimum(np.minimum(X, size[0] - X), np.minimum(Y, size[1] - Y)) - 0.05 * min(size)
    if solid_velocity:
        sv = np.stack([0.3 * np.sin(2 * Y) + 0 * X, -0.2 * np.cos(3 * X) + 0 * Y], axis=-1)
    else:
        sv = np.zeros((2 * Nx + 1, 2 * Ny + 1, 2))
    cx = (np.arange(Nx) + 0.5) * cs[0]
    cy = (np.arange(Ny) + 0.5) * cs[1]
    lphi = np.sqrt((cx[:, None] - 0.5 * size[0]) ** 2 + (cy[None, :] - 0.5 * size[1]) ** 2) \
        - 0.3 * min(size)
    rng = np.random.default_rng(seed)
    vx = rng.standard_normal((Nx + 1, Ny)).astype(vel_dtype)
    vy = rng.standard_normal((Nx, Ny + 1)).astype(vel_dtype)
    return dict(gres=(Nx, Ny), bound_size=size, cell_size=cs, sphi=sphi, sv=sv, lphi=lphi,
                vx=vx, vy=vy)


# ----------------------------------------------------------------------------
# viscosity scene (config 3): buckling-like block of viscous fluid above slabs
# ----------------------------------------------------------------------------
def _box_sdf(xp, X, Y, Z, centre, half):
    """signed distance to an axis-aligned box (negative inside); the semantics
    of the reference's box SDF (solver/sdf3D.py:86-109) for an unrotated box."""
    qx = xp.abs(X - centre[0]) - half[0]
    qy = xp.abs(Y - centre[1]) - half[1]
    qz = xp.abs(Z - centre[2]) - half[2]
    zero = 0 * (qx + qy + qz)
    outside = xp.sqrt(xp.maximum(qx, zero) ** 2 + xp.maximum(qy, zero) ** 2
                      + xp.maximum(qz, zero) ** 2)
    inside = xp.minimum(xp.maximum(xp.maximum(qx + zero, qy + zero), qz + zero), zero)
    return outside + inside


def viscosity_scene_3d(gres, seed=3, *, bound_size=(1.0, 1.0, 1.0), vel_dtype=np.float32,
                       x_range=None, device=None, noise=5e-2):
    """C3 of SURVEY.md 8(d): a container (flipped box), two obstacle slabs under a
    block of fluid; `lvol` is the analytic sub-cell coverage of the fluid block
    times the sub-cell volume (what the notebook's compute_fluid_volume produces,
    ipynb c6), velocities (-2,0,0)+noise inside the block.
    """
    xp = _xp(device)
    Nx, Ny, Nz = (int(g) for g in gres)
    a, b = (0, Nx) if x_range is None else x_range
    size = tuple(float(s) for s in bound_size)
    cs = (size[0] / Nx, size[1] / Ny, size[2] / Nz)
    dx = _axes(xp, 2 * a, 2 * b + 1, 0.5 * cs[0])
    dy = _axes(xp, 0, 2 * Ny + 1, 0.5 * cs[1])
    dz = _axes(xp, 0, 2 * Nz + 1, 0.5 * cs[2])
    X, Y, Z = _b3(dx, dy, dz)
    c = (0.5 * size[0], 0.5 * size[1], 0.5 * size[2])
    container = -_box_sdf(xp, X, Y, Z, c, (0.5 * size[0] - 1.6 * cs[0], 0.5 * size[1] - 1.6 * cs[1],
                                          0.5 * size[2] - 1.6 * cs[2]))
    slab1 = _box_sdf(xp, X, Y, Z, (0.30 * size[0], 0.33 * size[1], c[2]),
                     (0.16 * size[0], 0.04 * size[1], 0.6 * size[2]))
    slab2 = _box_sdf(xp, X, Y, Z, (0.72 * size[0], 0.30 * size[1], c[2]),
                     (0.14 * size[0], 0.05 * size[1], 0.6 * size[2]))
    sphi = xp.minimum(xp.minimum(container, slab1), slab2)
    sv = xp.stack_last([0 * sphi, 0 * sphi, 0 * sphi])

    # fluid block [lo, hi]; lvol(node) = overlap of the block with the sub-cell box
    # centred on the node, edge 0.5*cs (so interior nodes carry cell_vol/8).
    lo = (0.28 * size[0], 0.42 * size[1], 0.30 * size[2])
    hi = (0.74 * size[0], 0.80 * size[1], 0.72 * size[2])

    def overlap(P, l, h, half):
        return xp.clip(xp.minimum(P + half, 0 * P + h) - xp.maximum(P - half, 0 * P + l), 0.0, 2 * half)

    lvol = overlap(X, lo[0], hi[0], 0.25 * cs[0]) * overlap(Y, lo[1], hi[1], 0.25 * cs[1]) \
        * overlap(Z, lo[2], hi[2], 0.25 * cs[2])

    cx = _axes(xp, a, b, cs[0], 0.5 * cs[0])
    cy = _axes(xp, 0, Ny, cs[1], 0.5 * cs[1])
    cz = _axes(xp, 0, Nz, cs[2], 0.5 * cs[2])
    CX, CY, CZ = _b3(cx, cy, cz)
    blk = _box_sdf(xp, CX, CY, CZ, tuple(0.5 * (l + h) for l, h in zip(lo, hi)),
                   tuple(0.5 * (h - l) for l, h in zip(lo, hi)))
    lphi = blk + 0 * (CX + CY + CZ)

    fx = _axes(xp, a, b + 1, cs[0])
    fy = _axes(xp, 0, Ny + 1, cs[1])
    fz = _axes(xp, 0, Nz + 1, cs[2])

    def inside(PX, PY, PZ):
        s = _box_sdf(xp, PX, PY, PZ, tuple(0.5 * (l + h) for l, h in zip(lo, hi)),
                     tuple(0.5 * (h - l) + 1.5 * min(cs) for l, h in zip(lo, hi)))
        return (s < 0) * 1.0

    mx = inside(*_b3(fx, cy, cz))
    my = inside(*_b3(cx, fy, cz))
    mz = inside(*_b3(cx, cy, fz))
    shear = lambda P: 1.0 + 0.5 * xp.sin(6.0 * P)  # noqa: E731
    bx = -2.0 * mx * shear(_b3(fx, cy, cz)[1] + 0 * mx)
    by = -0.5 * my * shear(_b3(cx, fy, cz)[0] + 0 * my)
    bz = 0.3 * mz * shear(_b3(cx, cy, fz)[1] + 0 * mz)

    out = dict(gres=(Nx, Ny, Nz), x_range=(a, b), bound_size=size, cell_size=cs,
               sphi=sphi, sv=sv, lphi=lphi, lvol=lvol, dt=1.0 / 300.0, mu=1.0, rho=1000.0)
    if device is None:
        rng = np.random.default_rng(seed)
        nx = rng.standard_normal((Nx + 1, Ny, Nz))[a:b + 1]
        ny = rng.standard_normal((Nx, Ny + 1, Nz))[a:b]
        nz = rng.standard_normal((Nx, Ny, Nz + 1))[a:b]
        out["vx"] = (bx + noise * nx * mx).astype(vel_dtype)
        out["vy"] = (by + noise * ny * my).astype(vel_dtype)
        out["vz"] = (bz + noise * nz * mz).astype(vel_dtype)
    else:
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(seed * 104729 + a)
        tdt = {np.float32: torch.float32, np.float64: torch.float64}.get(vel_dtype, vel_dtype)
        for name, arr, m in (("vx", bx, mx), ("vy", by, my), ("vz", bz, mz)):
            n = torch.randn(arr.shape, generator=g, device=device, dtype=torch.float64)
            out[name] = (arr + noise * n * m).to(tdt)
    return out


def density_scene_3d(gres, seed=0, *, per_cell=4, bound_min=(-0.25, 0.1, 0.0), px_dtype=np.float64, rho0=1000.0,
                     dt=1.0 / 300.0):
    """Inputs of DensityCGSolver3D.solve: the pool scene of `pressure_scene_3d`
    (shifted by `bound_min`) filled with jittered particles below the free surface; particle masses
    rho0 * pvol * (1 +- 10 %), so the density right-hand side is non-trivial.  numpy only."""
    sc = pressure_scene_3d(gres, seed)
    Nx, Ny, Nz = (int(g) for g in gres)
    size = np.asarray(sc["bound_size"], np.float64)
    cs = size / np.array([Nx, Ny, Nz], np.float64)
    rng = np.random.default_rng(seed + 1000)
    n = per_cell * Nx * Ny * Nz
    pos = rng.uniform(0.6 * cs, size - 0.6 * cs, size=(n, 3))
    ci = np.minimum((pos / cs).astype(np.int64), np.array([Nx - 1, Ny - 1, Nz - 1]))
    lphi = np.asarray(sc["lphi"])
    sphi_c = np.asarray(sc["sphi"])[1::2, 1::2, 1::2]
    keep = (lphi[ci[:, 0], ci[:, 1], ci[:, 2]] < 0) & (sphi_c[ci[:, 0], ci[:, 1], ci[:, 2]] > 0)
    pos = pos[keep]
    pvol = float(np.prod(cs)) / per_cell
    pm = rho0 * pvol * (1.0 + 0.1 * rng.standard_normal(len(pos)))
    px = (pos + np.asarray(bound_min, np.float64)).astype(px_dtype)
    lvol = np.zeros_like(np.asarray(sc["sphi"]))        # accepted and unused by the density solver (:262-269)
    return dict(gres=(Nx, Ny, Nz), bound_min=tuple(float(b) for b in bound_min), bound_size=tuple(float(v) for v in size),
                sphi=sc["sphi"], sv=sc["sv"], lphi=sc["lphi"], lvol=lvol, px=px, pm=pm, pvol=pvol, rho0=float(rho0),
                dt=float(dt))


def particle_scene_3d(gres, seed=0, *, per_cell=3, bound_min=(-0.3, 0.0, -0.3)):
    """Inputs of the notebook's particle <-> grid transfers: jittered particles in the
    lower part of a box with the notebook's origin (BOUND_MIN = (-0.3, 0, -0.3), ipynb code cell 9), a few of
    them pushed up to / beyond the walls so that the index clamps are exercised; velocities and affine rows
    random.  Cubic cells of size gdx.  numpy only."""
    Nx, Ny, Nz = (int(g) for g in gres)
    gdx = 0.05
    size = np.array([Nx, Ny, Nz], np.float64) * gdx
    rng = np.random.default_rng(seed + 2000)
    n = per_cell * Nx * Ny * Nz // 2
    pos = rng.uniform([0.8 * gdx, 0.8 * gdx, 0.8 * gdx], [size[0] - 0.8 * gdx, 0.55 * size[1], size[2] - 0.8 * gdx],
                      size=(n, 3))
    k = max(4, n // 50)                 # stragglers next to / outside the walls
    pos[:k] = rng.uniform(-0.4 * gdx, 0.7 * gdx, size=(k, 3)) + rng.integers(0, 2, size=(k, 3)) * (size - 0.3 * gdx)
    px = pos + np.asarray(bound_min, np.float64)
    pvol = gdx ** 3 / per_cell
    pm = 1000.0 * pvol * (1.0 + 0.1 * rng.standard_normal(n))
    pv = rng.standard_normal((n, 3))
    aff = [2.0 * rng.standard_normal((n, 3)) for _ in range(3)]
    return dict(gres=(Nx, Ny, Nz), bound_min=tuple(float(b) for b in bound_min), bound_size=tuple(float(v) for v in size),
                gdx=gdx, px=px, pm=pm, pv=pv, pcx=aff[0], pcy=aff[1], pcz=aff[2], pvol=float(pvol))
