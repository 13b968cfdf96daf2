"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A numpy (fp64) restatement of the reference algorithm for the hot path of
SSTDV-Project/python-fluid-simulation: the CG pressure / viscosity solvers.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this file, and only as the checker.

Parity status: PINNED against tests/golden/*.npz, which were produced by
executing the reference's own source files under CPython with container-only
array/launch plumbing (tests/golden/make_goldens.py).
The reference ships no tests or fixtures of its own for this path.  Not pinned
against cupy/numba-CUDA execution itself (unavailable here): FMA contraction
and cp.sum reduction order may differ from this file at the 1e-16 level.

Every function cites the reference file:line it follows (paths relative to the
reference repo root).  Array conventions: SURVEY.md section 8.
"""
from __future__ import annotations

import numpy as np

F64 = np.float64


# =============================================================================
# solid fractions
# =============================================================================
def edge_in_fraction(lval, rval):
    """solver/SolidFractionCommon.py:4-16 (vectorised; returns fp64)."""
    lval = np.asarray(lval, F64)
    rval = np.asarray(rval, F64)
    l_in = lval < 0
    r_in = rval < 0
    with np.errstate(divide="ignore", invalid="ignore"):
        diff = -np.abs(lval - rval)
        out = np.where(l_in & r_in, 1.0,
                       np.where(~l_in & ~r_in, 0.0,
                                np.where(l_in & ~r_in, lval / diff, rval / diff)))
    return out


def tri_in_fraction(v0, v1, v2):
    """solver/SolidFractionCommon.py:18-50, branch for branch.

    As written in the reference this is 1.0 iff all three vertices are < 0 and
    0.0 otherwise: with two vertices inside, lines 31-39 pick the
    two *inside* vertices (edge fraction 1 -> 1-1 = 0); with one inside, lines
    40-48 pick the two *outside* vertices (edge fraction 0).  The general
    selection logic is kept so the restatement does not depend on that reading.
    """
    v = np.stack(np.broadcast_arrays(np.asarray(v0, F64), np.asarray(v1, F64), np.asarray(v2, F64)), axis=-1)
    vin = v < 0
    cnt = vin.sum(axis=-1)
    v0_in, v1_in = vin[..., 0], vin[..., 1]
    # in_count == 2 (:31-39)
    out_v = np.where(v0_in, np.where(v1_in, 2, 1), 0)
    a = np.take_along_axis(v, ((out_v + 1) % 3)[..., None], axis=-1)[..., 0]
    b = np.take_along_axis(v, ((out_v + 2) % 3)[..., None], axis=-1)[..., 0]
    two = 1.0 - edge_in_fraction(a, b)
    # in_count == 1 (:40-48)
    in_v = np.where(~v0_in, np.where(~v1_in, 2, 1), 0)
    a = np.take_along_axis(v, ((in_v + 1) % 3)[..., None], axis=-1)[..., 0]
    b = np.take_along_axis(v, ((in_v + 2) % 3)[..., None], axis=-1)[..., 0]
    one = edge_in_fraction(a, b)
    return np.where(cnt == 3, 1.0, np.where(cnt == 2, two, np.where(cnt == 1, one, 0.0)))


def face_in_fraction(bl, br, tl, tr):
    """solver/SolidFractionCommon.py:52-60."""
    bl, br, tl, tr = (np.asarray(a, F64) for a in (bl, br, tl, tr))
    ce = 0.25 * (bl + br + tl + tr)
    return 0.25 * (tri_in_fraction(bl, br, ce) + tri_in_fraction(br, tr, ce)
                   + tri_in_fraction(tr, tl, ce) + tri_in_fraction(tl, bl, ce))


def compute_solid_frac3d(gres, sphi, wx, wy, wz):
    """solver/SolidFraction3D.py:6-32.  Writes w*[0:N] only; the upper faces
    w*[N] are never written (lines 21,23,25 are commented out) -> stay as given."""
    Nx, Ny, Nz = (int(g) for g in gres)
    s = np.asarray(sphi, F64)
    n = lambda ox, oy, oz: s[ox:ox + 2 * Nx:2, oy:oy + 2 * Ny:2, oz:oz + 2 * Nz:2]  # noqa: E731
    blb, brb, tlb, trb = n(0, 0, 0), n(2, 0, 0), n(0, 2, 0), n(2, 2, 0)
    blf, brf, tlf = n(0, 0, 2), n(2, 0, 2), n(0, 2, 2)
    wx[:Nx, :Ny, :Nz] = 1.0 - face_in_fraction(tlb, blb, tlf, blf)   # :22
    wy[:Nx, :Ny, :Nz] = 1.0 - face_in_fraction(brb, blb, brf, blf)   # :24
    wz[:Nx, :Ny, :Nz] = 1.0 - face_in_fraction(trb, tlb, brb, blb)   # :26


def compute_solid_frac2d(gres, sphi, wx, wy):
    """solver/SolidFraction2D.py:6-26: for x<Nx-1, y<Ny-1 writes BOTH faces of the
    cell with true linear edge fractions.  Written in launch order semantics: a
    face shared by two cells is written twice with the same value."""
    Nx, Ny = (int(g) for g in gres)
    s = np.asarray(sphi, F64)
    n = lambda ox, oy: s[ox:ox + 2 * (Nx - 1):2, oy:oy + 2 * (Ny - 1):2]  # noqa: E731
    bl, br, tl, tr = n(0, 0), n(2, 0), n(0, 2), n(2, 2)
    wx[1:Nx, 0:Ny - 1] = 1.0 - edge_in_fraction(tr, br)     # :17
    wx[0:Nx - 1, 0:Ny - 1] = 1.0 - edge_in_fraction(tl, bl)  # :18
    wy[0:Nx - 1, 1:Ny] = 1.0 - edge_in_fraction(tr, tl)     # :19
    wy[0:Nx - 1, 0:Ny - 1] = 1.0 - edge_in_fraction(br, bl)  # :20
    # overlap check: the value written to wx[x+1,y] by cell (x,y) equals the one
    # written to wx[x+1,y] by cell (x+1,y) as its own left face (same two nodes).


# =============================================================================
# pressure, 3D
# =============================================================================
def _theta(phi, nphi):
    """min(1, max(0.01, phi/(phi-nphi))) -- solver/PressureCGSolver3D.py:75."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.minimum(1.0, np.maximum(0.01, phi / (phi - nphi)))


def pressure_rhs3d(cell_size, gres, vx, vy, vz, sphi, sv, lphi, b, wx, wy, wz):
    """solver/PressureCGSolver3D.py:6-50 (initialize_solver_kernel)."""
    Nx, Ny, Nz = (int(g) for g in gres)
    if min(Nx, Ny, Nz) < 3:
        return                       # no interior cell: the kernel's every thread returns at :8-10
    cs = [float(c) for c in cell_size]
    I = (slice(1, Nx - 1), slice(1, Ny - 1), slice(1, Nz - 1))
    sv = np.asarray(sv, F64)

    def dgrid(ox, oy, oz, comp):   # sv[2x+ox, 2y+oy, 2z+oz, comp] over interior cells
        return sv[2 + ox:2 * (Nx - 1) + ox:2, 2 + oy:2 * (Ny - 1) + oy:2, 2 + oz:2 * (Nz - 1) + oz:2, comp]

    def sh(a, dx, dy, dz):
        return np.asarray(a)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy, 1 + dz:Nz - 1 + dz].astype(F64)

    bv = np.zeros((Nx - 2, Ny - 2, Nz - 2))
    terms = [  # (w, v, sign, cs, sv sample)  in the reference's order :20-47
        (sh(wx, 1, 0, 0), sh(vx, 1, 0, 0), +1, cs[0], dgrid(2, 1, 1, 0)),
        (sh(wx, 0, 0, 0), sh(vx, 0, 0, 0), -1, cs[0], dgrid(0, 1, 1, 0)),
        (sh(wy, 0, 1, 0), sh(vy, 0, 1, 0), +1, cs[1], dgrid(1, 2, 1, 1)),
        (sh(wy, 0, 0, 0), sh(vy, 0, 0, 0), -1, cs[1], dgrid(1, 0, 1, 1)),
        (sh(wz, 0, 0, 1), sh(vz, 0, 0, 1), +1, cs[2], dgrid(1, 1, 2, 2)),
        (sh(wz, 0, 0, 0), sh(vz, 0, 0, 0), -1, cs[2], dgrid(1, 1, 0, 2)),
    ]
    for w, v, sgn, c, s in terms:
        bv = bv + sgn * (w * v / c)
        bv = bv - sgn * np.where(w < 1, w * s / c, 0.0)
    fluid = np.asarray(lphi, F64)[I] < 0
    b[I] = np.where(fluid, bv, 0.0)


def pressure_apply3d(gres, v, out, wx, wy, wz, lphi):
    """solver/PressureCGSolver3D.py:52-130 (matvecmul_kernel).  Boundary cells of
    `out` are not written (:55-57); non-fluid interior cells get 0 (:60-63)."""
    Nx, Ny, Nz = (int(g) for g in gres)
    if min(Nx, Ny, Nz) < 3:
        return
    I = (slice(1, Nx - 1), slice(1, Ny - 1), slice(1, Nz - 1))

    def sh(a, dx, dy, dz):
        return np.asarray(a, F64)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy, 1 + dz:Nz - 1 + dz]

    phi = sh(lphi, 0, 0, 0)
    val = np.zeros_like(phi)
    diag = np.zeros_like(phi)
    nbrs = [  # (neighbour offset, face weight) in the reference's order: +x -x +y -y +z -z
        ((1, 0, 0), sh(wx, 1, 0, 0)), ((-1, 0, 0), sh(wx, 0, 0, 0)),
        ((0, 1, 0), sh(wy, 0, 1, 0)), ((0, -1, 0), sh(wy, 0, 0, 0)),
        ((0, 0, 1), sh(wz, 0, 0, 1)), ((0, 0, -1), sh(wz, 0, 0, 0)),
    ]
    for off, w in nbrs:
        nphi = sh(lphi, *off)
        nf = nphi < 0
        val = val - np.where(nf, w * sh(v, *off), 0.0)
        diag = diag + np.where(nf, w, w / _theta(phi, nphi))
    val = val + diag * sh(v, 0, 0, 0)
    out[I] = np.where(phi < 0, val, 0.0)


def pressure_update3d(gres, cell_size, vx, vy, vz, pv, wx, wy, wz, sv, lphi):
    """solver/PressureCGSolver3D.py:132-153 (apply_pressure_kernel): x,y,z in
    [1, N-1]; in-place, result cast to the velocity dtype (Q11)."""
    Nx, Ny, Nz = (int(g) for g in gres)
    cs = [float(c) for c in cell_size]
    lphi = np.asarray(lphi, F64)
    pv = np.asarray(pv, F64)
    sv = np.asarray(sv, F64)
    C = (slice(1, Nx), slice(1, Ny), slice(1, Nz))
    for axis, (vel, w, c) in enumerate(((vx, wx, cs[0]), (vy, wy, cs[1]), (vz, wz, cs[2]))):
        M = tuple(slice(1 - (axis == a), (Nx, Ny, Nz)[a] - (axis == a)) for a in range(3))
        pc, pm = lphi[C], lphi[M]
        act = (pc < 0) | (pm < 0)
        theta = np.minimum(1.0, np.maximum(0.01, edge_in_fraction(pc, pm)))
        # sv sample: (2x,2y+1,2z+1,0) / (2x+1,2y,2z+1,1) / (2x+1,2y+1,2z,2)
        sx = slice(2 + (axis != 0), 2 * Nx + (axis != 0), 2)
        sy = slice(2 + (axis != 1), 2 * Ny + (axis != 1), 2)
        sz = slice(2 + (axis != 2), 2 * Nz + (axis != 2), 2)
        svs = sv[sx, sy, sz, axis]
        wv = np.asarray(w, F64)[C]
        old = np.asarray(vel)[C].astype(F64)
        new = old + (pv[C] - pv[M]) * c / theta
        new = wv * new + (1 - wv) * svs
        vel[C] = np.where(act, new, old).astype(vel.dtype)


def cg(apply, b, x, d, r, q, tol, max_iter, history=None, raise_on_fail=True, dot=None):
    """The CG loop shared by all solvers: solver/PressureCGSolver3D.py:198-223
    (and solver/ViscosityCGSolver3D.py:575-612, solver/PressureCGSolver2D.py:159-177).

    `x,d,r,q,b` are arrays or tuples of arrays (viscosity: 3 components).  Plain
    un-preconditioned CG, ABSOLUTE tolerance delta < tol**2 tested after the x/r
    update and before the direction update (Q1,Q2); A.x0 is evaluated even when
    x0 = 0 (Q5).  Returns (iterations, delta, alpha, beta).
    history receives [delta0, dq1, delta1, dq2, delta2, ...].
    """
    tup = isinstance(x, (tuple, list))
    X, D, R, Q, B = ((a if tup else (a,)) for a in (x, d, r, q, b))
    if dot is None:
        def dot(A_, B_):
            s = None
            for a_, b_ in zip(A_, B_):
                t = np.sum(a_ * b_)
                s = t if s is None else s + t
            return float(s)
    apply(X, Q)
    for d_, b_, q_, r_ in zip(D, B, Q, R):
        d_[...] = b_ - q_
        r_[...] = d_
    delta = dot(R, R)
    if history is not None:
        history.append(delta)
    alpha = beta = 0.0
    it = 0
    if not delta < tol ** 2:
        converged = False
        for it in range(1, int(max_iter) + 1):
            apply(D, Q)
            dq = dot(D, Q)
            alpha = delta / dq
            for x_, d_, r_, q_ in zip(X, D, R, Q):
                x_ += alpha * d_
                r_ -= alpha * q_
            old = delta
            delta = dot(R, R)
            if history is not None:
                history.extend((dq, delta))
            if delta < tol ** 2:
                converged = True
                break
            beta = delta / old
            for d_, r_ in zip(D, R):
                d_[...] = r_ + beta * d_
        if not converged and raise_on_fail:
            raise ValueError("Failed to converge!")
    return it, delta, alpha, beta


class PressureCGSolver3D:
    """solver/PressureCGSolver3D.py:173-226 on numpy arrays."""

    def __init__(self, gres, bound_size):
        self.gres = tuple(int(g) for g in gres)
        self.cell_size = np.broadcast_to(np.asarray(bound_size, F64), (3,)) / np.asarray(self.gres, F64)
        Nx, Ny, Nz = self.gres
        self.d, self.r, self.q, self.b, self.x = (np.zeros(self.gres) for _ in range(5))
        self.wx = np.zeros((Nx + 1, Ny, Nz))
        self.wy = np.zeros((Nx, Ny + 1, Nz))
        self.wz = np.zeros((Nx, Ny, Nz + 1))
        self.max_iter = Nx * Ny * Nz
        self.history = []
        self.iterations = 0

    def solve(self, vx, vy, vz, sphi, sv, lphi, wx=None, wy=None, wz=None, tol=1e-3, max_iter=None,
              raise_on_fail=True):
        if wx is None or wy is None or wz is None:
            compute_solid_frac3d(self.gres, sphi, self.wx, self.wy, self.wz)
            wx, wy, wz = self.wx, self.wy, self.wz
        self.x *= 0.0
        pressure_rhs3d(self.cell_size, self.gres, vx, vy, vz, sphi, sv, lphi, self.b, wx, wy, wz)
        self.history = []
        ap = lambda V, O: pressure_apply3d(self.gres, V[0], O[0], wx, wy, wz, lphi)  # noqa: E731
        self.iterations, self.delta, self.alpha, self.beta = cg(
            ap, self.b, self.x, self.d, self.r, self.q, tol,
            self.max_iter if max_iter is None else max_iter, self.history, raise_on_fail)
        pressure_update3d(self.gres, self.cell_size, vx, vy, vz, self.x, wx, wy, wz, sv, lphi)


# =============================================================================
# pressure, 2D (config 1)
# =============================================================================
def pressure_rhs2d(cell_size, gres, vx, vy, sphi, sv, lphi, b, wx, wy):
    """solver/PressureCGSolver2D.py:6-44."""
    Nx, Ny = (int(g) for g in gres)
    cs = [float(c) for c in cell_size]
    sv = np.asarray(sv, F64)

    def sh(a, dx, dy):
        return np.asarray(a)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy].astype(F64)

    def dgrid(ox, oy, comp):
        return sv[2 + ox:2 * (Nx - 1) + ox:2, 2 + oy:2 * (Ny - 1) + oy:2, comp]

    bv = np.zeros((Nx - 2, Ny - 2))
    for w, v, sgn, c, s in [
        (sh(wx, 1, 0), sh(vx, 1, 0), +1, cs[0], dgrid(2, 1, 0)),
        (sh(wx, 0, 0), sh(vx, 0, 0), -1, cs[0], dgrid(0, 1, 0)),
        (sh(wy, 0, 1), sh(vy, 0, 1), +1, cs[1], dgrid(1, 2, 1)),
        (sh(wy, 0, 0), sh(vy, 0, 0), -1, cs[1], dgrid(1, 0, 1)),
    ]:
        bv = bv + sgn * (w * v / c)
        bv = bv - sgn * np.where(w < 1, w * s / c, 0.0)
    I = (slice(1, Nx - 1), slice(1, Ny - 1))
    b[I] = np.where(np.asarray(lphi, F64)[I] < 0, bv, 0.0)


def pressure_apply2d(gres, v, out, wx, wy, lphi):
    """solver/PressureCGSolver2D.py:46-100."""
    Nx, Ny = (int(g) for g in gres)

    def sh(a, dx, dy):
        return np.asarray(a, F64)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy]

    phi = sh(lphi, 0, 0)
    val = np.zeros_like(phi)
    diag = np.zeros_like(phi)
    for off, w in [((1, 0), sh(wx, 1, 0)), ((-1, 0), sh(wx, 0, 0)),
                   ((0, 1), sh(wy, 0, 1)), ((0, -1), sh(wy, 0, 0))]:
        nphi = sh(lphi, *off)
        nf = nphi < 0
        val = val - np.where(nf, w * sh(v, *off), 0.0)
        diag = diag + np.where(nf, w, w / _theta(phi, nphi))
    val = val + diag * sh(v, 0, 0)
    out[1:Nx - 1, 1:Ny - 1] = np.where(phi < 0, val, 0.0)


def pressure_update2d(gres, cell_size, vx, vy, pv, wx, wy, sv, lphi):
    """solver/PressureCGSolver2D.py:102-120."""
    Nx, Ny = (int(g) for g in gres)
    cs = [float(c) for c in cell_size]
    lphi = np.asarray(lphi, F64)
    pv = np.asarray(pv, F64)
    sv = np.asarray(sv, F64)
    C = (slice(1, Nx), slice(1, Ny))
    for axis, (vel, w, c) in enumerate(((vx, wx, cs[0]), (vy, wy, cs[1]))):
        M = tuple(slice(1 - (axis == a), (Nx, Ny)[a] - (axis == a)) for a in range(2))
        pc, pm = lphi[C], lphi[M]
        act = (pc < 0) | (pm < 0)
        theta = np.minimum(1.0, np.maximum(0.01, edge_in_fraction(pc, pm)))
        sx = slice(2 + (axis != 0), 2 * Nx + (axis != 0), 2)
        sy = slice(2 + (axis != 1), 2 * Ny + (axis != 1), 2)
        svs = sv[sx, sy, axis]
        wv = np.asarray(w, F64)[C]
        old = np.asarray(vel)[C].astype(F64)
        new = old + (pv[C] - pv[M]) * c / theta
        new = wv * new + (1 - wv) * svs
        vel[C] = np.where(act, new, old).astype(vel.dtype)


class PressureCGSolver2D:
    """solver/PressureCGSolver2D.py:140-179; no error on non-convergence (Q3)."""

    def __init__(self, gres, bound_size):
        self.gres = tuple(int(g) for g in gres)
        self.cell_size = np.broadcast_to(np.asarray(bound_size, F64), (2,)) / np.asarray(self.gres, F64)
        Nx, Ny = self.gres
        self.d, self.r, self.q, self.b, self.x = (np.zeros(self.gres) for _ in range(5))
        self.wx = np.zeros((Nx + 1, Ny))
        self.wy = np.zeros((Nx, Ny + 1))
        self.max_iter = Nx * Ny
        self.history = []
        self.iterations = 0

    def solve(self, vx, vy, sphi, sv, lphi, wx=None, wy=None, tol=1e-3):
        if wx is None or wy is None:
            compute_solid_frac2d(self.gres, sphi, self.wx, self.wy)
            wx, wy = self.wx, self.wy
        self.x *= 0.0
        pressure_rhs2d(self.cell_size, self.gres, vx, vy, sphi, sv, lphi, self.b, wx, wy)
        self.history = []
        ap = lambda V, O: pressure_apply2d(self.gres, V[0], O[0], wx, wy, lphi)  # noqa: E731
        self.iterations, self.delta, self.alpha, self.beta = cg(
            ap, self.b, self.x, self.d, self.r, self.q, tol, self.max_iter, self.history,
            raise_on_fail=False)
        pressure_update2d(self.gres, self.cell_size, vx, vy, self.x, wx, wy, sv, lphi)


# =============================================================================
# viscosity, 3D
# =============================================================================
# Tap tables of the three operator rows (each entry was
# re-read against solver/ViscosityCGSolver3D.py:248-456).  A row is described
# relative to its own face: D = doubled-grid index of the face,
#   u: D=(2x,2y+1,2z+1)   v: D=(2x+1,2y,2z+1)   w: D=(2x+1,2y+1,2z)
# vol samples: c=D, R/L = D -/+ x, T/B = D -/+ y, F/K = D -/+ z.
# tap = (factor, vol sample, sign in matvec, component, (dx,dy,dz) into that
#        component's array, mask offset relative to D)
_VOL_OFF = {"c": (0, 0, 0), "R": (1, 0, 0), "L": (-1, 0, 0), "T": (0, 1, 0), "B": (0, -1, 0),
            "F": (0, 0, 1), "K": (0, 0, -1)}
VISC_ROWS = {
    0: dict(D0=(0, 1, 1), diag=(2, 2, 1, 1, 1, 1), taps=[
        (2, "R", -1, 0, (1, 0, 0), (2, 0, 0)), (2, "L", -1, 0, (-1, 0, 0), (-2, 0, 0)),      # :272-276
        (1, "T", -1, 0, (0, 1, 0), (0, 2, 0)), (1, "B", -1, 0, (0, -1, 0), (0, -2, 0)),      # :278-282
        (1, "F", -1, 0, (0, 0, 1), (0, 0, 2)), (1, "K", -1, 0, (0, 0, -1), (0, 0, -2)),      # :284-288
        (1, "T", -1, 1, (0, 1, 0), (1, 1, 0)), (1, "T", +1, 1, (-1, 1, 0), (-1, 1, 0)),      # :291-295
        (1, "B", +1, 1, (0, 0, 0), (1, -1, 0)), (1, "B", -1, 1, (-1, 0, 0), (-1, -1, 0)),    # :297-301
        (1, "F", -1, 2, (0, 0, 1), (1, 0, 1)), (1, "F", +1, 2, (-1, 0, 1), (-1, 0, 1)),      # :304-308
        (1, "K", +1, 2, (0, 0, 0), (1, 0, -1)), (1, "K", -1, 2, (-1, 0, 0), (-1, 0, -1)),    # :310-314
    ]),
    1: dict(D0=(1, 0, 1), diag=(1, 1, 2, 2, 1, 1), taps=[
        (1, "R", -1, 1, (1, 0, 0), (2, 0, 0)), (1, "L", -1, 1, (-1, 0, 0), (-2, 0, 0)),      # :342-346
        (2, "T", -1, 1, (0, 1, 0), (0, 2, 0)), (2, "B", -1, 1, (0, -1, 0), (0, -2, 0)),      # :348-352
        (1, "F", -1, 1, (0, 0, 1), (0, 0, 2)), (1, "K", -1, 1, (0, 0, -1), (0, 0, -2)),      # :354-358
        (1, "R", -1, 0, (1, 0, 0), (1, 1, 0)), (1, "R", +1, 0, (1, -1, 0), (1, -1, 0)),      # :361-365
        (1, "L", +1, 0, (0, 0, 0), (-1, 1, 0)), (1, "L", -1, 0, (0, -1, 0), (-1, -1, 0)),    # :367-371
        (1, "F", -1, 2, (0, 0, 1), (0, 1, 1)), (1, "F", +1, 2, (0, -1, 1), (0, -1, 1)),      # :374-378
        (1, "K", +1, 2, (0, 0, 0), (0, 1, -1)), (1, "K", -1, 2, (0, -1, 0), (0, -1, -1)),    # :380-384
    ]),
    2: dict(D0=(1, 1, 0), diag=(1, 1, 1, 1, 2, 2), taps=[
        (1, "R", -1, 2, (1, 0, 0), (2, 0, 0)), (1, "L", -1, 2, (-1, 0, 0), (-2, 0, 0)),      # :412-416
        (1, "T", -1, 2, (0, 1, 0), (0, 2, 0)), (1, "B", -1, 2, (0, -1, 0), (0, -2, 0)),      # :418-422
        (2, "F", -1, 2, (0, 0, 1), (0, 0, 2)), (2, "K", -1, 2, (0, 0, -1), (0, 0, -2)),      # :424-428
        (1, "R", -1, 0, (1, 0, 0), (1, 0, 1)), (1, "R", +1, 0, (1, 0, -1), (1, 0, -1)),      # :431-435
        (1, "L", +1, 0, (0, 0, 0), (-1, 0, 1)), (1, "L", -1, 0, (0, 0, -1), (-1, 0, -1)),    # :437-441
        (1, "T", -1, 1, (0, 1, 0), (0, 1, 1)), (1, "T", +1, 1, (0, 1, -1), (0, 1, -1)),      # :444-448
        (1, "B", +1, 1, (0, 0, 0), (0, -1, 1)), (1, "B", -1, 1, (0, 0, -1), (0, -1, -1)),    # :450-454
    ]),
}


def _face_shape(gres, axis):
    s = [int(g) for g in gres]
    s[axis] += 1
    return tuple(s)


def _row_views(gres, axis):
    """helpers that slice doubled-grid arrays / component arrays over the
    interior faces (1 <= idx <= shape-2 on every axis) of row `axis`."""
    shp = _face_shape(gres, axis)
    cnt = tuple(s - 2 for s in shp)
    D0 = VISC_ROWS[axis]["D0"]
    base = tuple(2 * 1 + D0[a] for a in range(3))   # doubled index of interior face idx=1

    def dg(G, off):
        sl = tuple(slice(base[a] + off[a], base[a] + off[a] + 2 * cnt[a], 2) for a in range(3))
        return G[sl]

    def comp(A, off):
        sl = tuple(slice(1 + off[a], 1 + off[a] + cnt[a]) for a in range(3))
        return A[sl]

    I = tuple(slice(1, 1 + cnt[a]) for a in range(3))
    return dg, comp, I


def visc_extrapolate3d(gres, num_iter, vx, vy, vz, sphi):
    """solver/ViscosityCGSolver3D.py:8-39,472-502: `num_iter` Jacobi sweeps of the
    6-neighbour average into invalid (sphi<0) interior faces."""
    sphi = np.asarray(sphi, F64)
    valids = [sphi[0::2, 1::2, 1::2] >= 0, sphi[1::2, 0::2, 1::2] >= 0, sphi[1::2, 1::2, 0::2] >= 0]
    for _ in range(num_iter):
        for v, valid in zip((vx, vy, vz), valids):
            n = v.shape
            if min(n) < 3:
                continue             # no interior face
            I = (slice(1, n[0] - 1), slice(1, n[1] - 1), slice(1, n[2] - 1))
            val = np.zeros(tuple(s - 2 for s in n))
            count = np.zeros(val.shape, dtype=np.int64)
            for off in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                sl = tuple(slice(1 + o, s - 1 + o) for o, s in zip(off, n))
                m = valid[sl]
                val = val + np.where(m, v[sl], 0.0)
                count = count + m
            upd = (~valid[I]) & (count > 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                newv = np.where(upd, val / count, v[I])
            v[I] = newv          # Jacobi: neighbours read above are all old values
            nv = valid.copy()
            nv[I] = valid[I] | upd
            valid[...] = nv


def _visc_row(axis, gres, scale, mu, V, sphi, vol, rhs):
    if min(_face_shape(gres, axis)) < 3:
        return (slice(0, 0),) * 3, np.zeros((0, 0, 0))      # no interior face in this row
    dg, comp, I = _row_views(gres, axis)
    row = VISC_ROWS[axis]
    vs = {k: dg(vol, o) for k, o in _VOL_OFF.items()}
    own = comp(V[axis], (0, 0, 0))
    solid = dg(sphi, (0, 0, 0)) < 0
    if rhs:
        val = own * vs["c"]                                            # :61
    else:
        fR, fL, fT, fB, fF, fK = row["diag"]
        m = lambda f, a: a if f == 1 else f * a  # noqa: E731
        diag = vs["c"] + scale * mu * (m(fR, vs["R"]) + m(fL, vs["L"]) + m(fT, vs["T"])
                                       + m(fB, vs["B"]) + m(fF, vs["F"]) + m(fK, vs["K"]))   # :268
        val = diag * own
    for fac, vname, sgn, c, off, moff in row["taps"]:
        k = (2 * scale * mu) if fac == 2 else (scale * mu)
        term = k * vs[vname] * comp(V[c], off)
        ms = dg(sphi, moff)
        if rhs:      # Dirichlet side: neighbour face inside solid, opposite sign (:64-106)
            val = val - sgn * np.where(ms < 0, term, 0.0)
        else:
            val = val + sgn * np.where(ms >= 0, term, 0.0)
    return I, np.where(solid, 0.0, val)


def visc_rhs3d(gres, scale, mu, vx, vy, vz, sphi, sv, vol, b_x, b_y, b_z):
    """solver/ViscosityCGSolver3D.py:41-246,504-513.  `sv` is accepted and unused (Q12)."""
    V = (np.asarray(vx, F64), np.asarray(vy, F64), np.asarray(vz, F64))
    for axis, b in enumerate((b_x, b_y, b_z)):
        I, val = _visc_row(axis, gres, scale, mu, V, np.asarray(sphi, F64), np.asarray(vol, F64), True)
        b[I] = val


def visc_apply3d(gres, scale, mu, vx, vy, vz, out_x, out_y, out_z, sphi, vol):
    """solver/ViscosityCGSolver3D.py:248-456,515-524."""
    V = (np.asarray(vx, F64), np.asarray(vy, F64), np.asarray(vz, F64))
    for axis, o in enumerate((out_x, out_y, out_z)):
        I, val = _visc_row(axis, gres, scale, mu, V, np.asarray(sphi, F64), np.asarray(vol, F64), False)
        o[I] = val


def visc_writeback3d(gres, vx, vy, vz, out_x, out_y, out_z, sphi, sv=None):
    """solver/ViscosityCGSolver3D.py:458-470,526-530: x,y,z in [1,N-1]."""
    Nx, Ny, Nz = (int(g) for g in gres)
    sphi = np.asarray(sphi, F64)
    C = (slice(1, Nx), slice(1, Ny), slice(1, Nz))
    for axis, (v, o) in enumerate(((vx, out_x), (vy, out_y), (vz, out_z))):
        sx = slice(2 + (axis != 0), 2 * Nx + (axis != 0), 2)
        sy = slice(2 + (axis != 1), 2 * Ny + (axis != 1), 2)
        sz = slice(2 + (axis != 2), 2 * Nz + (axis != 2), 2)
        m = sphi[sx, sy, sz] >= 0
        v[C] = np.where(m, np.asarray(o, F64)[C], v[C].astype(F64)).astype(v.dtype)


class ViscosityCGSolver3D:
    """solver/ViscosityCGSolver3D.py:532-613 on numpy arrays."""

    def __init__(self, gres, bound_size):
        self.gres = tuple(int(g) for g in gres)
        self.cell_size = np.broadcast_to(np.asarray(bound_size, F64), (3,)) / np.asarray(self.gres, F64)
        self.cell_vol = float(np.prod(self.cell_size))
        self.vol = np.zeros(tuple(2 * g + 1 for g in self.gres))
        for nm in "drqxb":
            for a, c in enumerate("xyz"):
                setattr(self, f"{nm}_{c}", np.zeros(_face_shape(self.gres, a)))
        self.max_iter = int(np.prod(self.gres))
        self.history = []
        self.iterations = 0

    def solve(self, dt, mu, rho, vx, vy, vz, sphi, sv, lphi, lvol, tol=1e-3, max_iter=None,
              raise_on_fail=True):
        scale = dt / self.cell_vol / rho                     # :567
        self.vol[...] = np.asarray(lvol, F64) / (self.cell_vol * 0.125)   # :568
        self.x_x[...] = vx
        self.x_y[...] = vy
        self.x_z[...] = vz
        visc_extrapolate3d(self.gres, 3, self.x_x, self.x_y, self.x_z, sphi)
        visc_rhs3d(self.gres, scale, mu, self.x_x, self.x_y, self.x_z, sphi, sv, self.vol,
                   self.b_x, self.b_y, self.b_z)
        self.history = []
        ap = lambda V, O: visc_apply3d(self.gres, scale, mu, V[0], V[1], V[2], O[0], O[1], O[2], sphi, self.vol)  # noqa: E731
        X = (self.x_x, self.x_y, self.x_z)
        self.iterations, self.delta, self.alpha, self.beta = cg(
            ap, (self.b_x, self.b_y, self.b_z), X, (self.d_x, self.d_y, self.d_z),
            (self.r_x, self.r_y, self.r_z), (self.q_x, self.q_y, self.q_z), tol,
            self.max_iter if max_iter is None else max_iter, self.history, raise_on_fail)
        visc_writeback3d(self.gres, vx, vy, vz, self.x_x, self.x_y, self.x_z, sphi, sv)


# =============================================================================
# the notebook's grid kernels that bracket the two solves
# (3D_viscous_fluid_sim.ipynb code cells 5 and 7; "ipynb cN:L" = line L of code cell N)
# =============================================================================
def nb_extrapolate(gres, num_iter, vx, vy, vz, mx, my, mz):
    """ipynb c7:1-64 (`extrapolate`, called at ipynb:4652 with num_iter=2): the same Jacobi
    sweep as the viscosity solver's, with validity = grid mass > 0 instead of sphi >= 0."""
    valids = [np.asarray(mx) > 0, np.asarray(my) > 0, np.asarray(mz) > 0]
    for _ in range(num_iter):
        for v, valid in zip((vx, vy, vz), valids):
            n = v.shape
            if min(n) < 3:
                continue
            I = (slice(1, n[0] - 1), slice(1, n[1] - 1), slice(1, n[2] - 1))
            val = np.zeros(tuple(s - 2 for s in n))
            count = np.zeros(val.shape, dtype=np.int64)
            for off in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                sl = tuple(slice(1 + o, s - 1 + o) for o, s in zip(off, n))
                m = valid[sl]
                val = val + np.where(m, v[sl].astype(F64), 0.0)
                count = count + m
            upd = (~valid[I]) & (count > 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                v[I] = np.where(upd, val / count, v[I].astype(F64)).astype(v.dtype)
            nv = valid.copy()
            nv[I] = valid[I] | upd
            valid[...] = nv


def nb_boundary_condition(gres, gv, gm, sphi, sv, dx, dv):
    """ipynb c5:1-146 (`boundary_condition_{x,y,z}`): dv = -(component of the solid-relative
    velocity's inward normal part) * (1 - sphi/dx) on interior faces closer than dx to a solid;
    0 elsewhere (array-boundary faces included).  gv, gm, dv are (x,y,z) triples of face arrays.

    The mass-weighted averages multiply velocity by mass in the INPUT dtype (fp32 in the
    notebook, numba/numpy scalar promotion) before accumulating in fp64.  `min(0, s)` is
    `s if s < 0 else 0` (NaN -> 0, e.g. where the averaged mass is 0).
    Reference quirk (not reproduced): the kernels store `dv[x,y,z] = 0` BEFORE comparing x,y,z
    with the array shape, so threads of the rounded-up launch grid store out of bounds -- on
    the GPU some of those stores race with interior faces of the next plane.  The intended
    (race-free) result is restated here; tests/golden/make_goldens_notebook.py drops those stores.
    """
    sphi = np.asarray(sphi, F64)
    sv = np.asarray(sv, F64)
    D0 = ((0, 1, 1), (1, 0, 1), (1, 1, 0))
    # per component: the two other components and their 4 (ix, iy) sample offsets, as written in c5
    #   x: vy,gmy at (x-ix, y+iy, z) ; vz,gmz at (x-ix, y, z+iy)
    #   y: vx,gmx at (x+iz, y-iy, z) ; vz,gmz at (x, y-iy, z+iz)
    #   z: vx,gmx at (x+ix, y, z-iz) ; vy,gmy at (x, y+ix, z-iz)
    taps = {
        0: ((1, [(-ix, iy, 0) for ix in range(2) for iy in range(2)]), (2, [(-ix, 0, iy) for ix in range(2) for iy in range(2)])),
        1: ((0, [(iz, -iy, 0) for iy in range(2) for iz in range(2)]), (2, [(0, -iy, iz) for iy in range(2) for iz in range(2)])),
        2: ((0, [(ix, 0, -iz) for iz in range(2) for ix in range(2)]), (1, [(0, ix, -iz) for iz in range(2) for ix in range(2)])),
    }
    for a in range(3):
        n = gv[a].shape
        dv[a][...] = 0
        if min(n) < 3:
            continue
        cnt = tuple(s - 2 for s in n)
        I = tuple(slice(1, 1 + c) for c in cnt)

        def dg(G, off, comp=None):
            sl = tuple(slice(2 + D0[a][k] + off[k], 2 + D0[a][k] + off[k] + 2 * cnt[k], 2) for k in range(3))
            return G[sl] if comp is None else G[sl + (comp,)]

        def sh(A, off):
            return np.asarray(A)[tuple(slice(1 + off[k], 1 + off[k] + cnt[k]) for k in range(3))]

        ndist = dg(sphi, (0, 0, 0)) / dx
        vel = [None, None, None]
        vel[a] = sh(gv[a], (0, 0, 0)).astype(F64)
        with np.errstate(divide="ignore", invalid="ignore"):
            for comp, offs in taps[a]:
                msum = np.zeros(cnt)
                vsum = np.zeros(cnt)
                for off in offs:
                    m_ = sh(gm[comp], off)
                    v_ = sh(gv[comp], off)
                    msum = msum + m_.astype(F64)
                    vsum = vsum + (v_ * m_).astype(F64)       # product in the arrays' own dtype
                vel[comp] = vsum / msum
            rel = [vel[c] - dg(sv, (0, 0, 0), c) for c in range(3)]
            e = np.eye(3, dtype=int)
            sn = [dg(sphi, tuple(e[c])) - dg(sphi, tuple(-e[c])) for c in range(3)]
            sn_inv = 1.0 / (sn[0] ** 2 + sn[1] ** 2 + sn[2] ** 2)
            s = sn[0] * rel[0] + sn[1] * rel[1] + sn[2] * rel[2]
            proj = np.where(s < 0, s, 0.0) * sn[a] * sn_inv
            out = -proj * (1.0 - ndist)
        dv[a][I] = np.where(ndist >= 1, 0.0, out).astype(dv[a].dtype)


# =============================================================================
# density solver, 3D -- solver/DensityCGSolver3D.py
# =============================================================================
def _particle_cell(px, bound_min, cell_size, bias):
    """base index and |gx - x| / cell_size weights of every particle (:17-22 / :235-239)."""
    x = np.asarray(px, F64)
    bmin, cs, bias = (np.asarray(a, F64) for a in (bound_min, cell_size, bias))
    gi = np.floor((x - bmin) / cs - bias).astype(np.int64)
    gx = (gi + bias) * cs + bmin
    return gi, np.abs(gx - x) / cs


def _corner_weights(w, ix, iy, iz):
    cw = lambda i, wd: i + ((-1) ** i) * (1 - wd)  # noqa: E731
    return cw(ix, w[:, 0]) * cw(iy, w[:, 1]) * cw(iz, w[:, 2])


def density_splat3d(bound_min, cell_size, gres, px, pm, pvol, gm, gvol):
    """initialize_density_kernel :8-36 (order of the atomic adds is unspecified in the reference)."""
    Nx, Ny, Nz = (int(g) for g in gres)
    gi, w = _particle_cell(px, bound_min, cell_size, (0.5, 0.5, 0.5))
    m = np.asarray(pm, F64)
    for ix in (0, 1):
        for iy in (0, 1):
            for iz in (0, 1):
                cx = np.clip(gi[:, 0] + ix, 0, Nx - 1)
                cy = np.clip(gi[:, 1] + iy, 0, Ny - 1)
                cz = np.clip(gi[:, 2] + iz, 0, Nz - 1)
                weight = _corner_weights(w, ix, iy, iz)
                np.add.at(gm, (cx, cy, cz), weight * m)
                np.add.at(gvol, (cx, cy, cz), weight * float(pvol))


def _nonsolid_frac(gres, wx, wy, wz):
    Nx, Ny, Nz = (int(g) for g in gres)
    a = lambda A, dx, dy, dz: np.asarray(A, F64)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy, 1 + dz:Nz - 1 + dz]  # noqa: E731
    return (a(wx, 0, 0, 0) + a(wx, 1, 0, 0) + a(wy, 0, 0, 0) + a(wy, 0, 1, 0) + a(wz, 0, 0, 0) + a(wz, 0, 0, 1)) / 6


def density_fix_volume3d(cell_size, gres, lvol, gvol, sphi, lphi, wx, wy, wz):
    """fix_volume_kernel :38-86 (`lvol` is unused there, as in the reference)."""
    Nx, Ny, Nz = (int(g) for g in gres)
    if min(Nx, Ny, Nz) < 3:
        return
    cs = np.asarray(cell_size, F64)
    cvol, dx = float(np.prod(cs)), float(np.min(cs))
    I = (slice(1, Nx - 1), slice(1, Ny - 1), slice(1, Nz - 1))
    sh = lambda A, ox, oy, oz: np.asarray(A, F64)[1 + ox:Nx - 1 + ox, 1 + oy:Ny - 1 + oy, 1 + oz:Nz - 1 + oz]  # noqa: E731
    fluid_vol = np.asarray(gvol, F64)[I]
    near_solid = np.asarray(sphi, F64)[3:2 * Nx - 2:2, 3:2 * Ny - 2:2, 3:2 * Nz - 2:2] < dx
    internal = sh(lphi, 0, 0, 0) < 0
    for off in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        internal = internal & (sh(lphi, *off) < 0)
    fluid_vol = np.where(internal & ~near_solid, cvol, fluid_vol)
    gvol[I] = np.minimum(fluid_vol, cvol * _nonsolid_frac(gres, wx, wy, wz))


def density_rhs3d(rho0, dt, gres, cell_size, gm, gvol, lphi, wx, wy, wz, b):
    """initialize_solver_kernel :88-116."""
    Nx, Ny, Nz = (int(g) for g in gres)
    if min(Nx, Ny, Nz) < 3:
        return
    cvol = float(np.prod(np.asarray(cell_size, F64)))
    I = (slice(1, Nx - 1), slice(1, Ny - 1), slice(1, Nz - 1))
    solid_vol = (1 - _nonsolid_frac(gres, wx, wy, wz)) * cvol
    solid_mass = rho0 * solid_vol
    cell_mass = np.asarray(gm, F64)[I] + solid_mass
    cell_vol = np.asarray(gvol, F64)[I] + solid_vol
    frac = cell_mass / np.maximum(cell_vol, 1e-10) / rho0
    frac = np.where(cell_mass < 1e-10, 1.0, frac)
    frac = np.maximum(0.5, np.minimum(1.5, frac))
    b[I] = np.where(np.asarray(lphi, F64)[I] >= 0, 0.0, (1 - frac) / dt)


def density_apply3d(gres, v, out, wx, wy, wz, lphi):
    """matvecmul_kernel :118-207: diag counts 1 per fluid neighbour (1/theta per non-fluid one), and the
    -z tap is weighted by wz[x,y,z+1] (:184, kept as written)."""
    Nx, Ny, Nz = (int(g) for g in gres)
    if min(Nx, Ny, Nz) < 3:
        return
    I = (slice(1, Nx - 1), slice(1, Ny - 1), slice(1, Nz - 1))
    sh = lambda a, dx, dy, dz: np.asarray(a, F64)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy, 1 + dz:Nz - 1 + dz]  # noqa: E731
    phi = sh(lphi, 0, 0, 0)
    val = np.zeros_like(phi)
    diag = np.zeros_like(phi)
    nbrs = [((1, 0, 0), sh(wx, 1, 0, 0)), ((-1, 0, 0), sh(wx, 0, 0, 0)),
            ((0, 1, 0), sh(wy, 0, 1, 0)), ((0, -1, 0), sh(wy, 0, 0, 0)),
            ((0, 0, 1), sh(wz, 0, 0, 1)), ((0, 0, -1), sh(wz, 0, 0, 1))]
    for off, w in nbrs:
        nphi = sh(lphi, *off)
        nf = nphi < 0
        val = val - np.where(nf, w * sh(v, *off), 0.0)
        diag = diag + np.where(nf, 1.0, 1.0 / _theta(phi, nphi))
    val = val + diag * sh(v, 0, 0, 0)
    out[I] = np.where(phi < 0, val, 0.0)


def density_displacement3d(gres, dt, cell_size, dx, dy, dz, pv, lphi):
    """compute_displacement_kernel :209-222 (x,y,z in [1, N-1])."""
    Nx, Ny, Nz = (int(g) for g in gres)
    cs = np.asarray(cell_size, F64)
    P, L = np.asarray(pv, F64), np.asarray(lphi, F64)
    c = (slice(1, Nx), slice(1, Ny), slice(1, Nz))
    for out, ax in ((dx, 0), (dy, 1), (dz, 2)):
        m = tuple(slice(0, n - 1) if a == ax else slice(1, n) for a, n in enumerate((Nx, Ny, Nz)))
        th = np.minimum(1.0, np.maximum(0.01, edge_in_fraction(L[c], L[m])))
        out[c] = (P[c] - P[m]) * dt * cs[ax] / th


def density_advect3d(px, d, bound_min, cell_size, grid_bias, axis):
    """apply_displacement_kernel :224-253: px[:, axis] += trilinear sample of d (in the kernel's add order)."""
    s = d.shape
    gi, w = _particle_cell(px, bound_min, cell_size, grid_bias)
    pos = px[:, axis].copy()
    for ix in (0, 1):
        for iy in (0, 1):
            for iz in (0, 1):
                cx = np.clip(gi[:, 0] + ix, 0, s[0] - 1)
                cy = np.clip(gi[:, 1] + iy, 0, s[1] - 1)
                cz = np.clip(gi[:, 2] + iz, 0, s[2] - 1)
                pos = (pos + _corner_weights(w, ix, iy, iz) * np.asarray(d, F64)[cx, cy, cz]).astype(px.dtype)
    px[:, axis] = pos


class DensityCGSolver3D:
    """solver/DensityCGSolver3D.py:298-350 on numpy arrays."""

    def __init__(self, gres, bound_min, bound_size):
        self.gres = tuple(int(g) for g in gres)
        self.bound_min = np.asarray(bound_min, F64)
        self.cell_size = np.broadcast_to(np.asarray(bound_size, F64), (3,)) / np.asarray(self.gres, F64)
        Nx, Ny, Nz = self.gres
        self.d, self.r, self.q, self.b, self.x, self.m, self.vol = (np.zeros(self.gres) for _ in range(7))
        self.wx, self.dx = np.zeros((Nx + 1, Ny, Nz)), np.zeros((Nx + 1, Ny, Nz))
        self.wy, self.dy = np.zeros((Nx, Ny + 1, Nz)), np.zeros((Nx, Ny + 1, Nz))
        self.wz, self.dz = np.zeros((Nx, Ny, Nz + 1)), np.zeros((Nx, Ny, Nz + 1))
        self.max_iter = Nx * Ny * Nz
        self.history = []
        self.iterations = 0

    def solve(self, rho0, dt, px, pm, pvol, vx, vy, vz, sphi, sv, lphi, lvol, wx=None, wy=None, wz=None, tol=1e-3,
              max_iter=None, raise_on_fail=True):
        if wx is None or wy is None or wz is None:
            compute_solid_frac3d(self.gres, sphi, self.wx, self.wy, self.wz)
            wx, wy, wz = self.wx, self.wy, self.wz
        self.m *= 0
        self.vol *= 0
        self.x *= 0
        density_splat3d(self.bound_min, self.cell_size, self.gres, px, pm, pvol, self.m, self.vol)
        density_fix_volume3d(self.cell_size, self.gres, lvol, self.vol, sphi, lphi, wx, wy, wz)
        density_rhs3d(rho0, dt, self.gres, self.cell_size, self.m, self.vol, lphi, wx, wy, wz, self.b)
        self.history = []
        ap = lambda V, O: density_apply3d(self.gres, V[0], O[0], wx, wy, wz, lphi)  # noqa: E731
        self.iterations, self.delta, self.alpha, self.beta = cg(
            ap, self.b, self.x, self.d, self.r, self.q, tol,
            self.max_iter if max_iter is None else max_iter, self.history, raise_on_fail)
        density_displacement3d(self.gres, dt, self.cell_size, self.dx, self.dy, self.dz, self.x, lphi)
        density_advect3d(px, self.dx, self.bound_min, self.cell_size, (0, 0.5, 0.5), 0)
        density_advect3d(px, self.dy, self.bound_min, self.cell_size, (0.5, 0, 0.5), 1)
        density_advect3d(px, self.dz, self.bound_min, self.cell_size, (0.5, 0.5, 0), 2)


# =============================================================================
# notebook particle <-> grid transfers -- 3D_viscous_fluid_sim.ipynb code cells 2,3,4,6
# The kernels keep float32 locals (x, gx, disp, w); the roundings below follow numba's typing with the
# notebook's container dtypes: bound_min / grid biases float32, cell sizes float64, particle arrays float64,
# grid mass / velocity float32, level set / volume float64.
# =============================================================================
F32 = np.float32


def _nb_particle_cell(px, bound_min, cell_size, bias, centre_offset=None):
    """float32 position, base index, float32 grid position (code cell 2: `gi`, `gx`)."""
    x32 = np.asarray(px).astype(F32)
    bmin32, cs = np.asarray(bound_min, F32), np.asarray(cell_size, F64)
    t = (x32 - bmin32).astype(F64) / cs                       # float32 difference, float64 quotient
    if bias is not None:
        t = t - np.asarray(bias, F32).astype(F64)
    gi = np.floor(t).astype(np.int64)
    off = np.asarray(bias, F32).astype(F64) if centre_offset is None else centre_offset
    gx32 = ((gi + off) * cs + bmin32.astype(F64)).astype(F32)
    return x32, gi, gx32


def nb_p2g_scatter(px, pm, pv, pca, gm, gv, bound_min, gres, grid_bias, cell_size, axis):
    """p2g_particle (code cell 2): APIC scatter of mass and momentum of component `axis` to its faces."""
    Nx, Ny, Nz = (int(g) for g in gres)
    cs = np.asarray(cell_size, F64)
    x32, gi, gx32 = _nb_particle_cell(px, bound_min, cs, grid_bias)
    disp = gx32 - x32                                          # float32
    w = (np.abs(disp).astype(F64) / cs).astype(F32).astype(F64)
    v32 = np.asarray(pv).astype(F32)
    m, pca = np.asarray(pm, F64), np.asarray(pca, F64)
    d64 = disp.astype(F64)
    for ix in (0, 1):
        for iy in (0, 1):
            for iz in (0, 1):
                cx = np.clip(gi[:, 0] + ix, 0, Nx - 1)
                cy = np.clip(gi[:, 1] + iy, 0, Ny - 1)
                cz = np.clip(gi[:, 2] + iz, 0, Nz - 1)
                wx = ix + ((-1) ** ix) * (1 - w[:, 0])
                wy = iy + ((-1) ** iy) * (1 - w[:, 1])
                wz = iz + ((-1) ** iz) * (1 - w[:, 2])
                cv = ((d64[:, 0] + ix * cs[0]) * pca[:, 0] + (d64[:, 1] + iy * cs[1]) * pca[:, 1]
                      + (d64[:, 2] + iz * cs[2]) * pca[:, 2])
                weight = wx * wy * wz
                np.add.at(gm, (cx, cy, cz), (weight * m).astype(gm.dtype))
                np.add.at(gv, (cx, cy, cz), (weight * m * (v32[:, axis].astype(F64) + cv)).astype(gv.dtype))


def nb_p2g_normalize(gm, gv):
    """p2g_grid (code cell 2): momentum -> velocity where mass landed."""
    m = gm > 0
    gv[m] = gv[m] / gm[m]


def nb_g2p_gather(bound_min, gres, grid_bias, cell_size, axis, px, pv, pca, gv):
    """g2p_particle (code cell 3): trilinear velocity and its affine row, in the kernel's accumulation order."""
    Nx, Ny, Nz = (int(g) for g in gres)
    cs = np.asarray(cell_size, F64)
    x32, gi, gx32 = _nb_particle_cell(px, bound_min, cs, grid_bias)
    w = (np.abs(gx32 - x32).astype(F64) / cs).astype(F32).astype(F64)
    pca[:, :] = 0
    vel = np.zeros(len(x32))
    G = np.asarray(gv)
    for ix in (0, 1):
        for iy in (0, 1):
            for iz in (0, 1):
                cx = np.clip(gi[:, 0] + ix, 0, Nx - 1)
                cy = np.clip(gi[:, 1] + iy, 0, Ny - 1)
                cz = np.clip(gi[:, 2] + iz, 0, Nz - 1)
                wx = 1 - ix + (2 * ix - 1) * w[:, 0]
                wy = 1 - iy + (2 * iy - 1) * w[:, 1]
                wz = 1 - iz + (2 * iz - 1) * w[:, 2]
                g = G[cx, cy, cz].astype(F64)
                vel = vel + wx * wy * wz * g
                pca[:, 0] += (2 * ix - 1) * wy * wz * g / cs[0]
                pca[:, 1] += wx * (2 * iy - 1) * wz * g / cs[1]
                pca[:, 2] += wx * wy * (2 * iz - 1) * g / cs[2]
    pv[:, axis] = vel


def nb_fluid_levelset(px, phi, bound_min, cell_size, gdx, gres):
    """compute_fluid_levelset (code cell 4): phi = min over particles within +-2 cells of |centre - x| - r."""
    Nx, Ny, Nz = (int(g) for g in gres)
    cs = np.asarray(cell_size, F64)
    r = gdx * 0.5 * np.sqrt(3.0) * 1.02
    phi[...] = gdx * 3
    x32, gi, _ = _nb_particle_cell(px, bound_min, cs, None, centre_offset=0.5)
    bmin = np.asarray(bound_min, F32).astype(F64)
    x64 = x32.astype(F64)
    for dx in range(-2, 3):
        for dy in range(-2, 3):
            for dz in range(-2, 3):
                ii = np.stack([np.clip(gi[:, 0] + dx, 0, Nx - 1), np.clip(gi[:, 1] + dy, 0, Ny - 1),
                               np.clip(gi[:, 2] + dz, 0, Nz - 1)], axis=1)
                gip = ((ii + 0.5) * cs + bmin - x64).astype(F32)
                n = np.zeros(len(x32))
                for d in range(3):
                    n = n + (gip[:, d] * gip[:, d]).astype(F64)           # float32 product, float64 sum
                np.minimum.at(phi, (ii[:, 0], ii[:, 1], ii[:, 2]), n ** 0.5 - r)


def nb_fluid_volume(bound_min, cell_size, gres, px, pvol, gvol):
    """compute_fluid_volume (code cell 6): trilinear splat of the particle volume onto the doubled-grid nodes,
    clamped to the node's cell volume."""
    Nx, Ny, Nz = (int(g) for g in gres)
    cs = np.asarray(cell_size, F64)
    gvol[...] = 0.0
    x32, gi, gx32 = _nb_particle_cell(px, bound_min, cs, None, centre_offset=0.0)
    w = (np.abs(gx32 - x32).astype(F64) / cs).astype(F32).astype(F64)
    for ix in (0, 1):
        for iy in (0, 1):
            for iz in (0, 1):
                cx = np.clip(gi[:, 0] + ix, 0, Nx - 1)
                cy = np.clip(gi[:, 1] + iy, 0, Ny - 1)
                cz = np.clip(gi[:, 2] + iz, 0, Nz - 1)
                weight = ((ix + ((-1) ** ix) * (1 - w[:, 0])) * (iy + ((-1) ** iy) * (1 - w[:, 1]))
                          * (iz + ((-1) ** iz) * (1 - w[:, 2])))
                np.add.at(gvol, (cx, cy, cz), weight * float(pvol))
    np.minimum(gvol, float(np.prod(cs)), out=gvol)


# =============================================================================
# rigid-body signed distances -- solver/sdf3D.py
# rb: one (10,4) float64 block; see generate_rb :277-305.  Scalar per-point restatements (test sizes only).
# =============================================================================
def _sdf_to_body(rb, pos):
    """inv_rigid (:31-40) then matvecmul4 (:19-29), in their accumulation order."""
    T, Rm = rb[1:5], rb[5:9]
    out = np.zeros(3)
    for i in range(3):
        t3 = 0.0
        for j in range(3):
            t3 -= Rm[j, i] * T[j, 3]
        tmp = 0.0
        for j in range(3):
            tmp += Rm[j, i] * pos[j]
        tmp += t3
        out[i] = tmp
    return out


def _sdf_to_world(rb, prb):
    """mat_TR (:12-17) then matvecmul4."""
    T, Rm = rb[1:5], rb[5:9]
    out = np.zeros(3)
    for i in range(3):
        tmp = 0.0
        for j in range(3):
            tmp += Rm[i, j] * prb[j]
        tmp += T[i, 3]
        out[i] = tmp
    return out


def _sdf_eval_one(rb, pos):
    kind, flipped = int(rb[0, 0] // 2), bool(rb[0, 0] % 2)
    if kind == 0:                                            # sphere_eval :53-66
        d = pos - np.array([rb[1, 3], rb[2, 3], rb[3, 3]])
        sd = (d[0] ** 2 + d[1] ** 2 + d[2] ** 2) ** 0.5 - rb[0, 1]
    elif kind == 1:                                          # box_eval :86-108
        prb = _sdf_to_body(rb, pos)
        tmp, mx = 0.0, -100
        for i in range(3):
            disp = abs(prb[i]) - rb[0, 1 + i] / 2
            if disp > 0:
                tmp += disp ** 2
            if mx < disp:
                mx = disp
        sd = tmp ** 0.5
        if mx < 0:
            sd += mx
    else:                                                    # cylinder_eval :148-172 (y_clip: see csrc/mfs_sdf.hip header)
        prb = _sdf_to_body(rb, pos)
        hh = rb[0, 2] / 2
        y_clip = -hh if prb[1] < -hh else (hh if prb[1] > hh else prb[1])
        sd = (prb[0] ** 2 + prb[2] ** 2) ** 0.5 - rb[0, 1]
        cap = y_clip == hh or y_clip == -hh
        if sd < 0:
            sd = abs(y_clip - prb[1]) if cap else max(sd, prb[1] - hh, -(prb[1] + hh))
        elif cap:
            sd = (sd ** 2 + abs(y_clip - prb[1]) ** 2) ** 0.5
    return -sd if flipped else sd


def sdf_evaluate(rb_d, sd, vel, position):
    """evaluate :260-270 -> evaluate_kernel :218-239."""
    vel *= 0
    P = position.reshape(-1, 3)
    S, V = sd.reshape(-1), vel.reshape(-1, 3)
    for p in range(P.shape[0]):
        min_sd, idx = 100, 0
        for i in range(rb_d.shape[0]):
            d = _sdf_eval_one(rb_d[i], P[p].astype(F64))
            if d < min_sd:
                min_sd, idx = d, i
        S[p] = min_sd
        if min_sd <= 0:
            V[p, :] = rb_d[idx, -1, :3]


def sdf_project(rb_d, position):
    """project :272-278 -> project_kernel :241-258 (in place; every body in turn)."""
    for p in range(position.shape[0]):
        pos = position[p].astype(F64)
        for i in range(rb_d.shape[0]):
            rb = rb_d[i]
            kind, flipped = int(rb[0, 0] // 2), bool(rb[0, 0] % 2)
            if kind == 0:                                    # sphere_project :68-84
                c = np.array([rb[1, 3], rb[2, 3], rb[3, 3]])
                d = pos - c
                dist = (d[0] ** 2 + d[1] ** 2 + d[2] ** 2) ** 0.5
                sd = dist - rb[0, 1]
                if flipped:
                    sd = -sd
                if sd < 0:
                    pos = d / dist * rb[0, 1] + c
            elif kind == 1:                                  # box_project :110-146
                prb = _sdf_to_body(rb, pos)
                h = rb[0, 1:4] / 2
                in_out = int(np.sum((prb > h) | (prb < -h)))
                if flipped:                                  # `% 2 and ~(in_out)`: ~ is bitwise, always true (:126)
                    pos = _sdf_to_world(rb, np.clip(prb, -h, h))
                elif in_out == 0:
                    index, dist_xyz = 0, 100
                    for k in range(3):
                        if h[k] - prb[k] < dist_xyz:
                            dist_xyz, index = h[k] - prb[k], k * 2
                        if prb[k] + h[k] < dist_xyz:
                            dist_xyz, index = prb[k] + h[k], k * 2 + 1
                    prb[index // 2] += dist_xyz * (-1) ** (index % 2)
                    pos = _sdf_to_world(rb, prb)
            else:                                            # cylinder_project :174-216
                prb = _sdf_to_body(rb, pos)
                hh = rb[0, 2] / 2
                y_clip = -hh if prb[1] < -hh else (hh if prb[1] > hh else prb[1])
                dist = (prb[0] ** 2 + prb[2] ** 2) ** 0.5
                sd = dist - rb[0, 1]
                if flipped:
                    if abs(y_clip) == hh or sd > 0:
                        if sd >= 0:
                            prb[0], prb[2] = prb[0] / dist * rb[0, 1], prb[2] / dist * rb[0, 1]
                        prb[1] = y_clip
                    pos = _sdf_to_world(rb, prb)
                elif sd < 0 and abs(y_clip) != hh:
                    mv = max(sd, prb[1] - hh, -(prb[1] + hh))
                    if mv == sd:
                        prb[0], prb[2] = prb[0] / dist * rb[0, 1], prb[2] / dist * rb[0, 1]
                    elif mv == prb[1] - hh:
                        prb[1] = hh
                    else:
                        prb[1] = -hh
                    pos = _sdf_to_world(rb, prb)
            pos = pos.astype(position.dtype).astype(F64)     # the reference writes into the array row
        position[p] = pos


# =============================================================================
# opt-in Jacobi-preconditioned CG (an EXTRA of the MI355X build, csrc/mfs_pcg.hip "optional Jacobi
# preconditioning"; the reference's CG is unpreconditioned).  Oracle of that extra only.
# =============================================================================
def pressure_diag3d(gres, wx, wy, wz, lphi):
    """the diagonal of the pressure operator (the `diag` accumulated in solver/PressureCGSolver3D.py:59-126);
    0 on boundary and non-fluid cells."""
    Nx, Ny, Nz = (int(g) for g in gres)
    out = np.zeros((Nx, Ny, Nz))
    if min(Nx, Ny, Nz) < 3:
        return out
    sh = lambda a, dx, dy, dz: np.asarray(a, F64)[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy, 1 + dz:Nz - 1 + dz]  # noqa: E731
    phi = sh(lphi, 0, 0, 0)
    diag = np.zeros_like(phi)
    for off, w in (((1, 0, 0), sh(wx, 1, 0, 0)), ((-1, 0, 0), sh(wx, 0, 0, 0)), ((0, 1, 0), sh(wy, 0, 1, 0)),
                   ((0, -1, 0), sh(wy, 0, 0, 0)), ((0, 0, 1), sh(wz, 0, 0, 1)), ((0, 0, -1), sh(wz, 0, 0, 0))):
        nphi = sh(lphi, *off)
        diag = diag + np.where(nphi < 0, w, w / _theta(phi, nphi))
    out[1:-1, 1:-1, 1:-1] = np.where(phi < 0, diag, 0.0)
    return out


def cg_jacobi(apply, diag, b, x, d, r, q, tol, max_iter, history=None):
    """Jacobi-preconditioned CG with the reference's stopping rule (r.r < tol^2 after the x / r update):
    z = r / diag (0 where diag = 0), delta = r.z.  history = [rr0, dq1, rr1, ...] like `cg`."""
    zof = lambda rv: np.divide(rv, diag, out=np.zeros_like(rv), where=diag != 0)  # noqa: E731
    x *= 0.0
    apply((x,), (q,))
    r[...] = b - q
    d[...] = zof(r)
    rr = float(np.sum(r * r))
    delta = float(np.sum(r * d))
    if history is not None:
        history.append(rr)
    it = 0
    if not rr < tol ** 2:
        for it in range(1, int(max_iter) + 1):
            apply((d,), (q,))
            dq = float(np.sum(d * q))
            alpha = delta / dq
            x += alpha * d
            r -= alpha * q
            z = zof(r)
            rr, rz = float(np.sum(r * r)), float(np.sum(r * z))
            if history is not None:
                history.extend((dq, rr))
            if rr < tol ** 2:
                break
            beta = rz / delta
            delta = rz
            d[...] = z + beta * d
        else:
            raise ValueError("Failed to converge!")
    return it, rr


def visc_diag3d(gres, scale, mu, sphi, vol):
    """the diagonal of the viscosity operator (`diag` of solver/ViscosityCGSolver3D.py:268, :338, :408) on the three
    component arrays; 0 on solid and array-boundary faces.  (Operand of the build's opt-in Jacobi loop.)"""
    sphi, vol = np.asarray(sphi, F64), np.asarray(vol, F64)
    out = []
    for axis in range(3):
        d = np.zeros(_face_shape(gres, axis))
        if min(d.shape) >= 3:
            dg, comp, I = _row_views(gres, axis)
            vs = {k: dg(vol, o) for k, o in _VOL_OFF.items()}
            fR, fL, fT, fB, fF, fK = VISC_ROWS[axis]["diag"]
            m = lambda f, a: a if f == 1 else f * a  # noqa: E731
            diag = vs["c"] + scale * mu * (m(fR, vs["R"]) + m(fL, vs["L"]) + m(fT, vs["T"]) + m(fB, vs["B"])
                                           + m(fF, vs["F"]) + m(fK, vs["K"]))
            d[I] = np.where(dg(sphi, (0, 0, 0)) < 0, 0.0, diag)
        out.append(d)
    return out


def visc_cg_jacobi(gres, scale, mu, B, X, sphi, vol, tol, max_iter, history=None):
    """Jacobi-preconditioned restatement of the viscosity CG loop (solver/ViscosityCGSolver3D.py:575-612 with z = r / diag,
    delta = r.z; the stopping rule stays sum(r.r) < tol^2 after the x / r update).  B, X: lists of the three component
    arrays; X holds the initial guess and receives the solution.  history = [rr0, dq1, rr1, ...]."""
    D = visc_diag3d(gres, scale, mu, sphi, vol)
    shp = [b.shape for b in B]
    zof = lambda R: [np.divide(r, d, out=np.zeros_like(r), where=d != 0) for r, d in zip(R, D)]  # noqa: E731
    dot = lambda P, Q: sum(float(np.sum(p * q)) for p, q in zip(P, Q))  # noqa: E731
    Q = [np.zeros(sh) for sh in shp]
    visc_apply3d(gres, scale, mu, *X, *Q, sphi, vol)
    R = [b - q for b, q in zip(B, Q)]
    Dv = zof(R)
    rr, delta = dot(R, R), dot(R, Dv)
    if history is not None:
        history.append(rr)
    it = 0
    if not rr < tol ** 2:
        for it in range(1, int(max_iter) + 1):
            for q in Q:
                q[...] = 0.0
            visc_apply3d(gres, scale, mu, *Dv, *Q, sphi, vol)
            dq = dot(Dv, Q)
            alpha = delta / dq
            for x, d in zip(X, Dv):
                x += alpha * d
            for r, q in zip(R, Q):
                r -= alpha * q
            Z = zof(R)
            rr, rz = dot(R, R), dot(R, Z)
            if history is not None:
                history.extend((dq, rr))
            if rr < tol ** 2:
                break
            beta = rz / delta
            delta = rz
            Dv = [z + beta * d for z, d in zip(Z, Dv)]
        else:
            raise ValueError("Failed to converge!")
    return it, rr
