"""Drop-ins for the notebook-local functions of 3D_viscous_fluid_sim.ipynb around the solves.
SURVEY.md section 8(f) rank 1: `extrapolate` (code cell 7, called at ipynb:4652) and
`apply_boundary_condition` (code cell 5, ipynb:4655); rank 3: the particle <-> grid transfers `p2g`
(code cell 2, ipynb:4604), `g2p` (cell 3, ipynb:4660), `compute_fluid_levelset` (cell 4, ipynb:4587)
and `compute_fluid_volume` (cell 6, ipynb:4588).  Same names and argument lists as the notebook's
definitions (its `edict` containers are any objects with the same attributes); PyTorch-ROCm tensors;
HIP kernels behind the C ABI.  (The gravity step, ipynb:4608, is `grid.y.v += -10 * dt` -- a tensor
expression.)"""
import math
import os

import numpy as np
import torch

from mfs import _lib, tensors as T


def extrapolate(gres, num_iter, vx, vy, vz, mx, my, mz):
    """`num_iter` Jacobi sweeps of the 6-neighbour average into faces that received no mass, in place."""
    g = T.as_gres(gres)
    vs = [T.dev(t, n, T.face_shape(g, a)) for a, (t, n) in enumerate(((vx, "vx"), (vy, "vy"), (vz, "vz")))]
    ms = [T.dev(t, n, T.face_shape(g, a)) for a, (t, n) in enumerate(((mx, "mx"), (my, "my"), (mz, "mz")))]
    if not (vs[0].dtype == vs[1].dtype == vs[2].dtype and ms[0].dtype == ms[1].dtype == ms[2].dtype):
        raise TypeError("velocity / mass components must share a dtype")
    lib = _lib.load()
    gi = _lib.i64x(g)
    nbytes = int(lib.mfs_visc_extrapolate3d_workspace_bytes(gi, T.code(vs[0])))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=vs[0].device)
    _lib.check(lib.mfs_grid_extrapolate3d(gi, int(num_iter), *[T.ptr(t) for t in vs], T.code(vs[0]),
                                          *[T.ptr(t) for t in ms], T.code(ms[0]), T.ptr(ws), nbytes, T.stream()),
               "mfs_grid_extrapolate3d")


def apply_boundary_condition(g, solid, dx):
    """`g` is the notebook's grid object (g.x.v, g.x.m, g.x.dv, ... per axis), `solid` its solid level set
    (solid.phi on the doubled grid, solid.v its velocity field), dx the grid spacing (GDX).
    Computes the free-slip corrections into g.*.dv and adds them to g.*.v, like the notebook."""
    shp = tuple(g.x.v.shape)
    gres = (shp[0] - 1, shp[1], shp[2])
    vs = [T.dev(t, n, T.face_shape(gres, a)) for a, (t, n) in enumerate(((g.x.v, "g.x.v"), (g.y.v, "g.y.v"), (g.z.v, "g.z.v")))]
    ms = [T.dev(t, n, T.face_shape(gres, a)) for a, (t, n) in enumerate(((g.x.m, "g.x.m"), (g.y.m, "g.y.m"), (g.z.m, "g.z.m")))]
    dvs = [T.dev(t, n, T.face_shape(gres, a)) for a, (t, n) in enumerate(((g.x.dv, "g.x.dv"), (g.y.dv, "g.y.dv"), (g.z.dv, "g.z.dv")))]
    sphi = T.dev(solid.phi, "solid.phi", T.doubled_shape(gres))
    sv = T.dev(solid.v, "solid.v", T.doubled_shape(gres) + (3,))
    for grp in (vs, ms, dvs):
        if not (grp[0].dtype == grp[1].dtype == grp[2].dtype):
            raise TypeError("the three components of a grid field must share a dtype")
    lib = _lib.load()
    _lib.check(lib.mfs_grid_boundary_condition3d(_lib.i64x(gres), *[T.ptr(t) for t in vs], T.code(vs[0]),
                                                 *[T.ptr(t) for t in ms], T.code(ms[0]), T.ptr(sphi), T.code(sphi),
                                                 T.ptr(sv), T.code(sv), float(dx), *[T.ptr(t) for t in dvs],
                                                 T.code(dvs[0]), T.stream()), "mfs_grid_boundary_condition3d")
    g.x.v += g.x.dv
    g.y.v += g.y.dv
    g.z.v += g.z.dv


# ------------------------------------------------------------------ particle <-> grid (rank 3)
def _particles(t, name, cols=3):
    t = T.dev(t, name)
    if t.dim() != 2 or t.shape[1] != cols:
        raise ValueError(f"{name}: expected shape (P, {cols}), got {tuple(t.shape)}")
    return t


def _f3(a):
    return _lib.f64x(T.as_f64_list(a, 3))


# ---- tile order of the particles (round 3): at millions of particles the scatters run one workgroup per tile of 8^3 cells
# with the tile's nodes in LDS (csrc/mfs_particles.hip).  The order is a permutation beside the particle arrays (particle i
# stays particle i), computed once per position update and cached on the particle container until p.x changes.
TILE_MIN_PARTICLES = int(os.environ.get("MFS_PARTICLE_TILE_MIN", "262144"))


_TILE_ORDERS = []      # [(key, perm, tile_start, work)], most recent first: the orders of the last two position tensors


def tile_order(p, gres, bound_min, cell_size):
    """(perm, tile_start) of the particle positions `p.x` (or the (P,3) tensor `p` itself) on the CELL grid `gres` -- int32
    device tensors -- or None below TILE_MIN_PARTICLES.  Cached by tensor identity and version: every consumer of one
    position update (level set, volume, density splat, p2g) shares one sort."""
    px = _particles(p.x if hasattr(p, "x") else p, "p.x")
    P = int(px.shape[0])
    if P < TILE_MIN_PARTICLES:
        return None
    g = tuple(int(v) for v in gres)
    geo = tuple(T.as_f64_list(bound_min, 3)) + tuple(T.as_f64_list(cell_size, 3))
    key = (px.data_ptr(), px._version, P, g, geo, str(px.device))
    for i, ent in enumerate(_TILE_ORDERS):
        if ent[0] == key:
            if i:
                _TILE_ORDERS.insert(0, _TILE_ORDERS.pop(i))
            return ent[1], ent[2]
    lib = _lib.load()
    nt = int(lib.mfs_particle_tiles3d(_lib.i64x(g)))
    bufs = None
    if len(_TILE_ORDERS) >= 2:           # recycle the older entry's buffers when they fit
        old = _TILE_ORDERS.pop()
        if old[1].numel() == P and old[2].numel() == nt + 1 and old[1].device == px.device:
            bufs = old[1:]
    if bufs is None:
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=px.device)  # noqa: E731
        bufs = (i32(P), i32(nt + 1), i32(2 * nt + P))
    perm, tstart, work = bufs
    _lib.check(lib.mfs_particle_tile_sort3d(_lib.i64x(g), _f3(bound_min), _f3(cell_size), T.ptr(px), T.code(px), P, T.ptr(perm),
                                            T.ptr(tstart), T.ptr(work), T.stream()), "mfs_particle_tile_sort3d")
    _TILE_ORDERS.insert(0, (key, perm, tstart, work))
    return perm, tstart


def _p2g_scatter_axis(lib, gres, g, gc, pc, axis, px, pm, pv, gm, gv, order):
    if order is None:
        _lib.check(lib.mfs_p2g_scatter3d(_lib.i64x(gres), _f3(g.bound_min), _f3(g.cell_size), _f3(gc.bias), axis,
                                         T.ptr(px), T.code(px), T.ptr(pm), T.code(pm), T.ptr(pv), T.code(pv), T.ptr(pc),
                                         T.code(pc), int(px.shape[0]), T.ptr(gm), T.ptr(gv), T.code(gm), T.stream()),
                   "mfs_p2g_scatter3d")
    else:
        _lib.check(lib.mfs_p2g_scatter3d_tiled(_lib.i64x(gres), _f3(g.bound_min), _f3(g.cell_size), _f3(gc.bias), axis,
                                               T.ptr(px), T.code(px), T.ptr(pm), T.code(pm), T.ptr(pv), T.code(pv), T.ptr(pc),
                                               T.code(pc), int(px.shape[0]), T.ptr(order[0]), T.ptr(order[1]), T.ptr(gm),
                                               T.ptr(gv), T.code(gm), T.stream()), "mfs_p2g_scatter3d_tiled")


def p2g(p, g):
    """Particle -> grid (code cell 2): APIC scatter of mass and momentum to the three face arrays, then
    momentum / mass.  `p`: num_particles, x, m, v, cx, cy, cz.  `g`: resolution, bound_min, cell_size and
    per axis g.x / g.y / g.z with m, v, bias.  The caller zeroes g.*.m and g.*.v first (ipynb:4597-4602)."""
    gres = T.as_gres(g.resolution)
    px, pv = _particles(p.x, "p.x"), _particles(p.v, "p.v")
    pm = T.dev(p.m, "p.m", (px.shape[0],))
    lib = _lib.load()
    comps = ((g.x, p.cx, 0), (g.y, p.cy, 1), (g.z, p.cz, 2))
    order = tile_order(p, gres, g.bound_min, g.cell_size)
    for gc, pc, axis in comps:
        pc = _particles(pc, "p.c" + "xyz"[axis])
        gm = T.dev(gc.m, "g.%s.m" % "xyz"[axis], T.face_shape(gres, axis))
        gv = T.dev(gc.v, "g.%s.v" % "xyz"[axis], T.face_shape(gres, axis))
        if gm.dtype != gv.dtype:
            raise TypeError("grid mass and velocity must share a dtype")
        _p2g_scatter_axis(lib, gres, g, gc, pc, axis, px, pm, pv, gm, gv, order)
    for gc, _, axis in comps:
        _lib.check(lib.mfs_p2g_normalize3d(int(gc.m.numel()), T.ptr(gc.m), T.ptr(gc.v), T.code(gc.m), T.stream()),
                   "mfs_p2g_normalize3d")


def p2g_scatter(p, g):
    """the scatter half of `p2g` alone (particle-sharded time step: the ranks' shares are added before the division)"""
    gres = T.as_gres(g.resolution)
    px, pv = _particles(p.x, "p.x"), _particles(p.v, "p.v")
    pm = T.dev(p.m, "p.m", (px.shape[0],))
    lib = _lib.load()
    if px.shape[0] == 0:
        return
    order = tile_order(p, gres, g.bound_min, g.cell_size)
    for gc, pc, axis in ((g.x, p.cx, 0), (g.y, p.cy, 1), (g.z, p.cz, 2)):
        pc = _particles(pc, "p.c" + "xyz"[axis])
        gm = T.dev(gc.m, "g.%s.m" % "xyz"[axis], T.face_shape(gres, axis))
        gv = T.dev(gc.v, "g.%s.v" % "xyz"[axis], T.face_shape(gres, axis))
        _p2g_scatter_axis(lib, gres, g, gc, pc, axis, px, pm, pv, gm, gv, order)


def p2g_normalize(g):
    """the division half of `p2g`: momentum / mass where mass landed"""
    lib = _lib.load()
    for gc in (g.x, g.y, g.z):
        _lib.check(lib.mfs_p2g_normalize3d(int(gc.m.numel()), T.ptr(gc.m), T.ptr(gc.v), T.code(gc.m), T.stream()),
                   "mfs_p2g_normalize3d")


def g2p(p, g):
    """Grid -> particle (code cell 3): p.v[:, axis] and the affine rows p.cx / p.cy / p.cz from g.*.v."""
    gres = T.as_gres(g.resolution)
    px, pv = _particles(p.x, "p.x"), _particles(p.v, "p.v")
    lib = _lib.load()
    for gc, pc, axis in ((g.x, p.cx, 0), (g.y, p.cy, 1), (g.z, p.cz, 2)):
        pc = _particles(pc, "p.c" + "xyz"[axis])
        gv = T.dev(gc.v, "g.%s.v" % "xyz"[axis], T.face_shape(gres, axis))
        _lib.check(lib.mfs_g2p_gather3d(_lib.i64x(gres), _f3(g.bound_min), _f3(g.cell_size), _f3(gc.bias), axis,
                                        T.ptr(px), T.code(px), T.ptr(pv), T.code(pv), T.ptr(pc), T.code(pc),
                                        int(px.shape[0]), T.ptr(gv), T.code(gv), T.stream()), "mfs_g2p_gather3d")


def compute_fluid_levelset(p, ls, gdx):
    """Particle level set on the cell grid (code cell 4): ls.phi = gdx * 3, then the atomic-min pass with
    radius gdx * 0.5 * sqrt(3) * 1.02."""
    gres = T.as_gres(ls.resolution)
    px = _particles(p.x, "p.x")
    phi = T.dev(ls.phi, "ls.phi", gres)
    r = gdx * 0.5 * math.sqrt(3.0) * 1.02
    phi.fill_(gdx * 3)
    lib = _lib.load()
    order = tile_order(p, gres, ls.bound_min, ls.cell_size)
    if order is not None:
        _lib.check(lib.mfs_fluid_levelset3d_tiled(_lib.i64x(gres), _f3(ls.bound_min), _f3(ls.cell_size), float(r), T.ptr(px),
                                                  T.code(px), int(px.shape[0]), T.ptr(order[0]), T.ptr(order[1]), T.ptr(phi),
                                                  T.code(phi), T.stream()), "mfs_fluid_levelset3d_tiled")
        return
    _lib.check(lib.mfs_fluid_levelset3d(_lib.i64x(gres), _f3(ls.bound_min), _f3(ls.cell_size), float(r), T.ptr(px),
                                        T.code(px), int(px.shape[0]), T.ptr(phi), T.code(phi), T.stream()),
               "mfs_fluid_levelset3d")


def compute_fluid_volume(p, fv, pvol):
    """Fluid volume on the doubled grid (code cell 6): zero, trilinear splat of `pvol`, clamp to the cell volume."""
    vres = T.as_gres(fv.resolution)
    px = _particles(p.x, "p.x")
    vol = T.dev(fv.vol, "fv.vol", vres)
    vol.zero_()
    lib = _lib.load()
    order = None
    if all(int(v) % 2 == 1 for v in vres):       # the doubled grid of a cell grid: tiles are tiles of cells
        gres = tuple((int(v) - 1) // 2 for v in vres)
        order = tile_order(p, gres, fv.bound_min, 2.0 * np.asarray(T.as_f64_list(fv.cell_size, 3)))
    if order is not None:
        _lib.check(lib.mfs_fluid_volume3d_tiled(_lib.i64x(vres), _f3(fv.bound_min), _f3(fv.cell_size), T.ptr(px), T.code(px),
                                                float(pvol), int(px.shape[0]), T.ptr(order[0]), T.ptr(order[1]), T.ptr(vol),
                                                T.code(vol), T.stream()), "mfs_fluid_volume3d_tiled")
        return
    _lib.check(lib.mfs_fluid_volume3d(_lib.i64x(vres), _f3(fv.bound_min), _f3(fv.cell_size), T.ptr(px), T.code(px),
                                      float(pvol), int(px.shape[0]), T.ptr(vol), T.code(vol), T.stream()),
               "mfs_fluid_volume3d")
