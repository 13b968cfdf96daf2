"""Drop-ins for the notebook-local grid functions that bracket the two solves in
3D_viscous_fluid_sim.ipynb: `extrapolate` (code cell 7, called at
ipynb:4652) and `apply_boundary_condition` (code cell 5, called at ipynb:4655).  Same names and
argument lists as the notebook's definitions; PyTorch-ROCm tensors; HIP kernels behind the C ABI.
(The gravity step between them, ipynb:4608, is `grid.y.v += -10 * dt` -- a tensor expression.)"""
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
