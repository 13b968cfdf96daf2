"""Drop-in for the reference's solver/DensityCGSolver3D.py on MI355X.

Same module functions, class, constructor and `solve` signature as the reference
(file:line cited per item), on PyTorch-ROCm tensors, calling the HIP kernels of
libmfs_hip.so through the C ABI (include/mfs.h).  The CG loop runs on the pressure
engine set up for the density operator (csrc/mfs_pcg.hip, mfs_pcg3d_setup_density):
device-resident scalars, no host sync per iteration.  No CPU path.
"""
import numpy as np
import torch

from mfs import _lib, tensors as T
from mfs.pcg import PcgEngine
from .SolidFraction3D import compute_solid_frac, edge_in_fraction  # noqa: F401  (reference line 6)


def _faces(g, wx, wy, wz, names=("wx", "wy", "wz")):
    wx = T.dev(wx, names[0], T.face_shape(g, 0))
    wy = T.dev(wy, names[1], T.face_shape(g, 1))
    wz = T.dev(wz, names[2], T.face_shape(g, 2))
    if not (wx.dtype == wy.dtype == wz.dtype):
        raise TypeError(f"{', '.join(names)} must share a dtype")
    return wx, wy, wz


def _particles(px):
    px = T.dev(px, "px")
    if px.dim() != 2 or px.shape[1] != 3:
        raise ValueError(f"px: expected shape (P, 3), got {tuple(px.shape)}")
    return px


def initialize_density(bound_min, cell_size, gres, px, pm, pvol, gm, gvol, sphi=None, lphi=None):
    """Scatter particle mass / volume to the cell centres (reference :255-260 -> kernel :8-36).
    `sphi`, `lphi` are accepted and unused, as in the reference."""
    g = T.as_gres(gres)
    px = _particles(px)
    pm = T.dev(pm, "pm", (px.shape[0],))
    gm, gvol = T.dev(gm, "gm", g), T.dev(gvol, "gvol", g)
    if gm.dtype != gvol.dtype:
        raise TypeError("gm and gvol must share a dtype")
    lib = _lib.load()
    # at millions of particles: one workgroup per tile of 8^3 cells, the tile's cells in LDS (csrc/mfs_density.hip) on the
    # tile order notebook_kernels keeps for the particle positions (same order as p2g / level set / volume use)
    import notebook_kernels as NK
    order = NK.tile_order(px, g, bound_min, cell_size)
    if order is not None:
        _lib.check(lib.mfs_density_splat3d_tiled(_lib.i64x(g), _lib.f64x(T.as_f64_list(bound_min, 3)),
                                                 _lib.f64x(T.as_f64_list(cell_size, 3)), T.ptr(px), T.code(px), T.ptr(pm),
                                                 T.code(pm), float(pvol), int(px.shape[0]), T.ptr(order[0]), T.ptr(order[1]),
                                                 T.ptr(gm), T.ptr(gvol), T.code(gm), T.stream()), "mfs_density_splat3d_tiled")
        return
    _lib.check(lib.mfs_density_splat3d(_lib.i64x(g), _lib.f64x(T.as_f64_list(bound_min, 3)),
                                       _lib.f64x(T.as_f64_list(cell_size, 3)), T.ptr(px), T.code(px), T.ptr(pm),
                                       T.code(pm), float(pvol), int(px.shape[0]), T.ptr(gm), T.ptr(gvol), T.code(gm),
                                       T.stream()), "mfs_density_splat3d")


def fix_volume(cell_size, gres, lvol, gvol, sphi, lphi, wx, wy, wz):
    """Clamp the splatted cell volume (reference :262-269 -> kernel :38-86); `lvol` is unused there."""
    g = T.as_gres(gres)
    gvol = T.dev(gvol, "gvol", g)
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    lphi = T.dev(lphi, "lphi", g)
    wx, wy, wz = _faces(g, wx, wy, wz)
    lib = _lib.load()
    _lib.check(lib.mfs_density_fix_volume3d(_lib.i64x(g), _lib.f64x(T.as_f64_list(cell_size, 3)), T.ptr(gvol),
                                            T.code(gvol), T.ptr(sphi), T.code(sphi), T.ptr(lphi), T.code(lphi),
                                            T.ptr(wx), T.ptr(wy), T.ptr(wz), T.code(wx), T.stream()),
               "mfs_density_fix_volume3d")


def initialize_solver(rho0, dt, gres, cell_size, gm, gvol, lphi, wx, wy, wz, b):
    """Right-hand side (reference :271-277 -> kernel :88-116)."""
    g = T.as_gres(gres)
    gm, gvol = T.dev(gm, "gm", g), T.dev(gvol, "gvol", g)
    if gm.dtype != gvol.dtype:
        raise TypeError("gm and gvol must share a dtype")
    lphi = T.dev(lphi, "lphi", g)
    wx, wy, wz = _faces(g, wx, wy, wz)
    b = T.dev(b, "b", g)
    lib = _lib.load()
    _lib.check(lib.mfs_density_rhs3d(_lib.i64x(g), float(rho0), float(dt), _lib.f64x(T.as_f64_list(cell_size, 3)),
                                     T.ptr(gm), T.ptr(gvol), T.code(gm), T.ptr(lphi), T.code(lphi), T.ptr(wx),
                                     T.ptr(wy), T.ptr(wz), T.code(wx), T.ptr(b), T.code(b), T.stream()),
               "mfs_density_rhs3d")


def matvecmul(gres, v, out, wx, wy, wz, lphi):
    """out = A v, the density solver's operator (reference :279-283 -> kernel :118-207)."""
    g = T.as_gres(gres)
    v, out = T.dev(v, "v", g), T.dev(out, "out", g)
    if v.dtype != out.dtype:
        raise TypeError("v and out must share a dtype")
    wx, wy, wz = _faces(g, wx, wy, wz)
    lphi = T.dev(lphi, "lphi", g)
    lib = _lib.load()
    _lib.check(lib.mfs_density_apply3d(_lib.i64x(g), T.ptr(v), T.ptr(out), T.code(v), T.ptr(wx), T.ptr(wy), T.ptr(wz),
                                       T.code(wx), T.ptr(lphi), T.code(lphi), T.stream()), "mfs_density_apply3d")


def compute_displacement(gres, dt, cell_size, dx, dy, dz, pv, lphi):
    """Face displacements from the solved field (reference :285-289 -> kernel :209-222)."""
    g = T.as_gres(gres)
    dx, dy, dz = _faces(g, dx, dy, dz, ("dx", "dy", "dz"))
    pv, lphi = T.dev(pv, "pv", g), T.dev(lphi, "lphi", g)
    lib = _lib.load()
    _lib.check(lib.mfs_density_displacement3d(_lib.i64x(g), float(dt), _lib.f64x(T.as_f64_list(cell_size, 3)),
                                              T.ptr(dx), T.ptr(dy), T.ptr(dz), T.code(dx), T.ptr(pv), T.code(pv),
                                              T.ptr(lphi), T.code(lphi), T.stream()), "mfs_density_displacement3d")


def apply_displacement(px, dx, bound_min, cell_size, grid_bias, axis):
    """px[:, axis] += trilinear sample of the face array (reference :291-296 -> kernel :224-253)."""
    px = _particles(px)
    dx = T.dev(dx, "dx")
    if dx.dim() != 3:
        raise ValueError("dx: expected a 3D face array")
    lib = _lib.load()
    _lib.check(lib.mfs_density_advect3d(T.ptr(px), T.code(px), int(px.shape[0]), T.ptr(dx), T.code(dx),
                                        _lib.i64x(tuple(dx.shape)), _lib.f64x(T.as_f64_list(bound_min, 3)),
                                        _lib.f64x(T.as_f64_list(cell_size, 3)), _lib.f64x(T.as_f64_list(grid_bias, 3)),
                                        int(axis), T.stream()), "mfs_density_advect3d")


class DensityCGSolver3D:
    """Reference :298-350.  `DensityCGSolver3D(buf, gres, bound_min, bound_size)`; shares the
    `CGSolverBuffer` with the pressure solver like the reference does (ipynb:777-779).
    `self.wx, self.wy, self.wz` are what the notebook hands to the pressure solve (ipynb:4648).
    Extras that do not change reference behaviour: `iterations`, `history`, `check_every`."""

    def __init__(self, buf, gres, bound_min, bound_size, check_every=32):
        self.gres = gres
        self._g = T.as_gres(gres)
        if len(self._g) != 3:
            raise ValueError("DensityCGSolver3D needs a 3D grid")
        self.bound_min = np.array(T.as_f64_list(bound_min, 3))
        self.cell_size = np.array(T.as_f64_list(bound_size, 3)) / np.array(self._g, dtype=np.float64)
        self.bias_x = np.array([0, 0.5, 0.5])
        self.bias_y = np.array([0.5, 0, 0.5])
        self.bias_z = np.array([0.5, 0.5, 0])
        self.buf = buf
        dt, device = buf.b.dtype, buf.b.device
        z = lambda shape: torch.zeros(shape, dtype=dt, device=device)  # noqa: E731
        self.m, self.vol, self.x = z(self._g), z(self._g), z(self._g)
        self.wx, self.wy, self.wz = (z(T.face_shape(self._g, a)) for a in range(3))
        self.dx, self.dy, self.dz = (z(T.face_shape(self._g, a)) for a in range(3))
        self.alpha = 0.0
        self.beta = 0.0
        self.delta = 0.0
        self.max_iter = int(np.prod(self._g))
        self.check_every = int(check_every)
        self.iterations = 0
        self._engine = PcgEngine(self._g, dt, device)

    @property
    def history(self):
        return self._engine.history()

    @property
    def history_truncated(self):
        """True if the last solve ran past the history buffer (8 191 iterations): `history` then holds its leading entries
        only; `iterations`, `delta`, `alpha`, `beta` are exact regardless (they come from the engine's scalar block)"""
        return self._engine.history_truncated()

    def solve(self, rho0, dt, px, pm, pvol, vx, vy, vz, sphi, sv, lphi, lvol, wx=None, wy=None, wz=None, tol=1e-3):
        g = self._g
        if wx is None or wy is None or wz is None:
            compute_solid_frac(self.gres, sphi, self.wx, self.wy, self.wz)
            wx, wy, wz = self.wx, self.wy, self.wz
        eng = self._engine
        with torch.cuda.device(self.x.device):
            self.m *= 0
            self.vol *= 0
            initialize_density(self.bound_min, self.cell_size, g, px, pm, pvol, self.m, self.vol, sphi, lphi)
            fix_volume(self.cell_size, g, lvol, self.vol, sphi, lphi, wx, wy, wz)
            initialize_solver(rho0, dt, g, self.cell_size, self.m, self.vol, lphi, wx, wy, wz, self.buf.b)
            eng.setup_density(lphi, wx, wy, wz)
            eng.bind(self.buf.b, self.x, self.buf.d, self.buf.r, self.buf.q)
            ok, self.iterations = eng.solve(tol, self.max_iter, self.check_every)     # x *= 0 happens inside (:320)
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            # self.x : -pressure * dt / rho / dx^2
            compute_displacement(g, dt, self.cell_size, self.dx, self.dy, self.dz, self.x, lphi)
            apply_displacement(px, self.dx, self.bound_min, self.cell_size, self.bias_x, 0)
            apply_displacement(px, self.dy, self.bound_min, self.cell_size, self.bias_y, 1)
            apply_displacement(px, self.dz, self.bound_min, self.cell_size, self.bias_z, 2)


class SlabDensityCGSolver3D(DensityCGSolver3D):
    """`DensityCGSolver3D` on N GPUs with REPLICATED particles (extension; one process per GPU): same constructor
    arguments plus `dist`, same `solve` signature on the GLOBAL arrays.  The CG loop -- all but a few milliseconds of the
    solve -- runs slab-decomposed along x (`mfs.dist.SlabCG`: window transport like the pressure solve, collectives as
    the fallback; the operator is the pressure stencil with the asymmetric -z tap); the particle splat, `fix_volume`,
    the RHS, the displacement and the particle update run replicated on every rank, exactly as on one GPU.  The splat's
    atomics make the replicas differ in the last bits, so rank 0's RHS is broadcast (one source of truth), and every rank
    receives every rank's owned planes of the solution.  Collective."""

    def __init__(self, buf, gres, bound_min, bound_size, dist, group=None, check_every=32, transport="auto"):
        from mfs.dist import SlabCG, SlabPartition
        from mfs.p2p import P2PWindow
        super().__init__(buf, gres, bound_min, bound_size, check_every)
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        g = self._g
        self.part = SlabPartition(g[0], self.world, self.rank)
        lo, hi = self.part.local_range
        lg = (hi - lo, g[1], g[2])
        dt, device = buf.b.dtype, buf.b.device
        self._lb, self._lx, self._ld, self._lr, self._lq = (torch.zeros(lg, dtype=dt, device=device) for _ in range(5))
        self._engine = PcgEngine(lg, dt, device)          # replaces the full-grid engine
        # transport as for the pressure solve: xGMI stores into HIP-IPC windows when the self-test passes, else collectives
        if transport not in ("auto", "p2p", "rccl"):
            raise ValueError("transport must be auto, p2p or rccl")
        self.window = None
        if transport != "rccl":
            self.window = P2PWindow(dist, lg[1] * lg[2] * buf.b.element_size(), device, group)
            if not self.window.ok:
                why, self.window = self.window.why, None
                if transport == "p2p":
                    raise _lib.MfsError(f"peer-to-peer transport unavailable: {why}")
        self._cg = SlabCG(self._engine, self.part, self._ld, dist, group, window=self.window, force_multi=True)
        self.transport = self._cg.mode

    def close(self):
        """release the window (collective, like construction)"""
        if self.window is not None:
            self.window.close()
            self.window = None

    def solve(self, rho0, dt, px, pm, pvol, vx, vy, vz, sphi, sv, lphi, lvol, wx=None, wy=None, wz=None, tol=1e-3):
        from mfs.dist import SlabPartition
        g = self._g
        if wx is None or wy is None or wz is None:
            compute_solid_frac(self.gres, sphi, self.wx, self.wy, self.wz)
            wx, wy, wz = self.wx, self.wy, self.wz
        eng, dist, group = self._engine, self.dist, self.group
        lo, hi = self.part.local_range
        lphi = T.dev(lphi, "lphi", g)
        wx, wy, wz = _faces(g, wx, wy, wz)
        with torch.cuda.device(self.x.device):
            self.m *= 0
            self.vol *= 0
            initialize_density(self.bound_min, self.cell_size, g, px, pm, pvol, self.m, self.vol, sphi, lphi)
            fix_volume(self.cell_size, g, lvol, self.vol, sphi, lphi, wx, wy, wz)
            initialize_solver(rho0, dt, g, self.cell_size, self.m, self.vol, lphi, wx, wy, wz, self.buf.b)
            if self.world > 1:
                dist.broadcast(self.buf.b, src=0, group=group)
            self._lb.copy_(self.buf.b[lo:hi])
            self._lb[0].zero_()          # ghost / boundary planes carry no equation on this rank
            self._lb[-1].zero_()
            eng.setup_density(lphi[lo:hi], wx[lo:hi + 1], wy[lo:hi], wz[lo:hi])
            eng.bind(self._lb, self._lx, self._ld, self._lr, self._lq)
            ok, self.iterations = self._cg.solve(tol, self.max_iter, self.check_every)
            self.transport = "p2p" if getattr(self._cg, "_p2p_active", False) else "rccl"
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            self.x.zero_()
            self.x[lo + 1:hi - 1] = self._lx[1:-1]
            if self.world > 1:
                for r in range(self.world):
                    a, b = SlabPartition(g[0], self.world, r).owned
                    dist.broadcast(self.x[a:b], src=r, group=group)
            compute_displacement(g, dt, self.cell_size, self.dx, self.dy, self.dz, self.x, lphi)
            apply_displacement(px, self.dx, self.bound_min, self.cell_size, self.bias_x, 0)
            apply_displacement(px, self.dy, self.bound_min, self.cell_size, self.bias_y, 1)
            apply_displacement(px, self.dz, self.bound_min, self.cell_size, self.bias_z, 2)


    def solve_sharded(self, bands, rho0, dt, px, pm, pvol, sphi, sv, lphi, lvol, wx=None, wy=None, wz=None, tol=1e-3,
                      reach=3, width=4):
        """The same solve with the particles SHARDED: `px`, `pm` are this rank's own particles (those in its x-range,
        `bands`: mfs.dist.SlabBands); every global-shaped array is maintained on this rank's planes plus `width` ghost
        planes only.  The splat's contributions beyond the range go to their owners (band reduce), the CG loop runs on
        the slab as in `solve`, the solution's ghost planes come from their owners, and the displacement moves the own
        particles.  No whole-grid collective.  Collective."""
        g = self._g
        if wx is None or wy is None or wz is None:
            compute_solid_frac(self.gres, sphi, self.wx, self.wy, self.wz)       # static scene data: every rank, whole grid
            wx, wy, wz = self.wx, self.wy, self.wz
        eng = self._engine
        lo, hi = self.part.local_range
        lphi = T.dev(lphi, "lphi", g)
        wx, wy, wz = _faces(g, wx, wy, wz)
        with torch.cuda.device(self.x.device):
            self.m *= 0
            self.vol *= 0
            initialize_density(self.bound_min, self.cell_size, g, px, pm, pvol, self.m, self.vol, sphi, lphi)
            bands.reduce([self.m, self.vol], "cell", reach, "sum")
            bands.ghosts([self.m, self.vol], "cell", width)
            fix_volume(self.cell_size, g, lvol, self.vol, sphi, lphi, wx, wy, wz)
            initialize_solver(rho0, dt, g, self.cell_size, self.m, self.vol, lphi, wx, wy, wz, self.buf.b)
            self._lb.copy_(self.buf.b[lo:hi])
            self._lb[0].zero_()          # ghost / boundary planes carry no equation on this rank
            self._lb[-1].zero_()
            eng.setup_density(lphi[lo:hi], wx[lo:hi + 1], wy[lo:hi], wz[lo:hi])
            eng.bind(self._lb, self._lx, self._ld, self._lr, self._lq)
            ok, self.iterations = self._cg.solve(tol, self.max_iter, self.check_every)
            self.transport = "p2p" if getattr(self._cg, "_p2p_active", False) else "rccl"
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            self.x.zero_()
            self.x[lo + 1:hi - 1] = self._lx[1:-1]
            bands.ghosts([self.x], "cell", width)
            compute_displacement(g, dt, self.cell_size, self.dx, self.dy, self.dz, self.x, lphi)
            apply_displacement(px, self.dx, self.bound_min, self.cell_size, self.bias_x, 0)
            apply_displacement(px, self.dy, self.bound_min, self.cell_size, self.bias_y, 1)
            apply_displacement(px, self.dz, self.bound_min, self.cell_size, self.bias_z, 2)
