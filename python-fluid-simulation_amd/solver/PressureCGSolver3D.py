"""Drop-in for the reference's solver/PressureCGSolver3D.py on MI355X.

Same module functions, class, constructor and `solve` signature as the reference
(file:line cited per item), operating on PyTorch-ROCm tensors and calling the
hand-written HIP kernels of libmfs_hip.so through the C ABI (include/mfs.h).
There is no CPU path: every call needs the GPU and the built library.
"""
import numpy as np
import torch

from mfs import _lib, tensors as T
from mfs.pcg import PcgEngine
from .SolidFraction3D import compute_solid_frac, edge_in_fraction  # noqa: F401  (reference line 4)


def _faces(g, wx, wy, wz):
    wx = T.dev(wx, "wx", T.face_shape(g, 0))
    wy = T.dev(wy, "wy", T.face_shape(g, 1))
    wz = T.dev(wz, "wz", T.face_shape(g, 2))
    if not (wx.dtype == wy.dtype == wz.dtype):
        raise TypeError("wx, wy, wz must share a dtype")
    return wx, wy, wz


def _vels(g, vx, vy, vz):
    vx = T.dev(vx, "vx", T.face_shape(g, 0))
    vy = T.dev(vy, "vy", T.face_shape(g, 1))
    vz = T.dev(vz, "vz", T.face_shape(g, 2))
    if not (vx.dtype == vy.dtype == vz.dtype):
        raise TypeError("vx, vy, vz must share a dtype")
    return vx, vy, vz


def initialize_solver(cell_size, gres, vx, vy, vz, sphi, sv, lphi, b, wx, wy, wz):
    """RHS b of the pressure system (reference :155-159 -> kernel :6-50)."""
    g = T.as_gres(gres)
    vx, vy, vz = _vels(g, vx, vy, vz)
    wx, wy, wz = _faces(g, wx, wy, wz)
    sv = T.dev(sv, "sv", T.doubled_shape(g) + (3,))
    lphi = T.dev(lphi, "lphi", g)
    b = T.dev(b, "b", g)
    lib = _lib.load()
    _lib.check(lib.mfs_pressure_rhs3d(_lib.i64x(g), _lib.f64x(T.as_f64_list(cell_size, 3)),
                                      T.ptr(vx), T.ptr(vy), T.ptr(vz), T.code(vx), T.ptr(sv), T.code(sv),
                                      T.ptr(lphi), T.code(lphi), T.ptr(wx), T.ptr(wy), T.ptr(wz), T.code(wx),
                                      T.ptr(b), T.code(b), T.stream()), "mfs_pressure_rhs3d")


def matvecmul(gres, v, out, wx, wy, wz, lphi):
    """out = A v, the 7-point ghost-fluid operator (reference :161-165 -> kernel :52-130)."""
    g = T.as_gres(gres)
    v = T.dev(v, "v", g)
    out = T.dev(out, "out", g)
    if v.dtype != out.dtype:
        raise TypeError("v and out must share a dtype")
    wx, wy, wz = _faces(g, wx, wy, wz)
    lphi = T.dev(lphi, "lphi", g)
    lib = _lib.load()
    _lib.check(lib.mfs_pressure_apply3d(_lib.i64x(g), T.ptr(v), T.ptr(out), T.code(v), T.ptr(wx), T.ptr(wy),
                                        T.ptr(wz), T.code(wx), T.ptr(lphi), T.code(lphi), T.stream()),
               "mfs_pressure_apply3d")


def apply_pressure(gres, cell_size, vx, vy, vz, pv, wx, wy, wz, sv, lphi):
    """In-place velocity update from the solved pressure (reference :167-171 -> kernel :132-153)."""
    g = T.as_gres(gres)
    vx, vy, vz = _vels(g, vx, vy, vz)
    wx, wy, wz = _faces(g, wx, wy, wz)
    pv = T.dev(pv, "pv", g)
    sv = T.dev(sv, "sv", T.doubled_shape(g) + (3,))
    lphi = T.dev(lphi, "lphi", g)
    lib = _lib.load()
    _lib.check(lib.mfs_pressure_update3d(_lib.i64x(g), _lib.f64x(T.as_f64_list(cell_size, 3)),
                                         T.ptr(vx), T.ptr(vy), T.ptr(vz), T.code(vx), T.ptr(pv), T.code(pv),
                                         T.ptr(wx), T.ptr(wy), T.ptr(wz), T.code(wx), T.ptr(sv), T.code(sv),
                                         T.ptr(lphi), T.code(lphi), T.stream()), "mfs_pressure_update3d")


class PressureCGSolver3D:
    """Reference :173-226.  `PressureCGSolver3D(buf, gres, bound_size)`; a scalar
    `bound_size` broadcasts exactly like the reference's `bound_size / gres`
    (the notebook passes GDX, ipynb:778).

    Extras that do not change reference behaviour: `iterations` and `history`
    ([delta0, dq1, delta1, ...]) after a solve; `check_every` = CG iterations
    enqueued between host looks at the device-resident convergence flag.
    """

    def __init__(self, buf, gres, bound_size, check_every=32, jacobi=None):
        """`jacobi=True` (or MFS_JACOBI=1) switches on the build's opt-in Jacobi preconditioner: fewer iterations,
        same stopping rule -- but no longer the reference's iteration (its CG is unpreconditioned)."""
        self.gres = gres
        self._g = T.as_gres(gres)
        if len(self._g) != 3:
            raise ValueError("PressureCGSolver3D needs a 3D grid")
        self.cell_size = np.array(T.as_f64_list(bound_size, 3)) / np.array(self._g, dtype=np.float64)
        self.buf = buf
        dt, device = buf.b.dtype, buf.b.device
        self.x = torch.zeros(self._g, dtype=dt, device=device)
        self.wx = torch.zeros(T.face_shape(self._g, 0), dtype=dt, device=device)
        self.wy = torch.zeros(T.face_shape(self._g, 1), dtype=dt, device=device)
        self.wz = torch.zeros(T.face_shape(self._g, 2), dtype=dt, device=device)
        self.alpha = 0.0
        self.beta = 0.0
        self.delta = 0.0
        self.max_iter = int(np.prod(self._g))
        self.check_every = int(check_every)
        self.iterations = 0
        self._engine = PcgEngine(self._g, dt, device)
        if jacobi is not None:
            self._engine.set_jacobi(jacobi)

    @property
    def history(self):
        return self._engine.history()

    @property
    def history_truncated(self):
        """True if the last solve ran past the history buffer (8 191 iterations): `history` then holds its leading entries
        only; `iterations`, `delta`, `alpha`, `beta` are exact regardless (they come from the engine's scalar block)"""
        return self._engine.history_truncated()

    def solve(self, vx, vy, vz, sphi, sv, lphi, wx=None, wy=None, wz=None, tol=1e-3):
        g = self._g
        if wx is None or wy is None or wz is None:
            compute_solid_frac(self.gres, sphi, self.wx, self.wy, self.wz)
            wx, wy, wz = self.wx, self.wy, self.wz
        eng = self._engine
        with torch.cuda.device(self.x.device):
            initialize_solver(self.cell_size, g, vx, vy, vz, sphi, sv, lphi, self.buf.b, wx, wy, wz)
            eng.setup(lphi, wx, wy, wz)
            eng.bind(self.buf.b, self.x, self.buf.d, self.buf.r, self.buf.q)
            ok, self.iterations = eng.solve(tol, self.max_iter, self.check_every)
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            # self.x : -pressure * dt / rho / cell_vol
            apply_pressure(g, self.cell_size, vx, vy, vz, self.x, wx, wy, wz, sv, lphi)


class SlabPressureCGSolver3D(PressureCGSolver3D):
    """One rank of a multi-GPU `PressureCGSolver3D` (extension: the reference is single-GPU).

    The GLOBAL grid `gres` is cut into contiguous x-slabs (`mfs.dist.SlabPartition`: array axis 0,
    the slowest-varying one); one process per GPU.  Every array handed to this object is the rank's
    LOCAL slab including one ghost / boundary plane on each side:

        cell arrays   planes [lo, hi)        of the global arrays, (lo, hi) = self.part.local_range
        x-face arrays planes [lo, hi + 1)    (vx, wx);  vy / vz / wy / wz: planes [lo, hi)
        doubled grid  planes [2 lo, 2 hi + 1)  (sphi, sv)

    `buf` is a `CGSolverBuffer(self.local_gres(gres, world, rank))`.  `solve` has the reference's
    signature and semantics on the global problem: same RHS, same CG iterates up to the summation
    order of the two dot products, `ValueError("Failed to converge!")` after prod(global gres)
    iterations, velocities updated in place on the faces this rank owns (local x-faces 1 .. L-1 and
    the y/z faces of local planes 1 .. L-1; the shared faces are computed identically by both
    neighbours).  Collective: every rank calls `solve` in step.

    transport: "p2p" = xGMI stores from the solver's kernels into HIP-IPC windows (mfs/p2p.py),
    "rccl" = torch.distributed collectives per iteration, "auto" = p2p when its self-test passes
    on every rank, else rccl."""

    @staticmethod
    def local_gres(gres, world, rank):
        from mfs.dist import SlabPartition
        g = T.as_gres(gres)
        return (SlabPartition(g[0], world, rank).local_planes, g[1], g[2])

    def __init__(self, buf, gres, bound_size, dist, group=None, transport="auto", check_every=32):
        from mfs.dist import SlabCG, SlabPartition
        from mfs.p2p import P2PWindow
        gg = T.as_gres(gres)
        self.global_gres = gg
        self.dist, self.group = dist, group
        self.part = SlabPartition(gg[0], dist.get_world_size(group), dist.get_rank(group))
        lg = (self.part.local_planes, gg[1], gg[2])
        if tuple(buf.b.shape) != lg:
            raise ValueError(f"buf must be a CGSolverBuffer of this rank's slab {lg} (see local_gres)")
        super().__init__(buf, lg, bound_size, check_every)
        # cell size and iteration cap are the GLOBAL problem's (reference :176, :190)
        self.cell_size = np.array(T.as_f64_list(bound_size, 3)) / np.array(gg, dtype=np.float64)
        self.max_iter = int(np.prod(gg))
        if transport not in ("auto", "p2p", "rccl"):
            raise ValueError("transport must be auto, p2p or rccl")
        self.window = None
        if transport != "rccl":
            self.window = P2PWindow(dist, lg[1] * lg[2] * buf.b.element_size(), buf.b.device, group)
            if not self.window.ok:
                why, self.window = self.window.why, None
                if transport == "p2p":
                    raise _lib.MfsError(f"peer-to-peer transport unavailable: {why}")
        self._cg = SlabCG(self._engine, self.part, buf.d, dist, group, window=self.window, force_multi=True)
        self.transport = self._cg.mode

    def solve(self, vx, vy, vz, sphi, sv, lphi, wx=None, wy=None, wz=None, tol=1e-3):
        g = self._g
        if wx is None or wy is None or wz is None:
            compute_solid_frac(g, sphi, self.wx, self.wy, self.wz)
            wx, wy, wz = self.wx, self.wy, self.wz
        eng = self._engine
        with torch.cuda.device(self.x.device):
            initialize_solver(self.cell_size, g, vx, vy, vz, sphi, sv, lphi, self.buf.b, wx, wy, wz)
            self.buf.b[0].zero_()         # ghost / boundary planes carry no equation on this rank
            self.buf.b[-1].zero_()
            eng.setup(lphi, wx, wy, wz)
            eng.bind(self.buf.b, self.x, self.buf.d, self.buf.r, self.buf.q)
            ok, self.iterations = self._cg.solve(tol, self.max_iter, self.check_every)
            self.transport = "p2p" if getattr(self._cg, "_p2p_active", False) else "rccl"   # what this solve actually used
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            self._cg.exchange(self.x)     # the neighbours' edge planes of the pressure, for the faces next to them
            apply_pressure(g, self.cell_size, vx, vy, vz, self.x, wx, wy, wz, sv, lphi)

    def close(self):
        """COLLECTIVE: release the peer-to-peer window (if any)."""
        w, self.window = self.window, None
        if w is not None:
            self._engine.attach_p2p(None)
            w.close()
