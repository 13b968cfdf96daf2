"""Drop-in for the reference's solver/ViscosityCGSolver3D.py on MI355X.

Same module functions (`extrapolate`, `initialize_solver`, `matvecmul`,
`apply_viscosity`), class, constructor and `solve` signature as the reference,
on PyTorch-ROCm tensors, calling the HIP kernels of libmfs_hip.so through the C
ABI (include/mfs.h).  No CPU path.
"""
import numpy as np
import torch

from mfs import _lib, tensors as T
from mfs.vcg import VcgEngine


def _comps(g, a, b, c, names):
    out = [T.dev(t, n, T.face_shape(g, ax)) for ax, (t, n) in enumerate(zip((a, b, c), names))]
    if not (out[0].dtype == out[1].dtype == out[2].dtype):
        raise TypeError(f"{names} must share a dtype")
    return out


def extrapolate(gres, num_iter, vx, vy, vz, sphi):
    """`num_iter` Jacobi sweeps of the 6-neighbour average into solid faces, in place
    (reference :472-502 -> kernel :8-39)."""
    g = T.as_gres(gres)
    vx, vy, vz = _comps(g, vx, vy, vz, ("vx", "vy", "vz"))
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    lib = _lib.load()
    gi = _lib.i64x(g)
    nbytes = int(lib.mfs_visc_extrapolate3d_workspace_bytes(gi, T.code(vx)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=vx.device)
    _lib.check(lib.mfs_visc_extrapolate3d(gi, int(num_iter), T.ptr(vx), T.ptr(vy), T.ptr(vz), T.code(vx), T.ptr(sphi),
                                          T.code(sphi), T.ptr(ws), nbytes, T.stream()), "mfs_visc_extrapolate3d")


def initialize_solver(gres, scale, mu, vx, vy, vz, sphi, sv, vol, b_x, b_y, b_z):
    """viscosity RHS (reference :504-513 -> kernels :41-246).  `sv` is accepted and unused,
    exactly as in the reference (Q12)."""
    g = T.as_gres(gres)
    vx, vy, vz = _comps(g, vx, vy, vz, ("vx", "vy", "vz"))
    b_x, b_y, b_z = _comps(g, b_x, b_y, b_z, ("b_x", "b_y", "b_z"))
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    vol = T.dev(vol, "vol", T.doubled_shape(g))
    lib = _lib.load()
    _lib.check(lib.mfs_visc_rhs3d(_lib.i64x(g), float(scale), float(mu), T.ptr(vx), T.ptr(vy), T.ptr(vz), T.code(vx),
                                  T.ptr(sphi), T.code(sphi), T.ptr(vol), T.code(vol), T.ptr(b_x), T.ptr(b_y),
                                  T.ptr(b_z), T.code(b_x), T.stream()), "mfs_visc_rhs3d")


def matvecmul(gres, scale, mu, vx, vy, vz, out_x, out_y, out_z, sphi, vol):
    """the coupled 3-component viscosity operator (reference :515-524 -> kernels :248-456)."""
    g = T.as_gres(gres)
    vx, vy, vz = _comps(g, vx, vy, vz, ("vx", "vy", "vz"))
    out_x, out_y, out_z = _comps(g, out_x, out_y, out_z, ("out_x", "out_y", "out_z"))
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    vol = T.dev(vol, "vol", T.doubled_shape(g))
    lib = _lib.load()
    _lib.check(lib.mfs_visc_apply3d(_lib.i64x(g), float(scale), float(mu), T.ptr(vx), T.ptr(vy), T.ptr(vz),
                                    T.code(vx), T.ptr(out_x), T.ptr(out_y), T.ptr(out_z), T.code(out_x), T.ptr(sphi),
                                    T.code(sphi), T.ptr(vol), T.code(vol), T.stream()), "mfs_visc_apply3d")


def apply_viscosity(gres, vx, vy, vz, out_x, out_y, out_z, sphi, sv):
    """copy the solution into the non-solid faces of vx,vy,vz, in place (reference :526-530 -> :458-470)."""
    g = T.as_gres(gres)
    vx, vy, vz = _comps(g, vx, vy, vz, ("vx", "vy", "vz"))
    out_x, out_y, out_z = _comps(g, out_x, out_y, out_z, ("out_x", "out_y", "out_z"))
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    lib = _lib.load()
    _lib.check(lib.mfs_visc_writeback3d(_lib.i64x(g), T.ptr(vx), T.ptr(vy), T.ptr(vz), T.code(vx), T.ptr(out_x),
                                        T.ptr(out_y), T.ptr(out_z), T.code(out_x), T.ptr(sphi), T.code(sphi),
                                        T.stream()), "mfs_visc_writeback3d")


class ViscosityCGSolver3D:
    """Reference :532-613.  `ViscosityCGSolver3D(gres, bound_size)`;
    `solve(dt, mu, rho, vx, vy, vz, sphi, sv, lphi, lvol, tol=1e-3)`.

    The 15 solver-owned CG arrays keep the reference's names and shapes (`x_x`,
    `d_y`, ...); each is a view into one flat per-vector allocation so the vector
    phases of the CG run as single launches over all three components.
    Extras: `iterations`, `history`; `precision` / MFS_PRECISION selects fp32 state.
    """

    def __init__(self, gres, bound_size, precision=None, device=None, check_every=32, jacobi=None):
        """`jacobi=True` (or MFS_VISC_JACOBI=1) switches on the build's opt-in Jacobi preconditioner: far fewer iterations
        where partly filled cells make the operator's diagonal span orders of magnitude, the same solution to `tol`, but
        NOT the reference's residual history (its CG is unpreconditioned, :575-612)."""
        self.gres = gres
        self._g = T.as_gres(gres)
        if len(self._g) != 3:
            raise ValueError("ViscosityCGSolver3D needs a 3D grid")
        self.cell_size = np.array(T.as_f64_list(bound_size, 3)) / np.array(self._g, dtype=np.float64)
        self.cell_vol = float(np.prod(self.cell_size))
        dt = T.state_dtype(precision)
        device = torch.device("cuda" if device is None else device)
        self._engine = VcgEngine(self._g, dt, device)
        if jacobi is not None:
            self._engine.set_jacobi(jacobi)
        self.vol = torch.zeros(T.doubled_shape(self._g), dtype=torch.float64, device=device)
        self._flat = {}
        for nm in "drqxb":
            flat, views = self._engine.new_vector()
            self._flat[nm] = flat
            for c, v in zip("xyz", views):
                setattr(self, f"{nm}_{c}", v)
        self.alpha = 0.0
        self.beta = 0.0
        self.delta = 0.0
        self.max_iter = int(np.prod(self._g))
        self.check_every = int(check_every)
        self.iterations = 0

    @property
    def history(self):
        return self._engine.history()

    @property
    def history_truncated(self):
        """True if the last solve ran past the history buffer (8 191 iterations): `history` then holds its leading entries
        only; `iterations`, `delta`, `alpha`, `beta` are exact regardless (they come from the engine's scalar block)"""
        return self._engine.history_truncated()

    def solve(self, dt, mu, rho, vx, vy, vz, sphi, sv, lphi, lvol, tol=1e-3):
        g = self._g
        scale = dt / self.cell_vol / rho                                   # :567
        lvol = T.dev(lvol, "lvol", T.doubled_shape(g))
        vx, vy, vz = _comps(g, vx, vy, vz, ("vx", "vy", "vz"))
        eng = self._engine
        with torch.cuda.device(self.vol.device):
            torch.div(lvol, self.cell_vol * 0.125, out=self.vol)           # :568
            self.x_x.copy_(vx)                                             # :569-571 (dtype cast)
            self.x_y.copy_(vy)
            self.x_z.copy_(vz)
            extrapolate(g, 3, self.x_x, self.x_y, self.x_z, sphi)          # :573
            initialize_solver(g, scale, mu, self.x_x, self.x_y, self.x_z, sphi, sv, self.vol,
                              self.b_x, self.b_y, self.b_z)                # :574
            eng.setup(scale, mu, sphi, self.vol)
            f = self._flat
            eng.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
            ok, self.iterations = eng.solve(tol, self.max_iter, self.check_every)   # :575-612
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            apply_viscosity(g, vx, vy, vz, self.x_x, self.x_y, self.x_z, sphi, sv)   # :613


def _extrapolate_slab(g, num_iter, comps, sphi, slab):
    """`extrapolate` on one rank's slab: the reference's Jacobi sweeps (:483-502) read the 6 neighbours' values AND
    validity of the previous sweep, so both ghost planes of both travel after every sweep."""
    lib = _lib.load()
    gi = _lib.i64x(g)
    for c, v in enumerate(comps):
        tmp = torch.empty_like(v)
        va = torch.empty(v.shape, dtype=torch.uint8, device=v.device)
        vb = torch.empty_like(va)
        _lib.check(lib.mfs_visc_valid3d(gi, c, T.ptr(sphi), T.code(sphi), T.ptr(va), T.stream()), "mfs_visc_valid3d")
        cur, oth, mcur, moth = v, tmp, va, vb
        for _ in range(int(num_iter)):
            _lib.check(lib.mfs_visc_extrapolate_sweep3d(gi, c, T.ptr(cur), T.ptr(oth), T.code(v), T.ptr(mcur),
                                                        T.ptr(moth), T.stream()), "mfs_visc_extrapolate_sweep3d")
            slab.exchange([oth, moth])
            cur, oth, mcur, moth = oth, cur, moth, mcur
        if cur is not v:
            v.copy_(cur)


class SlabViscosityCGSolver3D(ViscosityCGSolver3D):
    """One rank of a multi-GPU `ViscosityCGSolver3D` (extension: the reference is single-GPU).

    The GLOBAL grid `gres` is cut into contiguous x-slabs (`mfs.dist.SlabPartition`, as for the pressure solve);
    one process per GPU.  Every array handed to `solve` is the rank's LOCAL slab with one ghost / boundary cell
    plane on each side: with (lo, hi) = self.part.local_range, vx = global planes [lo, hi + 1), vy / vz / lphi =
    [lo, hi), sphi / sv / lvol = doubled planes [2 lo, 2 hi + 1).  `solve` has the reference's signature and
    semantics on the global problem: same extrapolation, RHS and CG iterates (up to the summation order of the two
    dot products across ranks), `ValueError("Failed to converge!")` after prod(global gres) iterations, velocities
    written in place on the local planes (the ghost planes receive the neighbours' values).  Collective: every
    rank calls `solve` in step.  Per iteration the edge planes of the direction vector (three components) and two dot
    products cross ranks: as xGMI stores from the library's kernels into HIP-IPC windows ("p2p", default when the
    window self-test passes) or as torch.distributed collectives (RCCL; "rccl") -- mfs.dist.SlabVCG."""

    @staticmethod
    def local_gres(gres, world, rank):
        from mfs.dist import SlabPartition
        g = T.as_gres(gres)
        return (SlabPartition(g[0], world, rank).local_planes, g[1], g[2])

    def __init__(self, gres, bound_size, dist, group=None, precision=None, device=None, check_every=32, transport="auto",
                 jacobi=None):
        from mfs.dist import SlabPartition, SlabVCG
        from mfs.p2p import P2PWindow
        gg = T.as_gres(gres)
        self.global_gres = gg
        self.dist, self.group = dist, group
        self.part = SlabPartition(gg[0], dist.get_world_size(group), dist.get_rank(group))
        lg = (self.part.local_planes, gg[1], gg[2])
        super().__init__(lg, bound_size, precision, device, check_every, jacobi)     # (Jacobi: window transport only, see SlabVCG)
        # cell size, cell volume and the iteration cap are the GLOBAL problem's (reference :535-537, :564)
        self.cell_size = np.array(T.as_f64_list(bound_size, 3)) / np.array(gg, dtype=np.float64)
        self.cell_vol = float(np.prod(self.cell_size))
        self.max_iter = int(np.prod(gg))
        # transport of the CG loop: "p2p" = xGMI stores into HIP-IPC windows (mfs/p2p.py), "rccl" = torch.distributed
        # collectives per iteration, "auto" = p2p when its self-test passes on every rank
        if transport not in ("auto", "p2p", "rccl"):
            raise ValueError("transport must be auto, p2p or rccl")
        self.window = None
        if transport != "rccl":
            self.window = P2PWindow(dist, self._engine.edge_plane_bytes(), self.vol.device, group)
            if not self.window.ok:
                why, self.window = self.window.why, None
                if transport == "p2p":
                    raise _lib.MfsError(f"peer-to-peer transport unavailable: {why}")
        self._cg = SlabVCG(self._engine, self.part, (self.d_x, self.d_y, self.d_z), dist, group, window=self.window)
        self.transport = self._cg.mode

    def close(self):
        """release the window (collective, like construction)"""
        if self.window is not None:
            self.window.close()
            self.window = None

    def solve(self, dt, mu, rho, vx, vy, vz, sphi, sv, lphi, lvol, tol=1e-3):
        g = self._g
        L = g[0]
        scale = dt / self.cell_vol / rho
        lvol = T.dev(lvol, "lvol", T.doubled_shape(g))
        sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
        vx, vy, vz = _comps(g, vx, vy, vz, ("vx", "vy", "vz"))
        eng = self._engine
        with torch.cuda.device(self.vol.device):
            torch.div(lvol, self.cell_vol * 0.125, out=self.vol)
            self.x_x.copy_(vx)
            self.x_y.copy_(vy)
            self.x_z.copy_(vz)
            _extrapolate_slab(g, 3, (self.x_x, self.x_y, self.x_z), sphi, self._cg)
            initialize_solver(g, scale, mu, self.x_x, self.x_y, self.x_z, sphi, sv, self.vol,
                              self.b_x, self.b_y, self.b_z)
            # ghost planes carry no equation on this rank (their rows belong to the neighbour)
            if self.part.right is not None:
                self.b_x[L - 1].zero_()
            for t in (self.b_y, self.b_z):
                t[0].zero_()
                t[L - 1].zero_()
            eng.setup(scale, mu, sphi, self.vol)
            f = self._flat
            f["q"].zero_()
            eng.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
            ok, self.iterations = self._cg.solve(tol, self.max_iter, self.check_every)
            st = eng.poll()
            self.alpha, self.beta, self.delta = st["alpha"], st["beta"], st["delta"]
            if not ok:
                raise ValueError("Failed to converge!")
            apply_viscosity(g, vx, vy, vz, self.x_x, self.x_y, self.x_z, sphi, sv)
