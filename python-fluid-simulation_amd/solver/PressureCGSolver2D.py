"""Drop-in for the reference's solver/PressureCGSolver2D.py on MI355X (BASELINE config 1)."""
import ctypes as C

import numpy as np
import torch

from mfs import _lib, tensors as T
from .SolidFraction2D import compute_solid_frac, edge_in_fraction  # noqa: F401  (reference line 4)


def _args2(g, vx, vy, wx, wy):
    vx = T.dev(vx, "vx", T.face_shape(g, 0))
    vy = T.dev(vy, "vy", T.face_shape(g, 1))
    wx = T.dev(wx, "wx", T.face_shape(g, 0))
    wy = T.dev(wy, "wy", T.face_shape(g, 1))
    if vx.dtype != vy.dtype or wx.dtype != wy.dtype:
        raise TypeError("vx/vy and wx/wy must share dtypes")
    return vx, vy, wx, wy


def initialize_solver(cell_size, gres, vx, vy, sphi, sv, lphi, b, wx, wy):
    """reference :122-126 -> kernel :6-44"""
    g = T.as_gres(gres)
    vx, vy, wx, wy = _args2(g, vx, vy, wx, wy)
    sv = T.dev(sv, "sv", T.doubled_shape(g) + (2,))
    lphi = T.dev(lphi, "lphi", g)
    b = T.dev(b, "b", g)
    lib = _lib.load()
    _lib.check(lib.mfs_pressure_rhs2d(_lib.i64x(g), _lib.f64x(T.as_f64_list(cell_size, 2)), T.ptr(vx), T.ptr(vy),
                                      T.code(vx), T.ptr(sv), T.code(sv), T.ptr(lphi), T.code(lphi), T.ptr(wx),
                                      T.ptr(wy), T.code(wx), T.ptr(b), T.code(b), T.stream()), "mfs_pressure_rhs2d")


def matvecmul(gres, v, out, wx, wy, lphi):
    """reference :128-132 -> kernel :46-100"""
    g = T.as_gres(gres)
    v, out = T.dev(v, "v", g), T.dev(out, "out", g)
    if v.dtype != out.dtype:
        raise TypeError("v and out must share a dtype")
    wx, wy = T.dev(wx, "wx", T.face_shape(g, 0)), T.dev(wy, "wy", T.face_shape(g, 1))
    lphi = T.dev(lphi, "lphi", g)
    lib = _lib.load()
    _lib.check(lib.mfs_pressure_apply2d(_lib.i64x(g), T.ptr(v), T.ptr(out), T.code(v), T.ptr(wx), T.ptr(wy),
                                        T.code(wx), T.ptr(lphi), T.code(lphi), T.stream()), "mfs_pressure_apply2d")


def apply_pressure(gres, cell_size, vx, vy, pv, wx, wy, sv, lphi):
    """reference :134-138 -> kernel :102-120"""
    g = T.as_gres(gres)
    vx, vy, wx, wy = _args2(g, vx, vy, wx, wy)
    pv = T.dev(pv, "pv", g)
    sv = T.dev(sv, "sv", T.doubled_shape(g) + (2,))
    lphi = T.dev(lphi, "lphi", g)
    lib = _lib.load()
    _lib.check(lib.mfs_pressure_update2d(_lib.i64x(g), _lib.f64x(T.as_f64_list(cell_size, 2)), T.ptr(vx), T.ptr(vy),
                                         T.code(vx), T.ptr(pv), T.code(pv), T.ptr(wx), T.ptr(wy), T.code(wx),
                                         T.ptr(sv), T.code(sv), T.ptr(lphi), T.code(lphi), T.stream()),
               "mfs_pressure_update2d")


class PressureCGSolver2D:
    """Reference :140-179.  Unlike the 3D solver it does NOT raise when `max_iter` is
    exhausted (no for-else in the reference, Q3): it applies whatever pressure it has."""

    def __init__(self, buf, gres, bound_size, check_every=32):
        self.gres = gres
        self._g = T.as_gres(gres)
        if len(self._g) != 2:
            raise ValueError("PressureCGSolver2D needs a 2D grid")
        self.cell_size = np.array(T.as_f64_list(bound_size, 2)) / np.array(self._g, dtype=np.float64)
        self.buf = buf
        dt, device = buf.b.dtype, buf.b.device
        self.x = torch.zeros(self._g, dtype=dt, device=device)
        self.wx = torch.zeros(T.face_shape(self._g, 0), dtype=dt, device=device)
        self.wy = torch.zeros(T.face_shape(self._g, 1), dtype=dt, device=device)
        self.alpha = 0.0
        self.beta = 0.0
        self.delta = 0.0
        self.max_iter = int(np.prod(self._g))
        self.check_every = int(check_every)
        self.iterations = 0
        self.converged = False
        self._lib = _lib.load()
        code = _lib.MFS_F32 if dt == torch.float32 else _lib.MFS_F64
        gi = _lib.i64x(self._g)
        nbytes = int(self._lib.mfs_pcg2d_workspace_bytes(gi, code))
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self._lib.mfs_pcg2d_create(C.byref(h), gi, code, T.ptr(self._ws), nbytes, T.stream()),
                       "mfs_pcg2d_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.mfs_pcg2d_destroy(h)
            except Exception:
                pass

    @property
    def history(self):
        cap = int(self._lib.mfs_pcg3d_history_capacity())
        buf = np.empty(cap, dtype=np.float64)
        n = self._lib.mfs_pcg2d_history(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), cap, T.stream())
        _lib.check(int(n), "mfs_pcg2d_history")
        return buf[: int(n)].copy()

    @property
    def history_truncated(self):
        """True if the last solve ran past the history buffer (8 191 iterations); `iterations` / `delta` stay exact"""
        return 2 * int(self.iterations) + 1 > int(self._lib.mfs_pcg3d_history_capacity())

    def solve(self, vx, vy, sphi, sv, lphi, wx=None, wy=None, tol=1e-3):
        g, lib = self._g, self._lib
        if wx is None or wy is None:
            compute_solid_frac(self.gres, sphi, self.wx, self.wy)
            wx, wy = self.wx, self.wy
        with torch.cuda.device(self.x.device):
            initialize_solver(self.cell_size, g, vx, vy, sphi, sv, lphi, self.buf.b, wx, wy)
            lphi_t = T.dev(lphi, "lphi", g)
            wx_t, wy_t = T.dev(wx, "wx", T.face_shape(g, 0)), T.dev(wy, "wy", T.face_shape(g, 1))
            _lib.check(lib.mfs_pcg2d_setup(self._h, T.ptr(lphi_t), T.code(lphi_t), T.ptr(wx_t), T.ptr(wy_t),
                                           T.code(wx_t)), "mfs_pcg2d_setup")
            vecs = [T.dev(a, n, g) for a, n in ((self.buf.b, "b"), (self.x, "x"), (self.buf.d, "d"),
                                                (self.buf.r, "r"), (self.buf.q, "q"))]
            _lib.check(lib.mfs_pcg2d_bind(self._h, *[T.ptr(t) for t in vecs]), "mfs_pcg2d_bind")
            it = C.c_int64()
            st = _lib.check(lib.mfs_pcg2d_solve(self._h, float(tol), self.max_iter, self.check_every, T.stream(),
                                                C.byref(it)), "mfs_pcg2d_solve")
            self.iterations, self.converged = it.value, st == _lib.MFS_OK
            done = C.c_int()
            d_, a_, b_ = C.c_double(), C.c_double(), C.c_double()
            _lib.check(lib.mfs_pcg2d_poll(self._h, T.stream(), C.byref(it), C.byref(done), C.byref(d_), C.byref(a_),
                                          C.byref(b_)), "mfs_pcg2d_poll")
            self.alpha, self.beta, self.delta = a_.value, b_.value, d_.value
            # self.x : -pressure * dt / rho / cell_vol
            apply_pressure(g, self.cell_size, vx, vy, self.x, wx, wy, sv, lphi)
