"""Python owner of one `mfs_pcg3d` engine handle (include/mfs.h): allocates the
device workspace with torch, keeps it alive, and exposes the C entry points."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, tensors as T


class PcgEngine:
    def __init__(self, gres, dtype, device=None):
        self.lib = _lib.load()
        self.gres = T.as_gres(gres)
        if len(self.gres) != 3:
            raise ValueError("PcgEngine is 3D")
        self.dtype = T.state_dtype(dtype)
        self.code = _lib.MFS_F32 if self.dtype == torch.float32 else _lib.MFS_F64
        self.device = torch.device("cuda" if device is None else device)
        g = _lib.i64x(self.gres)
        nbytes = int(self.lib.mfs_pcg3d_workspace_bytes(g, self.code))
        if nbytes <= 0:
            raise _lib.MfsError("mfs_pcg3d_workspace_bytes returned 0")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mfs_pcg3d_create(C.byref(h), g, self.code, T.ptr(self.workspace), nbytes, T.stream()),
                       "mfs_pcg3d_create")
        self.h = h
        # the engine's scalar block is the first 128 bytes of the workspace
        self.scalars = self.workspace[: _lib.NSCALARS * 8].view(torch.float64)
        assert self.scalars.data_ptr() == self.lib.mfs_pcg3d_scalars(self.h)
        self._bound = None

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.mfs_pcg3d_destroy(h)
            except Exception:
                pass

    def tune(self, variant=2, xchunk=0, blocks_per_cu=2, nontemporal=-1):
        _lib.check(self.lib.mfs_pcg3d_tune(self.h, int(variant), int(xchunk), int(blocks_per_cu), int(nontemporal)),
                   "mfs_pcg3d_tune")

    def set_prefetch(self, planes):
        _lib.check(self.lib.mfs_pcg3d_set_prefetch(self.h, int(planes)), "mfs_pcg3d_set_prefetch")

    def set_fuse(self, on):
        _lib.check(self.lib.mfs_pcg3d_set_fuse(self.h, int(bool(on))), "mfs_pcg3d_set_fuse")

    def loop_info(self):
        b = int(self.lib.mfs_pcg3d_loop_info(self.h))
        return dict(fused_direction_update=bool(b & 1), deferred_x_update=bool(b & 2), jacobi=bool(b & 4),
                    resident=bool(b & 8))

    def set_sparse(self, on):
        """single-domain solves from 2 M cells: live-chunk r update + sparse work list of the fused stencil launches"""
        _lib.check(self.lib.mfs_pcg3d_set_sparse(self.h, int(bool(on))), "mfs_pcg3d_set_sparse")

    def sparse_info(self):
        import ctypes
        out = (ctypes.c_int64 * 4)()
        _lib.check(self.lib.mfs_pcg3d_sparse_info(self.h, T.stream(), out), "mfs_pcg3d_sparse_info")
        return dict(live_chunks=int(out[0]), chunks=int(out[1]), listed_pairs=int(out[2]), pairs=int(out[3]))

    def set_resident(self, on):
        """small grids: run each batch of iterations as one resident launch (None = auto: whenever the grid fits)"""
        _lib.check(self.lib.mfs_pcg3d_set_resident(self.h, -1 if on is None else int(bool(on))), "mfs_pcg3d_set_resident")

    def set_lean(self, on):
        """close each iteration at the top of the next stencil launch instead of in a reduction tail (None = auto)"""
        _lib.check(self.lib.mfs_pcg3d_set_lean(self.h, -1 if on is None else int(bool(on))), "mfs_pcg3d_set_lean")

    def set_defer_x(self, on):
        _lib.check(self.lib.mfs_pcg3d_set_defer_x(self.h, -1 if on is None else int(bool(on))), "mfs_pcg3d_set_defer_x")

    def finish(self):
        """after iterate() calls of one's own: settle the deferred x update / parked direction vector"""
        _lib.check(self.lib.mfs_pcg3d_finish(self.h, T.stream()), "mfs_pcg3d_finish")

    def set_jacobi(self, on):
        """opt-in Jacobi preconditioning (NOT the reference's CG: fewer iterations, different residual history)"""
        _lib.check(self.lib.mfs_pcg3d_set_jacobi(self.h, int(bool(on))), "mfs_pcg3d_set_jacobi")

    def set_compress(self, on):
        _lib.check(self.lib.mfs_pcg3d_set_compress(self.h, int(bool(on))), "mfs_pcg3d_set_compress")

    # -- once per solve ------------------------------------------------------
    def setup(self, lphi, wx, wy, wz):
        g = self.gres
        lphi = T.dev(lphi, "lphi", g)
        wx, wy, wz = (T.dev(w, n, T.face_shape(g, a)) for a, (w, n) in enumerate(((wx, "wx"), (wy, "wy"), (wz, "wz"))))
        if not (wx.dtype == wy.dtype == wz.dtype):
            raise TypeError("wx, wy, wz must share a dtype")
        _lib.check(self.lib.mfs_pcg3d_setup(self.h, T.ptr(lphi), T.code(lphi), T.ptr(wx), T.ptr(wy), T.ptr(wz),
                                            T.code(wx), T.stream()), "mfs_pcg3d_setup")

    def setup_density(self, lphi, wx, wy, wz):
        """the density solver's operator (solver/DensityCGSolver3D.py:118-207) instead of the pressure one"""
        g = self.gres
        lphi = T.dev(lphi, "lphi", g)
        wx, wy, wz = (T.dev(w, n, T.face_shape(g, a)) for a, (w, n) in enumerate(((wx, "wx"), (wy, "wy"), (wz, "wz"))))
        if not (wx.dtype == wy.dtype == wz.dtype):
            raise TypeError("wx, wy, wz must share a dtype")
        _lib.check(self.lib.mfs_pcg3d_setup_density(self.h, T.ptr(lphi), T.code(lphi), T.ptr(wx), T.ptr(wy), T.ptr(wz),
                                                    T.code(wx), T.stream()), "mfs_pcg3d_setup_density")

    def bind(self, b, x, d, r, q):
        ts = [T.dev(a, n, self.gres) for a, n in ((b, "b"), (x, "x"), (d, "d"), (r, "r"), (q, "q"))]
        for t in ts:
            if t.dtype != self.dtype:
                raise TypeError(f"CG vectors must be {self.dtype}, got {t.dtype}")
        _lib.check(self.lib.mfs_pcg3d_bind(self.h, *[T.ptr(t) for t in ts]), "mfs_pcg3d_bind")
        self._bound = ts          # keep the tensors alive while the engine points at them

    # -- the hot kernel on its own ---------------------------------------------
    def apply(self, v, out, x_begin=None, x_end=None):
        v = T.dev(v, "v", self.gres)
        out = T.dev(out, "out", self.gres)
        if v.dtype != self.dtype or out.dtype != self.dtype:
            raise TypeError(f"apply operands must be {self.dtype}")
        xb = 1 if x_begin is None else int(x_begin)
        xe = self.gres[0] - 1 if x_end is None else int(x_end)
        _lib.check(self.lib.mfs_pcg3d_apply(self.h, T.ptr(v), T.ptr(out), xb, xe, T.stream()), "mfs_pcg3d_apply")

    # -- CG ----------------------------------------------------------------------
    def begin(self, tol):
        _lib.check(self.lib.mfs_pcg3d_begin(self.h, float(tol), T.stream()), "mfs_pcg3d_begin")

    def iterate(self, n):
        """enqueue n iterations.  On large grids the loop defers the last `x += alpha d` (and may park `d` in the
        engine's partner buffer): call finish() before reading x or d.  solve() does that itself."""
        _lib.check(self.lib.mfs_pcg3d_iterate(self.h, int(n), T.stream()), "mfs_pcg3d_iterate")

    def native_apply(self):
        _lib.check(self.lib.mfs_pcg3d_native_apply(self.h, T.stream()), "mfs_pcg3d_native_apply")

    def native_finish(self):
        _lib.check(self.lib.mfs_pcg3d_native_finish(self.h, T.stream()), "mfs_pcg3d_native_finish")

    def poll(self):
        it, done = C.c_int64(), C.c_int()
        delta, alpha, beta = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.mfs_pcg3d_poll(self.h, T.stream(), C.byref(it), C.byref(done), C.byref(delta),
                                           C.byref(alpha), C.byref(beta)), "mfs_pcg3d_poll")
        return dict(iterations=it.value, done=bool(done.value), delta=delta.value, alpha=alpha.value,
                    beta=beta.value)

    def poll_raw(self):
        """the scalar block as it stands, WITHOUT raising on the loop's error word (diagnostics after a failed solve)"""
        s = self.scalars.cpu()
        return dict(iterations=int(s[_lib.S_ITERS]), done=bool(s[_lib.S_DONE] != 0), delta=float(s[_lib.S_LASTRR]),
                    err=int(s[_lib.S_ERR]))

    def solve(self, tol, max_iter, check_every=32):
        it = C.c_int64()
        st = _lib.check(self.lib.mfs_pcg3d_solve(self.h, float(tol), int(max_iter), int(check_every), T.stream(),
                                                 C.byref(it)), "mfs_pcg3d_solve")
        return st == _lib.MFS_OK, it.value

    def history(self):
        cap = int(self.lib.mfs_pcg3d_history_capacity())
        buf = np.empty(cap, dtype=np.float64)
        n = self.lib.mfs_pcg3d_history(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)), cap, T.stream())
        _lib.check(int(n), "mfs_pcg3d_history")
        return buf[: int(n)].copy()

    def history_truncated(self):
        """True when the solve ran past the history buffer (capacity mfs_pcg3d_history_capacity() doubles = 8 191 iterations):
        history() then holds the LEADING entries only -- `iterations`, `delta`, alpha and beta come from the engine's scalar
        block (poll()), never from the history, and stay exact"""
        cap = int(self.lib.mfs_pcg3d_history_capacity())
        return 2 * int(self.poll_raw()["iterations"]) + 1 > cap

    # -- single phases (multi-GPU driver) ------------------------------------------
    def phase_apply(self, xb, xe, first):
        _lib.check(self.lib.mfs_pcg3d_phase_apply(self.h, int(xb), int(xe), int(bool(first)), T.stream()),
                   "mfs_pcg3d_phase_apply")

    def phase_apply2(self, xb, xe, xb2, xe2, first):
        _lib.check(self.lib.mfs_pcg3d_phase_apply2(self.h, int(xb), int(xe), int(xb2), int(xe2), int(bool(first)),
                                                   T.stream()), "mfs_pcg3d_phase_apply2")

    def phase_reduce(self, which):
        _lib.check(self.lib.mfs_pcg3d_phase_reduce(self.h, int(which), T.stream()), "mfs_pcg3d_phase_reduce")

    def phase_update_xr(self):
        _lib.check(self.lib.mfs_pcg3d_phase_update_xr(self.h, T.stream()), "mfs_pcg3d_phase_update_xr")

    def phase_update_r(self):
        _lib.check(self.lib.mfs_pcg3d_phase_update_r(self.h, T.stream()), "mfs_pcg3d_phase_update_r")

    def phase_update_x(self):
        _lib.check(self.lib.mfs_pcg3d_phase_update_x(self.h, T.stream()), "mfs_pcg3d_phase_update_x")

    def phase_update_d(self):
        _lib.check(self.lib.mfs_pcg3d_phase_update_d(self.h, T.stream()), "mfs_pcg3d_phase_update_d")

    def begin_local(self, tol):
        _lib.check(self.lib.mfs_pcg3d_begin_local(self.h, float(tol), T.stream()), "mfs_pcg3d_begin_local")

    def begin_finish(self):
        _lib.check(self.lib.mfs_pcg3d_begin_finish(self.h, T.stream()), "mfs_pcg3d_begin_finish")

    # -- slab loop over a peer-to-peer window (multi-GPU; COLLECTIVE calls) -----------------
    def attach_p2p(self, window):
        _lib.check(self.lib.mfs_pcg3d_attach_p2p(self.h, window.h if window is not None else None),
                   "mfs_pcg3d_attach_p2p")
        self._window = window     # keep it alive while the engine points at it

    def attach_rccl(self, own_window, comm):
        """collective transport of the slab loop: this rank's own one-rank window + an mfs.rccl.RcclComm (None, None detaches)"""
        _lib.check(self.lib.mfs_pcg3d_attach_rccl(self.h, own_window.h if own_window is not None else None,
                                                  comm.h if comm is not None else None), "mfs_pcg3d_attach_rccl")
        self._window, self._rccl = own_window, comm

    def slab_supported(self):
        return bool(self.lib.mfs_pcg3d_slab_supported(self.h))

    def slab_set_aux(self, on):
        _lib.check(self.lib.mfs_pcg3d_slab_set_aux(self.h, int(bool(on))), "mfs_pcg3d_slab_set_aux")

    def slab_begin(self, tol):
        _lib.check(self.lib.mfs_pcg3d_slab_begin(self.h, float(tol), T.stream()), "mfs_pcg3d_slab_begin")

    def slab_iterate(self, n):
        _lib.check(self.lib.mfs_pcg3d_slab_iterate(self.h, int(n), T.stream()), "mfs_pcg3d_slab_iterate")

    def slab_solve(self, tol, max_iter, check_every=32):
        it = C.c_int64()
        st = _lib.check(self.lib.mfs_pcg3d_slab_solve(self.h, float(tol), int(max_iter), int(check_every), T.stream(),
                                                      C.byref(it)), "mfs_pcg3d_slab_solve")
        return st == _lib.MFS_OK, it.value
