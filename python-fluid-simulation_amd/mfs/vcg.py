"""Python owner of one `mfs_vcg3d` engine handle (include/mfs.h): the viscosity CG."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, tensors as T


class VcgEngine:
    def __init__(self, gres, dtype, device=None):
        self.lib = _lib.load()
        self.gres = T.as_gres(gres)
        if len(self.gres) != 3:
            raise ValueError("VcgEngine is 3D")
        self.dtype = T.state_dtype(dtype)
        self.code = _lib.MFS_F32 if self.dtype == torch.float32 else _lib.MFS_F64
        self.device = torch.device("cuda" if device is None else device)
        g = _lib.i64x(self.gres)
        self.dofs = int(self.lib.mfs_vcg3d_dofs(g))
        self.face_shapes = [T.face_shape(self.gres, a) for a in range(3)]
        nbytes = int(self.lib.mfs_vcg3d_workspace_bytes(g, self.code))
        if nbytes <= 0:
            raise _lib.MfsError("mfs_vcg3d_workspace_bytes returned 0")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mfs_vcg3d_create(C.byref(h), g, self.code, T.ptr(self.workspace), nbytes, T.stream()),
                       "mfs_vcg3d_create")
        self.h = h
        # the engine's scalar block is the first bytes of the workspace (all-reduced in place by mfs.dist.SlabVCG)
        self.scalars = self.workspace[: _lib.NSCALARS * 8].view(torch.float64)
        assert self.scalars.data_ptr() == self.lib.mfs_vcg3d_scalars(self.h)
        self._bound = None

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.mfs_vcg3d_destroy(h)
            except Exception:
                pass

    def new_vector(self):
        """flat [x-faces | y-faces | z-faces] vector plus its three component views"""
        flat = torch.zeros(self.dofs, dtype=self.dtype, device=self.device)
        views, o = [], 0
        for shp in self.face_shapes:
            n = int(np.prod(shp))
            views.append(flat[o:o + n].view(shp))
            o += n
        return flat, views

    def setup(self, scale, mu, sphi, vol):
        sphi = T.dev(sphi, "sphi", T.doubled_shape(self.gres))
        vol = T.dev(vol, "vol", T.doubled_shape(self.gres))
        _lib.check(self.lib.mfs_vcg3d_setup(self.h, float(scale), float(mu), T.ptr(sphi), T.code(sphi), T.ptr(vol),
                                            T.code(vol), T.stream()), "mfs_vcg3d_setup")

    def _flat(self, t, name):
        t = T.dev(t, name, (self.dofs,))
        if t.dtype != self.dtype:
            raise TypeError(f"{name} must be {self.dtype}")
        return t

    def apply(self, v, out):
        v, out = self._flat(v, "v"), self._flat(out, "out")
        _lib.check(self.lib.mfs_vcg3d_apply(self.h, T.ptr(v), T.ptr(out), T.stream()), "mfs_vcg3d_apply")

    def bind(self, b, x, d, r, q):
        ts = [self._flat(a, n) for a, n in ((b, "b"), (x, "x"), (d, "d"), (r, "r"), (q, "q"))]
        _lib.check(self.lib.mfs_vcg3d_bind(self.h, *[T.ptr(t) for t in ts]), "mfs_vcg3d_bind")
        self._bound = ts

    def apply_kernel(self):
        """which kernel the CG applies take for the engine as bound: "march" | "tiled" | "scalar" (bit-identical)"""
        return {2: "march", 1: "tiled", 0: "scalar"}[int(self.lib.mfs_vcg3d_apply_kernel(self.h))]

    def begin(self, tol):
        _lib.check(self.lib.mfs_vcg3d_begin(self.h, float(tol), T.stream()), "mfs_vcg3d_begin")

    def iterate(self, n):
        _lib.check(self.lib.mfs_vcg3d_iterate(self.h, int(n), T.stream()), "mfs_vcg3d_iterate")

    def finish(self):
        """settle what the fused loop of `iterate` owes (last x update, d brought home); begin again before iterating on"""
        _lib.check(self.lib.mfs_vcg3d_finish(self.h, T.stream()), "mfs_vcg3d_finish")

    def set_compress(self, on):
        """compressed class access of the marching kernel (default on): bit-identical, fewer bytes where the liquid
        volume is 0 or 1 over whole z-vectors"""
        _lib.check(self.lib.mfs_vcg3d_set_compress(self.h, int(bool(on))), "mfs_vcg3d_set_compress")

    def class_census(self):
        """{"zero", "one", "mixed"}: z-vectors per class of the compressed class access (after setup)"""
        out = _lib.i64x([0, 0, 0])
        _lib.check(self.lib.mfs_vcg3d_class_census(self.h, out, T.stream()), "mfs_vcg3d_class_census")
        return {"zero": int(out[0]), "one": int(out[1]), "mixed": int(out[2])}

    def set_sparse(self, on):
        """single-domain solves from 2^21 unknowns: live-chunk vector phases + work list of the loop's march launches"""
        _lib.check(self.lib.mfs_vcg3d_set_sparse(self.h, int(bool(on))), "mfs_vcg3d_set_sparse")

    def sparse_info(self):
        import ctypes
        out = (ctypes.c_int64 * 4)()
        _lib.check(self.lib.mfs_vcg3d_sparse_info(self.h, T.stream(), out), "mfs_vcg3d_sparse_info")
        return dict(live_chunks=int(out[0]), chunks=int(out[1]), listed_pairs=int(out[2]), pairs=int(out[3]))

    def set_fuse(self, on):
        _lib.check(self.lib.mfs_vcg3d_set_fuse(self.h, int(bool(on))), "mfs_vcg3d_set_fuse")

    def loop_info(self):
        """{"fused": iterate() runs the 2-launch loop with the direction and x updates folded into the marching kernel}"""
        b = int(self.lib.mfs_vcg3d_loop_info(self.h))
        return {"fused": bool(b & 1), "merged_vector_phases": bool(b & 2), "jacobi": bool(b & 4), "resident": bool(b & 8)}

    def set_resident(self, on):
        """small grids: the whole CG loop as one resident launch per iterate() batch (default on where the grid qualifies)"""
        _lib.check(self.lib.mfs_vcg3d_set_resident(self.h, int(bool(on))), "mfs_vcg3d_set_resident")

    def set_jacobi(self, on):
        """opt-in Jacobi preconditioning (NOT the reference's CG: fewer iterations, another residual history); takes effect
        at the next setup()"""
        _lib.check(self.lib.mfs_vcg3d_set_jacobi(self.h, int(bool(on))), "mfs_vcg3d_set_jacobi")

    def set_merged(self, on):
        """small problems: r update, r.r, bookkeeping and x / direction updates in one launch (default on)"""
        _lib.check(self.lib.mfs_vcg3d_set_merged(self.h, int(bool(on))), "mfs_vcg3d_set_merged")

    # ---- slab decomposition (mfs/dist.py:SlabVCG): the phases of one iteration
    def set_slab(self, skip_top_x):
        _lib.check(self.lib.mfs_vcg3d_set_slab(self.h, int(bool(skip_top_x))), "mfs_vcg3d_set_slab")

    def begin_local(self, tol):
        _lib.check(self.lib.mfs_vcg3d_begin_local(self.h, float(tol), T.stream()), "mfs_vcg3d_begin_local")

    def begin_finish(self):
        _lib.check(self.lib.mfs_vcg3d_begin_finish(self.h, T.stream()), "mfs_vcg3d_begin_finish")

    def phase_apply(self):
        _lib.check(self.lib.mfs_vcg3d_phase_apply(self.h, T.stream()), "mfs_vcg3d_phase_apply")

    def phase_reduce(self, which):
        _lib.check(self.lib.mfs_vcg3d_phase_reduce(self.h, int(which), T.stream()), "mfs_vcg3d_phase_reduce")

    def phase_update_xr(self):
        _lib.check(self.lib.mfs_vcg3d_phase_update_xr(self.h, T.stream()), "mfs_vcg3d_phase_update_xr")

    def phase_update_d(self):
        _lib.check(self.lib.mfs_vcg3d_phase_update_d(self.h, T.stream()), "mfs_vcg3d_phase_update_d")

    # ---- the same loop over a peer-to-peer window (mfs/p2p.py): no host-side exchange inside the loop
    def edge_plane_bytes(self):
        """bytes of the three edge planes one neighbour receives per iteration (the window's plane size)"""
        return sum(int(np.prod(shp[1:])) for shp in self.face_shapes) * torch.empty(0, dtype=self.dtype).element_size()

    def attach_p2p(self, window):
        self._window = window            # keep it alive as long as the engine may use it
        _lib.check(self.lib.mfs_vcg3d_attach_p2p(self.h, window.h if window is not None else None), "mfs_vcg3d_attach_p2p")

    def slab_begin(self, tol):
        _lib.check(self.lib.mfs_vcg3d_slab_begin(self.h, float(tol), T.stream()), "mfs_vcg3d_slab_begin")

    def slab_iterate(self, n):
        _lib.check(self.lib.mfs_vcg3d_slab_iterate(self.h, int(n), T.stream()), "mfs_vcg3d_slab_iterate")

    def poll(self):
        it, done = C.c_int64(), C.c_int()
        delta, alpha, beta = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.mfs_vcg3d_poll(self.h, T.stream(), C.byref(it), C.byref(done), C.byref(delta),
                                           C.byref(alpha), C.byref(beta)), "mfs_vcg3d_poll")
        return dict(iterations=it.value, done=bool(done.value), delta=delta.value, alpha=alpha.value,
                    beta=beta.value)

    def poll_raw(self):
        """the scalar block as it stands, WITHOUT raising on the loop's error word (diagnostics after a failed solve)"""
        s = self.scalars.cpu()
        return dict(iterations=int(s[_lib.S_ITERS]), done=bool(s[_lib.S_DONE] != 0), delta=float(s[_lib.S_LASTRR]),
                    err=int(s[_lib.S_ERR]))

    def solve(self, tol, max_iter, check_every=32):
        it = C.c_int64()
        st = _lib.check(self.lib.mfs_vcg3d_solve(self.h, float(tol), int(max_iter), int(check_every), T.stream(),
                                                 C.byref(it)), "mfs_vcg3d_solve")
        return st == _lib.MFS_OK, it.value

    def history(self):
        cap = int(self.lib.mfs_pcg3d_history_capacity())
        buf = np.empty(cap, dtype=np.float64)
        n = self.lib.mfs_vcg3d_history(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)), cap, T.stream())
        _lib.check(int(n), "mfs_vcg3d_history")
        return buf[: int(n)].copy()

    def history_truncated(self):
        """True when the solve ran past the history buffer (capacity mfs_pcg3d_history_capacity() doubles = 8 191 iterations):
        history() then holds the LEADING entries only -- `iterations`, `delta`, alpha and beta come from the engine's scalar
        block (poll()), never from the history, and stay exact"""
        cap = int(self.lib.mfs_pcg3d_history_capacity())
        return 2 * int(self.poll_raw()["iterations"]) + 1 > cap
