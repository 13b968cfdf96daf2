"""ctypes access to the oracle's C restatement (oracle/mfs_oracle_c.c).
TEST INFRASTRUCTURE / bench cpu_baseline only -- never imported by the product."""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmfs_oracle.so")
_lib = None
_pd = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        lib = C.CDLL(_SO)
        lib.mfs_oracle_threads.restype = C.c_int
        lib.mfs_oracle_set_threads.argtypes = [C.c_int]
        lib.mfs_oracle_set_variant.restype = None
        lib.mfs_oracle_set_variant.argtypes = [C.c_int, C.c_int]
        lib.mfs_oracle_pressure_apply3d.restype = None
        lib.mfs_oracle_pressure_apply3d.argtypes = [C.POINTER(C.c_int64), _pd, _pd, _pd, _pd, _pd, _pd]
        lib.mfs_oracle_pressure_cg3d.restype = C.c_int64
        lib.mfs_oracle_pressure_cg3d.argtypes = [C.POINTER(C.c_int64), _pd, _pd, _pd, _pd, _pd, _pd, _pd, _pd, _pd,
                                                 C.c_double, C.c_int64, C.c_void_p, C.c_int64,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_int)]
        lib.mfs_oracle_visc_apply3d.restype = None
        lib.mfs_oracle_visc_apply3d.argtypes = [C.POINTER(C.c_int64), C.c_double, C.c_double, _pd, _pd, _pd, _pd, _pd, _pd,
                                                _pd, _pd]
        lib.mfs_oracle_visc_cg3d.restype = C.c_int64
        lib.mfs_oracle_visc_cg3d.argtypes = [C.POINTER(C.c_int64), C.c_double, C.c_double, _pd, _pd, _pd, _pd, _pd, _pd, _pd,
                                             C.c_double, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_double),
                                             C.POINTER(C.c_int)]
        _lib = lib
    return _lib


def available():
    try:
        _load()
        return True
    except Exception:
        return False


def threads():
    return int(_load().mfs_oracle_threads())


def set_variant(dot_variant=0, fma_mask=0):
    """rounding variant of the C oracle (mfs_oracle_set_variant): dot_variant 0..15 permutes the summation order of the dot
    products, fma_mask bit 0 / 1 / 2 fuses the multiply-adds of the dots / the vector updates / the operator.  (0, 0) is the
    oracle proper; the others are the same algorithm with different rounding (tests/test_history_envelope.py)."""
    _load().mfs_oracle_set_variant(int(dot_variant), int(fma_mask))


def _g(gres):
    return (C.c_int64 * 3)(*[int(v) for v in gres])


def apply(gres, v, out, wx, wy, wz, lphi):
    _load().mfs_oracle_pressure_apply3d(_g(gres), v, out, wx, wy, wz, lphi)


def cg(gres, b, lphi, wx, wy, wz, tol, max_iter, hist_cap=0):
    n = tuple(int(v) for v in gres)
    x, d, r, q = (np.zeros(n) for _ in range(4))
    hist = np.zeros(max(hist_cap, 1))
    delta, conv = C.c_double(), C.c_int()
    it = _load().mfs_oracle_pressure_cg3d(_g(gres), np.ascontiguousarray(b, np.float64), x, d, r, q, wx, wy, wz,
                                          np.ascontiguousarray(lphi, np.float64), tol, max_iter,
                                          hist.ctypes.data if hist_cap else None, hist_cap, C.byref(delta),
                                          C.byref(conv))
    return dict(iterations=int(it), x=x, delta=delta.value, converged=bool(conv.value),
                history=hist[: min(hist_cap, 2 * int(it) + 1)])


def time_cg(gres, b, lphi, wx, wy, wz, iters, nthreads=0):
    """wall time of `iters` CG iterations (tol=0 -> never converges) and the thread count used."""
    lib = _load()
    if nthreads:
        lib.mfs_oracle_set_threads(int(nthreads))
    t0 = time.perf_counter()
    cg(gres, b, lphi, wx, wy, wz, 0.0, iters)
    return time.perf_counter() - t0, threads()


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def visc_apply(gres, scale, mu, vx, vy, vz, ox, oy, oz, sphi, vol):
    """mfs_oracle_visc_apply3d: the three operator rows on doubled-grid sphi / vol; outputs written in place"""
    _load().mfs_oracle_visc_apply3d(_g(gres), float(scale), float(mu), _f64(vx), _f64(vy), _f64(vz), ox, oy, oz,
                                    _f64(sphi), _f64(vol))


def visc_cg(gres, scale, mu, b, x0, sphi, vol, tol, max_iter, hist_cap=0):
    """mfs_oracle_visc_cg3d on flat [x-faces | y-faces | z-faces] vectors; `x0` is the initial guess (copied)"""
    b = _f64(b).ravel()
    x = _f64(x0).ravel().copy()
    d, r, q = (np.zeros_like(x) for _ in range(3))
    hist = np.zeros(max(hist_cap, 1))
    delta, conv = C.c_double(), C.c_int()
    it = _load().mfs_oracle_visc_cg3d(_g(gres), float(scale), float(mu), b, x, d, r, q, _f64(sphi), _f64(vol), tol,
                                      max_iter, hist.ctypes.data if hist_cap else None, hist_cap, C.byref(delta),
                                      C.byref(conv))
    return dict(iterations=int(it), x=x, d=d, r=r, q=q, delta=delta.value, converged=bool(conv.value),
                history=hist[: min(hist_cap, 2 * int(it) + 1)])
