"""Drop-in for the reference's solver/SolidFraction2D.py (compute_solid_frac, 2D)."""
from mfs import _lib, tensors as T
from .SolidFractionCommon import edge_in_fraction  # noqa: F401


def compute_solid_frac(gres, sphi, wx, wy):
    """2D edge open-fractions (reference solver/SolidFraction2D.py:22-26 -> kernel :6-20):
    every cell with x<Nx-1, y<Ny-1 writes both faces per axis, true linear fractions."""
    g = T.as_gres(gres)
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    wx = T.dev(wx, "wx", T.face_shape(g, 0))
    wy = T.dev(wy, "wy", T.face_shape(g, 1))
    if wx.dtype != wy.dtype:
        raise TypeError("wx, wy must share a dtype")
    lib = _lib.load()
    _lib.check(lib.mfs_solid_frac2d(_lib.i64x(g), T.ptr(sphi), T.code(sphi), T.ptr(wx), T.ptr(wy), T.code(wx),
                                    T.stream()), "mfs_solid_frac2d")
