"""Drop-in for the reference's solver/SolidFraction3D.py (compute_solid_frac)."""
from mfs import _lib, tensors as T
from .SolidFractionCommon import edge_in_fraction, face_in_fraction, tri_in_fraction  # noqa: F401


def compute_solid_frac(gres, sphi, wx, wy, wz):
    """Face open-fractions wx,wy,wz from the solid level set on the doubled grid
    (reference solver/SolidFraction3D.py:28-32 -> kernel :6-26).  Writes w*[0:N];
    the upper faces w*[N] are left untouched, as in the reference."""
    g = T.as_gres(gres)
    sphi = T.dev(sphi, "sphi", T.doubled_shape(g))
    wx = T.dev(wx, "wx", T.face_shape(g, 0))
    wy = T.dev(wy, "wy", T.face_shape(g, 1))
    wz = T.dev(wz, "wz", T.face_shape(g, 2))
    if not (wx.dtype == wy.dtype == wz.dtype):
        raise TypeError("wx, wy, wz must share a dtype")
    lib = _lib.load()
    _lib.check(lib.mfs_solid_frac3d(_lib.i64x(g), T.ptr(sphi), T.code(sphi), T.ptr(wx), T.ptr(wy), T.ptr(wz),
                                    T.code(wx), T.stream()), "mfs_solid_frac3d")
