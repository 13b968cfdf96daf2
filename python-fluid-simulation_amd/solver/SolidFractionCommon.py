"""Host-side scalar versions of the three device functions of the reference's
solver/SolidFractionCommon.py:4-60.  The GPU kernels carry their own copies
(csrc/mfs_pressure.hip); these exist because the reference modules export the
names (`from .SolidFraction3D import compute_solid_frac, edge_in_fraction`)."""


def edge_in_fraction(lval, rval):
    l_in = lval < 0
    r_in = rval < 0
    if l_in and r_in:
        return 1
    if not l_in and not r_in:
        return 0
    diff = -abs(lval - rval)
    return lval / diff if l_in else rval / diff


def tri_in_fraction(v0, v1, v2):
    v = (v0, v1, v2)
    inside = [x < 0 for x in v]
    n = sum(inside)
    if n == 3:
        return 1.0
    if n == 0:
        return 0.0
    if n == 2:
        k = inside.index(False)
        return 1.0 - edge_in_fraction(v[(k + 1) % 3], v[(k + 2) % 3])
    k = inside.index(True)
    return edge_in_fraction(v[(k + 1) % 3], v[(k + 2) % 3])


def face_in_fraction(bl, br, tl, tr):
    ce = 0.25 * (bl + br + tl + tr)
    return 0.25 * (tri_in_fraction(bl, br, ce) + tri_in_fraction(br, tr, ce)
                   + tri_in_fraction(tr, tl, ce) + tri_in_fraction(tl, bl, ce))
