"""Randomised shapes and scenes: the optimised native loop (LDS march with compressed coefficients, fused
direction update, prefetch depth 2) against the same march with every one of those switched off (dense
coefficients, direction update as its own kernel, depth 1) and against the loop with the solution update deferred
into the next stencil launch -- bit for bit, since the per-block partial
sums are grouped identically -- and against the oracle on the first iterations.  (The direct-load
variant 0 groups its dot-product partials differently, so it is compared on the stencil output only,
tests/test_pressure_gpu.py / test_edge_cases_gpu.py.)  Shapes include odd sizes (scalar kernels), rows longer than
one tile, partial last tiles, and tiny x extents (marches shorter than the prefetch pipeline)."""
import numpy as np
import pytest
import torch

from mfs import scenes
from oracle import mfs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(3, 3, 4), (4, 9, 8), (5, 4, 12), (7, 33, 16), (9, 6, 260), (12, 70, 36), (33, 17, 24), (40, 5, 8),
          (6, 130, 8), (16, 16, 1028), (11, 13, 15), (3, 40, 64), (64, 3, 20)]


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


@pytest.mark.parametrize("gres", SHAPES)
@pytest.mark.parametrize("prec", ["fp64", "fp32"])
def test_optimised_loop_equals_plain_loop(gres, prec):
    import solver.CGSolverBuffer as B
    import solver.PressureCGSolver3D as P
    seed = sum(gres)
    sc = scenes.pressure_scene_3d(gres, seed=seed, vel_dtype=np.float32, solid_velocity=bool(seed & 1))
    res = []
    for mode in ("optimised", "plain", "deferred_x", "tailed", "deferred_x_tailed"):
        buf = B.CGSolverBuffer(gres, precision=prec, device=DEV)
        s = P.PressureCGSolver3D(buf, gres, sc["bound_size"], check_every=3)
        s.max_iter = 12                                  # a dozen iterations exercise every pipeline stage
        e = s._engine
        e.set_resident(False)                            # the resident small-grid loop groups its dot products differently:
                                                         # tests/test_resident_gpu.py compares it to rounding
        if mode == "plain":
            e.set_compress(False); e.set_fuse(False); e.set_prefetch(1)
        if mode.endswith("tailed"):                             # the x/r update's last workgroup closes the iteration (the form
            e.set_lean(False)                            # the default loop uses only with the deferred x update)
        if mode.startswith("deferred_x"):                         # x += alpha d rides in the NEXT stencil launch (default only
            e.set_defer_x(True)                          # beyond the Infinity Cache; forced here), flushed at the end
        v = [T(sc["vx"]), T(sc["vy"]), T(sc["vz"])]
        try:
            s.solve(*v, T(sc["sphi"]), T(sc["sv"]), T(sc["lphi"]), tol=1e-30)
        except ValueError:
            pass                                         # "Failed to converge!" after max_iter, as intended
        res.append((s.iterations, s.history, s.x.clone(), buf.d.clone(), buf.r.clone(), buf.q.clone()))
    a = res[0]
    for b in res[1:]:
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1], b[1])
        for i in range(2, 6):
            assert torch.equal(a[i], b[i]), ("x", "d", "r", "q")[i - 2]
    if prec == "fp64" and min(gres) >= 3:
        ref = O.PressureCGSolver3D(gres, sc["bound_size"])
        rv = [sc["vx"].copy(), sc["vy"].copy(), sc["vz"].copy()]
        ref.solve(*rv, sc["sphi"], sc["sv"], sc["lphi"], tol=1e-30, max_iter=12, raise_on_fail=False)
        n = min(len(a[1]), len(ref.history), 13)
        h0 = max(ref.history[0], 1e-300)
        np.testing.assert_allclose(a[1][:n], np.array(ref.history)[:n], rtol=1e-9, atol=1e-18 * h0)
