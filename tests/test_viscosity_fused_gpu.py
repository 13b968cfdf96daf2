"""The fused viscosity CG loop (csrc/mfs_vcg_march.h FUSE, mfs_vcg3d_iterate: the stencil launch of iteration j also forms
d_j = r + beta d_{j-1} and performs x += alpha d_{j-1}; 2 launches per iteration) against the three-launch loop it replaces:
BIT FOR BIT -- same x, d, r, q, residual history, iteration count and written-back velocities -- on shapes that span
several tiles and marches, partial last tiles, rows shorter and longer than a wave, both state precisions, a converged
solve and a loop stopped early (`iterate(n)` + `finish()`: the owed x and direction updates).  The three-launch loop itself
is pinned against the executed-reference goldens and the C oracle (tests/test_viscosity_gpu.py).  GPU only."""
import numpy as np
import pytest
import torch

from mfs import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(12, 12, 12), (24, 24, 24), (20, 24, 36), (40, 36, 32), (9, 70, 16), (16, 20, 64), (7, 5, 128), (33, 17, 8),
          (64, 64, 64), (5, 300, 4), (48, 80, 48)]


def _solver(gres, sc, dt, fuse):
    import solver.ViscosityCGSolver3D as V
    s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dt, device=DEV, check_every=8)
    s._engine.set_fuse(fuse)
    s._engine.set_resident(False)
    s._engine.set_compress(False)      # (round 3) the compressed march cuts its work into cost-balanced segments: another grouping of d.q.
                                       # The fused launch has no compressed form, so both loops are compared on dense access
    s._engine.set_merged(False)        # the reference form here is the THREE-launch loop (the merged vector phases group r.r differently)
    return s


def _prepare(s, sc, mu):
    """everything `solve` does before the CG loop (reference :567-574), engine bound and set up"""
    import solver.ViscosityCGSolver3D as V
    from mfs import tensors as T
    g = s._g
    scale = sc["dt"] / s.cell_vol / sc["rho"]
    torch.div(T.dev(sc["lvol"], "lvol", T.doubled_shape(g)), s.cell_vol * 0.125, out=s.vol)
    s.x_x.copy_(sc["vx"]); s.x_y.copy_(sc["vy"]); s.x_z.copy_(sc["vz"])
    V.extrapolate(g, 3, s.x_x, s.x_y, s.x_z, sc["sphi"])
    V.initialize_solver(g, scale, mu, s.x_x, s.x_y, s.x_z, sc["sphi"], sc["sv"], s.vol, s.b_x, s.b_y, s.b_z)
    s._engine.setup(scale, mu, sc["sphi"], s.vol)
    f = s._flat
    s._engine.bind(f["b"], f["x"], f["d"], f["r"], f["q"])


@pytest.mark.parametrize("dt", ["fp32", "fp64"])
@pytest.mark.parametrize("gres", SHAPES, ids=lambda g: "x".join(map(str, g)))
def test_fused_loop_equals_three_launch_loop(gres, dt):
    sc = scenes.viscosity_scene_3d(gres, seed=5, device=DEV, noise=0.3)
    mu = 40.0
    got = []
    for fuse in (1, 0):
        s = _solver(gres, sc, dt, fuse)
        vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
        if fuse:
            _prepare(s, sc, mu)
            # (rows of fewer than two vectors take the one-cell-per-lane kernels and the three-launch loop)
            assert s._engine.loop_info()["fused"] == (s._engine.apply_kernel() == "march"), "the fused loop is not what runs"
        s.solve(sc["dt"], mu, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=1e-7)
        torch.cuda.synchronize()
        got.append(dict(it=s.iterations, hist=s.history, v=(vx, vy, vz), **{k: s._flat[k].clone() for k in "xdrq"}))
    a, b = got
    assert a["it"] == b["it"] and a["it"] >= 2, (a["it"], b["it"])
    assert np.array_equal(a["hist"], b["hist"])
    for k in "xrqd":
        assert torch.equal(a[k], b[k]), f"{k}: max |diff| {float((a[k] - b[k]).abs().max())}"
    for p, q in zip(a["v"], b["v"]):
        assert torch.equal(p, q)


@pytest.mark.parametrize("dt", ["fp32", "fp64"])
@pytest.mark.parametrize("n", [1, 2, 7])
def test_stopped_loop_settles_what_it_owes(dt, n):
    """iterate(n) without convergence, then finish(): x has all n updates, d is d_n = r + beta d_{n-1}, home in the bound array"""
    gres = (20, 24, 36)
    sc = scenes.viscosity_scene_3d(gres, seed=9, device=DEV, noise=0.3)
    got = []
    for fuse in (1, 0):
        s = _solver(gres, sc, dt, fuse)
        _prepare(s, sc, 40.0)
        e = s._engine
        e.begin(1e-12)
        e.iterate(n)
        e.finish()
        torch.cuda.synchronize()
        st = e.poll()
        assert st["iterations"] == n and not st["done"]
        got.append(dict(hist=e.history(), **{k: s._flat[k].clone() for k in "xdrq"}))
    a, b = got
    assert np.array_equal(a["hist"], b["hist"])
    for k in "xrqd":
        assert torch.equal(a[k], b[k]), f"{k}: max |diff| {float((a[k] - b[k]).abs().max())}"


def test_two_solves_through_one_engine():
    """the partner buffer keeps last solve's direction: a second solve (other scene, other solid faces) must not see it"""
    gres = (24, 24, 24)
    s_f = s_u = None
    for seed in (3, 4):
        sc = scenes.viscosity_scene_3d(gres, seed=seed, device=DEV, noise=0.3)
        if s_f is None:
            s_f, s_u = _solver(gres, sc, "fp64", 1), _solver(gres, sc, "fp64", 0)
        outs = []
        for s in (s_f, s_u):
            vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
            s.solve(sc["dt"], 25.0, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=1e-8)
            outs.append((s.iterations, s._flat["x"].clone(), s._flat["d"].clone(), vx, vy, vz))
        assert outs[0][0] == outs[1][0]
        for p, q in zip(outs[0][1:], outs[1][1:]):
            assert torch.equal(p, q)
