"""Rigid-body signed distances on the MI355X against goldens produced by executing the
reference's solver/sdf3D.py, through the drop-in module's own generate_rb / set_vel_rb (so the packed body
layout is checked too).  Tolerance: 1e-13 (sqrt vs pow, FMA contraction off); float32 positions exact to 1 ulp."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names
from oracle import mfs_oracle as O
import solver.sdf3D as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
N = lambda t: t.detach().cpu().numpy()  # noqa: E731


def notebook_bodies():
    h = 0.45
    rb_d, rb_map = None, {}
    for nm, par, flip, c, ax, ang in [("cube", ['box', 0.5, 0.8, 0.5], True, [0, 0.5, 0], [0, 1, 0], 0),
                                      ("cube1", ['box', 0.67, 0.1, 1.0], False, [-0.34, h, 0], [0, 0, 1], -45),
                                      ("cube2", ['box', 0.67, 0.1, 1.0], False, [0.34, h, 0], [0, 0, 1], 45),
                                      ("cube3", ['box', 1.0, 0.1, 0.7], False, [0, h, -0.3], [1, 0, 0], 45),
                                      ("cube4", ['box', 1.0, 0.1, 0.7], False, [0, h, 0.3], [1, 0, 0], -45)]:
        rb_d, rb_map = S.generate_rb(rb_d, rb_map, nm, par, flip=flip, center=c, axis=ax, angle=ang, device=DEV)
    S.set_vel_rb(rb_d, 1, [0.3, -0.2, 0.1])
    return rb_d, rb_map


def test_generate_rb_layout_matches_reference():
    rb_d, rb_map = notebook_bodies()
    g = golden("sdf_a_notebook")
    assert rb_map == {"cube": 0, "cube1": 1, "cube2": 2, "cube3": 3, "cube4": 4}
    np.testing.assert_allclose(N(rb_d), g["rb_d"], rtol=0, atol=1e-16)


@pytest.mark.parametrize("name", golden_names("sdf_"))
def test_evaluate_and_project(name):
    g = golden(name)
    rb_d = T(g["rb_d"])
    pos = T(g["position"])
    n = pos.shape[0]
    sd = torch.full((n,), 9.0, dtype=torch.float64, device=DEV)
    vel = torch.full((n, 3), 9.0, dtype=torch.float64, device=DEV)
    S.evaluate(rb_d, sd, vel, pos)
    np.testing.assert_allclose(N(sd), g["sd"], rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(N(vel), g["vel"])
    proj = pos.clone()
    S.project(rb_d, proj)
    if proj.dtype == torch.float64:
        np.testing.assert_allclose(N(proj), g["projected"], rtol=0, atol=1e-15)
    else:
        np.testing.assert_allclose(N(proj), g["projected"], rtol=0, atol=1.2e-7)
    # a grid-shaped evaluation like the notebook's solid level set set-up (positions (..., 3), sd (...))
    P3 = pos[:1000].reshape(10, 10, 10, 3).contiguous()
    sd3 = torch.zeros(10, 10, 10, dtype=torch.float64, device=DEV)
    vel3 = torch.zeros(10, 10, 10, 3, dtype=torch.float64, device=DEV)
    S.evaluate(rb_d, sd3, vel3, P3)
    np.testing.assert_array_equal(N(sd3).reshape(-1), N(sd)[:1000])


def test_cylinder_against_the_restatement():
    """cylinder_eval's in-range branch is unpinned (the reference reads an unassigned variable there, sdf3D.py:148-160):
    the GPU kernel and the oracle agree on the intended reading."""
    rb_d, rb_map = S.generate_rb(None, {}, "can", ['cylinder', 0.2, 0.5], flip=True, center=[0, 0.5, 0], axis=[0, 1, 0],
                                 angle=0, device=DEV)
    rb_d, rb_map = S.generate_rb(rb_d, rb_map, "peg", ['cylinder', 0.05, 0.3], flip=False, center=[0.05, 0.5, 0.02],
                                 axis=[1, 0, 1], angle=35, device=DEV)
    rng = np.random.default_rng(5)
    pos = rng.uniform([-0.3, 0.1, -0.3], [0.3, 0.9, 0.3], size=(800, 3))
    sd, vel = torch.zeros(800, dtype=torch.float64, device=DEV), torch.zeros(800, 3, dtype=torch.float64, device=DEV)
    S.evaluate(rb_d, sd, vel, T(pos))
    rsd, rvel = np.zeros(800), np.zeros((800, 3))
    O.sdf_evaluate(N(rb_d), rsd, rvel, pos)
    np.testing.assert_allclose(N(sd), rsd, rtol=1e-13, atol=1e-15)
    proj, rproj = T(pos), pos.copy()
    S.project(rb_d, proj)
    O.sdf_project(N(rb_d), rproj)
    np.testing.assert_allclose(N(proj), rproj, rtol=0, atol=1e-14)


def test_no_bodies_and_misuse():
    """evaluate with an empty scene: every distance is the reference's initial 100, velocities 0; project: no-op"""
    rb = torch.zeros((0, 10, 4), dtype=torch.float64, device=DEV)
    pos = torch.rand((50, 3), dtype=torch.float64, device=DEV)
    sd, vel = torch.zeros(50, dtype=torch.float64, device=DEV), torch.ones((50, 3), dtype=torch.float64, device=DEV)
    S.evaluate(rb, sd, vel, pos)
    assert float(sd.min()) == 100.0 and float(vel.abs().max()) == 0.0
    before = pos.clone()
    S.project(rb, pos)
    assert torch.equal(pos, before)
    with pytest.raises(ValueError, match="rb_d"):
        S.project(torch.zeros((1, 9, 4), dtype=torch.float64, device=DEV), pos)
