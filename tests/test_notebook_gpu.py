"""SURVEY.md 8(f) rank 1 on the GPU: the notebook's `extrapolate` and `apply_boundary_condition` drop-ins against
goldens produced by executing the notebook's own cells, and against the oracle at a larger size."""
import types

import numpy as np
import pytest
import torch

from conftest import golden, golden_names
from oracle import mfs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _grid(vx, vy, vz, mx, my, mz):
    N = types.SimpleNamespace
    return N(x=N(v=vx, m=mx, dv=torch.full_like(vx, 9.0)), y=N(v=vy, m=my, dv=torch.full_like(vy, 9.0)),
             z=N(v=vz, m=mz, dv=torch.full_like(vz, 9.0)))


@pytest.mark.parametrize("name", golden_names("nb_"))
def test_notebook_kernels_vs_golden(name):
    import notebook_kernels as NK
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    ex = [T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])]
    NK.extrapolate(gres, 2, *ex, T(g["mx"]), T(g["my"]), T(g["mz"]))
    for a, k in zip(ex, ("ex_vx", "ex_vy", "ex_vz")):
        assert a.dtype == torch.float32
        np.testing.assert_array_equal(a.cpu().numpy(), g[k])            # fp32 results, bit for bit
    grid = _grid(T(g["ex_vx"]), T(g["ex_vy"]), T(g["ex_vz"]), T(g["mx"]), T(g["my"]), T(g["mz"]))
    solid = types.SimpleNamespace(phi=T(g["sphi"]), v=T(g["sv"]))
    NK.apply_boundary_condition(grid, solid, float(g["dx"]))
    for c, k in zip((grid.x, grid.y, grid.z), ("dvx", "dvy", "dvz")):
        np.testing.assert_allclose(c.dv.cpu().numpy(), g[k], rtol=2e-7, atol=1e-12)     # fp32 store, FMA contraction
    for c, k in zip((grid.x, grid.y, grid.z), ("bc_vx", "bc_vy", "bc_vz")):
        np.testing.assert_allclose(c.v.cpu().numpy(), g[k], rtol=3e-7, atol=1e-7)


def test_notebook_kernels_vs_oracle_64():
    import notebook_kernels as NK
    from mfs import scenes
    gres = (64, 48, 40)
    sc = scenes.viscosity_scene_3d(gres, seed=31, vel_dtype=np.float32, noise=0.2)
    rng = np.random.default_rng(7)
    v = [sc["vx"], sc["vy"], sc["vz"]]
    # mass almost everywhere (so the wall-adjacent faces carry defined averages), 10 % empty faces sprinkled in
    m = [((rng.uniform(size=a.shape) > 0.1) * rng.uniform(0.2, 1.5, size=a.shape)).astype(np.float32) for a in v]
    v = [(a + 0.5 * rng.standard_normal(a.shape)).astype(np.float32) for a in v]
    sv = np.stack([0.2 * np.sin(3 * sc["sphi"]), -0.1 + 0 * sc["sphi"], 0.05 * np.cos(2 * sc["sphi"])], axis=-1)
    ref = [a.copy() for a in v]
    O.nb_extrapolate(gres, 2, *ref, *m)
    dv = [np.zeros_like(a) for a in v]
    O.nb_boundary_condition(gres, ref, m, sc["sphi"], sv, sc["cell_size"][0], dv)
    dev = [T(a) for a in v]
    NK.extrapolate(gres, 2, *dev, *[T(a) for a in m])
    for a, b in zip(dev, ref):
        np.testing.assert_array_equal(a.cpu().numpy(), b)
    grid = _grid(*dev, *[T(a) for a in m])
    NK.apply_boundary_condition(grid, types.SimpleNamespace(phi=T(sc["sphi"]), v=T(sv)), sc["cell_size"][0])
    for c, d, r in zip((grid.x, grid.y, grid.z), dv, ref):
        np.testing.assert_allclose(c.dv.cpu().numpy(), d, rtol=2e-7, atol=1e-12)
        np.testing.assert_allclose(c.v.cpu().numpy(), r + d, rtol=3e-7, atol=1e-7)
        assert np.count_nonzero(d) > 100
