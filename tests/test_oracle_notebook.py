"""Oracle restatement of the notebook's grid kernels against goldens produced by
executing the notebook's own cell source (tests/golden/make_goldens_notebook.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import mfs_oracle as O


@pytest.mark.parametrize("name", golden_names("nb_"))
def test_notebook_grid_kernels(name):
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    ex = [g["in_vx"].copy(), g["in_vy"].copy(), g["in_vz"].copy()]
    O.nb_extrapolate(gres, 2, *ex, g["mx"], g["my"], g["mz"])
    for a, k in zip(ex, ("ex_vx", "ex_vy", "ex_vz")):
        assert a.dtype == g[k].dtype
        np.testing.assert_array_equal(a, g[k])
    dv = [np.full_like(g[k], 9.0) for k in ("dvx", "dvy", "dvz")]
    O.nb_boundary_condition(gres, [g["ex_vx"], g["ex_vy"], g["ex_vz"]], [g["mx"], g["my"], g["mz"]], g["sphi"], g["sv"],
                            float(g["dx"]), dv)
    for a, k in zip(dv, ("dvx", "dvy", "dvz")):
        np.testing.assert_allclose(a, g[k], rtol=2e-7, atol=1e-12)
        assert np.count_nonzero(g[k]) > 20
    for k_ex, k_dv, k_bc in (("ex_vx", "dvx", "bc_vx"), ("ex_vy", "dvy", "bc_vy"), ("ex_vz", "dvz", "bc_vz")):
        np.testing.assert_array_equal(g[k_ex] + g[k_dv], g[k_bc])       # g.x.v += g.x.dv
