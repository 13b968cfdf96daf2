"""The rigid-body SDF restatement (oracle/mfs_oracle.py sdf_*) against goldens produced by executing the
reference's solver/sdf3D.py (tests/golden/make_goldens_sdf.py)."""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import mfs_oracle as O


@pytest.mark.parametrize("name", golden_names("sdf_"))
def test_sdf_evaluate_and_project(name):
    g = golden(name)
    n = 600                                   # per-point Python loops: a slice is enough
    pos = g["position"][:n]
    sd, vel = np.zeros(n), np.zeros((n, 3))
    O.sdf_evaluate(g["rb_d"], sd, vel, pos)
    np.testing.assert_allclose(sd, g["sd"][:n], rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(vel, g["vel"][:n])
    proj = pos.copy()
    O.sdf_project(g["rb_d"], proj)
    tol = 1e-15 if proj.dtype == np.float64 else 0
    np.testing.assert_allclose(proj, g["projected"][:n], rtol=0, atol=tol)
    assert (vel != 0).any() and (np.abs(proj - pos) > 1e-6).any()
