"""The particle <-> grid restatements (oracle/mfs_oracle.py nb_p2g_*, nb_g2p_*, nb_fluid_*) against goldens
produced by executing the notebook's own cells (tests/golden/make_goldens_particles.py, pt_*)."""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import mfs_oracle as O

BIAS = {0: (0, .5, .5), 1: (.5, 0, .5), 2: (.5, .5, 0)}


def containers(g):
    gres = tuple(int(v) for v in g["gres"])
    bmin = np.asarray(g["bound_min"], np.float32)
    cs = np.asarray(g["bound_size"], np.float32) / np.asarray(gres, np.int64)          # float64, like the notebook's
    return gres, bmin, cs


@pytest.mark.parametrize("name", golden_names("pt_"))
def test_particle_transfers(name):
    g = golden(name)
    gres, bmin, cs = containers(g)
    assert cs.dtype == np.float64
    grids = {}
    for a, c in enumerate("xyz"):
        shape = tuple(np.array(gres) + np.eye(3, dtype=int)[a])
        gm, gv = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
        O.nb_p2g_scatter(g["px"], g["pm"], g["pv"], g["pc" + c], gm, gv, bmin, gres, BIAS[a], cs, a)
        O.nb_p2g_normalize(gm, gv)
        # fp32 accumulation in particle order: agreement to a few ulps of the largest entry
        np.testing.assert_allclose(gm, g[f"g{c}_m"], rtol=0, atol=2e-6 * np.abs(g[f"g{c}_m"]).max())
        np.testing.assert_allclose(gv, g[f"g{c}_v"], rtol=2e-4, atol=2e-5 * np.abs(g[f"g{c}_v"]).max())
        grids[c] = g[f"g{c}_v"]
    pv = np.array(g["pv"])
    for a, c in enumerate("xyz"):
        pca = np.array(g["pc" + c])
        O.nb_g2p_gather(bmin, gres, BIAS[a], cs, a, g["px"], pv, pca, grids[c])
        np.testing.assert_allclose(pca, g["g2p_c" + c], rtol=1e-12, atol=1e-12 * np.abs(g["g2p_c" + c]).max())
    np.testing.assert_allclose(pv, g["g2p_v"], rtol=1e-12, atol=1e-13)
    phi = np.zeros(gres)
    O.nb_fluid_levelset(g["px"], phi, bmin, cs, float(g["gdx"]), gres)
    np.testing.assert_allclose(phi, g["lphi"], rtol=1e-13, atol=1e-15)
    vres = tuple(2 * np.array(gres) + 1)
    vcs = np.asarray(g["bound_size"], np.float32) / (2 * np.asarray(gres, np.int64))
    vol = np.zeros(vres)
    O.nb_fluid_volume(bmin, vcs, vres, g["px"], float(g["pvol"]), vol)
    np.testing.assert_allclose(vol, g["lvol"], rtol=1e-11, atol=1e-15 * np.abs(g["lvol"]).max())
