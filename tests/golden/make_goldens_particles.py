#!/usr/bin/env python3
"""Generate tests/golden/pt_*.npz for the SURVEY.md section 8(f) rank-3 rows -- the notebook's particle <-> grid
transfers: `p2g` (code cell 2), `g2p` (cell 3), `compute_fluid_levelset` (cell 4), `compute_fluid_volume`
(cell 6) -- by EXECUTING THE NOTEBOOK'S OWN CELL SOURCE (container only), with the containers' dtypes as the
notebook builds them (code cell 9/10: BOUND_MIN / BOUND_SIZE / biases float32, cell sizes float64, particle
arrays float64, grid mass / velocity float32, level set and volume float64).

The cells are read out of /root/reference/3D_viscous_fluid_sim.ipynb at run time (nothing of the reference's
text is stored in this repo) and exec()'d with the container-only plumbing of tests/golden/refshim.
Needs /root/reference.
"""
import json
import math
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MFS_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, os.path.join(REPO, "python-fluid-simulation_amd"))

import numpy as np  # noqa: E402
import cupy as cp   # noqa: E402  (refshim)
from numba import cuda  # noqa: E402  (refshim)

from mfs import scenes  # noqa: E402


def notebook_namespace():
    nb = json.load(open(os.path.join(REF, "3D_viscous_fluid_sim.ipynb")))
    ns = {"cp": cp, "cuda": cuda, "np": np, "math": math}
    want = ("def p2g(p, g)", "def g2p(p, g)", "def compute_fluid_levelset(p, ls, gdx)", "def compute_fluid_volume(p, fv, pvol)")
    found = 0
    for c in nb["cells"]:
        if c["cell_type"] != "code":
            continue
        src = "".join(c["source"])
        if any(w in src for w in want):
            exec(compile(src, "<notebook cell>", "exec"), ns)
            found += 1
    assert found == 4, found
    return ns


def C(a, dtype=None):
    return cp.array(np.array(a, dtype=dtype, copy=True))


def gen(name, gres, seed, per_cell=3):
    ns = notebook_namespace()
    N = types.SimpleNamespace
    sc = scenes.particle_scene_3d(gres, seed, per_cell=per_cell)
    G = C(gres, np.int64)
    bmin, bsz = C(sc["bound_min"], np.float32), C(sc["bound_size"], np.float32)
    eye = np.eye(3, dtype=np.int64)
    p = N(num_particles=len(sc["px"]), x=C(sc["px"]), m=C(sc["pm"]), v=C(sc["pv"]), cx=C(sc["pcx"]), cy=C(sc["pcy"]),
          cz=C(sc["pcz"]), vol=sc["pvol"])

    def comp(a, bias):
        shape = tuple(np.array(gres) + eye[a])
        return N(resolution=C(np.array(gres) + eye[a], np.int64), bias=C(bias, np.float32),
                 m=cp.zeros(shape, dtype=cp.float32), v=cp.zeros(shape, dtype=cp.float32))
    g = N(resolution=G, bound_size=bsz, bound_min=bmin, cell_size=bsz / G, x=comp(0, [0, .5, .5]), y=comp(1, [.5, 0, .5]),
          z=comp(2, [.5, .5, 0]))
    assert np.asarray(g.cell_size).dtype == np.float64
    with np.errstate(all="ignore"):
        ns["p2g"](p, g)
    out = dict(kind="particles", gres=np.array(gres), bound_min=np.asarray(bmin), bound_size=np.asarray(bsz),
               px=sc["px"], pm=sc["pm"], pv=sc["pv"], pcx=sc["pcx"], pcy=sc["pcy"], pcz=sc["pcz"], pvol=sc["pvol"],
               gdx=sc["gdx"])
    for c in "xyz":
        out[f"g{c}_m"] = np.asarray(getattr(g, c).m)
        out[f"g{c}_v"] = np.asarray(getattr(g, c).v)
    # g2p from that grid (particle velocities and affine rows are overwritten)
    ns["g2p"](p, g)
    out.update(g2p_v=np.asarray(p.v), g2p_cx=np.asarray(p.cx), g2p_cy=np.asarray(p.cy), g2p_cz=np.asarray(p.cz))
    # fluid level set on the cell grid, fluid volume on the doubled grid
    ls = N(resolution=G, bound_size=bsz, bound_min=bmin, cell_size=bsz / G, phi=cp.zeros(gres))
    ns["compute_fluid_levelset"](p, ls, sc["gdx"])
    VG = C(2 * np.array(gres), np.int64)
    fv = N(resolution=VG + 1, bound_size=bsz, bound_min=bmin, cell_size=bsz / VG, vol=cp.zeros(tuple(2 * np.array(gres) + 1)))
    ns["compute_fluid_volume"](p, fv, p.vol)
    out.update(lphi=np.asarray(ls.phi), lvol=np.asarray(fv.vol))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  {name}: gres={gres} particles={p.num_particles} faces with mass "
          f"{[int((out[f'g{c}_m'] > 0).sum()) for c in 'xyz']} fluid cells {(out['lphi'] < 0).sum()} "
          f"vol cells {(out['lvol'] > 0).sum()}")


if __name__ == "__main__":
    gen("pt_a_12", (12, 12, 12), 31)
    gen("pt_b_10x14x12", (10, 14, 12), 32, per_cell=2)
