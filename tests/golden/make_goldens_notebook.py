#!/usr/bin/env python3
"""Generate tests/golden/nb_*.npz for the SURVEY.md section 8(f) rank-1 rows -- the notebook's grid kernels
that bracket the two solves: `extrapolate` (code cell 7) and `apply_boundary_condition` (code cell 5) --
by EXECUTING THE NOTEBOOK'S OWN CELL SOURCE (container only).

The cells are read out of /root/reference/3D_viscous_fluid_sim.ipynb as JSON at run time (nothing of the
reference's text is stored in this repo) and exec()'d with the same container-only plumbing as
make_goldens.py (tests/golden/refshim: numpy as the array container, a sequential per-thread launcher
for @cuda.jit bodies).  `edict` objects are replaced by SimpleNamespace.  Needs /root/reference.
"""
import json
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
    ns = {"cp": cp, "cuda": cuda, "np": np}
    want = ("def apply_boundary_condition(", "def extrapolate(gres, num_iter, vx, vy, vz, mx, my, mz)")
    found = 0
    for c in nb["cells"]:
        if c["cell_type"] != "code":
            continue
        src = "".join(c["source"])
        if any(w in src for w in want):
            exec(compile(src, "<notebook cell>", "exec"), ns)
            found += 1
    assert found == 2, found
    return ns


def C(a, dtype=None):
    return cp.array(np.array(a, dtype=dtype, copy=True))


def grid_inputs(gres, seed):
    """velocities / masses like the notebook's grid after p2g: mass > 0 in and around the fluid block,
    exactly 0 elsewhere; fp32 like the notebook's grid arrays (ipynb c10:76-78)."""
    sc = scenes.viscosity_scene_3d(gres, seed=seed, vel_dtype=np.float32, noise=0.2)
    rng = np.random.default_rng(seed + 50)
    out = dict(sphi=sc["sphi"], cell=sc["cell_size"])
    sv = np.stack([0.2 * np.sin(3 * sc["sphi"]), -0.1 + 0 * sc["sphi"], 0.05 * np.cos(2 * sc["sphi"])], axis=-1)
    out["sv"] = sv
    for c, k in zip("xyz", ("vx", "vy", "vz")):
        v = sc[k]
        m = ((np.abs(v) > 0) * rng.uniform(0.2, 1.5, size=v.shape)).astype(np.float32)
        out["v" + c] = (v + 0.3 * (m > 0) * rng.standard_normal(v.shape)).astype(np.float32)
        out["m" + c] = m
    return out


def gen(name, gres, seed):
    ns = notebook_namespace()
    g = grid_inputs(gres, seed)
    dx = float(g["cell"][0])
    # --- extrapolate(GRES, 2, ...)  (ipynb:4652)
    ex = [C(g["vx"]), C(g["vy"]), C(g["vz"])]
    ns["extrapolate"](C(gres, np.int64), 2, *ex, C(g["mx"]), C(g["my"]), C(g["mz"]))
    # --- apply_boundary_condition(grid, solid_levelset, GDX)  (ipynb:4655)
    N = types.SimpleNamespace
    res = lambda a: C(np.array(gres) + np.eye(3, dtype=np.int64)[a], np.int64)  # noqa: E731
    grid = N(x=N(v=C(ex[0]), m=C(g["mx"]), dv=cp.zeros(ex[0].shape, dtype=cp.float32), resolution=res(0)),
             y=N(v=C(ex[1]), m=C(g["my"]), dv=cp.zeros(ex[1].shape, dtype=cp.float32), resolution=res(1)),
             z=N(v=C(ex[2]), m=C(g["mz"]), dv=cp.zeros(ex[2].shape, dtype=cp.float32), resolution=res(2)))
    solid = N(phi=C(g["sphi"]), v=C(g["sv"]))
    cuda.ignore_oob = True      # boundary_condition_* store dv[x,y,z] = 0 before checking x,y,z against the shape
    try:
        with np.errstate(all="ignore"):
            ns["apply_boundary_condition"](grid, solid, dx)
    finally:
        cuda.ignore_oob = False
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="notebook_grid", gres=np.array(gres), dx=dx,
                        in_vx=g["vx"], in_vy=g["vy"], in_vz=g["vz"], mx=g["mx"], my=g["my"], mz=g["mz"],
                        sphi=g["sphi"], sv=g["sv"],
                        ex_vx=np.asarray(ex[0]), ex_vy=np.asarray(ex[1]), ex_vz=np.asarray(ex[2]),
                        dvx=np.asarray(grid.x.dv), dvy=np.asarray(grid.y.dv), dvz=np.asarray(grid.z.dv),
                        bc_vx=np.asarray(grid.x.v), bc_vy=np.asarray(grid.y.v), bc_vz=np.asarray(grid.z.v))
    nz = [int(np.count_nonzero(np.asarray(d))) for d in (grid.x.dv, grid.y.dv, grid.z.dv)]
    print(f"  {name}: gres={gres} nonzero dv {nz}, extrapolated faces "
          f"{[int((np.asarray(a) != b).sum()) for a, b in zip(ex, (g['vx'], g['vy'], g['vz']))]}")


if __name__ == "__main__":
    gen("nb_a_12", (12, 12, 12), 21)
    gen("nb_b_10x14x12", (10, 14, 12), 22)
