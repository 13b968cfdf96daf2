#!/usr/bin/env python3
"""Generate tests/golden/sdf_*.npz by EXECUTING THE REFERENCE'S OWN solver/sdf3D.py
(container only, tests/golden/refshim plumbing): a notebook-like scene -- a flipped container box, slanted
obstacle boxes (ipynb code cell 9), plus spheres -- built with the reference's generate_rb / set_vel_rb, then
its evaluate() and project() on random points in and around the bodies.  Cylinders are left out: the reference's
cylinder_eval reads an unassigned variable for points within the cylinder's height range (sdf3D.py:148-160).
Needs /root/reference."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MFS_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "refshim"))

import numpy as np  # noqa: E402
import cupy as cp   # noqa: E402  (refshim)

import solver.sdf3D as RS  # noqa: E402  (reference module)

assert RS.__file__.startswith(REF)


def scene(kind):
    if kind == "notebook":   # the five boxes of ipynb code cell 9
        h = 0.45
        return [("cube", ['box', 0.5, 0.8, 0.5], True, [0, 0.5, 0], [0, 1, 0], 0),
                ("cube1", ['box', 0.67, 0.1, 1.0], False, [-0.34, h, 0], [0, 0, 1], -45),
                ("cube2", ['box', 0.67, 0.1, 1.0], False, [0.34, h, 0], [0, 0, 1], 45),
                ("cube3", ['box', 1.0, 0.1, 0.7], False, [0, h, -0.3], [1, 0, 0], 45),
                ("cube4", ['box', 1.0, 0.1, 0.7], False, [0, h, 0.3], [1, 0, 0], -45)]
    return [("dome", ['sphere', 0.45], True, [0.02, 0.5, -0.01], [0, 1, 0], 0),
            ("ball", ['sphere', 0.12], False, [0.1, 0.4, 0.05], [0, 1, 0], 0),
            ("slab", ['box', 0.3, 0.06, 0.25], False, [-0.1, 0.6, 0.0], [1, 2, 0.5], 30)]


def gen(name, kind, seed, n=3000, dtype=np.float64):
    rb_d, rb_map = cp.zeros((0, 10, 4)), {}
    for nm, par, flip, c, ax, ang in scene(kind):
        rb_d, rb_map = RS.generate_rb(rb_d, rb_map, nm, par, flip=flip, center=c, axis=np.array(ax, dtype=np.float64),
                                      angle=ang)
    RS.set_vel_rb(rb_d, 1, cp.array([0.3, -0.2, 0.1]))
    rng = np.random.default_rng(seed)
    pos = rng.uniform([-0.4, -0.05, -0.4], [0.4, 1.05, 0.4], size=(n, 3)).astype(dtype)
    sd, vel = cp.zeros(n), cp.zeros((n, 3))
    RS.evaluate(rb_d, sd, vel, cp.array(pos))
    proj = cp.array(pos)
    RS.project(rb_d, proj)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="sdf3d", rb_d=np.asarray(rb_d), position=pos,
                        sd=np.asarray(sd), vel=np.asarray(vel), projected=np.asarray(proj))
    moved = int((np.abs(np.asarray(proj) - pos).max(axis=1) > 1e-12).sum())
    print(f"  {name}: bodies={rb_d.shape[0]} points={n} inside-solid={(np.asarray(sd) <= 0).sum()} moved={moved}")


if __name__ == "__main__":
    gen("sdf_a_notebook", "notebook", 41)
    gen("sdf_b_spheres", "mixed", 42)
    gen("sdf_c_notebook_f32", "notebook", 43, n=1500, dtype=np.float32)
