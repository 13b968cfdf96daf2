#!/usr/bin/env python3
"""Rounding envelopes of the residual history: how far does the reference's CG history move when ONLY
rounding changes?  The oracle's C restatement (oracle/mfs_oracle_c.c) is run on a golden's stored inputs in every rounding
variant it offers -- 16 summation orders of the dot products x {no FMA, FMA in the dots, in the vector updates, in the
operator, everywhere} = 80 runs, each the reference's algorithm statement for statement -- and per history entry k

    E_k = max over the variants of |h_k - golden_k| / golden_k        (golden = the EXECUTED reference's history)

is stored with the range of iteration counts and the spread of the converged field.  tests/test_history_envelope.py
regenerates the ensemble (CPU) and holds the HIP solvers to `dev_k <= 4 E_k + 1e-9` for EVERY k of the whole solve (GPU).

Needs no reference import: inputs and the golden history come from the committed tests/golden/*.npz.
    python tests/golden/make_envelope.py            # rewrites tests/golden/envelope_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "python-fluid-simulation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import cbaseline as CB  # noqa: E402

NAMES = ("p3d_f_40x36x32_sv", "p3d_d_20", "v3d_d_24", "v3d_c_16_mu50")
VARIANTS = [(dv, fm) for dv in range(16) for fm in (0, 1, 2, 4, 7)]


def run_variant(g, name, dv, fm):
    """one run of the C oracle on golden `g` in rounding variant (dv, fm): (history, iterations, x)"""
    gres = tuple(int(v) for v in g["gres"])
    CB.set_variant(dv, fm)
    try:
        cap = 4 * len(g["history"]) + 64
        if name.startswith("p3d"):
            res = CB.cg(gres, g["b"], g["lphi"], g["wx"], g["wy"], g["wz"], float(g["tol"]), int(np.prod(gres)), cap)
        else:
            cell_vol = float(np.prod(g["bound_size"] / g["gres"]))
            scale, mu = float(g["dt"]) / cell_vol / float(g["rho"]), float(g["mu"])
            vol = g["lvol"] / (cell_vol * 0.125)
            b = np.concatenate([g[k].ravel() for k in ("bx", "by", "bz")])
            x0 = np.concatenate([g[k].ravel() for k in ("ex", "ey", "ez")])
            res = CB.visc_cg(gres, scale, mu, b, x0, g["sphi"], vol, float(g["tol"]), int(np.prod(gres)), cap)
    finally:
        CB.set_variant(0, 0)
    assert res["converged"], (name, dv, fm)
    return res["history"], res["iterations"], res["x"].ravel()


def envelope(name):
    with np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    hg = np.asarray(g["history"], np.float64)
    xg = g["x"].ravel() if "x" in g else np.concatenate([g[k].ravel() for k in ("x_x", "x_y", "x_z")])
    E = np.zeros(len(hg))
    iters, xdev = [], []
    for dv, fm in VARIANTS:
        h, it, x = run_variant(g, name, dv, fm)
        n = min(len(h), len(hg))
        E[:n] = np.maximum(E[:n], np.abs(h[:n] - hg[:n]) / np.abs(hg[:n]))
        iters.append(it)
        xdev.append(float(np.max(np.abs(x - xg)) / np.max(np.abs(xg))))
    return dict(E=E, iters_min=min(iters), iters_max=max(iters), iters=np.array(iters), x_dev_max=max(xdev),
                golden_iters=int(g["iters"]), variants=np.array(VARIANTS))


def main():
    for name in NAMES:
        env = envelope(name)
        np.savez_compressed(os.path.join(HERE, f"envelope_{name}.npz"), **env)
        E = env["E"]
        print(f"{name}: golden {env['golden_iters']} iterations, ensemble {env['iters_min']}..{env['iters_max']}; "
              f"E_k at k = 1, 10%, 50%, last: {E[1]:.1e} {E[len(E) // 10]:.1e} {E[len(E) // 2]:.1e} {E[-1]:.1e}; "
              f"max {E.max():.1e}; converged x differs by up to {env['x_dev_max']:.1e} of its maximum")


if __name__ == "__main__":
    main()
