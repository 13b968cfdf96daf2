"""The "chaotic history" argument as an assertion.

CG on the reference's operators is rounding-chaotic beyond a leading window (tests/test_oracle_sensitivity.py), which is
why the parity tests compare histories entry by entry only over that window.  Here the deviation is bounded over the WHOLE
solve: tests/golden/envelope_<name>.npz (tests/golden/make_envelope.py) holds, per history entry k, how far the oracle's C
restatement moves from the executed reference's history when only ROUNDING changes -- 80 variants: 16 summation orders of
the dot products x 5 fused-multiply-add placements,

    E_k = max over the variants |h_k - golden_k| / golden_k,

and the range of iteration counts over the ensemble.

  CPU: the ensemble regenerated here reproduces the committed envelope (the oracle is deterministic: fixed-order dot
       products whatever the thread count), variant 0 IS the golden history to 1e-10, and the envelope is what the argument
       says -- tight at the start, O(1) later on the pressure scenes, tight throughout on the well-conditioned viscosity.
  GPU: the HIP solvers (fp64 state, default engine) satisfy  dev_k <= 4 E_k + 1e-9  for EVERY k, and their iteration count
       lies inside the ensemble's range.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden

sys.path.insert(0, GOLDEN_DIR)
import make_envelope as ME  # noqa: E402

NAMES = ME.NAMES


def load_env(name):
    with np.load(os.path.join(GOLDEN_DIR, f"envelope_{name}.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", NAMES)
def test_envelope_fixture_reproduces(name):
    env, now = load_env(name), ME.envelope(name)
    assert len(ME.VARIANTS) >= 8 and np.array_equal(env["variants"], now["variants"])
    assert int(env["iters_min"]) == now["iters_min"] and int(env["iters_max"]) == now["iters_max"]
    # bit-identical on the box that wrote it; libm's fma and the compiler may differ in the last bits elsewhere, and
    # the history amplifies those -- the regenerated envelope must stay within a factor 4 wherever it is resolved
    a, b = env["E"], now["E"]
    assert a.shape == b.shape
    big = np.maximum(a, b) > 1e-9
    assert np.all(b[big] <= 4 * a[big] + 1e-9) and np.all(a[big] <= 4 * b[big] + 1e-9)


@pytest.mark.parametrize("name", NAMES)
def test_variant_zero_is_the_golden_history(name):
    g = golden(name)
    h, it, _ = ME.run_variant(g, name, 0, 0)
    n = min(21, len(h), len(g["history"]))
    np.testing.assert_allclose(h[:n], g["history"][:n], rtol=1e-10)
    env = load_env(name)
    assert int(env["iters_min"]) <= it <= int(env["iters_max"])
    assert int(env["iters_min"]) <= int(g["iters"]) <= int(env["iters_max"])


def test_the_envelope_says_what_the_window_argument_says():
    """pressure pool scenes and the mu = 50 viscosity scene: entry-wise agreement to 1e-9 holds over the first 10 iterations
    and is lost (E_k > 1e-2) later -- the window of the parity tests is the operator's conditioning, not slack;
    viscosity at mu = 1: 1e-12 throughout."""
    for name in NAMES:
        E = load_env(name)["E"]
        assert E[:21].max() < 1e-9, (name, E[:21].max())
        if name == "v3d_d_24":
            assert E.max() < 1e-12
        else:
            assert E.max() > 1e-2, (name, E.max())


# ------------------------------------------------------------------ GPU --------------------------------------------
def _gpu_history(name):
    import torch
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    dev = "cuda:0"
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    if name.startswith("p3d"):
        import solver.CGSolverBuffer as B
        import solver.PressureCGSolver3D as P
        s = P.PressureCGSolver3D(B.CGSolverBuffer(gres, precision="fp64", device=dev), gres, g["bound_size"])
        s.solve(T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"]), T(g["sphi"]), T(g["sv"]), T(g["lphi"]), tol=float(g["tol"]))
    else:
        import solver.ViscosityCGSolver3D as V
        s = V.ViscosityCGSolver3D(gres, g["bound_size"], precision="fp64", device=dev)
        s.solve(float(g["dt"]), float(g["mu"]), float(g["rho"]), T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"]), T(g["sphi"]),
                T(g["sv"]), T(g["lphi"]), T(g["lvol"]), tol=float(g["tol"]))
    return np.asarray(s.history, np.float64), int(s.iterations), g


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_history_stays_inside_the_rounding_envelope(name):
    h, it, g = _gpu_history(name)
    env = load_env(name)
    hg, E = np.asarray(g["history"], np.float64), env["E"]
    n = min(len(h), len(hg))
    dev = np.abs(h[:n] - hg[:n]) / np.abs(hg[:n])
    bad = np.nonzero(dev > 4 * E[:n] + 1e-9)[0]
    assert bad.size == 0, (name, [(int(k), float(dev[k]), float(E[k])) for k in bad[:8]])
    assert int(env["iters_min"]) <= it <= int(env["iters_max"]), (it, int(env["iters_min"]), int(env["iters_max"]))
    assert len(h) == 2 * it + 1
