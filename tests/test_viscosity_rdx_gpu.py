"""Small problems: the viscosity CG's two vector phases, the r.r reduction and the bookkeeping in ONE launch whose resident
workgroups exchange their partial sums (csrc/mfs_cg_core.h k_update_rdx; 2 launches per iteration) against the three-launch
loop: the same arithmetic per element, r.r grouped by 128 workgroups instead of the update kernel's grid -- so equal to
rounding (fp64 state 1e-11 on the history), same iteration count on a converged solve.  And the one failure that can
happen on healthy hardware: a launch that is not fully resident times out clean and the solve carries on in the three-launch
loop, ending bit for bit where that loop ends.  GPU only."""
import os

import numpy as np
import pytest

from conftest import require_default_engine
import torch

from mfs import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _solver(gres, sc, dt, merged, check_every=8):
    import solver.ViscosityCGSolver3D as V
    s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dt, device=DEV, check_every=check_every)
    s._engine.set_merged(merged)
    s._engine.set_resident(False)       # (round 3: small grids default to the resident loop; this file pins the merged vector phases)
    return s


def _solve(s, sc, mu, tol):
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    s.solve(sc["dt"], mu, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=tol)
    torch.cuda.synchronize()
    return dict(it=s.iterations, hist=s.history, v=(vx, vy, vz), **{k: s._flat[k].clone() for k in "xdrq"})


@pytest.mark.parametrize("dt", ["fp32", "fp64"])
@pytest.mark.parametrize("gres", [(12, 12, 12), (20, 24, 36), (48, 80, 48), (33, 17, 8), (9, 11, 13), (64, 64, 64)],
                         ids=lambda g: "x".join(map(str, g)))
def test_merged_vector_phases_match_three_launch_loop(gres, dt):
    require_default_engine("test_merged_vector_phases_match_three_launch_loop")
    sc = scenes.viscosity_scene_3d(gres, seed=5, device=DEV, noise=0.3)
    a_s, b_s = _solver(gres, sc, dt, True), _solver(gres, sc, dt, False)
    a, b = _solve(a_s, sc, 40.0, 1e-7), _solve(b_s, sc, 40.0, 1e-7)
    assert a_s._engine.loop_info()["merged_vector_phases"] and not b_s._engine.loop_info()["merged_vector_phases"]
    n = min(len(a["hist"]), len(b["hist"]), 41)
    np.testing.assert_allclose(a["hist"][:n], b["hist"][:n], rtol=1e-11 if dt == "fp64" else 1e-5)
    assert abs(a["it"] - b["it"]) <= 1, (a["it"], b["it"])
    for k in "x":            # converged at tol = 1e-7: the two loops may stop one iteration apart
        ref = b[k].double()
        assert float((a[k].double() - ref).abs().max()) <= (1e-7 if dt == "fp64" else 1e-4) * float(ref.abs().max()), k
    for p, q in zip(a["v"], b["v"]):
        assert float((p.double() - q.double()).abs().max()) <= (1e-7 if dt == "fp64" else 1e-4) * float(q.double().abs().max())


def test_large_grids_keep_the_three_launch_loop():
    from mfs.vcg import VcgEngine
    eng = VcgEngine((128, 128, 128), torch.float32, DEV)
    vecs = [eng.new_vector()[0] for _ in range(5)]
    eng.bind(*vecs)
    dbl = tuple(2 * v + 1 for v in (128, 128, 128))
    one = torch.ones(dbl, dtype=torch.float64, device=DEV)
    eng.setup(1e-3, 1.0, one, one)
    assert not eng.loop_info()["merged_vector_phases"]


def test_a_launch_that_is_not_fully_resident_falls_back(monkeypatch):
    """fault injection: workgroup 5 of every merged launch never publishes its record -- what a GPU shared with other work
    does to a launch that needs all its workgroups at once.  Nothing may have been written by such a launch: the solve
    carries on in the three-launch loop and ends bit for bit where that loop ends."""
    require_default_engine("test_a_launch_that_is_not_fully_resident_falls_back")
    gres = (20, 24, 36)
    sc = scenes.viscosity_scene_3d(gres, seed=9, device=DEV, noise=0.3)
    ref = _solve(_solver(gres, sc, "fp64", False), sc, 40.0, 1e-8)
    monkeypatch.setenv("MFS_RDX_TEST_DROP_WG", "5")
    monkeypatch.setenv("MFS_RDX_TIMEOUT_MS", "30")
    s = _solver(gres, sc, "fp64", True)
    got = _solve(s, sc, 40.0, 1e-8)
    assert not s._engine.loop_info()["merged_vector_phases"]          # switched off for good
    assert got["it"] == ref["it"]
    assert np.array_equal(got["hist"], ref["hist"])
    for k in "xrd":
        assert torch.equal(got[k], ref[k]), k
    for p, q in zip(got["v"], ref["v"]):
        assert torch.equal(p, q)
