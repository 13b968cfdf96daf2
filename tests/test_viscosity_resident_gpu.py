"""Round 3: the viscosity CG loop of a SMALL grid as one resident launch per batch (csrc/mfs_vcg_resident.h) against the
launch-per-phase loop (marching kernel | r update | direction + x update).  The arithmetic per element is the same -- the rows
are the same code -- and only the grouping of the two dot products differs, so:

  fp64 state   residual history 1e-11 rel over the first 20 iterations, iteration count +-1, x to 1e-7 of its maximum
  fp32 state   history 1e-5, x 1e-4
  batches      iterate(3) + iterate(5) == iterate(8), bit for bit (the state makes the round trip through the bound arrays)
  a launch that is not fully resident (fault injection: one workgroup never shows up) times out at its first dot product
  having written nothing: the solve carries on in the launch-per-phase loop and ends bit for bit where that loop ends.
GPU only."""
import numpy as np
import pytest

from conftest import require_default_engine
import torch

from mfs import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _solver(gres, dt, resident, check_every=8):
    import solver.ViscosityCGSolver3D as V
    sc = None
    s = V.ViscosityCGSolver3D(gres, (1.0, 1.0, 1.0), precision=dt, device=DEV, check_every=check_every)
    s._engine.set_resident(resident)
    return s


def _solve(s, sc, mu, tol):
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    s.solve(sc["dt"], mu, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=tol)
    torch.cuda.synchronize()
    return dict(it=s.iterations, hist=np.asarray(s.history), v=(vx, vy, vz), **{k: s._flat[k].clone() for k in "xdrq"})


GRIDS = [(12, 12, 12), (20, 24, 36), (48, 80, 48), (33, 17, 8), (9, 11, 13), (16, 16, 14), (40, 36, 32)]


@pytest.mark.parametrize("dt", ["fp32", "fp64"])
@pytest.mark.parametrize("gres", GRIDS, ids=lambda g: "x".join(map(str, g)))
def test_resident_loop_matches_launch_per_phase_loop(gres, dt):
    require_default_engine("test_resident_loop_matches_launch_per_phase_loop")
    sc = scenes.viscosity_scene_3d(gres, seed=5, device=DEV, noise=0.3)
    a_s, b_s = _solver(gres, dt, True), _solver(gres, dt, False)
    a, b = _solve(a_s, sc, 40.0, 1e-7), _solve(b_s, sc, 40.0, 1e-7)
    assert a_s._engine.loop_info()["resident"], "the grid was expected to qualify for the resident loop"
    assert not b_s._engine.loop_info()["resident"]
    n = min(len(a["hist"]), len(b["hist"]), 41)
    np.testing.assert_allclose(a["hist"][:n], b["hist"][:n], rtol=1e-11 if dt == "fp64" else 1e-5)
    assert abs(a["it"] - b["it"]) <= 1, (a["it"], b["it"])
    assert len(a["hist"]) == 2 * a["it"] + 1 and a_s.delta == a["hist"][-1]
    ref = b["x"].double()
    assert float((a["x"].double() - ref).abs().max()) <= (1e-7 if dt == "fp64" else 1e-4) * float(ref.abs().max())
    for p, q in zip(a["v"], b["v"]):
        assert float((p.double() - q.double()).abs().max()) <= (1e-7 if dt == "fp64" else 1e-4) * float(q.double().abs().max())


@pytest.mark.parametrize("dt", ["fp32", "fp64"])
def test_state_round_trip_between_batches(dt):
    """iterate(3) + iterate(5) == iterate(8): x, r, d, q and the history, bit for bit -- and a batch of the launch-per-phase
    loop can follow a resident one (the state in the arrays is the reference's state after every iteration)"""
    require_default_engine("test_state_round_trip_between_batches")
    from mfs.vcg import VcgEngine
    gres = (20, 24, 36)
    sc = scenes.viscosity_scene_3d(gres, seed=9, device=DEV, noise=0.3)
    tdt = torch.float32 if dt == "fp32" else torch.float64
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale, vol = sc["dt"] / cell_vol / sc["rho"], sc["lvol"] / (cell_vol * 0.125)
    outs = []
    for batches in ((8,), (3, 5), (1, 1, 6)):
        eng = VcgEngine(gres, tdt, DEV)
        eng.setup(scale, 40.0, sc["sphi"], vol)
        vecs = [eng.new_vector()[0] for _ in range(5)]
        g = torch.Generator(device=DEV).manual_seed(1)
        vecs[0].copy_(torch.randn(vecs[0].shape, generator=g, device=DEV, dtype=torch.float64).to(tdt))
        # b only on interior, non-solid faces (as the RHS kernel leaves it); x = 0
        _, bv = eng.new_vector()
        valid = [sc["sphi"][0::2, 1::2, 1::2] >= 0, sc["sphi"][1::2, 0::2, 1::2] >= 0, sc["sphi"][1::2, 1::2, 0::2] >= 0]
        o = 0
        for t, m in zip(bv, valid):
            n = t.numel()
            blk = vecs[0][o:o + n].view(t.shape)
            keep = torch.zeros_like(blk, dtype=torch.bool)
            keep[1:-1, 1:-1, 1:-1] = m[1:-1, 1:-1, 1:-1]
            blk.mul_(keep)
            o += n
        eng.bind(*vecs)
        assert eng.loop_info()["resident"]
        eng.begin(0.0)
        for nb in batches:
            eng.iterate(nb)
        torch.cuda.synchronize()
        outs.append(([v.clone() for v in vecs], np.asarray(eng.history())))
    for vs, h in outs[1:]:
        np.testing.assert_array_equal(h, outs[0][1])
        for a, b in zip(vs, outs[0][0]):
            assert torch.equal(a, b)
    assert len(outs[0][1]) == 17


def test_the_default_engine_takes_the_resident_loop_on_the_notebook_grid():
    require_default_engine("test_the_default_engine_takes_the_resident_loop_on_the_notebook_grid")
    from mfs.vcg import VcgEngine
    eng = VcgEngine((48, 80, 48), torch.float64, DEV)
    vecs = [eng.new_vector()[0] for _ in range(5)]
    eng.bind(*vecs)
    dbl = tuple(2 * v + 1 for v in (48, 80, 48))
    one = torch.ones(dbl, dtype=torch.float64, device=DEV)
    eng.setup(1e-3, 1.0, one, one)
    assert eng.loop_info()["resident"]
    big = VcgEngine((128, 128, 128), torch.float32, DEV)
    vecs = [big.new_vector()[0] for _ in range(5)]
    big.bind(*vecs)
    one = torch.ones(tuple(2 * v + 1 for v in (128, 128, 128)), dtype=torch.float64, device=DEV)
    big.setup(1e-3, 1.0, one, one)
    assert not big.loop_info()["resident"]


def test_a_launch_that_is_not_fully_resident_falls_back(monkeypatch):
    """fault injection: workgroup 5 of every resident launch never shows up -- what a GPU shared with other work does to a
    launch that needs all its workgroups at once.  Nothing may have been written by such a launch: the poll switches the engine
    to the launch-per-phase loop and the solve ends bit for bit where that loop ends."""
    require_default_engine("test_a_launch_that_is_not_fully_resident_falls_back")
    gres = (20, 24, 36)
    sc = scenes.viscosity_scene_3d(gres, seed=9, device=DEV, noise=0.3)
    ref = _solve(_solver(gres, "fp64", False), sc, 40.0, 1e-8)
    monkeypatch.setenv("MFS_VRES_TEST_DROP_WG", "5")
    monkeypatch.setenv("MFS_VRES_FIRST_TIMEOUT_MS", "20")
    s = _solver(gres, "fp64", True)
    got = _solve(s, sc, 40.0, 1e-8)
    assert not s._engine.loop_info()["resident"], "the engine should have left the resident loop"
    assert got["it"] == ref["it"]
    assert np.array_equal(got["hist"], ref["hist"])
    for k in "xrd":
        assert torch.equal(got[k], ref[k]), k
    for p, q in zip(got["v"], ref["v"]):
        assert torch.equal(p, q)


def test_two_solves_through_one_engine():
    require_default_engine("test_two_solves_through_one_engine")
    gres = (20, 24, 36)
    sc = scenes.viscosity_scene_3d(gres, seed=9, device=DEV, noise=0.3)
    s = _solver(gres, "fp64", True)
    a = _solve(s, sc, 40.0, 1e-8)
    b = _solve(s, sc, 40.0, 1e-8)
    assert s._engine.loop_info()["resident"]
    assert a["it"] == b["it"] and np.array_equal(a["hist"], b["hist"])
    for k in "xrdq":
        assert torch.equal(a[k], b[k]), k
