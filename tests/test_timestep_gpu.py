"""Whole time steps of the notebook's loop (ipynb:4571-4667, solver == 'apic') on the MI355X drop-ins
(notebook_sim.NotebookSimulation) against goldens produced by executing the notebook's cells and the reference's
solver/ package step by step (tests/golden/make_goldens_step.py).  This is BASELINE config 5's pipeline on one GPU
at test size.  Tolerances: the three CG solves stop at the reference's absolute tol = 1e-3 and their iterates are
chaotic in rounding, so the state after a step agrees to solver-tolerance level, not to rounding."""
import numpy as np
import pytest
import torch

from conftest import golden, require_default_engine
import notebook_sim as NSIM
import solver.sdf3D as sdf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = lambda t: t.detach().cpu().numpy()  # noqa: E731


def build(g, **kw):
    gres = tuple(int(v) for v in g["gres"])
    gdx = float(g["gdx"])
    size = np.array(gres) * gdx
    rb_d, rb_map = sdf.generate_rb(None, {}, 'cube', ['box', size[0] - 2 * gdx, size[1] - 2 * gdx, size[2] - 2 * gdx], flip=True,
                                   center=[0, size[1] / 2, 0], axis=[0., 1, 0], angle=0, device=DEV)
    rb_d, rb_map = sdf.generate_rb(rb_d, rb_map, 'ramp', ['box', 0.45, 0.05, 0.8], flip=False, center=[-0.12, 0.2, 0],
                                   axis=[0., 0, 1], angle=-35)
    np.testing.assert_allclose(N(rb_d), g["rb_d"], rtol=0, atol=1e-16)
    sim = NSIM.NotebookSimulation(gres, gdx, [-0.3, 0, -0.3], rb_d, g["px0"], float(g["pdx"]), rho=float(g["rho"]),
                                  mu=float(g["mu"]), dt=float(g["dt"]), device=DEV, **kw)
    sim.particle.v.copy_(torch.as_tensor(g["pv0"], device=DEV))
    return sim


def test_scene_setup_matches_reference():
    g = golden("step_a_12x16x12")
    sim = build(g)
    np.testing.assert_allclose(N(sim.solid_levelset.phi), g["sphi"], rtol=1e-13, atol=1e-15)


def test_two_full_steps():
    g = golden("step_a_12x16x12")
    sim = build(g)
    timings = {}
    for s in range(int(g["steps"])):
        dt = sim.step(timings=timings)
        assert dt == pytest.approx(float(g["dts"][s]), rel=1e-12)
        px, pv = N(sim.particle.x), N(sim.particle.v)
        move = np.abs(g[f"px{s + 1}"] - g["px0"]).max()
        # positions: the step moves particles by `move`; agreement to 1e-4 of that
        np.testing.assert_allclose(px, g[f"px{s + 1}"], rtol=0, atol=1e-4 * move * (s + 1))
        np.testing.assert_allclose(pv, g[f"pv{s + 1}"], rtol=0, atol=2e-3 * np.abs(g[f"pv{s + 1}"]).max())
        np.testing.assert_allclose(N(sim.fluid_levelset.phi), g[f"lphi{s + 1}"], rtol=0, atol=1e-4 * float(g["gdx"]))
        gvy = g[f"gvy{s + 1}"]
        np.testing.assert_allclose(N(sim.grid.y.v), gvy, rtol=0, atol=5e-3 * np.abs(gvy).max())
    assert sim.iterations == 2 and set(timings) >= {"density", "viscosity", "pressure", "p2g", "g2p"}
    assert sim.PressureSolver.iterations > 0 and sim.ViscositySolver.iterations > 0 and sim.DensitySolver.iterations > 0


def test_two_full_steps_with_the_jacobi_option():
    """NotebookSimulation(..., jacobi=True): the three CG solves Jacobi-preconditioned (opt-in; NOT the reference's iterations).
    The PARTICLES -- the state the simulation carries from step to step -- agree with the executed-reference goldens within the
    tolerances of the default path (positions 1e-4 of the move, velocities 2e-3, level set 1e-4 cell), with a fraction of the CG
    iterations.  The raw grid velocities differ on nearly empty faces by the reference's own truncation error there (a tiny
    diagonal turns a large error into a tiny residual: tests/test_viscosity_jacobi_gpu.py), hence no grid assertion here."""
    g = golden("step_a_12x16x12")
    sim, ref = build(g, jacobi=True), build(g)
    for s in range(int(g["steps"])):
        dt = sim.step()
        ref.step()
        assert dt == pytest.approx(float(g["dts"][s]), rel=1e-12)
        px, pv = N(sim.particle.x), N(sim.particle.v)
        move = np.abs(g[f"px{s + 1}"] - g["px0"]).max()
        np.testing.assert_allclose(px, g[f"px{s + 1}"], rtol=0, atol=1e-4 * move * (s + 1))
        np.testing.assert_allclose(pv, g[f"pv{s + 1}"], rtol=0, atol=2e-3 * np.abs(g[f"pv{s + 1}"]).max())
        np.testing.assert_allclose(N(sim.fluid_levelset.phi), g[f"lphi{s + 1}"], rtol=0, atol=1e-4 * float(g["gdx"]))
        for a, b in ((sim.DensitySolver, ref.DensitySolver), (sim.PressureSolver, ref.PressureSolver),
                     (sim.ViscositySolver, ref.ViscositySolver)):
            assert a.iterations <= b.iterations and (b.iterations == 0 or 2 * a.iterations <= b.iterations + 2), (a.iterations, b.iterations)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_two_full_steps_slab_decomposed(world, tmp_path):
    """BASELINE config 5 on N ranks (notebook_sim.SlabNotebookSimulation; the ranks are processes sharing the one GPU of
    the box): viscosity and pressure CG slab-decomposed, the other stages replicated -- against the same goldens with
    the same tolerances as the single-GPU step, and every rank ends the step with the same state."""
    from test_p2p_gpu import _run_ranks
    g = golden("step_a_12x16x12")
    res = _run_ranks("step_a_12x16x12", world, tmp_path, "f64", P2P_TEST_MODE="timestep")
    for r in res:
        assert int(r["p_iters"]) > 0 and int(r["v_iters"]) > 0
        assert {"viscosity", "pressure", "broadcast", "gather", "density", "p2g", "g2p"} <= set(str(x) for x in r["stages"])
        for s in range(int(g["steps"])):
            assert float(r[f"dt{s + 1}"]) == pytest.approx(float(g["dts"][s]), rel=1e-12)
            move = np.abs(g[f"px{s + 1}"] - g["px0"]).max()
            np.testing.assert_allclose(r[f"px{s + 1}"], g[f"px{s + 1}"], rtol=0, atol=1e-4 * move * (s + 1))
            np.testing.assert_allclose(r[f"pv{s + 1}"], g[f"pv{s + 1}"], rtol=0, atol=2e-3 * np.abs(g[f"pv{s + 1}"]).max())
            np.testing.assert_allclose(r[f"lphi{s + 1}"], g[f"lphi{s + 1}"], rtol=0, atol=1e-4 * float(g["gdx"]))
            gvy = g[f"gvy{s + 1}"]
            np.testing.assert_allclose(r[f"gvy{s + 1}"], gvy, rtol=0, atol=5e-3 * np.abs(gvy).max())
    for r in res[1:]:       # the gathered grid velocities are the same arrays on every rank
        for s in range(int(g["steps"])):
            np.testing.assert_array_equal(r[f"gvy{s + 1}"], res[0][f"gvy{s + 1}"])


@pytest.mark.parametrize("world", [1, 2, 3])
def test_two_full_steps_particles_sharded(world, tmp_path):
    """BASELINE config 5 with the particle stages sharded too (notebook_sim.ShardedNotebookSimulation): every rank owns
    the particles of its x-range, scatters / gathers only them, and exchanges plane BANDS (no whole-grid collective).
    Against the executed-reference goldens with the single-GPU tolerances: particles gathered by id, grid fields
    assembled from the ranks' own planes; the per-rank particle counts sum to the total; nothing is lost in migration."""
    from test_p2p_gpu import _run_ranks
    g = golden("step_a_12x16x12")
    res = _run_ranks("step_a_12x16x12", world, tmp_path, "f64", P2P_TEST_MODE="timestep_sharded")
    total = g["px0"].shape[0]
    cuts = sorted((int(r["own_lo"]), int(r["own_hi"])) for r in res)
    assert cuts[0][0] == 0 and cuts[-1][1] == int(g["gres"][0]) and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    for s in range(int(g["steps"])):
        move = np.abs(g[f"px{s + 1}"] - g["px0"]).max()
        lphi = np.zeros_like(g[f"lphi{s + 1}"])
        gvy = np.zeros_like(g[f"gvy{s + 1}"])
        for r in res:
            assert float(r[f"dt{s + 1}"]) == pytest.approx(float(g["dts"][s]), rel=1e-12)
            assert int(r[f"counts{s + 1}"].sum()) == total
            np.testing.assert_allclose(r[f"px{s + 1}"], g[f"px{s + 1}"], rtol=0, atol=1e-4 * move * (s + 1))
            np.testing.assert_allclose(r[f"pv{s + 1}"], g[f"pv{s + 1}"], rtol=0, atol=2e-3 * np.abs(g[f"pv{s + 1}"]).max())
            a, b = int(r["own_lo"]), int(r["own_hi"])
            lphi[a:b] = r[f"lphi{s + 1}"][a:b]
            gvy[a:b] = r[f"gvy{s + 1}"][a:b]
        np.testing.assert_allclose(lphi, g[f"lphi{s + 1}"], rtol=0, atol=1e-4 * float(g["gdx"]))
        want = g[f"gvy{s + 1}"]
        np.testing.assert_allclose(gvy, want, rtol=0, atol=5e-3 * np.abs(want).max())
    for r in res:
        assert int(r["p_iters"]) > 0 and int(r["v_iters"]) > 0 and int(r["d_iters"]) > 0
        if world > 1:
            assert int(r["band_bytes"]) > 0
    if world > 1:      # more than one rank really holds particles
        assert (res[0]["counts2"] > 0).sum() >= 2, res[0]["counts2"]


def test_sharded_step_on_an_rccl_group(tmp_path):
    """the sharded time step as the multi-GPU launchers run it -- on an "nccl" (RCCL) group, which has no CPU backend:
    the CFL all-reduce, the migration count exchange and gather_particles must move DEVICE operands (round 2 they were
    host tensors and raised on the first step()).  One rank (RCCL admits one rank per device); the collectives that a
    one-rank run short-cuts are driven directly in the worker (rccl_world1_mode), device planes travel through
    batch_isend_irecv.  Results against the executed-reference goldens as in the gloo runs."""
    from test_p2p_gpu import _run_ranks
    g = golden("step_a_12x16x12")
    res = _run_ranks("step_a_12x16x12", 1, tmp_path, "f64", P2P_TEST_MODE="rccl_world1", P2P_TEST_BACKEND="nccl")
    r = res[0]
    for s in range(int(g["steps"])):
        move = np.abs(g[f"px{s + 1}"] - g["px0"]).max()
        assert float(r[f"dt{s + 1}"]) == pytest.approx(float(g["dts"][s]), rel=1e-12)
        assert int(r[f"counts{s + 1}"].sum()) == g["px0"].shape[0]
        np.testing.assert_allclose(r[f"px{s + 1}"], g[f"px{s + 1}"], rtol=0, atol=1e-4 * move * (s + 1))
        np.testing.assert_allclose(r[f"pv{s + 1}"], g[f"pv{s + 1}"], rtol=0, atol=2e-3 * np.abs(g[f"pv{s + 1}"]).max())


@pytest.mark.parametrize("world", [2, 3])
def test_two_full_steps_particles_sharded_with_the_jacobi_option(world, tmp_path):
    """the sharded time step with all three solves Jacobi-preconditioned THROUGH THE WINDOW SLAB LOOPS (pressure, density:
    mfs_pcg3d_slab_*; viscosity: mfs_vcg3d_slab_*): the particles against the executed-reference goldens within the default
    path's tolerances, with fewer CG iterations than the reference's loops need on this scene"""
    require_default_engine("test_two_full_steps_particles_sharded_with_the_jacobi_option")
    from test_p2p_gpu import _run_ranks
    g = golden("step_a_12x16x12")
    res = _run_ranks("step_a_12x16x12", world, tmp_path, "f64", P2P_TEST_MODE="timestep_sharded", P2P_TEST_JACOBI="1",
                     P2P_TEST_TRANSPORT="p2p")
    total = g["px0"].shape[0]
    for s in range(int(g["steps"])):
        move = np.abs(g[f"px{s + 1}"] - g["px0"]).max()
        for r in res:
            assert int(r[f"counts{s + 1}"].sum()) == total
            np.testing.assert_allclose(r[f"px{s + 1}"], g[f"px{s + 1}"], rtol=0, atol=1e-4 * move * (s + 1))
            np.testing.assert_allclose(r[f"pv{s + 1}"], g[f"pv{s + 1}"], rtol=0, atol=2e-3 * np.abs(g[f"pv{s + 1}"]).max())
    for r in res:       # the last step's solves: 32 / 33 / 32 iterations with the reference's loops (density, viscosity, pressure)
        assert 0 < int(r["p_iters"]) <= 16 and 0 < int(r["v_iters"]) <= 12 and 0 < int(r["d_iters"]) <= 20, \
            (int(r["d_iters"]), int(r["v_iters"]), int(r["p_iters"]))


def test_sharded_step_with_migration_matches_single_gpu(tmp_path):
    """a 32^3 buckling-like scene whose fluid block moves one cell per step across the slab cuts: four steps on 3 ranks
    with sharded particles against the same four steps on one GPU (no golden at this size: the single-GPU path is the
    one pinned by the goldens above).  Particles really change owner; none is lost; positions and velocities agree to
    solver-tolerance level."""
    from test_p2p_gpu import _run_ranks
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    one = _run_ranks("step_a_12x16x12", 1, tmp_path / "a", "f64", P2P_TEST_MODE="timestep_synth", P2P_TEST_SINGLE="1")[0]
    res = _run_ranks("step_a_12x16x12", 3, tmp_path / "b", "f64", P2P_TEST_MODE="timestep_synth")
    assert sum(int(r["arrivals"]) for r in res) > 0, "no particle crossed a cut: the test would not exercise migration"
    px0 = one["px1"]
    for s in range(int(one["steps"])):
        move = np.abs(one[f"px{s + 1}"] - px0).max() + 2.0 * float(one["dt1"])
        for r in res:
            assert float(r[f"dt{s + 1}"]) == pytest.approx(float(one[f"dt{s + 1}"]), rel=1e-9)
            np.testing.assert_allclose(r[f"px{s + 1}"], one[f"px{s + 1}"], rtol=0, atol=2e-4 * move * (s + 1))
            np.testing.assert_allclose(r[f"pv{s + 1}"], one[f"pv{s + 1}"], rtol=0, atol=5e-3 * np.abs(one[f"pv{s + 1}"]).max())
            assert int(r[f"counts{s + 1}"].sum()) == px0.shape[0]
