"""The notebook's particle <-> grid transfers on the MI355X against goldens produced by
executing the notebook's own cells (tests/golden/make_goldens_particles.py, pt_*), with the notebook's container
dtypes.  Tolerances: gathers are order-exact (1e-12); scatters add with fp atomics in arbitrary order -- fp32
grid arrays to a few fp32 ulps of the array maximum, fp64 arrays to 1e-11; the level set (an atomic min) is exact
up to sqrt vs pow (1e-13)."""
import types

import numpy as np
import pytest
import torch

from conftest import golden, golden_names
import notebook_kernels as K

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
N = lambda t: t.detach().cpu().numpy()  # noqa: E731
NS = types.SimpleNamespace


def containers(g):
    gres = tuple(int(v) for v in g["gres"])
    bmin = np.asarray(g["bound_min"], np.float32)
    bsz = np.asarray(g["bound_size"], np.float32)
    eye = np.eye(3, dtype=int)

    def comp(a, bias):
        shape = tuple(np.array(gres) + eye[a])
        return NS(bias=np.asarray(bias, np.float32), m=torch.zeros(shape, dtype=torch.float32, device=DEV),
                  v=torch.zeros(shape, dtype=torch.float32, device=DEV))
    grid = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64),
              x=comp(0, [0, .5, .5]), y=comp(1, [.5, 0, .5]), z=comp(2, [.5, .5, 0]))
    p = NS(num_particles=len(g["px"]), x=T(g["px"]), m=T(g["pm"]), v=T(g["pv"]), cx=T(g["pcx"]), cy=T(g["pcy"]),
           cz=T(g["pcz"]), vol=float(g["pvol"]))
    return gres, bmin, bsz, grid, p


@pytest.mark.parametrize("name", golden_names("pt_"))
def test_p2g_then_g2p(name):
    g = golden(name)
    gres, bmin, bsz, grid, p = containers(g)
    K.p2g(p, grid)
    for c in "xyz":
        gm, gv = N(getattr(grid, c).m), N(getattr(grid, c).v)
        np.testing.assert_allclose(gm, g[f"g{c}_m"], rtol=0, atol=4e-6 * np.abs(g[f"g{c}_m"]).max())
        np.testing.assert_allclose(gv, g[f"g{c}_v"], rtol=5e-4, atol=5e-5 * np.abs(g[f"g{c}_v"]).max())
        assert ((gm > 0) == (g[f"g{c}_m"] > 0)).all()
    # gather from the GOLDEN grid (so that the comparison is order-exact)
    for c in "xyz":
        getattr(grid, c).v.copy_(T(g[f"g{c}_v"]))
    K.g2p(p, grid)
    np.testing.assert_allclose(N(p.v), g["g2p_v"], rtol=1e-12, atol=1e-13)
    for c in "xyz":
        np.testing.assert_allclose(N(getattr(p, "c" + c)), g["g2p_c" + c], rtol=1e-12,
                                   atol=1e-12 * np.abs(g["g2p_c" + c]).max())


@pytest.mark.parametrize("name", golden_names("pt_"))
def test_fluid_levelset_and_volume(name):
    g = golden(name)
    gres, bmin, bsz, grid, p = containers(g)
    ls = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64),
            phi=torch.zeros(gres, dtype=torch.float64, device=DEV))
    K.compute_fluid_levelset(p, ls, float(g["gdx"]))
    np.testing.assert_allclose(N(ls.phi), g["lphi"], rtol=1e-13, atol=1e-15)
    vres = tuple(2 * np.array(gres) + 1)
    fv = NS(resolution=vres, bound_min=bmin, bound_size=bsz, cell_size=bsz / (2 * np.asarray(gres, np.int64)),
            vol=torch.full(vres, 3.0, dtype=torch.float64, device=DEV))
    K.compute_fluid_volume(p, fv, p.vol)
    np.testing.assert_allclose(N(fv.vol), g["lvol"], rtol=0, atol=1e-11 * np.abs(g["lvol"]).max())
    # the level set and volume are what the solvers take: lphi < 0 cells exist and every volume is within its cell
    assert (N(ls.phi) < 0).any() and N(fv.vol).max() <= float(np.prod(fv.cell_size)) * (1 + 1e-15)


def test_transfers_feed_the_solvers():
    """p2g -> ... -> DensityCGSolver3D / PressureCGSolver3D accept what the transfers produce (smoke of the hand-over)."""
    from solver.CGSolverBuffer import CGSolverBuffer
    from solver.DensityCGSolver3D import DensityCGSolver3D
    g, d = golden("pt_a_12"), golden("d3d_a_12")
    gres, bmin, bsz, grid, p = containers(g)
    ls = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64),
            phi=torch.zeros(gres, dtype=torch.float64, device=DEV))
    K.compute_fluid_levelset(p, ls, float(g["gdx"]))
    buf = CGSolverBuffer(gres, precision="fp64", device=DEV)
    ds = DensityCGSolver3D(buf, gres, bmin, bsz)
    sphi = T(np.full(tuple(2 * np.array(gres) + 1), 1.0))       # no solid anywhere near
    sv = torch.zeros(tuple(2 * np.array(gres) + 1) + (3,), dtype=torch.float64, device=DEV)
    lvol = torch.zeros(tuple(2 * np.array(gres) + 1), dtype=torch.float64, device=DEV)
    x0 = p.x.clone()
    ds.solve(1000.0, float(d["dt"]), p.x, p.m, p.vol, None, None, None, sphi, sv, ls.phi, lvol)
    assert ds.iterations > 0 and torch.isfinite(p.x).all() and not torch.equal(p.x, x0)


def test_empty_particle_sets_and_misuse():
    """zero particles are a no-op everywhere (the reference launches zero blocks); wrong shapes fail loudly"""
    gres = (6, 7, 8)
    bmin, bsz = np.zeros(3, np.float32), np.asarray(gres, np.float32) * 0.1
    eye = np.eye(3, dtype=int)
    comp = lambda a, b: NS(bias=np.asarray(b, np.float32), m=torch.zeros(tuple(np.array(gres) + eye[a]), dtype=torch.float32, device=DEV),  # noqa: E731
                           v=torch.ones(tuple(np.array(gres) + eye[a]), dtype=torch.float32, device=DEV))
    grid = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64), x=comp(0, [0, .5, .5]),
              y=comp(1, [.5, 0, .5]), z=comp(2, [.5, .5, 0]))
    e3 = lambda: torch.zeros((0, 3), dtype=torch.float64, device=DEV)  # noqa: E731
    p = NS(num_particles=0, x=e3(), m=torch.zeros(0, dtype=torch.float64, device=DEV), v=e3(), cx=e3(), cy=e3(), cz=e3(), vol=1e-3)
    K.p2g(p, grid)
    K.g2p(p, grid)
    assert float(grid.x.m.abs().max()) == 0.0 and float(grid.x.v.min()) == 1.0      # untouched
    ls = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=grid.cell_size, phi=torch.zeros(gres, dtype=torch.float64, device=DEV))
    K.compute_fluid_levelset(p, ls, 0.1)
    assert torch.allclose(ls.phi, torch.full_like(ls.phi, 0.3))                      # the pre-fill gdx * 3 everywhere
    vres = tuple(2 * np.array(gres) + 1)
    fv = NS(resolution=vres, bound_min=bmin, bound_size=bsz, cell_size=bsz / (2 * np.asarray(gres, np.int64)),
            vol=torch.ones(vres, dtype=torch.float64, device=DEV))
    K.compute_fluid_volume(p, fv, p.vol)
    assert float(fv.vol.abs().max()) == 0.0
    with pytest.raises(ValueError, match="shape"):
        K.p2g(NS(num_particles=2, x=torch.zeros((2, 2), dtype=torch.float64, device=DEV), m=p.m, v=p.v, cx=p.cx, cy=p.cy, cz=p.cz), grid)
    with pytest.raises(TypeError, match="GPU"):
        K.compute_fluid_levelset(NS(x=torch.zeros((1, 3), dtype=torch.float64)), ls, 0.1)


# ------------------------------------------------------------------ tile-sorted scatters (round 3) ------------------
def _grid_like(gres, bmin, bsz):
    eye = np.eye(3, dtype=int)

    def comp(a, bias):
        shape = tuple(np.array(gres) + eye[a])
        return NS(bias=np.asarray(bias, np.float32), m=torch.zeros(shape, dtype=torch.float32, device=DEV),
                  v=torch.zeros(shape, dtype=torch.float32, device=DEV))
    return NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64),
              x=comp(0, [0, .5, .5]), y=comp(1, [.5, 0, .5]), z=comp(2, [.5, .5, 0]))


@pytest.mark.parametrize("name", golden_names("pt_"))
def test_tiled_scatters_against_the_goldens(name, monkeypatch):
    """the tile-sorted forms (what runs at >= 262 144 particles) forced onto the golden scenes: same tolerances as the
    atomic forms above"""
    monkeypatch.setattr(K, "TILE_MIN_PARTICLES", 1)
    g = golden(name)
    gres, bmin, bsz, grid, p = containers(g)
    K.p2g(p, grid)
    assert K.tile_order(p, gres, grid.bound_min, grid.cell_size) is not None
    for c in "xyz":
        gm, gv = N(getattr(grid, c).m), N(getattr(grid, c).v)
        np.testing.assert_allclose(gm, g[f"g{c}_m"], rtol=0, atol=4e-6 * np.abs(g[f"g{c}_m"]).max())
        np.testing.assert_allclose(gv, g[f"g{c}_v"], rtol=5e-4, atol=5e-5 * np.abs(g[f"g{c}_v"]).max())
        assert ((gm > 0) == (g[f"g{c}_m"] > 0)).all()
    ls = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64),
            phi=torch.zeros(gres, dtype=torch.float64, device=DEV))
    K.compute_fluid_levelset(p, ls, float(g["gdx"]))
    np.testing.assert_allclose(N(ls.phi), g["lphi"], rtol=1e-13, atol=1e-15)
    vres = tuple(2 * np.array(gres) + 1)
    fv = NS(resolution=vres, bound_min=bmin, bound_size=bsz, cell_size=bsz / (2 * np.asarray(gres, np.int64)),
            vol=torch.full(vres, 3.0, dtype=torch.float64, device=DEV))
    K.compute_fluid_volume(p, fv, p.vol)
    np.testing.assert_allclose(N(fv.vol), g["lvol"], rtol=0, atol=1e-11 * np.abs(g["lvol"]).max())


def _synthetic(n=96, ppc=8, seed=0):
    """a block of (n/2)^3 cells at `ppc` jittered particles per cell in an n^3 grid, velocities and affine rows random"""
    rng = np.random.default_rng(seed)
    gres = (n, n, n)
    bmin, bsz = np.asarray([-0.5, 0.0, -0.5], np.float32), np.ones(3, np.float32)
    dx = 1.0 / n
    k = int(round(ppc ** (1 / 3)))
    ax = (np.arange(n // 2 * k) + 0.5) * (dx / k)
    X = np.stack(np.meshgrid(ax - 0.25, ax + 0.45, ax - 0.25, indexing="ij"), axis=-1).reshape(-1, 3)
    X = X + rng.standard_normal(X.shape) * dx * 0.15
    P = X.shape[0]
    p = NS(num_particles=P, x=T(X), m=T(np.full(P, 1000.0 * (dx / k) ** 3)), v=T(rng.standard_normal((P, 3))),
           cx=T(rng.standard_normal((P, 3))), cy=T(rng.standard_normal((P, 3))), cz=T(rng.standard_normal((P, 3))), vol=(dx / k) ** 3)
    return gres, bmin, bsz, p, dx


def _run_all(p, gres, bmin, bsz, dx):
    grid = _grid_like(gres, bmin, bsz)
    K.p2g_scatter(p, grid)
    ls = NS(resolution=gres, bound_min=bmin, bound_size=bsz, cell_size=bsz / np.asarray(gres, np.int64),
            phi=torch.zeros(gres, dtype=torch.float64, device=DEV))
    K.compute_fluid_levelset(p, ls, dx)
    vres = tuple(2 * np.array(gres) + 1)
    fv = NS(resolution=vres, bound_min=bmin, bound_size=bsz, cell_size=bsz / (2 * np.asarray(gres, np.int64)),
            vol=torch.zeros(vres, dtype=torch.float64, device=DEV))
    K.compute_fluid_volume(p, fv, p.vol)
    return dict(xm=grid.x.m, xv=grid.x.v, ym=grid.y.m, yv=grid.y.v, zm=grid.z.m, zv=grid.z.v, phi=ls.phi, vol=fv.vol)


def _close_fields(a, b):
    for k in a:
        x, y = a[k].double(), b[k].double()
        if k == "phi":
            assert torch.equal(x, y), k                       # a minimum does not depend on the order of its candidates
        else:
            tol = 1e-11 if a[k].dtype == torch.float64 else 2e-5
            assert float((x - y).abs().max()) <= tol * float(y.abs().max()), (k, float((x - y).abs().max()), float(y.abs().max()))


def test_tile_sort_is_a_permutation_by_tile():
    gres, bmin, bsz, p, dx = _synthetic(64)
    cs = bsz / np.asarray(gres, np.int64)
    import notebook_kernels as KK
    old = KK.TILE_MIN_PARTICLES
    KK.TILE_MIN_PARTICLES = 1
    try:
        perm, tstart = K.tile_order(p, gres, bmin, cs)
    finally:
        KK.TILE_MIN_PARTICLES = old
    P = p.num_particles
    perm_h, ts = perm.cpu().numpy(), tstart.cpu().numpy()
    assert sorted(perm_h.tolist()) == list(range(P))
    assert ts[0] == 0 and ts[-1] == P and (np.diff(ts) >= 0).all()
    x = p.x.cpu().numpy().astype(np.float32)
    cell = np.floor((x - bmin).astype(np.float64) / cs).astype(np.int64).clip(0, np.array(gres) - 1)
    nt = [(g_ + 7) // 8 for g_ in gres]
    tile = ((cell[:, 0] // 8) * nt[1] + cell[:, 1] // 8) * nt[2] + cell[:, 2] // 8
    seg = np.searchsorted(ts, np.arange(P), side="right") - 1          # tile of every slot of perm
    assert (tile[perm_h] == seg).all()


def test_tiled_scatters_match_the_atomic_forms_at_scale(monkeypatch):
    """884 736 particles on a 96^3 grid (the default path above 262 144 particles) against the per-particle atomics: masses,
    momenta and volumes to rounding of the sums, the level set bit for bit"""
    gres, bmin, bsz, p, dx = _synthetic(96)
    assert p.num_particles >= K.TILE_MIN_PARTICLES
    tiled = _run_all(p, gres, bmin, bsz, dx)
    assert K._TILE_ORDERS and K._TILE_ORDERS[0][0][0] == p.x.data_ptr()
    monkeypatch.setattr(K, "TILE_MIN_PARTICLES", 1 << 40)
    atomic = _run_all(p, gres, bmin, bsz, dx)
    _close_fields(tiled, atomic)


def test_a_stale_tile_order_is_slower_not_wrong(monkeypatch):
    """particles that moved up to three cells since the sort: contributions beyond the staged nodes take the global atomics"""
    gres, bmin, bsz, p, dx = _synthetic(64)
    monkeypatch.setattr(K, "TILE_MIN_PARTICLES", 1)
    cs = bsz / np.asarray(gres, np.int64)
    K.tile_order(p, gres, bmin, cs)
    stale = K._TILE_ORDERS[0]
    g_ = torch.Generator(device=DEV).manual_seed(2)
    p.x += (torch.rand(p.x.shape, generator=g_, device=DEV, dtype=torch.float64) - 0.5) * 6 * dx
    K._TILE_ORDERS[0] = ((p.x.data_ptr(), p.x._version) + stale[0][2:],) + stale[1:]      # pretend it is current
    got = _run_all(p, gres, bmin, bsz, dx)
    assert K._TILE_ORDERS[0][1] is stale[1]                   # the stale order was used
    monkeypatch.setattr(K, "TILE_MIN_PARTICLES", 1 << 40)
    want = _run_all(p, gres, bmin, bsz, dx)
    _close_fields(got, want)
