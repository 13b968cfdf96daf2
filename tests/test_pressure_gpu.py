"""Parity of the HIP pressure path (through the C ABI) against the oracle, the
golden vectors, and size-independent properties at BASELINE sizes.  GPU only.

Tolerances (stated, with the evidence for each):
  per-kernel outputs, fp64      1e-12 rel: same fp64 operations in the same order;
                                only FMA contraction differs.
  CG residual history           The ghost-fluid operator (theta clamp 0.01, quarter
                                face weights) is ill conditioned and CG on it is
                                CHAOTIC in rounding: the oracle ITSELF, with nothing
                                changed but the summation order of its dot products,
                                departs from its own history by >1e-2 after ~30
                                iterations (tests/test_oracle_sensitivity.py shows
                                this on the CPU).  Histories are therefore compared
                                entry by entry over a leading window --
                                  fp64 state: first 10 iterations at 1e-9 rel,
                                  fp32 state: first  8 iterations at 1e-5 rel
                                (north_star's bar) -- and over the WHOLE history for
                                the well-conditioned all-fluid case (fp64 1e-11,
                                fp32 1e-5).  Measured on MI355X (tools/diag_history.py,
                                profiles/r01_history_deviation.txt): fp64 <=3e-12 and
                                fp32 <=7e-7 inside those windows.
  solution x, output velocities 1e-4 rel to the field maximum for the pool scenes
                                (the oracle's own sensitivity is ~1e-5 at tol=1e-3),
                                1e-12 (fp64) / 2e-6 (fp32) for the all-fluid case.
  iteration count               exact where the history is stable (all-fluid, capped
                                runs); +-10 % (fp64) / -20..+50 % (fp32 storage needs
                                a few more iterations to reach the ABSOLUTE tol).
"""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names
from mfs import _lib, scenes
from oracle import mfs_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
W64, RT64 = 10, 1e-9      # fp64 state: leading window (iterations) and rtol
W32, RT32 = 8, 1e-5       # fp32 state


def hist_window(h, hg, iters, rtol):
    n = min(2 * iters + 1, len(h), len(hg))
    np.testing.assert_allclose(np.asarray(h)[:n], np.asarray(hg)[:n], rtol=rtol)


def stable(name):
    return "allfluid" in name


def T(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=DEV)
    return t if dtype is None else t.to(dtype)


def close(a, b, rtol, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    np.testing.assert_allclose(a.astype(np.float64), b, rtol=rtol, atol=rtol * scale, err_msg=what)


@pytest.fixture(scope="module")
def mods():
    from mfs import _lib
    _lib.load()
    import solver.CGSolverBuffer as B
    import solver.PressureCGSolver3D as P
    import solver.SolidFraction3D as S
    return B, P, S


@pytest.mark.parametrize("name", golden_names("p3d_"))
@pytest.mark.parametrize("wdt", [torch.float64, torch.float32])
def test_solid_frac_rhs_apply_vs_golden(mods, name, wdt):
    B, P, S = mods
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    Nx, Ny, Nz = gres
    wx = torch.full((Nx + 1, Ny, Nz), -3.0, dtype=wdt, device=DEV)
    wy = torch.full((Nx, Ny + 1, Nz), -3.0, dtype=wdt, device=DEV)
    wz = torch.full((Nx, Ny, Nz + 1), -3.0, dtype=wdt, device=DEV)
    S.compute_solid_frac(gres, T(g["sphi"]), wx, wy, wz)
    # quarters are exact in fp32 too -> bit-exact in both dtypes
    assert np.array_equal(wx[:Nx].cpu().numpy(), g["wx"][:Nx])
    assert np.array_equal(wy[:, :Ny].cpu().numpy(), g["wy"][:, :Ny])
    assert np.array_equal(wz[:, :, :Nz].cpu().numpy(), g["wz"][:, :, :Nz])
    assert (wx[Nx] == -3).all() and (wy[:, Ny] == -3).all() and (wz[:, :, Nz] == -3).all()  # untouched (Q8)

    gwx, gwy, gwz = T(g["wx"]), T(g["wy"]), T(g["wz"])
    b = torch.full(gres, 5.0, dtype=torch.float64, device=DEV)
    P.initialize_solver(g["bound_size"] / g["gres"], gres, T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"]),
                        T(g["sphi"]), T(g["sv"]), T(g["lphi"]), b, gwx, gwy, gwz)
    bi = b.cpu().numpy()
    close(bi[1:-1, 1:-1, 1:-1], g["b"][1:-1, 1:-1, 1:-1], 1e-12, "rhs")
    assert (bi[0] == 5).all() and (bi[:, 0] == 5).all() and (bi[:, :, -1] == 5).all()   # boundary untouched

    out = torch.full(gres, 7.0, dtype=torch.float64, device=DEV)
    P.matvecmul(gres, T(g["rv"]), out, gwx, gwy, gwz, T(g["lphi"]))
    close(out, g["qr"], 1e-12, "matvecmul")          # includes the untouched 7.0 boundary


@pytest.mark.parametrize("name", golden_names("p3d_"))
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_engine_apply_matches_reference_operator(mods, name, dt):
    """the per-iteration kernel (precomputed coefficients) == the operator from lphi,w."""
    from mfs.pcg import PcgEngine
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
    v = T(g["rv"], dt)
    out = torch.full(gres, 7.0, dtype=dt, device=DEV)
    eng.apply(v, out)
    ref = np.full(gres, 7.0)
    O.pressure_apply3d(gres, v.cpu().numpy().astype(np.float64), ref, g["wx"], g["wy"], g["wz"], g["lphi"])
    close(out, ref, 1e-12 if dt == torch.float64 else 2e-7, "engine apply")
    # compressed coefficient access (default) == dense access, bit for bit
    eng.set_compress(False)
    outd = torch.full(gres, 7.0, dtype=dt, device=DEV)
    eng.apply(v, outd)
    assert torch.equal(out, outd)
    eng.set_compress(True)
    # plane-range form: two halves == whole
    out2 = torch.full(gres, 7.0, dtype=dt, device=DEV)
    mid = gres[0] // 2
    eng.apply(v, out2, 1, mid)
    eng.apply(v, out2, mid, gres[0] - 1)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("name", golden_names("p3d_"))
def test_solve_fp64_vs_golden(mods, name):
    B, P, S = mods
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    buf = B.CGSolverBuffer(gres, precision="fp64", device=DEV)
    s = P.PressureCGSolver3D(buf, gres, g["bound_size"], check_every=7)
    vx, vy, vz = T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])
    s.solve(vx, vy, vz, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), tol=float(g["tol"]))
    hist_window(s.history, g["history"], W64, RT64)
    if stable(name):
        assert s.iterations == int(g["iters"])
        close(s.history, g["history"], 1e-11, "history")
        ftol, vtol = 1e-12, 1e-12
        assert abs(s.delta - float(g["delta"])) <= 1e-9 * float(g["delta"])
        assert abs(s.alpha - float(g["alpha"])) <= 1e-8 * abs(float(g["alpha"]))
        assert abs(s.beta - float(g["beta"])) <= 1e-8 * abs(float(g["beta"]))
    else:
        assert abs(s.iterations - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
        ftol = vtol = 1e-4
    assert s.delta < float(g["tol"]) ** 2 and s.delta == s.history[-1]
    close(s.x, g["x"], ftol, "x")
    for a, k in ((vx, "out_vx"), (vy, "out_vy"), (vz, "out_vz")):
        assert a.dtype == torch.as_tensor(g[k]).dtype
        close(a, g[k], max(vtol, 2e-7 if a.dtype == torch.float32 else 0), k)
    # the reference's stopping rule, re-evaluated with the ORACLE operator on the GPU's solution
    q = np.zeros(gres)
    O.pressure_apply3d(gres, s.x.cpu().numpy(), q, g["wx"], g["wy"], g["wz"], g["lphi"])
    true_delta = float(((g["b"] - q) ** 2).sum())
    assert true_delta < 1.01 * float(g["tol"]) ** 2, true_delta
    # caller-supplied weights (the notebook passes DensitySolver.wx, ipynb:4648): same result
    buf2 = B.CGSolverBuffer(gres, precision="fp64", device=DEV)
    s2 = P.PressureCGSolver3D(buf2, gres, g["bound_size"])
    v2 = [T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])]
    s2.solve(*v2, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), wx=T(g["wx"]), wy=T(g["wy"]), wz=T(g["wz"]),
             tol=float(g["tol"]))
    assert s2.iterations == s.iterations and torch.equal(v2[0], vx) and torch.equal(s2.x, s.x)


@pytest.mark.parametrize("name", golden_names("p3d_"))
def test_solve_fp32_state_vs_golden(mods, name):
    B, P, S = mods
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    buf = B.CGSolverBuffer(gres, precision="fp32", device=DEV)
    s = P.PressureCGSolver3D(buf, gres, g["bound_size"])
    vx, vy, vz = T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])
    s.solve(vx, vy, vz, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), tol=float(g["tol"]))
    hist_window(s.history, g["history"], W32, RT32)
    it = int(g["iters"])
    if stable(name):
        assert s.iterations == it
        close(s.history, g["history"], RT32, "history")
        ftol = 2e-6
    else:
        assert 0.8 * it - 2 <= s.iterations <= 1.5 * it + 2
        ftol = 1e-4
    assert s.delta < float(g["tol"]) ** 2
    close(s.x, g["x"], ftol, "x fp32")
    for a, k in ((vx, "out_vx"), (vy, "out_vy"), (vz, "out_vz")):
        close(a, g[k], ftol, k)


def test_seeded_scene_vs_oracle_nonmultiple_sizes(mods):
    """ragged sizes (odd Nz -> scalar kernels; fp32 velocities; solid velocity)"""
    B, P, S = mods
    for gres, seed in (((9, 11, 13), 21), ((6, 5, 7), 22), ((32, 8, 12), 23)):
        sc = scenes.pressure_scene_3d(gres, seed=seed, solid_velocity=True)
        ref = O.PressureCGSolver3D(gres, sc["bound_size"])
        rv = [sc["vx"].copy(), sc["vy"].copy(), sc["vz"].copy()]
        ref.solve(*rv, sc["sphi"], sc["sv"], sc["lphi"])
        buf = B.CGSolverBuffer(gres, device=DEV)
        s = P.PressureCGSolver3D(buf, gres, sc["bound_size"])
        v = [T(sc["vx"]), T(sc["vy"]), T(sc["vz"])]
        s.solve(*v, T(sc["sphi"]), T(sc["sv"]), T(sc["lphi"]))
        assert abs(s.iterations - ref.iterations) <= max(2, ref.iterations // 10)
        hist_window(s.history, ref.history, W64, RT64)
        close(s.x, ref.x, 1e-4)
        for a, b in zip(v, rv):
            close(a, b, 1e-4)


def test_failed_to_converge_raises(mods):
    B, P, S = mods
    g = golden("p3d_a_12")
    gres = tuple(int(v) for v in g["gres"])
    buf = B.CGSolverBuffer(gres, device=DEV)
    s = P.PressureCGSolver3D(buf, gres, g["bound_size"])
    s.max_iter = 5
    with pytest.raises(ValueError, match="Failed to converge!"):
        s.solve(T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"]), T(g["sphi"]), T(g["sv"]), T(g["lphi"]))
    assert s.iterations == 5
    np.testing.assert_allclose(s.history, g["history"][:11], rtol=1e-9)


def test_already_converged_rhs_does_no_iterations(mods):
    B, P, S = mods
    gres = (8, 8, 8)
    sc = scenes.pressure_scene_3d(gres, seed=1)
    buf = B.CGSolverBuffer(gres, device=DEV)
    s = P.PressureCGSolver3D(buf, gres, sc["bound_size"])
    z = lambda k: torch.zeros_like(T(sc[k]))  # noqa: E731
    vx, vy, vz = z("vx"), z("vy"), z("vz")
    s.solve(vx, vy, vz, T(sc["sphi"]), T(sc["sv"]), T(sc["lphi"]))
    assert s.iterations == 0 and s.delta == 0.0
    assert not vx.any() and not vy.any() and not vz.any()


@pytest.mark.parametrize("dt,N", [(torch.float32, 256), (torch.float64, 128)])
def test_full_size_properties(mods, dt, N):
    """BASELINE sizes: properties that need no oracle run -- symmetry <Au,v>=<u,Av>,
    positivity, A.0=0, boundary untouched, plane-range decomposition, determinism,
    and agreement of the precomputed-coefficient kernel with the direct operator."""
    B, P, S = mods
    from mfs.pcg import PcgEngine
    gres = (N, N, N)
    sc = scenes.pressure_scene_3d(gres, seed=4, device=DEV)
    wx = torch.zeros((N + 1, N, N), dtype=torch.float64, device=DEV)
    wy = torch.zeros((N, N + 1, N), dtype=torch.float64, device=DEV)
    wz = torch.zeros((N, N, N + 1), dtype=torch.float64, device=DEV)
    S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
    assert set(torch.unique(wx).tolist()) <= {0.0, 0.25, 0.5, 0.75, 1.0}
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(sc["lphi"], wx, wy, wz)
    gen = torch.Generator(device=DEV).manual_seed(0)
    u = torch.randn(gres, generator=gen, device=DEV, dtype=dt)
    v = torch.randn(gres, generator=gen, device=DEV, dtype=dt)
    for t in (u, v):   # boundary cells carry no DOF in the CG (d is 0 there)
        t[0] = 0; t[-1] = 0; t[:, 0] = 0; t[:, -1] = 0; t[:, :, 0] = 0; t[:, :, -1] = 0
    Au = torch.full(gres, 7.0, dtype=dt, device=DEV)
    Av = torch.full(gres, 7.0, dtype=dt, device=DEV)
    eng.apply(u, Au)
    eng.apply(v, Av)
    for t in (Au, Av):
        assert (t[0] == 7).all() and (t[-1] == 7).all() and (t[:, 0] == 7).all() and (t[:, :, -1] == 7).all()
        t[0] = 0; t[-1] = 0; t[:, 0] = 0; t[:, -1] = 0; t[:, :, 0] = 0; t[:, :, -1] = 0
    uAv = (u.double() * Av.double()).sum().item()
    vAu = (v.double() * Au.double()).sum().item()
    uAu = (u.double() * Au.double()).sum().item()
    tol = 1e-11 if dt == torch.float64 else 2e-6
    assert abs(uAv - vAu) <= tol * max(abs(uAv), abs(uAu)), (uAv, vAu)
    assert uAu > 0
    # direct operator from lphi / w agrees
    if dt == torch.float64:
        ref = torch.zeros(gres, dtype=dt, device=DEV)
        P.matvecmul(gres, u, ref, wx, wy, wz, sc["lphi"])
        assert torch.allclose(ref, Au, rtol=1e-12, atol=1e-12 * Au.abs().max().item())
    # A.0 = 0, determinism
    z = torch.zeros(gres, dtype=dt, device=DEV)
    Az = torch.full(gres, 7.0, dtype=dt, device=DEV)
    eng.apply(z, Az)
    assert (Az[1:-1, 1:-1, 1:-1] == 0).all()
    Au2 = torch.full(gres, 7.0, dtype=dt, device=DEV)
    eng.set_compress(False)          # dense coefficient access: identical bits
    eng.apply(u, Au2)
    eng.set_compress(True)
    Au2[0] = 0; Au2[-1] = 0; Au2[:, 0] = 0; Au2[:, -1] = 0; Au2[:, :, 0] = 0; Au2[:, :, -1] = 0
    assert torch.equal(Au, Au2)


def test_full_size_solve_runs_and_reduces_divergence(mods):
    """128^3 fp32 state, end to end through the drop-in class: converges, is
    reproducible bit for bit, and the CG energy decreases (residual history sane)."""
    B, P, S = mods
    gres = (128, 128, 128)
    sc = scenes.pressure_scene_3d(gres, seed=9, device=DEV)
    outs = []
    for _ in range(2):
        buf = B.CGSolverBuffer(gres, precision="fp32", device=DEV)
        s = P.PressureCGSolver3D(buf, gres, sc["bound_size"])
        v = [sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()]
        s.solve(*v, sc["sphi"], sc["sv"], sc["lphi"], tol=1e-2)
        outs.append((s.iterations, s.history, v))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))
    h = outs[0][1]
    assert h[-1] < 1e-4 and h[-1] < 1e-6 * h[0]
    assert (h[1::2] > 0).all()          # d.Ad > 0 : operator positive on the Krylov directions


@pytest.mark.parametrize("name", ["p3d_d_20", "p3d_a_12", "p3d_e_allfluid_12"])
@pytest.mark.parametrize("prec", ["fp64", "fp32"])
def test_fused_direction_update_is_bit_identical(mods, name, prec):
    """native loop with d = r + beta d folded into the stencil launch (default) vs the 3-kernel form"""
    B, P, S = mods
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    res = []
    for fuse in (True, False):
        buf = B.CGSolverBuffer(gres, precision=prec, device=DEV)
        s = P.PressureCGSolver3D(buf, gres, g["bound_size"], check_every=5)
        s._engine.set_fuse(fuse)
        s._engine.set_resident(False)     # both launch-per-phase forms (the resident loop: tests/test_resident_gpu.py)
        v = [T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])]
        s.solve(*v, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), tol=float(g["tol"]))
        res.append((s.iterations, s.history, s.x.clone(), buf.d.clone(), buf.r.clone(), buf.q.clone(), v))
    a, b = res
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    for i in (2, 3, 4, 5):
        assert torch.equal(a[i], b[i]), i          # x, d (brought home from the partner buffer), r, q
    assert all(torch.equal(p_, q_) for p_, q_ in zip(a[6], b[6]))


# ------------------------------------------------------------------ sparse work list + live chunks (round 3) ----------
def _blob_problem(gres, centre, radius, seed):
    """a ball of liquid in an otherwise empty box (most tiles of the march are air): lphi, unit weights with a few
    quarter weights inside, a random right-hand side in the liquid"""
    dev = DEV
    g = torch.Generator(device=dev).manual_seed(seed)
    ax = [torch.arange(n, device=dev, dtype=torch.float64) + 0.5 for n in gres]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")
    lphi = torch.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2) - radius
    def w(shape):
        q = torch.randint(1, 5, shape, generator=g, device=dev).double() * 0.25
        return torch.where(torch.rand(shape, generator=g, device=dev) < 0.9, torch.ones(shape, dtype=torch.float64, device=dev), q)
    wx, wy, wz = w((gres[0] + 1, gres[1], gres[2])), w((gres[0], gres[1] + 1, gres[2])), w((gres[0], gres[1], gres[2] + 1))
    b = torch.randn(gres, generator=g, device=dev, dtype=torch.float64) * (lphi < 0)
    b[0] = 0; b[-1] = 0; b[:, 0] = 0; b[:, -1] = 0; b[:, :, 0] = 0; b[:, :, -1] = 0
    return lphi, wx, wy, wz, b


@pytest.mark.parametrize("dt", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_sparse_work_list_and_live_chunks_match_the_dense_loop(dt, monkeypatch):
    """round 3: behind the initial residual a single-domain solve lists the (tile, plane) pairs of the march that compute
    anything and the 32-cell chunks that hold a live unknown; the fused stencil launches and the r update visit only
    those.  A ball of liquid in a 160 x 96 x 144 box (2.2 M cells: the lists are built from 2^21 cells on) -- and then, THROUGH
    THE SAME ENGINE, a ball elsewhere (the partner buffer of the direction vector still holds the first solve's liquid) --
    against the engine with the lists off: same history to rounding (dot products group differently), same solution."""
    from mfs.pcg import PcgEngine
    gres = (160, 96, 144)
    outs = {}
    for sparse in ("1", "0"):
        monkeypatch.setenv("MFS_SPARSE", sparse)
        eng = PcgEngine(gres, dt, DEV)
        res = []
        for centre, radius, seed in (((50.0, 40.0, 60.0), 22.0, 1), ((110.0, 60.0, 50.0), 18.0, 2)):
            lphi, wx, wy, wz, b = _blob_problem(gres, centre, radius, seed)
            eng.setup(lphi.to(dt), wx.to(dt), wy.to(dt), wz.to(dt))
            bt = b.to(dt)
            x, d, r, q = (torch.zeros(gres, dtype=dt, device=DEV) for _ in range(4))
            q[1:-1, 1:-1, 1:-1] = 5.0                  # stale q: begin must overwrite every computed cell
            eng.bind(bt, x, d, r, q)
            ok, it = eng.solve(1e-6 if dt == torch.float64 else 1e-3, 4000, 16)
            torch.cuda.synchronize()
            assert ok
            # a ball in a box: most lanes of the listed (tile, plane) pairs are dead, so the solve switches its listed launches to
            # the lane-masking form at its first look at the scalar block (and the engine's next solve starts with it)
            info = eng.sparse_info()
            assert (info["listed_pairs"] > 0 and info["live_chunks"] > 0) == (sparse == "1"), info
            assert float(eng.scalars[_lib.S_LANE]) == (1.0 if sparse == "1" else 0.0)
            res.append((it, np.asarray(eng.history()), x.clone(), q.clone()))
        outs[sparse] = res
    for (it_s, h_s, x_s, q_s), (it_d, h_d, x_d, q_d) in zip(outs["1"], outs["0"]):
        assert abs(it_s - it_d) <= 1, (it_s, it_d)
        n = min(len(h_s), len(h_d), 41)
        np.testing.assert_allclose(h_s[:n], h_d[:n], rtol=1e-10 if dt == torch.float64 else 1e-4)
        tol = 1e-8 if dt == torch.float64 else 1e-3
        assert float((x_s - x_d).abs().max()) <= tol * float(x_d.abs().max())
        air = x_d == 0
        assert torch.equal(x_s[air], x_d[air])         # nothing leaks into the air


@pytest.mark.parametrize("dt", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_sparse_lists_in_the_fused_jacobi_loop(dt):
    """the opt-in Jacobi loop's fused single-domain form takes the solve's lists as well (z = r / diag is 0 wherever r is: a dead
    vector stays dead; the stored z and the partner d buffer are cleared when the lists are built).  Ball-in-a-box problem at
    2.2 M cells, two solves through one engine (the ball moves), against the same loop with the lists off: same history to
    rounding, same solution, nothing leaks into the air."""
    from mfs.pcg import PcgEngine
    gres = (160, 96, 144)
    outs = {}
    for sparse in (True, False):
        eng = PcgEngine(gres, dt, DEV)
        eng.set_jacobi(True)
        eng.set_sparse(sparse)
        res = []
        for centre, radius, seed in (((50.0, 40.0, 60.0), 22.0, 1), ((110.0, 60.0, 70.0), 18.0, 2)):
            lphi, wx, wy, wz, b = _blob_problem(gres, centre, radius, seed)
            eng.setup(lphi.to(dt), wx.to(dt), wy.to(dt), wz.to(dt))
            x, d, r, q = (torch.zeros(gres, dtype=dt, device=DEV) for _ in range(4))
            q[1:-1, 1:-1, 1:-1] = 5.0
            eng.bind(b.to(dt), x, d, r, q)
            ok, it = eng.solve(1e-6 if dt == torch.float64 else 1e-3, 4000, 16)
            torch.cuda.synchronize()
            assert ok and eng.loop_info()["jacobi"] and eng.loop_info()["fused_direction_update"]
            info = eng.sparse_info()
            assert (info["listed_pairs"] > 0 and info["live_chunks"] > 0) == sparse, info
            res.append((it, np.asarray(eng.history()), x.clone()))
        outs[sparse] = res
    for (it_s, h_s, x_s), (it_d, h_d, x_d) in zip(outs[True], outs[False]):
        assert abs(it_s - it_d) <= 1, (it_s, it_d)
        n = min(len(h_s), len(h_d), 41)
        np.testing.assert_allclose(h_s[:n], h_d[:n], rtol=1e-10 if dt == torch.float64 else 1e-4)
        assert float((x_s - x_d).abs().max()) <= (1e-8 if dt == torch.float64 else 1e-3) * float(x_d.abs().max())
        air = x_d == 0
        assert torch.equal(x_s[air], x_d[air])


def test_sparse_lists_are_off_for_small_grids_and_on_from_two_million_cells():
    from mfs.pcg import PcgEngine
    for gres, on in (((24, 20, 16), False), ((160, 96, 144), True)):
        lphi, wx, wy, wz, b = _blob_problem(gres, (12.0, 10.0, 8.0), 6.0, 3)
        eng = PcgEngine(gres, torch.float64, DEV)
        eng.setup(lphi, wx, wy, wz)
        x, d, r, q = (torch.zeros(gres, dtype=torch.float64, device=DEV) for _ in range(4))
        eng.bind(b, x, d, r, q)
        ok, it = eng.solve(1e-8, 2000, 8)
        assert ok and it > 0
        info = eng.sparse_info()
        if not on:
            assert info == dict(live_chunks=0, chunks=0, listed_pairs=0, pairs=0), info
            continue
        assert 0 < info["live_chunks"] < info["chunks"] // 20 and 0 < info["listed_pairs"] < info["pairs"] // 20, info
        eng.set_sparse(False)
        eng.begin(1e-8)
        assert eng.sparse_info()["pairs"] == 0
