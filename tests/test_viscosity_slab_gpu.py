"""Slab-decomposed viscosity solve (solver.ViscosityCGSolver3D.SlabViscosityCGSolver3D, mfs/dist.py:SlabVCG) on ONE
MI355X: 1, 2 and 3 ranks -- separate processes sharing the card; halo planes and dot products through HIP-IPC
windows ("p2p") or collectives over gloo ("rccl"; RCCL on a node) -- each solving its x-slab of a golden scene.  The assembled RHS, solution and written-back velocities are compared with
the golden outputs of the reference's own `solve` on the whole grid, and the residual history with the golden one."""
import numpy as np
import pytest

from conftest import golden
from test_p2p_gpu import _run_ranks

pytestmark = pytest.mark.gpu


def _assemble(g, res):
    """owned faces of every rank -> global arrays (faces no rank owns keep the golden INPUT values / zeros)"""
    gres = tuple(int(v) for v in g["gres"])
    Nx, Ny, Nz = gres
    shp = [(Nx + 1, Ny, Nz), (Nx, Ny + 1, Nz), (Nx, Ny, Nz + 1)]
    out = {}
    for base, init in (("b", None), ("x", ("ex", "ey", "ez")), ("v", ("in_vx", "in_vy", "in_vz"))):
        out[base] = [np.zeros(s) if init is None else np.array(g[k], dtype=np.float64) for s, k in
                     zip(shp, init or (None,) * 3)]
    last = max(int(r["hi"]) for r in res)
    for r in res:
        lo, hi = int(r["lo"]), int(r["hi"])
        L = hi - lo
        top_u = L if hi == last else L - 1          # the last rank owns its u plane L-1 (global face Nx-1)
        for base, keys in (("b", ("b_x", "b_y", "b_z")), ("x", ("x_x", "x_y", "x_z")), ("v", ("vx", "vy", "vz"))):
            out[base][0][lo + 1:lo + top_u] = r[keys[0]][1:top_u]
            out[base][1][lo + 1:hi - 1] = r[keys[1]][1:L - 1]
            out[base][2][lo + 1:hi - 1] = r[keys[2]][1:L - 1]
    return out


@pytest.mark.parametrize("name,world,dtname,transport", [
    ("v3d_a_12", 1, "f64", "p2p"), ("v3d_a_12", 2, "f64", "p2p"), ("v3d_b_10x12x14", 2, "f64", "p2p"),
    ("v3d_b_10x12x14", 3, "f64", "p2p"), ("v3d_a_12", 3, "f32", "p2p"), ("v3d_c_16_mu50", 2, "f64", "p2p"),
    ("v3d_a_12", 2, "f64", "rccl"), ("v3d_b_10x12x14", 3, "f64", "rccl"), ("v3d_a_12", 3, "f32", "rccl"),
])
def test_slab_viscosity_matches_reference_outputs(name, world, dtname, transport, tmp_path):
    _check_against_goldens(name, world, dtname, transport, tmp_path)


@pytest.mark.parametrize("name,world,dtname", [("v3d_a_12", 1, "f64"), ("v3d_b_10x12x14", 2, "f64"), ("v3d_b_10x12x14", 3, "f64"),
                                               ("v3d_a_12", 3, "f32"), ("v3d_c_16_mu50", 2, "f64")])
def test_slab_viscosity_sparse_lists(name, world, dtname, tmp_path):
    """round 3: the window slab loop with the solve's sparse lists (live chunks for the vector phases, busy (tile, plane) pairs
    for the march launches), forced onto these small grids with MFS_VISC_SPARSE_MIN=1: the same checks against the goldens"""
    _check_against_goldens(name, world, dtname, "p2p", tmp_path, MFS_VISC_SPARSE_MIN="1", MFS_VISC_RESIDENT="0", MFS_RDX="0")


def _check_against_goldens(name, world, dtname, transport, tmp_path, **env):
    g = golden(name)
    res = _run_ranks(name, world, tmp_path, dtname, P2P_TEST_MODE="viscosity", P2P_TEST_TRANSPORT=transport, **env)
    for r in res:          # the solve's sparse lists: built exactly when forced onto these small grids
        live_chunks, chunks, listed, pairs = (int(v) for v in r["sparse"])
        assert (chunks > 0 and live_chunks > 0) == ("MFS_VISC_SPARSE_MIN" in env), (r["sparse"], env)
    assert all(str(r["transport"]) == transport for r in res)
    a = _assemble(g, res)
    for r in res:          # ghost planes of q and r stay exactly 0: the local dot products count owned faces only
        L = int(r["hi"]) - int(r["lo"])
        if int(r["hi"]) != max(int(q["hi"]) for q in res):
            assert not r["q_x"][L - 1].any() and not r["r_x"][L - 1].any() and not r["b_x"][L - 1].any()
        assert not r["q_x"][0].any() and not r["r_x"][0].any()
        for k in ("q_y", "r_y", "b_y"):
            assert not r[k][0].any() and not r[k][L - 1].any()
    hists = [r["hist"] for r in res]
    for h in hists[1:]:
        np.testing.assert_array_equal(h, hists[0])        # every rank took the same all-reduced scalars
    h, hg, it = hists[0], g["history"], int(g["iters"])
    stable = "mu50" not in name
    f64 = dtname == "f64"
    bscale = max(np.abs(g[k]).max() for k in ("bx", "by", "bz"))
    for arr, k in zip(a["b"], ("bx", "by", "bz")):
        np.testing.assert_allclose(arr, g[k], rtol=0, atol=(1e-12 if f64 else 3e-6) * bscale, err_msg=k)
    n = min(len(h), len(hg), 21 if f64 else 17)
    np.testing.assert_allclose(h[:n], hg[:n], rtol=1e-9 if f64 else 1e-5)
    if f64 and stable:
        assert int(res[0]["iters"]) == it
        np.testing.assert_allclose(h, hg, rtol=1e-9)
    else:
        assert 0.8 * it - 2 <= int(res[0]["iters"]) <= 1.5 * it + 2
    vscale = max(np.abs(g[k]).max() for k in ("x_x", "x_y", "x_z"))
    ftol = (1e-10 if f64 else 2e-5) if stable else 1e-3
    for arr, k in zip(a["x"], ("x_x", "x_y", "x_z")):
        np.testing.assert_allclose(arr, g[k], rtol=0, atol=ftol * vscale, err_msg=k)
    for arr, k in zip(a["v"], ("out_vx", "out_vy", "out_vz")):
        np.testing.assert_allclose(arr, g[k].astype(np.float64), rtol=0, atol=max(ftol, 1e-6) * vscale, err_msg=k)


@pytest.mark.parametrize("name,world,dtname", [("v3d_a_12", 1, "f64"), ("v3d_a_12", 2, "f64"), ("v3d_c_16_mu50", 2, "f64"),
                                               ("v3d_b_10x12x14", 3, "f64"), ("v3d_d_24", 3, "f32")])
def test_slab_viscosity_jacobi_matches_single_domain_jacobi(name, world, dtname, tmp_path):
    """the opt-in Jacobi iteration through the window slab loop (r.r AND r.z all-reduced through the windows, the generic
    Jacobi kernels on the all-reduced scalars) against the single-domain Jacobi solve of the same golden scene"""
    import torch
    import solver.ViscosityCGSolver3D as V
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")  # noqa: E731
    s = V.ViscosityCGSolver3D(gres, g["bound_size"], precision={"f64": "fp64", "f32": "fp32"}[dtname], device="cuda:0",
                              check_every=8, jacobi=True)
    vx, vy, vz = T(g["in_vx"]), T(g["in_vy"]), T(g["in_vz"])
    s.solve(float(g["dt"]), float(g["mu"]), float(g["rho"]), vx, vy, vz, T(g["sphi"]), T(g["sv"]), T(g["lphi"]), T(g["lvol"]),
            tol=float(g["tol"]))
    h0, it0 = s.history, s.iterations
    assert it0 < int(g["iters"])
    res = _run_ranks(name, world, tmp_path, dtname, P2P_TEST_MODE="viscosity", P2P_TEST_TRANSPORT="p2p", MFS_VISC_JACOBI="1")
    assert all(str(r["transport"]) == "p2p" for r in res)
    hists = [r["hist"] for r in res]
    for h in hists[1:]:
        np.testing.assert_array_equal(h, hists[0])
    h = hists[0]
    f64 = dtname == "f64"
    n = min(len(h), len(h0), 17)
    np.testing.assert_allclose(h[:n], h0[:n], rtol=1e-9 if f64 else 2e-5)
    assert abs(int(res[0]["iters"]) - it0) <= max(1, it0 // 10)
    a = _assemble(g, res)
    ref = [vx, vy, vz]
    vscale = max(float(t.double().abs().max()) for t in ref)
    for arr, t in zip(a["v"], ref):
        np.testing.assert_allclose(arr, t.double().cpu().numpy(), rtol=0, atol=(1e-6 if f64 else 1e-3) * vscale)
