"""The peer-to-peer slab loop (csrc/mfs_pcg_slab.h, mfs/p2p.py) on ONE MI355X: 1, 2 and 3
ranks -- separate processes that share the card -- exchange halo planes and dot products
through HIP-IPC windows, and the assembled solution is compared with the single-domain
native solve and with the golden residual history (which pins it to the reference).
Real multi-GPU runs differ only in the windows sitting on different cards."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, golden, require_default_engine
from mfs.pcg import PcgEngine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WORKER = os.path.join(REPO, "tests", "p2p_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _native(g, dt):
    gres = tuple(int(v) for v in g["gres"])
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
    b = T(g["b"]).to(dt)
    x, d, r, q = (torch.zeros(gres, dtype=dt, device=DEV) for _ in range(4))
    eng.bind(b, x, d, r, q)
    ok, it = eng.solve(float(g["tol"]), int(np.prod(gres)), 16)
    assert ok
    return it, eng.history(), x.cpu().numpy().astype(np.float64)


def _run_ranks(name, world, tmp_path, dtname, solves=1, **extra_env):
    port = _free_port()
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    out = str(tmp_path / f"{name}_w{world}")
    env = dict(os.environ, MFS_P2P_TIMEOUT_MS="4000", P2P_TEST_SOLVES=str(solves), **extra_env)
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), path, out, dtname], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    return [np.load(f"{out}.rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("name,world,dtname", [
    ("p3d_e_allfluid_12", 1, "f64"), ("p3d_e_allfluid_12", 2, "f64"), ("p3d_d_20", 2, "f64"), ("p3d_d_20", 3, "f64"),
    ("p3d_d_20", 2, "f32"),
    ("p3d_a_12", 5, "f64"),          # two owned planes per rank: both are edge planes, no interior launch
    ("p3d_d_20", 5, "f64"),          # uneven slabs (4,4,3,4,3 owned planes), three ranks with two neighbours
])
def test_slab_p2p_matches_single_domain(name, world, dtname, tmp_path):
    require_default_engine("test_slab_p2p_matches_single_domain")
    g = golden(name)
    dt = torch.float64 if dtname == "f64" else torch.float32
    it0, h0, x0 = _native(g, dt)
    res = _run_ranks(name, world, tmp_path, dtname, solves=2 if world == 2 and dtname == "f64" else 1)
    gres = tuple(int(v) for v in g["gres"])
    x = np.zeros(gres)
    for r in res:
        assert int(r["done"]) == 1 and str(r["alloc"]) in ("uncached", "fine-grained")
        lo, hi = int(r["lo"]), int(r["hi"])
        x[lo + 1:hi - 1] = r["x"][1:-1]
        if lo > 0:          # the ghost plane holds the neighbour's edge plane after cg.exchange(x)
            np.testing.assert_array_equal(r["x"][0], [rr for rr in res if int(rr["hi"]) - 1 == lo + 1][0]["x"][-2])
    hists = [r["hist"] for r in res]
    for h in hists[1:]:      # every rank took bit-identical scalars
        np.testing.assert_array_equal(h, hists[0])
    assert len({int(r["iters"]) for r in res}) == 1
    h = hists[0]
    if dtname == "f64":
        n = min(21, len(h), len(h0))
        np.testing.assert_allclose(h[:n], h0[:n], rtol=1e-10)           # vs the single-domain HIP solve
        np.testing.assert_allclose(h[:n], g["history"][:n], rtol=1e-9)  # vs the golden (reference) history
        if "allfluid" in name:
            assert int(res[0]["iters"]) == it0 == int(g["iters"])
            np.testing.assert_allclose(h, h0, rtol=1e-9)
            np.testing.assert_allclose(x, x0, rtol=0, atol=1e-11 * np.abs(x0).max())
        else:
            assert abs(int(res[0]["iters"]) - it0) <= max(2, it0 // 10)
            np.testing.assert_allclose(x, x0, rtol=0, atol=1e-4 * np.abs(x0).max())
    else:   # fp32 state: leading window at north_star's 1e-5, converged field at 1e-3 of its maximum
        n = min(17, len(h), len(h0))
        np.testing.assert_allclose(h[:n], h0[:n], rtol=1e-5)
        np.testing.assert_allclose(x, x0, rtol=0, atol=1e-3 * np.abs(x0).max())


@pytest.mark.parametrize("name,world", [("p3d_d_20", 2), ("p3d_d_20", 3), ("p3d_a_12", 5)])
def test_slab_p2p_deferred_x_update(name, world, tmp_path):
    """the slab loop with the solution update deferred into the next iteration's edge / interior launches (its default
    on production-size slabs, forced here): same history bit for bit, same solution as the undeferred loop."""
    require_default_engine("test_slab_p2p_deferred_x_update")
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    out = {}
    for defer in ("0", "1"):
        # the deferred form also with the edge-plane sends on the second stream (its default on production-size planes)
        res = _run_ranks(name, world, tmp_path, "f64", MFS_DEFER_X=defer, MFS_SLAB_AUX_STREAM=defer)
        x = np.zeros(gres)
        for r in res:
            lo, hi = int(r["lo"]), int(r["hi"])
            x[lo + 1:hi - 1] = r["x"][1:-1]
        out[defer] = (res[0]["hist"], x, int(res[0]["iters"]))
    np.testing.assert_array_equal(out["0"][0], out["1"][0])
    assert out["0"][2] == out["1"][2]
    np.testing.assert_array_equal(out["0"][1], out["1"][1])


@pytest.mark.parametrize("name,world,dtname", [("p3d_d_20", 2, "f64"), ("p3d_f_40x36x32_sv", 2, "f64"), ("p3d_d_20", 3, "f32"),
                                               ("p3d_a_12", 5, "f64"), ("p3d_f_40x36x32_sv", 1, "f64")])
def test_slab_p2p_sparse_lists(name, world, dtname, tmp_path):
    """round 3: the slab loops build the solve's sparse lists too (live chunks of the owned planes for the r update, the
    (tile, plane) pairs of the interior launch that hold a live vector) -- forced onto these small grids with
    MFS_SPARSE_MIN=1, two solves through the same engines (deferred x update and second stream on), against the same
    loops with the lists off: same history to rounding (the dot products group differently), same solution."""
    require_default_engine("test_slab_p2p_sparse_lists")
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    out = {}
    for sparse in ("1", "0"):
        res = _run_ranks(name, world, tmp_path, dtname, solves=2, MFS_SPARSE=sparse, MFS_SPARSE_MIN="1", MFS_DEFER_X="1",
                         MFS_SLAB_AUX_STREAM="1")
        x = np.zeros(gres)
        for r in res:
            assert int(r["done"]) == 1
            lo, hi = int(r["lo"]), int(r["hi"])
            x[lo + 1:hi - 1] = r["x"][1:-1]
            live_chunks, chunks, listed, pairs = (int(v) for v in r["sparse"])
            if sparse == "1":      # the lists WERE built: chunks of the owned planes always, pairs where there is an interior launch
                assert chunks > 0 and 0 < live_chunks <= chunks, r["sparse"]
                assert (pairs > 0 and 0 < listed <= pairs) == (hi - lo - 2 > 2), (r["sparse"], lo, hi)
            else:
                assert chunks == 0 and pairs == 0, r["sparse"]
        for r in res[1:]:
            np.testing.assert_array_equal(r["hist"], res[0]["hist"])
        out[sparse] = (res[0]["hist"], x, int(res[0]["iters"]))
    (h1, x1, it1), (h0, x0, it0) = out["1"], out["0"]
    assert abs(it1 - it0) <= max(2, it0 // 10)
    n = min(21 if dtname == "f64" else 17, len(h1), len(h0))
    np.testing.assert_allclose(h1[:n], h0[:n], rtol=1e-10 if dtname == "f64" else 1e-5)
    np.testing.assert_allclose(x1, x0, rtol=0, atol=(1e-4 if dtname == "f64" else 1e-3) * np.abs(x0).max())
    air = x0 == 0
    np.testing.assert_array_equal(x1[air], 0.0)


@pytest.mark.parametrize("name,world", [("p3d_d_20", 2), ("p3d_b_10x12x14_sv", 2), ("p3d_d_20", 3)])
def test_slab_solver_class_matches_reference_outputs(name, world, tmp_path):
    """SlabPressureCGSolver3D.solve (the reference's solve signature on a rank's slab): RHS, pressure and the
    in-place velocity update against the golden outputs of the reference's own solve on the whole grid."""
    require_default_engine("test_slab_solver_class_matches_reference_outputs")
    g = golden(name)
    gres = tuple(int(v) for v in g["gres"])
    res = _run_ranks(name, world, tmp_path, "f64", P2P_TEST_MODE="solver")
    x, b = np.zeros(gres), np.zeros(gres)
    vx, vy, vz = (np.array(g[k], dtype=np.float32) for k in ("in_vx", "in_vy", "in_vz"))
    for r in res:
        assert str(r["transport"]) == "p2p"
        lo, hi = int(r["lo"]), int(r["hi"])
        L = hi - lo
        x[lo + 1:hi - 1] = r["x"][1:-1]
        b[lo + 1:hi - 1] = r["b"][1:-1]
        vx[lo + 1:lo + L] = r["vx"][1:L]              # x-faces 1 .. L-1 of the slab
        vy[lo + 1:hi] = r["vy"][1:L]                  # y / z faces of local cell planes 1 .. L-1 (the reference updates
        vz[lo + 1:hi] = r["vz"][1:L]                  # cell plane N-1 too, :135; shared planes are computed by both ranks)
    np.testing.assert_allclose(b, g["b"], rtol=0, atol=1e-12 * np.abs(g["b"]).max())
    h = res[0]["hist"]
    n = min(21, len(h), len(g["history"]))
    np.testing.assert_allclose(h[:n], g["history"][:n], rtol=1e-9)
    assert abs(int(res[0]["iters"]) - int(g["iters"])) <= max(2, int(g["iters"]) // 10)
    np.testing.assert_allclose(x, g["x"], rtol=0, atol=1e-4 * np.abs(g["x"]).max())
    for a, k in ((vx, "out_vx"), (vy, "out_vy"), (vz, "out_vz")):
        np.testing.assert_allclose(a, g[k], rtol=0, atol=1e-4 * max(np.abs(g[k]).max(), 1e-30))


def test_slab_solver_class_rccl_style_loop_over_gloo(tmp_path):
    """the fallback transport of the same class (collectives per iteration; gloo here, RCCL on a node)."""
    g = golden("p3d_e_allfluid_12")
    # one rank: gloo moves no device planes (send/recv of GPU tensors is RCCL's job on a node); the all-reduces
    # on the device-resident scalars and the phase-by-phase loop are what this covers
    res = _run_ranks("p3d_e_allfluid_12", 1, tmp_path, "f64", P2P_TEST_MODE="solver", P2P_TEST_TRANSPORT="rccl")
    assert all(str(r["transport"]) == "rccl" for r in res)
    assert int(res[0]["iters"]) == int(g["iters"])
    np.testing.assert_allclose(res[0]["hist"], g["history"], rtol=1e-9)


@pytest.mark.parametrize("mode,extra", [("lost_peer", {}), ("lost_peer_viscosity", {}),
                                        ("lost_peer", {"MFS_FUSE_D": "0"})])
def test_lost_peer_is_reported_not_hung(mode, extra, tmp_path):
    """fault injection (pressure and viscosity window loops): the neighbour never joins the solve -> every wait gives up
    after MFS_P2P_TIMEOUT_MS, later kernels return at their top, and the next poll raises with status MFS_E_TIMEOUT (-4).
    Third case: with the fused direction update off the engine cannot take the window loop and SlabCG falls back to the
    collective loop (the configuration of round 1's stuck run, gpurun_out/suite_nofuse.log): its bounded waits must
    report the same status within MFS_COLLECTIVE_TIMEOUT_S."""
    require_default_engine("test_lost_peer_is_reported_not_hung")
    port = _free_port()
    out = str(tmp_path / "lost")
    env = dict(os.environ, MFS_P2P_TIMEOUT_MS="400", MFS_COLLECTIVE_TIMEOUT_S="3", P2P_TEST_MODE=mode, **extra)
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), "2", str(port), "-", out, "f64"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        logs = [p.communicate(timeout=120)[0] for p in procs]
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    secs, outcome, mode_line = open(out + ".rank0.txt").read().strip().split("\n")
    assert outcome.startswith("MfsError") and "status -4" in outcome and "timed out" in outcome, outcome
    assert float(secs) < 20.0
    assert mode_line == ("mode=rccl" if extra else "mode=p2p"), mode_line      # the downgrade is visible


def _native_jacobi(g, dt):
    gres = tuple(int(v) for v in g["gres"])
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
    eng.set_jacobi(True)
    b = T(g["b"]).to(dt)
    x, d, r, q = (torch.zeros(gres, dtype=dt, device=DEV) for _ in range(4))
    eng.bind(b, x, d, r, q)
    ok, it = eng.solve(float(g["tol"]), int(np.prod(gres)), 16)
    assert ok
    return it, eng.history(), x.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("name,world,dtname,defer", [
    ("p3d_d_20", 1, "f64", "0"), ("p3d_d_20", 2, "f64", "0"), ("p3d_d_20", 3, "f64", "1"), ("p3d_d_20", 2, "f32", "0"),
    ("p3d_a_12", 5, "f64", "1"),          # two owned planes per rank: no interior launch
])
def test_slab_p2p_jacobi(name, world, dtname, defer, tmp_path):
    """the opt-in Jacobi loop through the window slab loop (z = r / diag as the operand of the edge and interior direction
    updates, r.r AND r.z all-reduced in the tail of the r / z update) against the single-domain Jacobi solve"""
    require_default_engine("test_slab_p2p_jacobi")
    g = golden(name)
    dt = torch.float64 if dtname == "f64" else torch.float32
    it0, h0, x0 = _native_jacobi(g, dt)
    assert it0 < int(g["iters"])            # it IS the preconditioned iteration
    res = _run_ranks(name, world, tmp_path, dtname, solves=2 if world == 2 and dtname == "f64" else 1, MFS_JACOBI="1",
                     MFS_DEFER_X=defer, MFS_SLAB_AUX_STREAM=defer)
    gres = tuple(int(v) for v in g["gres"])
    x = np.zeros(gres)
    for r in res:
        assert int(r["done"]) == 1
        lo, hi = int(r["lo"]), int(r["hi"])
        x[lo + 1:hi - 1] = r["x"][1:-1]
    hists = [r["hist"] for r in res]
    for h in hists[1:]:      # every rank took bit-identical scalars
        np.testing.assert_array_equal(h, hists[0])
    assert len({int(r["iters"]) for r in res}) == 1
    h = hists[0]
    n = min(21, len(h), len(h0))
    np.testing.assert_allclose(h[:n], h0[:n], rtol=1e-10 if dtname == "f64" else 1e-5)
    assert abs(int(res[0]["iters"]) - it0) <= max(2, it0 // 10)
    np.testing.assert_allclose(x, x0, rtol=0, atol=(1e-4 if dtname == "f64" else 1e-3) * np.abs(x0).max())


@pytest.mark.parametrize("name,dtname", [("p3d_d_20", "f64"), ("p3d_d_20", "f32"), ("p3d_f_40x36x32_sv", "f64")])
def test_native_collective_loop_on_a_one_rank_communicator(name, dtname, tmp_path):
    """round 3: the collective transport as a NATIVE loop -- the window loop's four launches with
    ncclSend / ncclRecv of the edge planes and one ncclAllReduce per dot product between them, enqueued from C
    (csrc/mfs_rccl.h).  RCCL admits one rank per device, so: a one-rank communicator of the build's own (bootstrap over a
    real "nccl" torch.distributed group) -- every launch, every RCCL call except the peer transfers -- against the window
    loop on a one-rank window (same launches and partial sums: bit for bit) and the phase-by-phase collective loop
    (rounding).  solve() converges to the golden's iteration count and solution."""
    require_default_engine("test_native_collective_loop_on_a_one_rank_communicator")
    g = golden(name)
    r = _run_ranks(name, 1, tmp_path, dtname, P2P_TEST_MODE="rccl_native", P2P_TEST_BACKEND="nccl")[0]
    assert str(r["window_mode"]) == "p2p" and str(r["native_mode"]) == "rccl" and str(r["phases_mode"]) == "rccl"
    np.testing.assert_array_equal(r["native_hist"], r["window_hist"])
    np.testing.assert_array_equal(r["native_x"], r["window_x"])
    tol = 1e-9 if dtname == "f64" else 1e-4
    n = 21
    np.testing.assert_allclose(r["native_hist"][:n], r["phases_hist"][:n], rtol=tol)
    assert int(r["solve_ok"]) == 1
    it = int(g["iters"])
    assert abs(int(r["solve_iters"]) - it) <= max(2, it // 10) if dtname == "f64" else int(r["solve_iters"]) <= 1.5 * it + 2
    xg = g["x"]
    np.testing.assert_allclose(r["solve_x"], xg, rtol=0, atol=1e-4 * np.abs(xg).max())
    hg = g["history"]
    w = 17 if dtname == "f32" else 21
    np.testing.assert_allclose(r["solve_hist"][:w], hg[:w], rtol=1e-9 if dtname == "f64" else 1e-5)
