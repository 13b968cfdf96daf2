"""Slab-decomposed CG driver over `gloo` (world_size 2 and 3) on the CPU: partition,
halo exchange, scalar all-reduces and the overlap ordering, against the
single-domain oracle solve of the same problem.  The local compute is a numpy
stand-in injected into mfs.dist.SlabCG (tests/dist_worker.py); on the GPU the same
driver runs the HIP engine (tests/test_dist_gpu.py covers world_size 1 on RCCL)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from mfs import scenes
from mfs.dist import SlabPartition
from oracle import mfs_oracle as O

import dist_worker


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _global_problem(gres, seed, all_fluid):
    sc = scenes.pressure_scene_3d(gres, seed=seed, vel_dtype=np.float64, all_fluid=all_fluid, solid_velocity=True)
    Nx, Ny, Nz = gres
    wx, wy, wz = np.zeros((Nx + 1, Ny, Nz)), np.zeros((Nx, Ny + 1, Nz)), np.zeros((Nx, Ny, Nz + 1))
    O.compute_solid_frac3d(gres, sc["sphi"], wx, wy, wz)
    b = np.zeros(gres)
    O.pressure_rhs3d(sc["cell_size"], gres, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
    x, d, r, q = (np.zeros(gres) for _ in range(4))
    hist = []
    ap = lambda V, Q: O.pressure_apply3d(gres, V[0], Q[0], wx, wy, wz, sc["lphi"])  # noqa: E731
    it, *_ = O.cg(ap, b, x, d, r, q, 1e-3, Nx * Ny * Nz, hist)
    gl = dict(gres=np.array(gres), cell_size=np.array(sc["cell_size"]), lphi=sc["lphi"], wx=wx, wy=wy, wz=wz,
              vx=sc["vx"], vy=sc["vy"], vz=sc["vz"], sphi=sc["sphi"], sv=sc["sv"])
    return gl, dict(b=b, x=x, hist=np.array(hist), iters=it)


def test_partition_covers_computed_planes_once():
    for nx in (5, 18, 64, 257, 512):
        for world in (1, 2, 3, 4, 8):
            if nx - 2 < world:
                with pytest.raises(ValueError):
                    SlabPartition(nx, world, 0)
                continue
            parts = [SlabPartition(nx, world, r) for r in range(world)]
            owned = [p.owned for p in parts]
            assert owned[0][0] == 1 and owned[-1][1] == nx - 1
            assert all(a[1] == b[0] for a, b in zip(owned, owned[1:]))          # contiguous, disjoint
            assert all(p.local_range == (p.owned[0] - 1, p.owned[1] + 1) for p in parts)
            assert parts[0].left is None and parts[-1].right is None
            sizes = [b - a for a, b in owned]
            assert max(sizes) - min(sizes) <= 1                                  # balanced


@pytest.mark.parametrize("world,overlap", [(2, True), (2, False), (3, True)])
@pytest.mark.parametrize("all_fluid", [True, False])
def test_slab_cg_matches_single_domain(world, overlap, all_fluid):
    gres = (14, 8, 10)
    gl, ref = _global_problem(gres, seed=13, all_fluid=all_fluid)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "problem.npz")
        np.savez(path, **gl)
        mp.spawn(dist_worker.run, args=(world, _free_port(), path, 1e-3, overlap, 4000), nprocs=world, join=True)
        outs = [dict(np.load(f"{path}.rank{r}.npz")) for r in range(world)]
    for o in outs:
        lo, hi = int(o["lo"]), int(o["hi"])
        assert int(o["done"]) == 1
        # local RHS == global RHS on owned planes, exactly 0 on the ghost/boundary planes
        np.testing.assert_array_equal(o["b"][1:-1], ref["b"][lo + 1:hi - 1])
        assert not o["b"][0].any() and not o["b"][-1].any()
        # ghost planes of q and r never become non-zero (no double counting in the dots)
        assert not o["q"][0].any() and not o["q"][-1].any() and not o["r"][0].any() and not o["r"][-1].any()
    # every rank sees the same history (all-reduced scalars)
    for o in outs[1:]:
        np.testing.assert_array_equal(o["hist"], outs[0]["hist"])
    h, hr = outs[0]["hist"], ref["hist"]
    n = min(21, len(h), len(hr))
    np.testing.assert_allclose(h[:n], hr[:n], rtol=1e-10)            # leading window (see test_oracle_sensitivity)
    if all_fluid:                                                    # stable case: the whole history
        assert int(outs[0]["iters"]) == ref["iters"]
        np.testing.assert_allclose(h, hr, rtol=1e-9)
        xtol = 1e-11
    else:
        assert abs(int(outs[0]["iters"]) - ref["iters"]) <= max(2, ref["iters"] // 10)
        xtol = 1e-4
    scale = np.abs(ref["x"]).max()
    for o in outs:
        lo, hi = int(o["lo"]), int(o["hi"])
        # owned planes AND the ghost planes (x ghosts accumulate the neighbour's update)
        np.testing.assert_allclose(o["x"], ref["x"][lo:hi], rtol=0, atol=xtol * scale)


@pytest.mark.parametrize("name,world", [("v3d_b_10x12x14", 2), ("v3d_a_12", 3)])
def test_slab_viscosity_cg_matches_reference_golden(name, world):
    """mfs.dist.SlabVCG (three staggered components, u-plane ownership, per-sweep ghost exchange of the extrapolation)
    over gloo with the oracle as the local compute, against the golden outputs of the reference's own solve."""
    import conftest
    g = conftest.golden(name)
    gpath = os.path.join(conftest.REPO, "tests", "golden", name + ".npz")
    gres = tuple(int(v) for v in g["gres"])
    Nx = gres[0]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "problem.npz")
        with open(gpath, "rb") as fi, open(path, "wb") as fo:
            fo.write(fi.read())
        mp.spawn(dist_worker.run_viscosity, args=(world, _free_port(), path, float(g["tol"]), 4000), nprocs=world,
                 join=True)
        outs = [dict(np.load(f"{path}.rank{r}.npz")) for r in range(world)]
    asm = {k: [np.array(g[n]) for n in names] for k, names in
           (("e", ("ex", "ey", "ez")), ("x", ("ex", "ey", "ez")), ("b", ("bx", "by", "bz")))}
    for arr in asm["b"]:
        arr[...] = 0.0
    for o in outs:
        lo, hi = int(o["lo"]), int(o["hi"])
        L = hi - lo
        assert int(o["done"]) == 1
        top_u = L if hi == Nx else L - 1
        if hi != Nx:
            assert not o["q_x"][L - 1].any() and not o["r_x"][L - 1].any()
        for c in "yz":
            assert not o[f"q_{c}"][0].any() and not o[f"q_{c}"][L - 1].any() and not o[f"r_{c}"][0].any()
        for k in ("e", "x", "b"):
            asm[k][0][lo + 1:lo + top_u] = o[f"{k}_x"][1:top_u]
            asm[k][1][lo + 1:hi - 1] = o[f"{k}_y"][1:L - 1]
            asm[k][2][lo + 1:hi - 1] = o[f"{k}_z"][1:L - 1]
    for o in outs[1:]:
        np.testing.assert_array_equal(o["hist"], outs[0]["hist"])
    for k, names in (("e", ("ex", "ey", "ez")), ("b", ("bx", "by", "bz")), ("x", ("x_x", "x_y", "x_z"))):
        scale = max(np.abs(g[n]).max() for n in names)
        for arr, n in zip(asm[k], names):
            np.testing.assert_allclose(arr, g[n], rtol=0, atol=(1e-10 if k == "x" else 1e-13) * scale, err_msg=n)
    assert int(outs[0]["iters"]) == int(g["iters"])
    np.testing.assert_allclose(outs[0]["hist"], g["history"], rtol=1e-9)


def test_collective_loop_reports_a_lost_peer():
    """the "rccl"-style loop of SlabCG (every configuration that cannot take the window loop lands on it): a peer
    that never joins the solve must surface as MfsTimeout -- status MFS_E_TIMEOUT like the window loop -- within
    MFS_COLLECTIVE_TIMEOUT_S, not as a rank blocked in all_reduce (round-1 record: gpurun_out/suite_nofuse.log)."""
    gres = (14, 8, 10)
    gl, _ = _global_problem(gres, seed=13, all_fluid=True)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "problem.npz")
        np.savez(path, **gl)
        mp.spawn(dist_worker.run_lost_peer, args=(2, _free_port(), path, 2.0), nprocs=2, join=True)
        secs, outcome = open(path + ".rank0.txt").read().strip().split("\n")
    assert outcome.startswith("MfsTimeout") and "status -4" in outcome and "timed out" in outcome, outcome
    assert float(secs) < 15.0, secs


@pytest.mark.parametrize("world", [2, 3, 4])
def test_slab_bands_reduce_ghosts_migrate(world):
    """mfs.dist.SlabBands over gloo: band reduce (sum / min), ghost fetch and particle migration for the three array kinds
    ("cell", "xface", "doubled"), thin slabs included (world 4 on 12 planes: bands span two neighbours) -- against the
    single-process scatter of all particles."""
    nx = 12
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "bands")
        mp.spawn(dist_worker.run_bands, args=(world, _free_port(), path, nx), nprocs=world, join=True)
        outs = [dict(np.load(f"{path}.rank{r}.npz")) for r in range(world)]
    cell, wgt = outs[0]["cell"], outs[0]["wgt"]
    for kind, n, scale in (("cell", nx, 1), ("xface", nx + 1, 1), ("doubled", 2 * nx + 1, 2)):
        for op in ("sum", "min"):
            full = np.zeros((n, 3, 2)) if op == "sum" else np.full((n, 3, 2), 9.0)
            for c, w in zip(cell, wgt):
                for dxx in range(-2 * scale, 2 * scale + 1):
                    pl = min(max(c * scale + dxx, 0), n - 1)
                    if op == "sum":
                        full[pl] += w
                    else:
                        full[pl] = np.minimum(full[pl], w)
            covered = np.zeros(n, bool)
            for o in outs:
                a, b = o[f"{kind}_own"]
                assert not covered[a:b].any()
                covered[a:b] = True
                np.testing.assert_allclose(o[f"{kind}_{op}_owned"], full[a:b], rtol=1e-12, atol=1e-12)
                w_ = 4 * scale
                lo, hi = max(0, a - w_), min(n, b + w_)
                np.testing.assert_allclose(o[f"{kind}_{op}_ghosted"][lo:hi], full[lo:hi], rtol=1e-12, atol=1e-12)
            assert covered.all()
    # migration: every particle ends on the rank that owns its new cell, with its fields, exactly once
    seen = np.concatenate([o["mig_ids"] for o in outs])
    assert sorted(seen.tolist()) == list(range(len(cell)))
    for r, o in enumerate(outs):
        assert (o["mig_expected_owner"][o["mig_ids"]] == r).all()
        np.testing.assert_array_equal(o["mig_w"], wgt[o["mig_ids"]])
        np.testing.assert_array_equal(o["mig_v"][:, 2], 3 * wgt[o["mig_ids"]])
        assert int(o["bytes_moved"]) > 0
