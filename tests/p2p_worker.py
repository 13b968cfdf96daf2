"""Worker of tests/test_p2p_gpu.py: ONE rank of a slab-decomposed pressure solve whose
ranks all sit on the same MI355X (the GPU box has one card), talking through HIP-IPC
windows (mfs/p2p.py).  torch.distributed (gloo) only bootstraps the windows.
usage: p2p_worker.py RANK WORLD PORT GOLDEN_NPZ OUT_PREFIX DTYPE"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "python-fluid-simulation_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from mfs.dist import SlabCG, SlabPartition  # noqa: E402
from mfs.p2p import P2PWindow  # noqa: E402
from mfs.pcg import PcgEngine  # noqa: E402


def solver_mode(rank, world, path, out, dtname, dev):
    """The drop-in class: SlabPressureCGSolver3D.solve on this rank's slab of the golden scene."""
    from solver.CGSolverBuffer import CGSolverBuffer
    from solver.PressureCGSolver3D import SlabPressureCGSolver3D
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    lg = SlabPressureCGSolver3D.local_gres(gres, world, rank)
    buf = CGSolverBuffer(lg, precision={"f64": "fp64", "f32": "fp32"}[dtname], device=dev)
    s = SlabPressureCGSolver3D(buf, gres, g["bound_size"], dist, transport=os.environ.get("P2P_TEST_TRANSPORT", "p2p"))
    lo, hi = s.part.local_range
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    vx, vy, vz = T(g["in_vx"][lo:hi + 1]), T(g["in_vy"][lo:hi]), T(g["in_vz"][lo:hi])
    s.solve(vx, vy, vz, T(g["sphi"][2 * lo:2 * hi + 1]), T(g["sv"][2 * lo:2 * hi + 1]), T(g["lphi"][lo:hi]),
            tol=float(g["tol"]))
    torch.cuda.synchronize()
    np.savez(f"{out}.rank{rank}.npz", vx=vx.cpu().numpy(), vy=vy.cpu().numpy(), vz=vz.cpu().numpy(),
             x=s.x.cpu().numpy().astype(np.float64), hist=s.history, iters=s.iterations, lo=lo, hi=hi,
             transport=s.transport, b=buf.b.cpu().numpy().astype(np.float64))
    s.close()


def viscosity_mode(rank, world, path, out, dtname, dev):
    """SlabViscosityCGSolver3D.solve on this rank's slab of a golden viscosity scene (collectives over gloo)."""
    from solver.ViscosityCGSolver3D import SlabViscosityCGSolver3D
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    s = SlabViscosityCGSolver3D(gres, g["bound_size"], dist, precision={"f64": "fp64", "f32": "fp32"}[dtname], device=dev,
                                transport=os.environ.get("P2P_TEST_TRANSPORT", "auto"))
    lo, hi = s.part.local_range
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    vx, vy, vz = T(g["in_vx"][lo:hi + 1]), T(g["in_vy"][lo:hi]), T(g["in_vz"][lo:hi])
    s.solve(float(g["dt"]), float(g["mu"]), float(g["rho"]), vx, vy, vz, T(g["sphi"][2 * lo:2 * hi + 1]),
            T(g["sv"][2 * lo:2 * hi + 1]), T(g["lphi"][lo:hi]), T(g["lvol"][2 * lo:2 * hi + 1]), tol=float(g["tol"]))
    torch.cuda.synchronize()
    c = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    np.savez(f"{out}.rank{rank}.npz", vx=vx.cpu().numpy(), vy=vy.cpu().numpy(), vz=vz.cpu().numpy(),
             x_x=c(s.x_x), x_y=c(s.x_y), x_z=c(s.x_z), b_x=c(s.b_x), b_y=c(s.b_y), b_z=c(s.b_z),
             q_x=c(s.q_x), r_x=c(s.r_x), q_y=c(s.q_y), r_y=c(s.r_y),
             hist=s.history, iters=s.iterations, lo=lo, hi=hi, transport=s.transport,
             sparse=np.array([v for v in s._engine.sparse_info().values()], dtype=np.int64))
    s.close()


def density_mode(rank, world, path, out, dtname, dev):
    """SlabDensityCGSolver3D.solve (replicated particles, CG loop slab-decomposed) on a golden density scene."""
    from solver.CGSolverBuffer import CGSolverBuffer
    from solver.DensityCGSolver3D import SlabDensityCGSolver3D
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    buf = CGSolverBuffer(gres, precision={"f64": "fp64", "f32": "fp32"}[dtname], device=dev)
    s = SlabDensityCGSolver3D(buf, gres, g["bound_min"], g["bound_size"], dist, transport=os.environ.get("P2P_TEST_TRANSPORT", "auto"))
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    px = T(g["px"])
    s.solve(float(g["rho0"]), float(g["dt"]), px, T(g["pm"]), float(g["pvol"]), None, None, None, T(g["sphi"]), T(g["sv"]),
            T(g["lphi"]), T(g["lvol"]), tol=float(g["tol"]))
    torch.cuda.synchronize()
    c = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    np.savez(f"{out}.rank{rank}.npz", px=c(px), x=c(s.x), dx=c(s.dx), dy=c(s.dy), dz=c(s.dz), hist=s.history,
             iters=s.iterations, delta=s.delta, lq=c(s._lq), lr=c(s._lr), transport=s.transport)
    s.close()


def timestep_mode(rank, world, path, out, dtname, dev):
    """SlabNotebookSimulation: whole time steps with the two hot-path solves slab-decomposed (BASELINE config 5)."""
    import notebook_sim as NSIM
    import solver.sdf3D as sdf
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    gdx = float(g["gdx"])
    size = np.array(gres) * gdx
    rb_d, rb_map = sdf.generate_rb(None, {}, 'cube', ['box', size[0] - 2 * gdx, size[1] - 2 * gdx, size[2] - 2 * gdx], flip=True,
                                   center=[0, size[1] / 2, 0], axis=[0., 1, 0], angle=0, device=dev)
    rb_d, rb_map = sdf.generate_rb(rb_d, rb_map, 'ramp', ['box', 0.45, 0.05, 0.8], flip=False, center=[-0.12, 0.2, 0],
                                   axis=[0., 0, 1], angle=-35)
    sim = NSIM.SlabNotebookSimulation(gres, gdx, [-0.3, 0, -0.3], rb_d, g["px0"], float(g["pdx"]), rho=float(g["rho"]),
                                      mu=float(g["mu"]), dt=float(g["dt"]), device=dev, dist=dist,
                                      transport=os.environ.get("P2P_TEST_TRANSPORT", "auto"))
    sim.particle.v.copy_(torch.as_tensor(g["pv0"], device=dev))
    res, timings = {}, {}
    for s in range(int(g["steps"])):
        res[f"dt{s + 1}"] = sim.step(timings=timings)
        res[f"px{s + 1}"] = sim.particle.x.cpu().numpy()
        res[f"pv{s + 1}"] = sim.particle.v.cpu().numpy()
        res[f"lphi{s + 1}"] = sim.fluid_levelset.phi.cpu().numpy()
        res[f"gvy{s + 1}"] = sim.grid.y.v.cpu().numpy()
    np.savez(f"{out}.rank{rank}.npz", transport=sim.PressureSolver.transport, p_iters=sim.PressureSolver.iterations,
             v_iters=sim.ViscositySolver.iterations, stages=np.array(sorted(timings)), **res)
    sim.close()


def timestep_sharded_mode(rank, world, path, out, dtname, dev):
    """ShardedNotebookSimulation: whole time steps with the particles sharded by x-slab as well (BASELINE config 5)."""
    import notebook_sim as NSIM
    import solver.sdf3D as sdf
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    gdx = float(g["gdx"])
    size = np.array(gres) * gdx
    rb_d, rb_map = sdf.generate_rb(None, {}, 'cube', ['box', size[0] - 2 * gdx, size[1] - 2 * gdx, size[2] - 2 * gdx], flip=True,
                                   center=[0, size[1] / 2, 0], axis=[0., 1, 0], angle=0, device=dev)
    rb_d, rb_map = sdf.generate_rb(rb_d, rb_map, 'ramp', ['box', 0.45, 0.05, 0.8], flip=False, center=[-0.12, 0.2, 0],
                                   axis=[0., 0, 1], angle=-35)
    sim = NSIM.ShardedNotebookSimulation(gres, gdx, [-0.3, 0, -0.3], rb_d, g["px0"], float(g["pdx"]), rho=float(g["rho"]),
                                         mu=float(g["mu"]), dt=float(g["dt"]), device=dev, dist=dist,
                                         transport=os.environ.get("P2P_TEST_TRANSPORT", "auto"),
                                         jacobi=os.environ.get("P2P_TEST_JACOBI", "0") == "1")
    sim.set_particle_velocities(g["pv0"])
    res, timings = {}, {}
    for s in range(int(g["steps"])):
        res[f"dt{s + 1}"] = sim.step(timings=timings)
        ids, x, v, counts = sim.gather_particles()
        assert ids.numpy().tolist() == list(range(sim.total_particles)), "a particle was lost or duplicated in migration"
        res[f"px{s + 1}"], res[f"pv{s + 1}"], res[f"counts{s + 1}"] = x.numpy(), v.numpy(), np.array(counts)
        res[f"lphi{s + 1}"] = sim.fluid_levelset.phi.cpu().numpy()       # valid on this rank's planes (+ ghosts)
        res[f"gvy{s + 1}"] = sim.grid.y.v.cpu().numpy()
    a, b = sim.bands.owned("cell")
    np.savez(f"{out}.rank{rank}.npz", transport=sim.PressureSolver.transport, p_iters=sim.PressureSolver.iterations,
             v_iters=sim.ViscositySolver.iterations, d_iters=sim.DensitySolver.iterations, stages=np.array(sorted(timings)),
             own_lo=a, own_hi=b, band_bytes=sim.bands.bytes_moved, **res)
    sim.close()


def timestep_synth_mode(rank, world, out, dev):
    """a buckling-like synthetic scene (tools/bench_timestep.py's, at 32^3 with ~33 k particles moving at -2 in x, so
    that particles cross the slab cuts every step): ShardedNotebookSimulation on `world` ranks; world == 1 with
    P2P_TEST_SINGLE=1 runs the plain single-GPU NotebookSimulation as the reference of the comparison"""
    import notebook_sim as NSIM
    import solver.sdf3D as sdf
    N, steps = 32, 4
    gdx = 1.0 / N
    bmin = [-0.5, 0.0, -0.5]
    rb_d, rb_map = sdf.generate_rb(None, {}, 'cube', ['box', 1 - 4 * gdx, 1 - 4 * gdx, 1 - 4 * gdx], flip=True, center=[0, 0.5, 0], device=dev)
    h = 0.35
    for nm, par, c, ax, ang in (("p1", ['box', 0.67, 0.05, 1.2], [-0.42, h, 0], [0, 0, 1], -45), ("p2", ['box', 0.67, 0.05, 1.2], [0.42, h, 0], [0, 0, 1], 45)):
        rb_d, rb_map = sdf.generate_rb(rb_d, rb_map, nm, par, flip=False, center=c, axis=ax, angle=ang)
    px = NSIM.add_box([0.0, 0.7, 0.0], [0.5, 0.5, 0.5], gdx / 2, np.random.default_rng(0))
    single = os.environ.get("P2P_TEST_SINGLE") == "1"
    if single:
        sim = NSIM.NotebookSimulation((N, N, N), gdx, bmin, rb_d, px, gdx / 2, mu=1.0, device=dev)
        sim.particle.v[:, 0] = -2.0
    else:
        sim = NSIM.ShardedNotebookSimulation((N, N, N), gdx, bmin, rb_d, px, gdx / 2, mu=1.0, device=dev, dist=dist,
                                             transport=os.environ.get("P2P_TEST_TRANSPORT", "auto"))
        sim.particle.v[:, 0] = -2.0
    res = {}
    moved = 0
    for s in range(steps):
        before = None if single else sim.particle.id.clone()
        res[f"dt{s + 1}"] = sim.step()
        if single:
            res[f"px{s + 1}"], res[f"pv{s + 1}"] = sim.particle.x.cpu().numpy(), sim.particle.v.cpu().numpy()
        else:
            ids, x, v, counts = sim.gather_particles()
            assert ids.numpy().tolist() == list(range(sim.total_particles))
            res[f"px{s + 1}"], res[f"pv{s + 1}"], res[f"counts{s + 1}"] = x.numpy(), v.numpy(), np.array(counts)
            now = set(sim.particle.id.cpu().numpy().tolist())
            moved += len(now - set(before.cpu().numpy().tolist()))
    np.savez(f"{out}.rank{rank}.npz", steps=steps, arrivals=moved, **res)
    if not single:
        sim.close()


def rccl_world1_mode(rank, world, path, out, dtname, dev):
    """every collective form of the sharded time step on an RCCL group.  World 1 (all one GPU admits),
    so the exchanges that `world == 1` short-cuts are driven directly: the count all-gather, the scalar all-reduce, the
    table broadcast of gather_particles, and a batched isend / irecv of DEVICE planes to this very rank (SlabBands._run's
    non-staged branch).  Then two whole steps of ShardedNotebookSimulation, as tools/bench_timestep.py runs them."""
    from mfs.dist import SlabBands
    assert dist.get_backend() == "nccl" and world == 1
    B = SlabBands(dist, None, 12, device=dev)
    c = B.exchange_counts(torch.tensor([3, 4, 5], dtype=torch.int64, device=dev))
    assert c.shape == (1, 3) and c.tolist() == [[3, 4, 5]] and not c.is_cuda
    c = B.exchange_counts(torch.tensor([7], dtype=torch.int64))           # a host operand: moved to the device for the collective
    assert c.tolist() == [[7]]
    assert B.allreduce_scalar(2.5, "max") == 2.5 and B.allreduce_scalar(-1.0, "sum") == -1.0
    src = torch.arange(2 * 5 * 3, dtype=torch.float64, device=dev).reshape(2, 5, 3)
    dst = torch.zeros_like(src)
    B._run([(src, 0)], [(dst, 0)])                                       # device planes through RCCL send / recv
    torch.cuda.synchronize()
    assert torch.equal(src, dst) and B.bytes_moved == src.numel() * 8
    timestep_sharded_mode(rank, world, path, out, dtname, dev)


def rccl_native_mode(rank, world, path, out, dtname, dev, tdt):
    """the NATIVE collective slab loop (csrc/mfs_rccl.h) on a one-rank RCCL communicator (all one GPU admits) against the
    window loop on a one-rank window and the phase-by-phase collective loop: same launches, same partial sums -- the window
    loop's history bit for bit, the phase loop's to rounding; begin / iterate / finish and solve()"""
    from mfs.rccl import RcclComm
    assert dist.get_backend() == "nccl" and world == 1
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    gres = tuple(int(v) for v in g["gres"])
    part = SlabPartition(gres[0], 1, 0)
    T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), device=dev).to(dt)  # noqa: E731
    eng = PcgEngine(gres, tdt, dev)
    eng.setup(T(g["lphi"]), T(g["wx"]), T(g["wy"]), T(g["wz"]))
    bt = T(g["b"], tdt)
    x, d, r, q = (torch.zeros(gres, dtype=tdt, device=dev) for _ in range(4))
    eng.bind(bt, x, d, r, q)
    win = P2PWindow(dist, gres[1] * gres[2] * bt.element_size(), dev)
    assert win.ok, win.why
    comm = RcclComm(dist, dev)
    res = {}
    for name, cg in (("window", SlabCG(eng, part, d, dist, force_multi=True, window=win)),
                     ("native", SlabCG(eng, part, d, dist, force_multi=True, rccl=comm)),
                     ("phases", SlabCG(eng, part, d, dist, force_multi=True))):
        cg.begin(0.0)
        cg.iterate(3)
        cg.iterate(9)
        cg.finish()
        torch.cuda.synchronize()
        res[name + "_hist"], res[name + "_x"], res[name + "_mode"] = eng.history()[:25], x.cpu().numpy().astype(np.float64), cg.mode
    # ... and a whole solve through the native loop
    cg = SlabCG(eng, part, d, dist, force_multi=True, rccl=comm)
    ok, it = cg.solve(float(g["tol"]), int(np.prod(gres)), 8)
    torch.cuda.synchronize()
    res["solve_ok"], res["solve_iters"], res["solve_x"], res["solve_hist"] = int(ok), it, x.cpu().numpy().astype(np.float64), eng.history()
    np.savez(f"{out}.rank{rank}.npz", **res)
    comm.close()
    win.close()


def _close_after_fault(win):
    """teardown of the fault-injection modes: the group may be broken by the timed-out collective (gloo closes the
    pair), so the window's closing barrier is best-effort and the process leaves without a collective teardown."""
    try:
        win.close()
    except Exception as exc:  # noqa: BLE001
        print(f"window close after the injected fault: {type(exc).__name__}", flush=True)
    sys.stdout.flush()
    os._exit(0)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    path, out, dtname = sys.argv[4], sys.argv[5], sys.argv[6]
    tdt = {"f64": torch.float64, "f32": torch.float32}[dtname]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from mfs.dist import pg_timeout
    if os.environ.get("P2P_TEST_BACKEND") == "nccl":
        # a REAL RCCL group (one rank: RCCL refuses two ranks on one device) -- the launchers' backend on the GPUs
        # (bench.py, tools/bench_timestep.py); every small host-side collective must then run on device operands
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev,
                                timeout=pg_timeout())
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=pg_timeout())
    if os.environ.get("P2P_TEST_MODE") == "rccl_native":
        try:
            rccl_native_mode(rank, world, path, out, dtname, dev, tdt)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "rccl_world1":
        try:
            rccl_world1_mode(rank, world, path, out, dtname, dev)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "lost_peer_viscosity":
        # fault injection, viscosity window loop: rank 1 maps its window and never takes part in the solve
        from mfs import _lib
        from mfs.dist import SlabVCG
        from mfs.vcg import VcgEngine
        import time as _t
        try:
            lg = (8, 12, 8)
            eng = VcgEngine(lg, torch.float64, dev)
            win = P2PWindow(dist, eng.edge_plane_bytes(), dev)
            assert win.ok, win.why
            if rank == 0:
                dbl = tuple(2 * v + 1 for v in lg)
                one = lambda sh: torch.ones(sh, dtype=torch.float64, device=dev)  # noqa: E731
                eng.setup(1e-3, 1.0, one(dbl), one(dbl))
                vecs = {n: eng.new_vector() for n in "bxdrq"}
                vecs["b"][0].fill_(1.0)
                eng.bind(*[vecs[n][0] for n in "bxdrq"])
                cg = SlabVCG(eng, SlabPartition(14, 2, 0), vecs["d"][1], dist, window=win)
                t0 = _t.perf_counter()
                try:
                    cg.begin(1e-12)
                    cg.iterate(4)
                    eng.poll()
                    outcome = "no error"
                except _lib.MfsError as exc:
                    outcome = "MfsError: " + str(exc)
                with open(f"{out}.rank0.txt", "w") as f:
                    f.write(f"{_t.perf_counter() - t0:.3f}\n{outcome}\n")
                    f.write(f"mode={cg.mode}\n")
            _close_after_fault(win)
        finally:
            pass
        return
    if os.environ.get("P2P_TEST_MODE") == "lost_peer":
        # fault injection: rank 1 maps its window and then never takes part in the solve
        from mfs import _lib
        import time as _t
        try:
            lg = (8, 12, 8)
            win = P2PWindow(dist, lg[1] * lg[2] * 8, dev)
            assert win.ok, win.why
            if rank == 0:
                eng = PcgEngine(lg, torch.float64, dev)
                one = lambda *sh: torch.ones(sh, dtype=torch.float64, device=dev)  # noqa: E731
                eng.setup(-one(*lg), one(lg[0] + 1, lg[1], lg[2]), one(lg[0], lg[1] + 1, lg[2]), one(lg[0], lg[1], lg[2] + 1))
                b, x, d, r, q = one(*lg), *(torch.zeros(lg, dtype=torch.float64, device=dev) for _ in range(4))
                eng.bind(b, x, d, r, q)
                cg = SlabCG(eng, SlabPartition(14, 2, 0), d, dist, window=win)
                t0 = _t.perf_counter()
                try:
                    cg.begin(1e-9)            # (collective fallback, e.g. MFS_FUSE_D=0: the bounded all-reduce raises here)
                    cg.iterate(8)
                    eng.poll()
                    outcome = "no error"
                except _lib.MfsError as exc:
                    outcome = "MfsError: " + str(exc)
                with open(f"{out}.rank0.txt", "w") as f:
                    f.write(f"{_t.perf_counter() - t0:.3f}\n{outcome}\n")
                    f.write(f"mode={cg.mode}\n")
            _close_after_fault(win)
        finally:
            pass
        return
    if os.environ.get("P2P_TEST_MODE") == "density":
        try:
            density_mode(rank, world, path, out, dtname, dev)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "timestep_synth":
        try:
            timestep_synth_mode(rank, world, out, dev)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "timestep_sharded":
        try:
            timestep_sharded_mode(rank, world, path, out, dtname, dev)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "timestep":
        try:
            timestep_mode(rank, world, path, out, dtname, dev)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "viscosity":
        try:
            viscosity_mode(rank, world, path, out, dtname, dev)
        finally:
            dist.destroy_process_group()
        return
    if os.environ.get("P2P_TEST_MODE") == "solver":
        try:
            solver_mode(rank, world, path, out, dtname, dev)
        finally:
            dist.destroy_process_group()
        return
    try:
        with np.load(path, allow_pickle=False) as z:
            g = {k: z[k] for k in z.files}
        gres = tuple(int(v) for v in g["gres"])
        part = SlabPartition(gres[0], world, rank)
        lo, hi = part.local_range
        lg = (hi - lo, gres[1], gres[2])
        T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), device=dev).to(dt)  # noqa: E731
        b = np.array(g["b"][lo:hi], dtype=np.float64)
        b[0] = 0.0          # ghost / boundary planes carry no equation on this rank
        b[-1] = 0.0
        eng = PcgEngine(lg, tdt, dev)
        eng.setup(T(g["lphi"][lo:hi]), T(g["wx"][lo:hi + 1]), T(g["wy"][lo:hi]), T(g["wz"][lo:hi]))
        bt = T(b, tdt)
        x, d, r, q = (torch.zeros(lg, dtype=tdt, device=dev) for _ in range(4))
        eng.bind(bt, x, d, r, q)
        win = P2PWindow(dist, lg[1] * lg[2] * bt.element_size(), dev)
        assert win.ok, win.why
        cg = SlabCG(eng, part, d, dist, window=win)
        assert cg.mode == "p2p"
        reps = int(os.environ.get("P2P_TEST_SOLVES", "1"))
        for _ in range(reps):            # a second solve re-uses the window (epoch handling)
            cg.begin(float(g["tol"]))
            st = eng.poll()
            n = 0
            while not st["done"] and n < 4000:
                cg.iterate(8)
                n += 8
                st = eng.poll()
            cg.finish()                  # callers of begin / iterate settle the deferred x update themselves
        torch.cuda.synchronize()
        xc = x.cpu()                     # gloo moves host tensors; on the GPUs (bench.py) RCCL moves device planes
        cg.exchange(xc)
        x.copy_(xc)
        torch.cuda.synchronize()
        np.savez(f"{out}.rank{rank}.npz", x=x.cpu().numpy().astype(np.float64), hist=eng.history(),
                 iters=st["iterations"], done=int(st["done"]), lo=lo, hi=hi, q=q.cpu().numpy().astype(np.float64),
                 alloc=win.alloc_kind, sparse=np.array([v for v in eng.sparse_info().values()], dtype=np.int64))
        win.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
