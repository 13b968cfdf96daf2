"""GPU: whole time steps of the notebook's loop (BASELINE config 5's pipeline on ONE MI355X): a buckling-like
scene scaled to an N^3 grid -- flipped container box, four slanted obstacle plates (ipynb code cell 9), a fluid
block of (N/2)^3 cells at 8 particles per cell -- stepped with notebook_sim.NotebookSimulation; per-stage
wall-clock (synchronised), CG iteration counts.   usage: python tools/bench_timestep.py [N] [steps] [mu]
Under `python -m torch.distributed.run --nproc-per-node R ...` (one rank per GPU, RCCL) the two hot-path solves run
slab-decomposed (notebook_sim.SlabNotebookSimulation); MFS_BENCH_SHARED_GPU=1 = rehearsal with all ranks on cuda:0 / gloo."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
import notebook_sim as NSIM
import solver.sdf3D as sdf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mu = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
shared = os.environ.get("MFS_BENCH_SHARED_GPU", "0") == "1"
dev = "cuda:0" if (world == 1 or shared) else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
torch.cuda.set_device(dev)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if shared:
        from mfs.dist import pg_timeout
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout())
    else:
        from mfs.dist import pg_timeout
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev), timeout=pg_timeout())
gdx = 1.0 / N
size = np.array([1.0, 1.0, 1.0])
bmin = [-0.5, 0.0, -0.5]
rb_d, rb_map = sdf.generate_rb(None, {}, 'cube', ['box', 1 - 4 * gdx, 1 - 4 * gdx, 1 - 4 * gdx], flip=True, center=[0, 0.5, 0], device=dev)
h = 0.35
for nm, par, c, ax, ang in (("p1", ['box', 0.67, 0.05, 1.2], [-0.42, h, 0], [0, 0, 1], -45), ("p2", ['box', 0.67, 0.05, 1.2], [0.42, h, 0], [0, 0, 1], 45),
                            ("p3", ['box', 1.2, 0.05, 0.67], [0, h, -0.42], [1, 0, 0], 45), ("p4", ['box', 1.2, 0.05, 0.67], [0, h, 0.42], [1, 0, 0], -45)):
    rb_d, rb_map = sdf.generate_rb(rb_d, rb_map, nm, par, flip=False, center=c, axis=ax, angle=ang)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
px = NSIM.add_box([0.0, 0.7, 0.0], [0.5, 0.5, 0.5], gdx / 2, rng)
if dist is None:
    sim = NSIM.NotebookSimulation((N, N, N), gdx, bmin, rb_d, px, gdx / 2, mu=mu, device=dev, precision=os.environ.get("MFS_PRECISION"),
                                  jacobi=os.environ.get("MFS_TIMESTEP_JACOBI", "0") == "1")
else:
    # MFS_TIMESTEP_PARTICLES=replicated: the round-1 form (every rank holds all particles, whole-grid broadcasts)
    cls = NSIM.SlabNotebookSimulation if os.environ.get("MFS_TIMESTEP_PARTICLES") == "replicated" else NSIM.ShardedNotebookSimulation
    sim = cls((N, N, N), gdx, bmin, rb_d, px, gdx / 2, mu=mu, device=dev, precision=os.environ.get("MFS_PRECISION"), dist=dist,
              jacobi=os.environ.get("MFS_TIMESTEP_JACOBI", "0") == "1")     # (honoured by the window slab loops)
sim.particle.v[:, 0] = -2.0
torch.cuda.synchronize()
t_setup = time.perf_counter() - t0
sim.step()                                  # warm-up step (allocations, first launches)
tim, its = {}, []
t0 = time.perf_counter()
for _ in range(steps):
    sim.step(timings=tim)
    its.append((sim.DensitySolver.iterations, sim.ViscositySolver.iterations, sim.PressureSolver.iterations))
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
if dist is not None:
    tt = torch.tensor([t_all], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_all = tt.item()
if rank == 0:
  print(json.dumps({"workload": f"notebook time step {N}^3, {getattr(sim, 'total_particles', sim.particle.num_particles)} particles, mu={mu}",
                  "ranks": world, "decomposition": "single GPU" if dist is None else
                  f"density, viscosity and pressure CG loops on x-slabs x{world} (transports {sim.DensitySolver.transport}/{sim.ViscositySolver.transport}/{sim.PressureSolver.transport}), particle stages "
                  + ("sharded by x-slab: this rank holds %d of %d particles, %.1f MB of plane bands moved" % (sim.particle.num_particles, sim.total_particles, sim.bands.bytes_moved / 1e6)
                     if hasattr(sim, "bands") else "replicated")
                  + (" [REHEARSAL: ranks share one GPU]" if shared else ""),
                  "state_precision": os.environ.get("MFS_PRECISION", "fp64"), "steps": steps,
                  "s_per_step": round(t_all / steps, 4), "setup_s": round(t_setup, 2),
                  "stage_ms_per_step": {k: round(v / steps * 1e3, 2) for k, v in tim.items()},
                  "cg_iterations(density,viscosity,pressure)": its}))
if dist is not None:
    sim.close()
    dist.destroy_process_group()
