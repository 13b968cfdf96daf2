"""GPU, under torch.distributed.run (MFS_BENCH_SHARED_GPU=1: all ranks on cuda:0 over gloo): reproducibility of the two
slab loops of the pressure CG on bench.py's weak-scaling problem -- each loop run twice from the same start, residual
histories compared bit for bit within a loop and between the loops.  usage: ... tools/slab_repro.py [dtype] [iters]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch, torch.distributed as dist
from mfs import scenes, dist as mdist
from mfs.pcg import PcgEngine
from mfs.p2p import P2PWindow
import solver.PressureCGSolver3D as P
import solver.SolidFraction3D as S
dts = sys.argv[1] if len(sys.argv) > 1 else "f32"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 6
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
shared = os.environ.get("MFS_BENCH_SHARED_GPU", "0") == "1"
dev = torch.device("cuda", 0 if shared else int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("gloo" if shared else "nccl", rank=rank, world_size=world)
tdt = torch.float32 if dts == "f32" else torch.float64
ggrid = {1: (256, 256, 256), 2: (512, 256, 256), 4: (512, 512, 256), 8: (512, 512, 512)}.get(world, (256 * world, 256, 256))
if os.environ.get("SLAB_REPRO_GRID"):
    ggrid = tuple(int(v) for v in os.environ["SLAB_REPRO_GRID"].split(","))
part = mdist.SlabPartition(ggrid[0], world, rank)
lo, hi = part.local_range
lg = (hi - lo, ggrid[1], ggrid[2])
sc = scenes.pressure_scene_3d(ggrid, seed=0, x_range=(lo, hi), device=dev)
wx, wy, wz = (torch.zeros(tuple(lg[a] + (a == c) for a in range(3)), dtype=tdt, device=dev) for c in range(3))
S.compute_solid_frac(lg, sc["sphi"], wx, wy, wz)
b, x, d, r, q = (torch.zeros(lg, dtype=tdt, device=dev) for _ in range(5))
P.initialize_solver(sc["cell_size"], lg, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
eng = PcgEngine(lg, tdt, dev)
eng.setup(sc["lphi"], wx, wy, wz)
eng.bind(b, x, d, r, q)
cg_r = mdist.SlabCG(eng, part, d, dist)
win = P2PWindow(dist, lg[1] * lg[2] * b.element_size(), dev)
assert win.ok, win.why
cg_p = mdist.SlabCG(eng, part, d, dist, window=win)
runs, sums = {}, {}
L = lg[0]
for name, cg in (("rccl1", cg_r), ("p2p1", cg_p), ("rccl2", cg_r), ("p2p2", cg_p)):
    cg.begin(0.0); cg.iterate(V)
    if name.startswith("p2p"):
        cg.finish()
    torch.cuda.synchronize()
    runs[name] = eng.history()[: 2 * V + 1].copy()
    # where a difference sits: plane sums (exact in fp64) of q and r on the edge planes / the interior, of x on owned planes
    f = lambda t: float(t.double().sum().item())  # noqa: E731
    sums[name] = [f(q[1]), f(q[L - 2]), f(q[2:L - 2]), f(r[1]), f(r[L - 2]), f(r[2:L - 2]), f(x[1:L - 1]), f(d[1]), f(d[L - 2]), f(d[2:L-2])]
allsums = [None] * world
dist.all_gather_object(allsums, sums)
rel = lambda a, c: float(np.max(np.abs(a - c) / np.abs(c)))  # noqa: E731
if rank == 0:
    print(json.dumps({"world": world, "dtype": dts, "grid": ggrid, "iters": V, "loop_info": eng.loop_info(),
                      "rccl_vs_rccl": rel(runs["rccl1"], runs["rccl2"]), "p2p_vs_p2p": rel(runs["p2p1"], runs["p2p2"]),
                      "p2p_vs_rccl": rel(runs["p2p1"], runs["rccl1"]),
                      "plane_sum_labels": ["q1", "qL-2", "q_int", "r1", "rL-2", "r_int", "x_owned", "d1", "dL-2", "d_int"],
                      "p2p1_minus_p2p2_per_rank": [[a - c for a, c in zip(sr["p2p1"], sr["p2p2"])] for sr in allsums],
                      "p2p1_minus_rccl1_per_rank": [[a - c for a, c in zip(sr["p2p1"], sr["rccl1"])] for sr in allsums],
                      "per_entry_p2p_vs_rccl": [float(v) for v in np.abs(runs["p2p1"] - runs["rccl1"]) / np.abs(runs["rccl1"])]}))
win.close()
dist.destroy_process_group()
