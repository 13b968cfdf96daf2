"""GPU, under torch.distributed.run (MFS_BENCH_SHARED_GPU=1: all ranks on cuda:0 over gloo): reproducibility of the
slab-decomposed viscosity solve -- the whole `SlabViscosityCGSolver3D.solve` twice per transport on an N^3 scene;
histories bit-identical within a transport, to summation-order level between them.  usage: ... tools/vslab_repro.py [N] [dtype]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch, torch.distributed as dist
from mfs import scenes
import solver.ViscosityCGSolver3D as V
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dts = sys.argv[2] if len(sys.argv) > 2 else "f32"
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
shared = os.environ.get("MFS_BENCH_SHARED_GPU", "0") == "1"
dev = torch.device("cuda", 0 if shared else int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("gloo" if shared else "nccl", rank=rank, world_size=world)
gres = (N, N, N)
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
runs = {}
for tr in ("p2p", "rccl"):
    s = V.SlabViscosityCGSolver3D(gres, sc["bound_size"], dist, precision={"f32": "fp32", "f64": "fp64"}[dts], device=dev, transport=tr)
    lo, hi = s.part.local_range
    for rep in (1, 2):
        vx, vy, vz = sc["vx"][lo:hi + 1].clone(), sc["vy"][lo:hi].clone(), sc["vz"][lo:hi].clone()
        s.solve(sc["dt"], 50.0, sc["rho"], vx, vy, vz, sc["sphi"][2 * lo:2 * hi + 1], sc["sv"][2 * lo:2 * hi + 1],
                sc["lphi"][lo:hi], sc["lvol"][2 * lo:2 * hi + 1], tol=1e-3)
        torch.cuda.synchronize()
        runs[f"{tr}{rep}"] = (s.history.copy(), vx.double().sum().item())
    s.close()
    del s
if rank == 0:
    n = min(len(runs["p2p1"][0]), len(runs["rccl1"][0]), 41)
    rel = lambda a, c: float(np.max(np.abs(a[:n] - c[:n]) / np.abs(c[:n])))  # noqa: E731
    print(json.dumps({"world": world, "N": N, "dtype": dts, "iterations": [(len(v[0]) - 1) // 2 for v in runs.values()],
                      "p2p_identical": bool(np.array_equal(runs["p2p1"][0], runs["p2p2"][0]) and runs["p2p1"][1] == runs["p2p2"][1]),
                      "rccl_identical": bool(np.array_equal(runs["rccl1"][0], runs["rccl2"][0]) and runs["rccl1"][1] == runs["rccl2"][1]),
                      "p2p_vs_rccl_first_entries": rel(runs["p2p1"][0], runs["rccl1"][0])}))
dist.destroy_process_group()
