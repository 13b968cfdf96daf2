"""GPU: ViscosityCGSolver3D (BASELINE config 3: 128^3 buckling-like scene) -- per-apply and
per-iteration time against the algorithmic traffic of SURVEY.md 8(d) (apply 16 N^3, iteration 43 N^3 scalars).
usage: python tools/bench_viscosity.py [N] [dtype] [iters]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
from mfs import scenes
from mfs.vcg import VcgEngine
import solver.ViscosityCGSolver3D as V
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dts = sys.argv[2] if len(sys.argv) > 2 else "f32"
dt = {"f32": torch.float32, "f64": torch.float64}[dts]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0"); gres = (N, N, N); esz = 4 if dts == "f32" else 8
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dts, device=dev)
# full solve once (parity-style run), then fixed-iteration timing
vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
t0 = time.perf_counter()
s.solve(sc["dt"], 50.0, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=1e-3)
torch.cuda.synchronize(); t_solve = time.perf_counter() - t0
eng = s._engine
f = s._flat
eng.begin(0.0); eng.iterate(10); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.iterate(iters); torch.cuda.synchronize(); t_it = (time.perf_counter() - t0) / iters
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.apply(f["d"], f["q"]); a.record()
for _ in range(50): eng.apply(f["d"], f["q"])
e.record(); torch.cuda.synchronize(); t_ap = a.elapsed_time(e) / 50 * 1e-3
# the slab-decomposed loop's per-iteration kernels on this one GPU (1 rank: no exchange, no all-reduce -- what the
# multi-GPU loop costs before its collectives): apply, 2 reductions, x/r update, direction update as separate phases
from mfs.dist import SlabPartition, SlabVCG
cgs = SlabVCG(eng, SlabPartition(N, 1, 0), (s.d_x, s.d_y, s.d_z))
cgs.begin(0.0); cgs.iterate(10); torch.cuda.synchronize()
t0 = time.perf_counter(); cgs.iterate(iters); torch.cuda.synchronize(); t_slab = (time.perf_counter() - t0) / iters
# ... and the same loop as the library enqueues it over a peer-to-peer window (1-rank window: the all-reduces go through
# the window to this rank itself, no planes move): what the multi-GPU window loop costs before xGMI
import torch.distributed as dist
from mfs.p2p import P2PWindow
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
win = P2PWindow(dist, eng.edge_plane_bytes(), dev)
t_win = float("nan")
if win.ok:
    cgw = SlabVCG(eng, SlabPartition(N, 1, 0), (s.d_x, s.d_y, s.d_z), dist, window=win)
    cgw.begin(0.0); cgw.iterate(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); cgw.iterate(iters); torch.cuda.synchronize(); t_win = (time.perf_counter() - t0) / iters
    eng.attach_p2p(None)
    win.close()
dist.destroy_process_group()
cells = N ** 3
out = {"workload": f"ViscosityCGSolver3D {N}^3 {dts}", "solve_iterations": s.iterations, "solve_s": round(t_solve, 4),
       "iter_us": round(t_it * 1e6, 2), "iters_per_s": round(1 / t_it, 1), "Mcells_per_s": round(cells / t_it / 1e6, 1),
       "iter_GBs_alg(43N^3)": round(43 * cells * esz / t_it / 1e9, 1),
       "slab_phase_loop_iter_us": round(t_slab * 1e6, 2), "slab_window_loop_iter_us": round(t_win * 1e6, 2),
       "apply_us": round(t_ap * 1e6, 2), "apply_GBs_alg(16N^3)": round(16 * cells * esz / t_ap / 1e9, 1)}
print(json.dumps(out))
