"""GPU: does replaying the CG loop as a captured HIP graph beat launching it (small grids)?
usage: python tools/graph_probe.py [N] [dtype]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
from mfs.pcg import PcgEngine
import solver.PressureCGSolver3D as P, solver.SolidFraction3D as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[2] if len(sys.argv) > 2 else "f64"]
dev = torch.device("cuda:0"); gres = (N, N, N)
sc = scenes.pressure_scene_3d(gres, seed=0, device=dev)
wx = torch.zeros((N + 1, N, N), dtype=dt, device=dev); wy = torch.zeros((N, N + 1, N), dtype=dt, device=dev)
wz = torch.zeros((N, N, N + 1), dtype=dt, device=dev)
S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
b, x, d, r, q = (torch.zeros(gres, dtype=dt, device=dev) for _ in range(5))
P.initialize_solver(sc["cell_size"], gres, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
eng = PcgEngine(gres, dt, dev); eng.setup(sc["lphi"], wx, wy, wz); eng.bind(b, x, d, r, q)
CH, REP = 32, 20
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eng.begin(0.0); eng.iterate(2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP): eng.iterate(CH)
    torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / (REP * CH)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.iterate(CH)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP): g.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / (REP * CH)
print(f"{N}^3 {dt}: eager {t_eager*1e6:.2f} us/iteration, graph replay {t_graph*1e6:.2f} us/iteration, iterations {eng.poll()['iterations']}")
