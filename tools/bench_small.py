"""GPU: wall-clock of whole PressureCGSolver3D.solve() calls on the reference notebook's grid
(48x80x48 = 184,320 cells, fp64 state, tol 1e-3) -- the one quantity the reference publishes
(0.747 s mean per solve on a 24 GB GeForce; different scene and hardware: context only)."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
from mfs import scenes
import solver.CGSolverBuffer as B, solver.PressureCGSolver3D as P
dev = torch.device("cuda:0")
gres = (48, 80, 48)
prec = sys.argv[1] if len(sys.argv) > 1 else "fp64"
sc = scenes.pressure_scene_3d(gres, seed=0, bound_size=(0.6, 1.0, 0.6), device=dev)
buf = B.CGSolverBuffer(gres, precision=prec, device=dev)
for ce in (8, 32, 128):
    s = P.PressureCGSolver3D(buf, gres, 0.0125 * np.array(gres), check_every=ce)
    ts = []
    for rep in range(6):
        v = [sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.solve(*v, sc["sphi"], sc["sv"], sc["lphi"], tol=1e-3)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(json.dumps({"grid": gres, "precision": prec, "check_every": ce, "iterations": s.iterations,
                      "solve_ms_min": round(min(ts) * 1e3, 3), "solve_ms_median": round(sorted(ts)[3] * 1e3, 3),
                      "us_per_iteration": round(min(ts) / max(s.iterations, 1) * 1e6, 2)}))
