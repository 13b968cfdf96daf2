"""GPU: iterations and time of a whole pressure solve with and without the opt-in Jacobi preconditioner
(synthetic pool scene, fp64 state, the reference's default tol = 1e-3).   usage: python tools/jacobi_compare.py [N]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
from solver.CGSolverBuffer import CGSolverBuffer
from solver.PressureCGSolver3D import PressureCGSolver3D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0"); gres = (N, N, N)
sc = scenes.pressure_scene_3d(gres, seed=0, device=dev)
out = {}
for jac in (False, True):
    buf = CGSolverBuffer(gres, precision="fp64", device=dev)
    s = PressureCGSolver3D(buf, gres, sc["bound_size"], jacobi=jac)
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    s.solve(vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"])          # warm-up (allocations)
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.solve(vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"])
    torch.cuda.synchronize()
    out["jacobi" if jac else "reference_cg"] = {"iterations": s.iterations, "solve_ms": round((time.perf_counter() - t0) * 1e3, 2),
                                                 "delta": s.delta}
print(json.dumps({"workload": f"PressureCGSolver3D {N}^3 fp64 pool scene, tol 1e-3", **out}))
