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
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
out = {}
for jac, fuse in ((False, True), (True, False), (True, True)):
    buf = CGSolverBuffer(gres, precision=prec, device=dev)
    s = PressureCGSolver3D(buf, gres, sc["bound_size"], jacobi=jac)
    s._engine.set_fuse(fuse)
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    s.solve(vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"])          # warm-up (allocations)
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.solve(vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    e = s._engine                     # the loop alone: a fixed number of iterations on the bound problem (tol 0), no set-up, no polls
    e.begin(0.0); e.iterate(20); torch.cuda.synchronize()
    t1 = time.perf_counter(); e.iterate(200); torch.cuda.synchronize()
    loop_us = (time.perf_counter() - t1) / 200 * 1e6
    e.finish()
    key = "reference_cg" if not jac else ("jacobi_fused" if fuse else "jacobi_three_launch")
    out[key] = {"iterations": s.iterations, "solve_ms": round(ms, 2), "solve_us_per_iteration": round(ms * 1e3 / max(1, s.iterations), 2), "loop_us_per_iteration": round(loop_us, 2),
                "delta": s.delta, "loop": s._engine.loop_info()}
print(json.dumps({"workload": f"PressureCGSolver3D {N}^3 {prec} pool scene, tol 1e-3", **out}))
