"""GPU: the viscosity CG loop on a small (launch-bound) grid -- time per iteration with the merged vector phases
(k_update_rdx, default) and with the three-launch loop, and the fixed cost of a whole solve() around the loop.
usage: python tools/visc_small_iter.py Nx Ny Nz [f64|f32]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
import solver.ViscosityCGSolver3D as V
gres = tuple(int(v) for v in sys.argv[1:4]); dts = sys.argv[4] if len(sys.argv) > 4 else "f64"
dev = torch.device("cuda:0")
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
out = {"grid": list(gres), "dtype": dts}
for merged in ("resident", True, False, "resident", True, False):
    s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision={"f64": "fp64", "f32": "fp32"}[dts], device=dev)
    e = s._engine
    e.set_resident(merged == "resident")
    e.set_merged(bool(merged))
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    s.solve(sc["dt"], 50.0, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=1e-3)   # warm-up
    vx, vy, vz = sc["vx"].clone(), sc["vy"].clone(), sc["vz"].clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.solve(sc["dt"], 50.0, sc["rho"], vx, vy, vz, sc["sphi"], sc["sv"], sc["lphi"], sc["lvol"], tol=1e-3)
    torch.cuda.synchronize(); t_solve = time.perf_counter() - t0
    e.begin(0.0); e.iterate(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); e.iterate(1000); torch.cuda.synchronize(); t_it = (time.perf_counter() - t0) / 1000
    key = "resident" if merged == "resident" else ("merged" if merged else "three_launch")
    out.setdefault(key, []).append({"loop_us_per_iteration": round(t_it * 1e6, 2), "solve_ms": round(t_solve * 1e3, 3),
                                    "solve_iterations": s.iterations, "info": e.loop_info()})
print(json.dumps(out))
