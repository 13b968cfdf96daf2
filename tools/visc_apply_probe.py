"""GPU: time the viscosity CG's operator launch alone (mfs_vcg3d_phase_apply: the per-iteration kernel + the boundary
slabs launch) on the direction vector of a begun solve.  usage: python tools/visc_apply_probe.py [N] [dtype] [reps]
Environment knobs of csrc/mfs_visc.hip apply (MFS_VISC_XCD, MFS_VISC_TILED, ...); MFS_PROBE_LIB = another build of the library."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes, _lib
if os.environ.get("MFS_PROBE_LIB"):          # experiments: another build of the library (path relative to the repo)
    _lib.LIB_PATH = os.path.join(REPO, os.environ["MFS_PROBE_LIB"])
import solver.ViscosityCGSolver3D as V
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dts = sys.argv[2] if len(sys.argv) > 2 else "f32"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0"); gres = (N, N, N); esz = 4 if dts == "f32" else 8
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dts, device=dev)
scale = sc["dt"] / s.cell_vol / sc["rho"]
torch.div(sc["lvol"], s.cell_vol * 0.125, out=s.vol)
s.x_x.copy_(sc["vx"]); s.x_y.copy_(sc["vy"]); s.x_z.copy_(sc["vz"])
V.extrapolate(gres, 3, s.x_x, s.x_y, s.x_z, sc["sphi"])
V.initialize_solver(gres, scale, 50.0, s.x_x, s.x_y, s.x_z, sc["sphi"], sc["sv"], s.vol, s.b_x, s.b_y, s.b_z)
eng = s._engine
eng.setup(scale, 50.0, sc["sphi"], s.vol)
f = s._flat
eng.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
eng.begin_local(0.0); eng.begin_finish()
for _ in range(5): eng.phase_apply()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): eng.phase_apply()
e.record(); torch.cuda.synchronize()
t = a.elapsed_time(e) / reps * 1e-3
print(json.dumps({"N": N, "dtype": dts, "apply_launches_us": round(t * 1e6, 2),
                  "GBs_alg(13N^3+3B)": round((13 * esz + 3) * N ** 3 / t / 1e9, 1),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("MFS_VISC")}}))
