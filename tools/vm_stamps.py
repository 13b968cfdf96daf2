"""GPU: where a step of the viscosity march kernel spends its cycles -- reads the in-kernel stamps of a diagnostic build
(tools/build_variant.sh stamps "-DMFS_VM_STAMPS"; MFS_LIB=.../libmfs_hip_stamps.so).  Shares, not lengths: the stamps'
fences forbid overlaps the product kernel has.  usage: MFS_LIB=... python tools/vm_stamps.py N [f32|f64]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
from mfs import scenes
import solver.ViscosityCGSolver3D as V
N = int(sys.argv[1]); dts = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0"); gres = (N, N, N)
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dts, device=dev)
scale = sc["dt"] / s.cell_vol / sc["rho"]
torch.div(sc["lvol"], s.cell_vol * 0.125, out=s.vol)
s.x_x.copy_(sc["vx"]); s.x_y.copy_(sc["vy"]); s.x_z.copy_(sc["vz"])
V.extrapolate(gres, 3, s.x_x, s.x_y, s.x_z, sc["sphi"])
V.initialize_solver(gres, scale, 50.0, s.x_x, s.x_y, s.x_z, sc["sphi"], sc["sv"], s.vol, s.b_x, s.b_y, s.b_z)
e = s._engine
e.setup(scale, 50.0, sc["sphi"], s.vol)
f = s._flat
e.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
e.begin(0.0); e.iterate(3)
for _ in range(5): e.phase_apply()
torch.cuda.synchronize()
off = 256 + 16384 * 8                      # scalars, history, then the d.q partial sums (csrc/mfs_cg_core.h core_carve)
part = e.workspace[off:off + 8192 * 8].view(torch.float64).cpu().numpy()
st = part[4096:4096 + 64 * 4 * 12].reshape(64, 4, 12)
names = ["prologue", "barrier", "issue own+halo loads", "u image reads", "u rows", "u store+issue", "v image reads", "v rows",
         "v store+issue", "w image reads", "w rows", "w store+issue+publish"]
tot = st.sum(axis=2)
print(json.dumps({"N": N, "dtype": dts, "kernel": e.apply_kernel(), "ticks_per_wave_mean": float(tot.mean()),
                  "share_by_segment": {n: round(float(st[:, :, k].sum() / st.sum()), 4) for k, n in enumerate(names)},
                  "share_by_wave_of_wg": [round(float(st[:, w, :].sum() / st.sum()), 4) for w in range(4)],
                  "wave0_vs_wave3_barrier_share": [round(float(st[:, w, 1].sum() / st[:, w, :].sum()), 4) for w in (0, 3)]}))
