"""GPU: the notebook's OWN scene (3D_viscous_fluid_sim.ipynb code cell 9: 48 x 80 x 48 cells, GDX 0.0125, PDX 0.00625,
container box + four slanted plates, a 0.3^3 fluid block, RHO 1000, MU 1, DT 1/300) stepped with
notebook_sim.NotebookSimulation (the `apic` branch: CG viscosity solve).  Per-stage wall clock per step, for the
context numbers of SURVEY.md section 6 (the reference's committed run prints p2g / visco(U-Net) / press seconds per
step on its GeForce).   usage: python tools/run_notebook_scene.py [steps]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
import notebook_sim as NSIM
import solver.sdf3D as sdf
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
GDX, PDX = 0.0125, 0.00625
rb_d, rb_map = sdf.generate_rb(None, {}, 'cube', ['box', 0.5, 0.8, 0.5], flip=True, center=[0, 0.5, 0], axis=[0, 1, 0], angle=0, device=dev)
h = 0.7
for nm, par, c, ax, ang in (("cube1", ['box', 0.67, 0.1, 1.0], [-0.34, h, 0], [0, 0, 1], -45), ("cube2", ['box', 0.67, 0.1, 1.0], [0.34, h, 0], [0, 0, 1], 45),
                            ("cube3", ['box', 1.0, 0.1, 0.7], [0, h, -0.3], [1, 0, 0], 45), ("cube4", ['box', 1.0, 0.1, 0.7], [0, h, 0.3], [1, 0, 0], -45)):
    rb_d, rb_map = sdf.generate_rb(rb_d, rb_map, nm, par, flip=False, center=c, axis=ax, angle=ang)


def oob_filter(px):
    p = torch.as_tensor(px, dtype=torch.float64, device=dev)
    sd = torch.zeros(p.shape[0], dtype=torch.float64, device=dev)
    vel = torch.zeros_like(p)
    sdf.evaluate(rb_d, sd, vel, p)
    return (sd >= 0).cpu().numpy()


px = NSIM.add_box([0.0, 0.65, 0.0], [0.3, 0.3, 0.3], PDX, np.random.default_rng(0), keep=oob_filter)
JAC = len(sys.argv) > 2 and sys.argv[2] == "jacobi"      # usage: run_notebook_scene.py [steps] [jacobi]
sim = NSIM.NotebookSimulation((48, 80, 48), GDX, [-0.3, 0, -0.3], rb_d, px, PDX, rho=1000, mu=1.0, dt=1 / 300, device=dev, jacobi=JAC)
sim.step()
tim, its = {}, []
t0 = time.perf_counter()
for _ in range(steps):
    sim.step(timings=tim)
    its.append((sim.DensitySolver.iterations, sim.ViscositySolver.iterations, sim.PressureSolver.iterations))
torch.cuda.synchronize()
t = time.perf_counter() - t0
its = np.array(its)
print(json.dumps({"scene": "notebook code cell 9 (48x80x48, %d particles)" % sim.particle.num_particles, "steps": steps,
                  "s_per_step": round(t / steps, 4), "stage_ms_per_step": {k: round(v / steps * 1e3, 3) for k, v in tim.items()},
                  "mean_cg_iterations(density,viscosity,pressure)": [round(float(v), 1) for v in its.mean(axis=0)],
                  "max_cg_iterations": [int(v) for v in its.max(axis=0)],
                  "particles_after": {"mean": [round(float(v), 5) for v in sim.particle.x.mean(dim=0)],
                                      "min": [round(float(v), 5) for v in sim.particle.x.min(dim=0).values],
                                      "max": [round(float(v), 5) for v in sim.particle.x.max(dim=0).values],
                                      "speed_max": round(float(sim.particle.v.norm(dim=1).max()), 4)},
                  "reference_context": "committed notebook output on a GeForce: press 0.745 s, visco (U-Net) 0.88 s, p2g 6.2 ms per step"}))
