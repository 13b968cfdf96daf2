"""GPU: in-process A/B of an engine knob of the viscosity CG (environment variable read at engine creation): two
solvers on the same scene, timed alternately (A, B, A, B, ...) so that machine state drifts hit both alike.
usage: python tools/visc_ab.py N dtype VAR valueA valueB [iters] [rounds]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
import solver.ViscosityCGSolver3D as V
N, dts, var, va, vb = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 100
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 4
dev = torch.device("cuda:0"); gres = (N, N, N)
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
engs = []
for val in (va, vb):
    os.environ[var] = val
    s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dts, device=dev)
    scale = sc["dt"] / s.cell_vol / sc["rho"]
    torch.div(sc["lvol"], s.cell_vol * 0.125, out=s.vol)
    s.x_x.copy_(sc["vx"]); s.x_y.copy_(sc["vy"]); s.x_z.copy_(sc["vz"])
    V.extrapolate(gres, 3, s.x_x, s.x_y, s.x_z, sc["sphi"])
    V.initialize_solver(gres, scale, 50.0, s.x_x, s.x_y, s.x_z, sc["sphi"], sc["sv"], s.vol, s.b_x, s.b_y, s.b_z)
    s._engine.setup(scale, 50.0, sc["sphi"], s.vol)
    f = s._flat
    s._engine.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
    s._engine.begin(0.0); s._engine.iterate(10)
    engs.append(s)
torch.cuda.synchronize()
res = {va: [], vb: []}
for _ in range(rounds):
    for val, s in zip((va, vb), engs):
        t0 = time.perf_counter(); s._engine.iterate(iters); torch.cuda.synchronize()
        res[val].append(round((time.perf_counter() - t0) / iters * 1e6, 2))
# knobs that the library reads per launch: the SAME engine (the first) with the variable toggled between loops --
# no placement difference at all (at 256^3 the first-created solver of a process runs ~6 % faster than the second)
same = {va: [], vb: []}
for _ in range(rounds):
    for val in (va, vb):
        os.environ[var] = val
        s = engs[0]
        t0 = time.perf_counter(); s._engine.iterate(iters); torch.cuda.synchronize()
        same[val].append(round((time.perf_counter() - t0) / iters * 1e6, 2))
# where a difference sits: the phases of one iteration (collective-loop form of the same kernels), event-timed
ph = {}
for val, s in zip((va, vb), engs):
    e = s._engine
    out = {}
    for name, fn in (("apply", e.phase_apply), ("xr", e.phase_update_xr), ("d", e.phase_update_d)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        out[name] = round(a.elapsed_time(b) / 20 * 1e3, 1)
    f = s._flat
    out["addr_mod_2MiB_KiB(b,x,d,r,q)"] = [int(f[n].data_ptr() % (2 << 20)) // 1024 for n in "bxdrq"]
    ph[val] = out
print(json.dumps({"N": N, "dtype": dts, "knob": var, "iter_us": res, "same_engine_toggled_iter_us": same, "phase_us": ph}))
