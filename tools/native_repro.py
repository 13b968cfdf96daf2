"""GPU: run-to-run reproducibility of the single-GPU CG loops (pressure 256^3, viscosity 192^3; fp32 and fp64 state):
each loop three times from the same start; residual histories and solutions must be bit-identical.
usage: python tools/native_repro.py"""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
from mfs import scenes
from mfs.pcg import PcgEngine
import solver.PressureCGSolver3D as P
import solver.SolidFraction3D as S
import solver.ViscosityCGSolver3D as V
dev = torch.device("cuda:0")
out = {}
for dts in ("f32", "f64"):
    tdt = torch.float32 if dts == "f32" else torch.float64
    g = (256, 256, 256)
    sc = scenes.pressure_scene_3d(g, seed=0, device=dev)
    wx, wy, wz = (torch.zeros(tuple(g[a] + (a == c) for a in range(3)), dtype=tdt, device=dev) for c in range(3))
    S.compute_solid_frac(g, sc["sphi"], wx, wy, wz)
    b, x, d, r, q = (torch.zeros(g, dtype=tdt, device=dev) for _ in range(5))
    P.initialize_solver(sc["cell_size"], g, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
    eng = PcgEngine(g, tdt, dev); eng.setup(sc["lphi"], wx, wy, wz); eng.bind(b, x, d, r, q)
    hs, xs = [], []
    for rep in range(3):
        eng.begin(0.0); eng.iterate(60); eng.finish(); torch.cuda.synchronize()
        hs.append(eng.history()[:121].copy()); xs.append(x.clone())
    out["pressure_" + dts] = [bool(np.array_equal(hs[0], h)) and bool(torch.equal(xs[0], xx)) for h, xx in zip(hs[1:], xs[1:])]
    del sc, eng, b, x, d, r, q, wx, wy, wz
    torch.cuda.empty_cache()
gres = (192, 192, 192)
sc = scenes.viscosity_scene_3d(gres, seed=3, device=dev)
for dts in ("f32", "f64"):
    s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=dts, device=dev)
    scale = sc["dt"] / s.cell_vol / sc["rho"]
    torch.div(sc["lvol"], s.cell_vol * 0.125, out=s.vol)
    s.x_x.copy_(sc["vx"]); s.x_y.copy_(sc["vy"]); s.x_z.copy_(sc["vz"])
    V.extrapolate(gres, 3, s.x_x, s.x_y, s.x_z, sc["sphi"])
    x0 = s._flat["x"].clone()
    V.initialize_solver(gres, scale, 50.0, s.x_x, s.x_y, s.x_z, sc["sphi"], sc["sv"], s.vol, s.b_x, s.b_y, s.b_z)
    s._engine.setup(scale, 50.0, sc["sphi"], s.vol)
    f = s._flat; s._engine.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
    hs, xs = [], []
    for rep in range(3):
        f["x"].copy_(x0)
        s._engine.begin(0.0); s._engine.iterate(40); torch.cuda.synchronize()
        hs.append(s._engine.history()[:81].copy()); xs.append(f["x"].clone())
    out["viscosity_" + dts] = [bool(np.array_equal(hs[0], h)) and bool(torch.equal(xs[0], xx)) for h, xx in zip(hs[1:], xs[1:])]
    del s
print(json.dumps(out))
