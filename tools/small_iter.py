"""GPU: CG iteration time of the pressure engine on a small (launch-bound) grid -- the reference notebook's own
48x80x48 by default.  usage: python tools/small_iter.py [Nx Ny Nz] [f32|f64] [iters]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
from mfs.pcg import PcgEngine
import solver.PressureCGSolver3D as P, solver.SolidFraction3D as S
a = sys.argv[1:]
gres = tuple(int(v) for v in a[:3]) if len(a) >= 3 else (48, 80, 48)
a = a[3:] if len(a) >= 3 else a
dts = a[0] if a else "f64"
iters = int(a[1]) if len(a) > 1 else 2000
dt = {"f32": torch.float32, "f64": torch.float64}[dts]
dev = torch.device("cuda:0")
sc = scenes.pressure_scene_3d(gres, seed=0, device=dev)
wx = torch.zeros((gres[0] + 1, gres[1], gres[2]), dtype=dt, device=dev)
wy = torch.zeros((gres[0], gres[1] + 1, gres[2]), dtype=dt, device=dev)
wz = torch.zeros((gres[0], gres[1], gres[2] + 1), dtype=dt, device=dev)
S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
b, x, d, r, q = (torch.zeros(gres, dtype=dt, device=dev) for _ in range(5))
P.initialize_solver(sc["cell_size"], gres, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
eng = PcgEngine(gres, dt, dev); eng.setup(sc["lphi"], wx, wy, wz); eng.bind(b, x, d, r, q)
eng.begin(0.0); eng.iterate(50); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.iterate(iters); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / iters * 1e6)
print(json.dumps({"grid": gres, "dtype": dts, "cells": gres[0] * gres[1] * gres[2], "loop": eng.loop_info(),
                  "us_per_iteration": [round(t, 2) for t in ts]}))
