"""GPU: resident vs launch-per-phase loop on one grid; prints where the vectors differ.  usage: res_debug.py Nx Ny Nz [fp64|fp32] [iters] [check_every]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
import test_resident_gpu as TR
g = tuple(int(v) for v in sys.argv[1:4]); prec = sys.argv[4] if len(sys.argv) > 4 else "fp64"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 12
ce = int(sys.argv[6]) if len(sys.argv) > 6 else 5
a = TR._run(g, prec, True, iters, ce); b = TR._run(g, prec, False, iters, ce)
print("info", a["info"], "iters", a["iters"], b["iters"])
print("hist rel err", np.max(np.abs(a["hist"] - b["hist"]) / np.abs(b["hist"])))
for k in ("x", "d", "r", "q"):
    e = (a[k] - b[k]).abs()
    i = int(e.argmax()); idx = np.unravel_index(i, g)
    print(k, "max err", float(e.max()), "scale", float(b[k].abs().max()), "at", idx, "a", float(a[k].flatten()[i]), "b", float(b[k].flatten()[i]),
          "n bad", int((e > 1e-8 * float(b[k].abs().max())).sum()))
