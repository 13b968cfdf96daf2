"""GPU: sweep the stencil-kernel tuning knobs (mfs_pcg3d_tune) and time each setting.
usage: python tools/apply_sweep.py [N] [dtype] [reps]"""
import itertools, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
from mfs.pcg import PcgEngine
import solver.SolidFraction3D as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
gres = (N, N, N)
sc = scenes.pressure_scene_3d(gres, seed=0, device=dev)
wx = torch.zeros((N + 1, N, N), dtype=dt, device=dev); wy = torch.zeros((N, N + 1, N), dtype=dt, device=dev)
wz = torch.zeros((N, N, N + 1), dtype=dt, device=dev)
S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
eng = PcgEngine(gres, dt, dev)
eng.setup(sc["lphi"], wx, wy, wz)
lphi = sc["lphi"]; del sc; torch.cuda.empty_cache()
g = torch.Generator(device=dev).manual_seed(1)
v = torch.randn(gres, generator=g, device=dev, dtype=dt)
ref = torch.zeros(gres, dtype=dt, device=dev)
eng.tune(0, 16, 8, 0); eng.apply(v, ref); torch.cuda.synchronize()
esz = 4 if dt == torch.float32 else 8
alg = (6 * N**3 + 3 * N**2) * esz
out = torch.zeros(gres, dtype=dt, device=dev)
cfgs = [(0, 16, b, 0) for b in (4, 8, 16)]
cfgs += [(var, xc, b, nt) for var in (1, 2) for xc in (0, 16) for b in (1, 2, 3, 4, 6) for nt in (0, 1)]
res = []
for var, xc, b, nt in cfgs:
    eng.tune(var, xc, b, nt)
    out.zero_(); eng.apply(v, out); torch.cuda.synchronize()
    ok = torch.equal(out, ref)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): eng.apply(v, out)
    s.record()
    for _ in range(reps): eng.apply(v, out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    res.append((us, var, xc, b, nt, ok))
    print(f"var {var} xchunk {xc:3d} blocks/CU {b:2d} nt {nt}: {us:8.2f} us  {alg/us/1e3:7.1f} GB/s  {alg/us/1e3/80:5.1f}%  equal={ok}", flush=True)
res.sort()
print("BEST:", res[:5])
