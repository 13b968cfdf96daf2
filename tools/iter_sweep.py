"""GPU: time whole CG iterations (256^3 fp32 by default) under different cache-hint settings.
usage: python tools/iter_sweep.py [N] [dtype] [iters] [nt bits ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
from mfs import scenes
from mfs.pcg import PcgEngine
import solver.PressureCGSolver3D as P, solver.SolidFraction3D as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0"); gres = (N, N, N)
sc = scenes.pressure_scene_3d(gres, seed=0, device=dev)
wx = torch.zeros((N + 1, N, N), dtype=dt, device=dev); wy = torch.zeros((N, N + 1, N), dtype=dt, device=dev)
wz = torch.zeros((N, N, N + 1), dtype=dt, device=dev)
S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
b, x, d, r, q = (torch.zeros(gres, dtype=dt, device=dev) for _ in range(5))
P.initialize_solver(sc["cell_size"], gres, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
eng = PcgEngine(gres, dt, dev); eng.setup(sc["lphi"], wx, wy, wz); eng.bind(b, x, d, r, q)
del sc; torch.cuda.empty_cache()
# config = "variant,bpc,nt,compress" strings
# config = "variant,bpc,nt,compress[,fuse,pd]"
cfgs = sys.argv[4:] or ["2,2,15,1,1,2", "2,2,15,1,1,1", "2,2,15,1,0,2", "2,2,15,1,0,1", "2,2,15,0,1,2", "2,2,15,0,0,2"]
ref = None
for c in cfgs:
    f = [int(t) for t in c.split(",")] + [1, 2]
    var, bpc, nt, comp, fuse, pd = f[:6]
    eng.set_compress(comp); eng.set_fuse(fuse); eng.set_prefetch(pd)
    eng.tune(var, 0, bpc, nt)
    eng.begin(0.0); eng.iterate(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.iterate(iters); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
    a_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    o = torch.zeros_like(q); eng.apply(d, o); a_.record()
    for _ in range(20): eng.apply(d, o)
    e_.record(); torch.cuda.synchronize()
    if ref is None: ref = o.clone()
    print(f"variant {var} bpc {bpc} nt {nt:2d} compress {comp} fuse {fuse} pd {pd}: {t*1e6:8.2f} us/iter  {N**3/t/1e9:7.2f} Gcell/s   apply b2b {a_.elapsed_time(e_)/20*1e3:7.2f} us  equal={torch.equal(o, ref)}", flush=True)
