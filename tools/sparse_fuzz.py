"""GPU: randomised geometry against the sparse paths of round 3 (a soak, not part of the test suite): pressure solves of
random sets of liquid balls (touching walls, overlapping, thin) with the sparse lists on and off through ONE engine each, and
viscosity solves of liquid boxes whose faces are snapped to vector / tile boundaries with compressed class access + lists on and
off.  Prints one line per case and a verdict.   usage: python tools/sparse_fuzz.py [cases] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import numpy as np, torch
from mfs.pcg import PcgEngine
from mfs import scenes
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
bad = 0


def pressure_case(k, jacobi=False):
    global bad
    gres = [(160, 96, 144), (130, 128, 128), (96, 160, 160), (258, 96, 96)][k % 4]
    dt = torch.float64 if k % 2 == 0 else torch.float32
    ax = [torch.arange(n, device=dev, dtype=torch.float64) + 0.5 for n in gres]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")
    lphi = torch.full(gres, 1e3, dtype=torch.float64, device=dev)
    for _ in range(int(rng.integers(1, 4))):
        c = [rng.uniform(0, n) for n in gres]               # centres anywhere, incl. outside margins: balls cut by the walls
        rad = rng.uniform(3, 0.3 * min(gres))
        sq = rng.uniform(0.15, 1.0)                            # squashed in z: thin sheets
        lphi = torch.minimum(lphi, torch.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + ((Z - c[2]) / sq) ** 2) - rad)
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    def w(shape):
        qq = torch.randint(0, 5, shape, generator=g, device=dev).double() * 0.25          # incl. weight 0 (closed faces)
        return torch.where(torch.rand(shape, generator=g, device=dev) < 0.85, torch.ones(shape, dtype=torch.float64, device=dev), qq)
    wx, wy, wz = w((gres[0] + 1, gres[1], gres[2])), w((gres[0], gres[1] + 1, gres[2])), w((gres[0], gres[1], gres[2] + 1))
    b = torch.randn(gres, generator=g, device=dev, dtype=torch.float64) * (lphi < 0)
    b[0] = 0; b[-1] = 0; b[:, 0] = 0; b[:, -1] = 0; b[:, :, 0] = 0; b[:, :, -1] = 0
    out = {}
    for sparse in (True, False):
        eng = PcgEngine(gres, dt, dev)
        eng.set_sparse(sparse)
        if jacobi:
            eng.set_jacobi(True)
        eng.setup(lphi.to(dt), wx.to(dt), wy.to(dt), wz.to(dt))
        x, d, r, q = (torch.zeros(gres, dtype=dt, device=dev) for _ in range(4))
        eng.bind(b.to(dt), x, d, r, q)
        # to convergence.  These scenes are badly conditioned (closed faces, sheets two cells thin, theta clamped at 0.01): a
        # 1e-16 difference in a dot product grows about tenfold per iteration in fp64 state, so the history
        # is compared over a leading window and the CONVERGED field at the tests' 1e-4 of its maximum
        ok_, it = eng.solve(1e-6 if dt == torch.float64 else 1e-3, 20000, 16)
        torch.cuda.synchronize()
        out[sparse] = (np.asarray(eng.history()), x.clone(), eng.sparse_info(), float(eng.scalars[13]), it, ok_)
    (h1, x1, info, lane, it1, c1), (h0, x0, _, _, it0, c0) = out[True], out[False]
    n = min(len(h1), len(h0), 13)
    hdev = float(np.max(np.abs(h1[:n] - h0[:n]) / np.abs(h0[:n])))
    xdev = float((x1 - x0).abs().max() / x0.abs().max())
    ok = c1 and c0 and hdev < (1e-9 if dt == torch.float64 else 1e-4) and xdev < 1e-4 and abs(it1 - it0) <= max(3, it0 // 10) and \
        info["listed_pairs"] > 0 and torch.equal(x1[x0 == 0], x0[x0 == 0])
    bad += not ok
    print(f"pressure{' jacobi' if jacobi else ''} {k:2d} {str(gres):16s} {str(dt)[6:]:8s} fluid {float((lphi < 0).double().mean()):.3f} pairs {info['listed_pairs']}/{info['pairs']} "
          f"chunks {info['live_chunks']}/{info['chunks']} lane-mask {int(lane)} | iterations {it1} / {it0}: history dev (first {n}) {hdev:.1e} "
          f"converged x dev {xdev:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)


def viscosity_case(k):
    global bad
    import solver.ViscosityCGSolver3D as V
    gres = (96, 96, 96)
    prec = "fp64" if k % 2 == 0 else "fp32"
    sc = scenes.viscosity_scene_3d(gres, seed=5, device=dev)
    D = [2 * n + 1 for n in gres]
    # a liquid box with faces ON doubled-grid nodes that are multiples of 8 (= vector boundaries of both precisions) or anywhere
    lo = [int(rng.integers(6, D[a] // 2)) for a in range(3)]
    hi = [int(rng.integers(lo[a] + 6, D[a] - 6)) for a in range(3)]
    if k % 3 != 2:
        lo[2] -= lo[2] % 8; hi[2] -= hi[2] % 8; hi[2] = max(hi[2], lo[2] + 8)
    lvol = torch.zeros_like(sc["lvol"])
    cv = float(np.prod(np.array(sc["bound_size"]) / np.array(gres))) * 0.125
    lvol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = cv
    if k % 4 == 1:
        lvol[lo[0]:hi[0], lo[1]:hi[1], hi[2]] = 0.3 * cv       # a partially filled layer exactly behind a vector boundary
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    v = [t.clone() for t in (sc["vx"], sc["vy"], sc["vz"])]
    for t in v:
        t.copy_(torch.randn(t.shape, generator=g, device=dev, dtype=torch.float64).to(t.dtype))
    outs = {}
    for sparse in ("1", "0"):
        os.environ["MFS_VISC_SPARSE"] = sparse
        os.environ["MFS_VISC_COMPRESS"] = sparse
        s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=prec, device=dev)
        vv = [t.clone() for t in v]
        s.solve(sc["dt"], 5.0, sc["rho"], *vv, sc["sphi"], sc["sv"], sc["lphi"], lvol)
        torch.cuda.synchronize()
        outs[sparse] = (s.iterations, np.asarray(s.history), vv, s._engine.sparse_info())
    (i1, h1, v1, info), (i0, h0, v0, _) = outs["1"], outs["0"]
    n = min(len(h1), len(h0), 31)
    hdev = float(np.max(np.abs(h1[:n] - h0[:n]) / np.abs(h0[:n])))
    vdev = max(float((a - b_).abs().max()) for a, b_ in zip(v1, v0)) / max(float(b_.abs().max()) for b_ in v0)
    tol_h, tol_v = (1e-9, 1e-6) if prec == "fp64" else (2e-3, 1e-3)      # (leading window of the history; converged velocities)
    ok = abs(i1 - i0) <= max(1, i0 // 20) and hdev < tol_h and vdev < tol_v and info["listed_pairs"] > 0
    bad += not ok
    print(f"viscosity {k:2d} box nodes {lo}..{hi} {prec} pairs {info['listed_pairs']}/{info['pairs']} chunks {info['live_chunks']}/{info['chunks']} | "
          f"iterations {i1} / {i0}: history dev {hdev:.1e} velocity dev {vdev:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)


for k in range(cases):
    pressure_case(k)
for k in range(cases):
    pressure_case(k, jacobi=True)      # the opt-in Jacobi loop's fused form takes the lists too
for k in range(cases):
    viscosity_case(k)
print("VERDICT:", "all cases agree" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
