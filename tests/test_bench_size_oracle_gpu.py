"""The kernels the bench times, at the sizes it times them, against the oracle's C restatement
(oracle/mfs_oracle_c.c -- pinned by the executed-reference goldens in tests/test_oracle_c.py).  GPU only.

VERDICT r1: the goldens span at most a few tiles of the march kernels; what BASELINE.json's metric is quoted on --
`PressureCGSolver3D` 256^3 with the default engine (automatic nontemporal loads, compressed coefficient access, fused
direction update, deferred x update) -- was compared with nothing.  Here: the first 10 CG iterations' residual history
(fp32 state: north_star's 1e-5 rel; fp64 state: 1e-9) and x after finish(), at 256^3 (pressure, BASELINE config 2's
solver at the headline size) and 128^3 (viscosity, config 3).  The oracle starts from the same stored right-hand side /
initial guess (state-precision values), computes in fp64, and costs ~50 ms per 10 iterations at 256^3.
"""
import numpy as np
import pytest

from conftest import require_default_engine
import torch

from mfs import scenes
from oracle import cbaseline as CB

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = lambda t: t.double().cpu().numpy()  # noqa: E731


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.float64, 1e-9)], ids=["f32", "f64"])
def test_pressure_256_default_engine_vs_c_oracle(dt, tol):
    require_default_engine("test_pressure_256_default_engine_vs_c_oracle")
    import solver.PressureCGSolver3D as P
    import solver.SolidFraction3D as S
    from mfs.pcg import PcgEngine
    gres = (256, 256, 256)
    iters = 10
    sc = scenes.pressure_scene_3d(gres, seed=0, device=DEV)                       # bench.py's workload
    wx = torch.zeros((gres[0] + 1, gres[1], gres[2]), dtype=dt, device=DEV)
    wy = torch.zeros((gres[0], gres[1] + 1, gres[2]), dtype=dt, device=DEV)
    wz = torch.zeros((gres[0], gres[1], gres[2] + 1), dtype=dt, device=DEV)
    S.compute_solid_frac(gres, sc["sphi"], wx, wy, wz)
    b, x, d, r, q = (torch.zeros(gres, dtype=dt, device=DEV) for _ in range(5))
    P.initialize_solver(sc["cell_size"], gres, sc["vx"], sc["vy"], sc["vz"], sc["sphi"], sc["sv"], sc["lphi"], b, wx, wy, wz)
    lphi = sc["lphi"]
    del sc
    eng = PcgEngine(gres, dt, DEV)
    eng.setup(lphi, wx, wy, wz)
    eng.bind(b, x, d, r, q)
    form = eng.loop_info()
    assert form["fused_direction_update"] and form["deferred_x_update"] and not form["jacobi"], form   # the timed form
    eng.begin(0.0)
    eng.iterate(iters)
    eng.finish()
    torch.cuda.synchronize()
    h = eng.history()[: 2 * iters + 1]
    ref = CB.cg(gres, H(b), H(lphi), H(wx), H(wy), H(wz), 0.0, iters, 2 * iters + 1)
    assert ref["iterations"] == iters and len(h) == 2 * iters + 1
    np.testing.assert_allclose(h, ref["history"], rtol=tol)
    xr = ref["x"]
    np.testing.assert_allclose(H(x), xr, rtol=0, atol=tol * np.abs(xr).max())


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.float64, 1e-9)], ids=["f32", "f64"])
def test_viscosity_128_default_engine_vs_c_oracle(dt, tol):
    """BASELINE config 3: `ViscosityCGSolver3D` 128^3, buckling-like scene; the CG applies run the x-marching kernel"""
    require_default_engine("test_viscosity_128_default_engine_vs_c_oracle")
    import solver.ViscosityCGSolver3D as V
    gres = (128, 128, 128)
    iters = 10
    sc = scenes.viscosity_scene_3d(gres, seed=3, device=DEV)
    prec = "fp32" if dt == torch.float32 else "fp64"
    s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=prec, device=DEV)
    scale = sc["dt"] / s.cell_vol / sc["rho"]
    mu = sc["mu"]
    torch.div(sc["lvol"], s.cell_vol * 0.125, out=s.vol)
    s.x_x.copy_(sc["vx"]); s.x_y.copy_(sc["vy"]); s.x_z.copy_(sc["vz"])
    V.extrapolate(gres, 3, s.x_x, s.x_y, s.x_z, sc["sphi"])
    V.initialize_solver(gres, scale, mu, s.x_x, s.x_y, s.x_z, sc["sphi"], sc["sv"], s.vol, s.b_x, s.b_y, s.b_z)
    e = s._engine
    e.setup(scale, mu, sc["sphi"], s.vol)
    f = s._flat
    e.bind(f["b"], f["x"], f["d"], f["r"], f["q"])
    assert e.apply_kernel() == "march"
    x0, b = H(f["x"]), H(f["b"])                       # state-precision values: where both loops start
    # fp32 state stores the class samples in fp32: the oracle gets the same stored values
    vol = H(s.vol) if dt == torch.float64 else s.vol.float().double().cpu().numpy()
    e.begin(0.0)
    e.iterate(iters)
    e.finish()                                         # (a fused loop, MFS_VISC_FUSE=1, would owe the last x update)
    torch.cuda.synchronize()
    h = e.history()[: 2 * iters + 1]
    ref = CB.visc_cg(gres, scale, mu, b, x0, H(sc["sphi"]), vol, 0.0, iters, 2 * iters + 1)
    assert ref["iterations"] == iters and len(h) == 2 * iters + 1
    np.testing.assert_allclose(h, ref["history"], rtol=tol)
    for name in ("x", "r"):
        got, want = H(f[name]), ref[name]
        np.testing.assert_allclose(got, want, rtol=0, atol=tol * np.abs(want).max(), err_msg=name)
