"""The x-marching vector kernel of the viscosity CG (csrc/mfs_vcg_march.h) against the one-cell-per-lane kernel it
replaces (bit for bit: both evaluate the rows with the same vcg_row_s) and against the doubled-grid operator
(`matvecmul`, pinned by the goldens) -- on shapes that span several tiles, partial last tiles, rows shorter and
longer than a wave, and both state precisions.  GPU only."""
import os

import numpy as np
import pytest
import torch

from mfs import _lib, scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(gres, dt, march):
    from mfs.vcg import VcgEngine
    old = os.environ.get("MFS_VISC_MARCH")
    os.environ["MFS_VISC_MARCH"] = "1" if march else "0"
    try:
        return VcgEngine(gres, dt, DEV)
    finally:
        if old is None:
            os.environ.pop("MFS_VISC_MARCH", None)
        else:
            os.environ["MFS_VISC_MARCH"] = old


def _direction(eng, sc, seed):
    """a CG direction vector: random on non-solid interior faces, exactly 0 elsewhere (what `d` is in the loop)"""
    gen = torch.Generator(device=DEV).manual_seed(seed)
    d, dv = eng.new_vector()
    valid = [sc["sphi"][0::2, 1::2, 1::2] >= 0, sc["sphi"][1::2, 0::2, 1::2] >= 0, sc["sphi"][1::2, 1::2, 0::2] >= 0]
    for t, m in zip(dv, valid):
        t[1:-1, 1:-1, 1:-1] = torch.randn(t[1:-1, 1:-1, 1:-1].shape, generator=gen, device=DEV, dtype=torch.float64).to(t.dtype)
        t.mul_(m)
    return d, dv


SHAPES = [(12, 12, 12), (24, 24, 24), (20, 24, 36), (40, 36, 32), (9, 70, 16), (16, 20, 64), (7, 5, 128), (33, 17, 8),
          (64, 64, 64), (5, 300, 4), (48, 80, 48)]


@pytest.mark.parametrize("dt", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("gres", SHAPES, ids=lambda g: "x".join(map(str, g)))
def test_march_equals_scalar_kernel_and_reference_operator(gres, dt):
    import solver.ViscosityCGSolver3D as V
    sc = scenes.viscosity_scene_3d(gres, seed=7, device=DEV, noise=0.3)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale = sc["dt"] / cell_vol / sc["rho"]
    vol = sc["lvol"] / (cell_vol * 0.125)
    mu = 37.0
    outs, dqs = [], []
    for march in (1, 0):
        eng = _engine(gres, dt, march)
        eng.setup(scale, mu, sc["sphi"], vol)
        d, dv = _direction(eng, sc, seed=11)
        vecs = [eng.new_vector()[0] for _ in range(4)]
        q = vecs[3]
        q.fill_(3.0)
        eng.bind(vecs[0], vecs[1], d, vecs[2], q)
        eng.phase_apply()
        eng.phase_reduce(0)
        torch.cuda.synchronize()
        outs.append(q.clone())
        dqs.append(float(eng.scalars[_lib.S_DQ]))
        want_dq = float((d.double() * torch.where(q == 3.0, torch.zeros_like(q), q).double()).sum())
        assert abs(dqs[-1] - want_dq) <= 1e-11 * max(abs(want_dq), 1e-300), (march, dqs[-1], want_dq)
    assert torch.equal(outs[0], outs[1]), f"march kernel differs from the scalar kernel: max |diff| {float((outs[0] - outs[1]).abs().max())}"
    assert abs(dqs[0] - dqs[1]) <= 1e-12 * abs(dqs[1])
    # ... and the doubled-grid operator of the drop-in module (fp64, goldens) on the same operand
    eng = _engine(gres, dt, 1)
    d, dv = _direction(eng, sc, seed=11)
    ref = [torch.full(t.shape, 3.0, dtype=torch.float64, device=DEV) for t in dv]
    V.matvecmul(gres, scale, mu, *[t.double() for t in dv], *ref, sc["sphi"], vol)
    _, qv = eng.new_vector()
    o = 0
    for t, rf in zip(qv, ref):
        n = t.numel()
        got = outs[0][o:o + n].view(t.shape).double()
        o += n
        tol = 1e-12 if dt == torch.float64 else 3e-7
        assert torch.allclose(got, rf, rtol=0, atol=tol * float(rf.abs().max()))


def test_march_is_what_runs_by_default():
    """the default engine takes the marching kernel for CG applies on an aligned grid (and says so)"""
    from mfs.vcg import VcgEngine
    eng = VcgEngine((16, 16, 16), torch.float32, DEV)
    assert eng.apply_kernel() == "march"
    eng = VcgEngine((16, 16, 14), torch.float32, DEV)      # Nz % 4 != 0: one cell per lane
    assert eng.apply_kernel() == "scalar"


@pytest.mark.parametrize("dt", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("gres", [(12, 12, 12), (20, 24, 36), (9, 70, 16), (7, 5, 128), (10, 14, 256), (48, 80, 48), (6, 10, 512)],
                         ids=lambda g: "x".join(map(str, g)))
def test_march_geometries_give_the_same_bits(gres, dt, monkeypatch):
    """every geometry of the marching kernel -- tiles of 256 or 512 z-vectors (512: one workgroup of eight waves per CU, what
    long rows take), ring of 4 plane slots or 3 (a second barrier per plane: fp64 rows up to Nz = 512) -- gives the bits of
    the one-cell-per-lane kernel"""
    sc = scenes.viscosity_scene_3d(gres, seed=7, device=DEV, noise=0.3)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale = sc["dt"] / cell_vol / sc["rho"]
    vol = sc["lvol"] / (cell_vol * 0.125)
    outs, dqs, kinds = [], [], []
    for blk in ("scalar", "256", "512", "5123", "auto"):
        if blk in ("scalar", "auto"):
            monkeypatch.delenv("MFS_VISC_MARCH_BLOCK", raising=False)
        else:
            monkeypatch.setenv("MFS_VISC_MARCH_BLOCK", blk)
        eng = _engine(gres, dt, 0 if blk == "scalar" else 1)
        eng.setup(scale, 37.0, sc["sphi"], vol)
        d, dv = _direction(eng, sc, seed=11)
        vecs = [eng.new_vector()[0] for _ in range(4)]
        q = vecs[3]
        q.fill_(3.0)
        eng.bind(vecs[0], vecs[1], d, vecs[2], q)
        kinds.append(eng.apply_kernel())
        eng.phase_apply()
        eng.phase_reduce(0)
        torch.cuda.synchronize()
        outs.append(q.clone())
        dqs.append(float(eng.scalars[_lib.S_DQ]))
    assert kinds[0] == "scalar" and kinds[1:] == ["march"] * 4, kinds      # (a forced geometry that does not fit falls back to the automatic one)
    for o, dq in zip(outs[1:], dqs[1:]):
        assert torch.equal(o, outs[0]), f"max |diff| {float((o - outs[0]).abs().max())}"
        assert abs(dq - dqs[0]) <= 1e-12 * abs(dqs[0])
