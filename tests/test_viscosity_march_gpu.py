"""The x-marching vector kernel of the viscosity CG (csrc/mfs_vcg_march.h) against the one-cell-per-lane kernel it
replaces (bit for bit: both evaluate the rows with the same vcg_row_s) and against the doubled-grid operator
(`matvecmul`, pinned by the goldens) -- on shapes that span several tiles, partial last tiles, rows shorter and
longer than a wave, and both state precisions.  GPU only."""
import os

import numpy as np
import pytest

from conftest import require_default_engine
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
    require_default_engine("test_march_is_what_runs_by_default")
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
    require_default_engine("test_march_geometries_give_the_same_bits")
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


def _torch_census(gres, vol, vec):
    """the class definition of k_vcg_classify restated on the doubled-grid volume (torch, fp of the state)"""
    Nx, Ny, Nz = gres
    nzv = Nz // vec
    # compact class arrays: class p = parity bits (x odd -> 4, y odd -> 2, z odd -> 1); index = node >> 1
    cls = {p: vol[(p >> 2) & 1::2, (p >> 1) & 1::2, p & 1::2] for p in range(1, 8)}

    def win(p, dx, dy):
        a = cls[p]
        # vectors (x, y, zv): x in [1, Nx-2], y in [1, Ny-2], cells zv*vec .. zv*vec+vec-1 (z < Nz always inside the array)
        blk = a[1 + dx:Nx - 1 + dx, 1 + dy:Ny - 1 + dy, :Nz]
        return blk.reshape(Nx - 2, Ny - 2, nzv, vec)

    parts = [win(p, 0, 0) for p in range(1, 8)] + [win(7, -1, 0), win(7, 0, -1), win(1, 1, 0), win(1, 0, 1), win(2, 1, 0), win(4, 0, 1)]
    allv = torch.cat(parts, dim=-1)
    zero = (allv == 0).all(dim=-1) & ~torch.signbit(allv).any(dim=-1)
    # ... ZERO also covers the three operands the rows take from the neighbouring vectors: C at z0-1, EXZ and EYZ at z0+vec
    pz = lambda a: (a == 0) & ~torch.signbit(a)  # noqa: E731
    c7, c2, c4 = (cls[p][1:Nx - 1, 1:Ny - 1] for p in (7, 2, 4))
    left = torch.ones_like(zero)
    left[:, :, 1:] = pz(c7[:, :, vec - 1:Nz - 1:vec])
    right = torch.ones_like(zero)
    right[:, :, :-1] = pz(c2[:, :, vec:Nz:vec]) & pz(c4[:, :, vec:Nz:vec])
    zero = zero & left & right
    bulk = vol[vol > 0].max() if bool((vol > 0).any()) else vol.new_tensor(0.0)     # k_vcg_bulk_value
    one = (allv == bulk).all(dim=-1) & ~zero
    return {"zero": int(zero.sum()), "one": int(one.sum()), "mixed": int((~zero & ~one).sum())}


@pytest.mark.parametrize("dt", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("gres", SHAPES + [(10, 14, 256), (6, 10, 512)], ids=lambda g: "x".join(map(str, g)))
def test_compressed_class_access_is_bit_identical(gres, dt):
    """round 3: the marching kernel reads the seven class arrays only for z-vectors whose samples are not uniformly 0 or 1
    (k_vcg_classify; class bits ride in the packed mask bytes).  Same bits as dense access -- q and d.q -- on the 11 shapes
    of this file plus two long-row geometries; the census of the classes equals the definition restated in torch; and the
    scene really holds all three kinds wherever the grid is big enough to have bulk liquid."""
    require_default_engine("test_compressed_class_access_is_bit_identical")
    sc = scenes.viscosity_scene_3d(gres, seed=7, device=DEV, noise=0.3)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale = sc["dt"] / cell_vol / sc["rho"]
    vol = sc["lvol"] / (cell_vol * 0.125)
    outs, dqs = [], []
    for comp in (True, False):
        eng = _engine(gres, dt, 1)
        eng.set_compress(comp)
        eng.setup(scale, 37.0, sc["sphi"], vol)
        d, dv = _direction(eng, sc, seed=11)
        vecs = [eng.new_vector()[0] for _ in range(4)]
        q = vecs[3]
        q.fill_(3.0)
        eng.bind(vecs[0], vecs[1], d, vecs[2], q)
        if eng.apply_kernel() != "march":
            pytest.skip("rows too short for the marching kernel (Nz < 2 vectors): nothing to compress")
        eng.phase_apply()
        eng.phase_reduce(0)
        torch.cuda.synchronize()
        outs.append(q.clone())
        dqs.append(float(eng.scalars[_lib.S_DQ]))
        if comp:
            census = eng.class_census()
    assert torch.equal(outs[0], outs[1]), f"compressed access changes q: max |diff| {float((outs[0] - outs[1]).abs().max())}"
    # d.q: the same products, grouped by the cost-balanced segments of the compressed launch instead of equal ones
    assert abs(dqs[0] - dqs[1]) <= 1e-13 * abs(dqs[1])
    vec = 4 if dt == torch.float32 else 2
    want = _torch_census(gres, vol.to(dt), vec)
    assert census == want, (census, want)
    assert sum(census.values()) == (gres[0] - 2) * (gres[1] - 2) * (gres[2] // vec)
    assert census["zero"] > 0 and census["mixed"] > 0
    if min(gres) >= 24 and dt == torch.float32:
        # bulk-liquid vectors: this synthetic scene's interior volumes are products of three rounded overlaps -- uniform once
        # rounded to fp32, a few ulps apart in fp64 (the notebook's clamped volumes are uniform in either)
        assert census["one"] > 0, census


@pytest.mark.parametrize("dt", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_compressed_class_access_on_hostile_volumes(dt):
    """volumes that LOOK uniform and are not: -0.0 (bit pattern differs from the constant), bulk value + 1 ulp, a single odd sample
    at each of the neighbour positions a step reads, uniform 1.0 and uniform 0.0 fields -- compressed == dense, bit for bit"""
    require_default_engine("test_compressed_class_access_on_hostile_volumes")
    gres = (12, 16, 32)
    sc = scenes.viscosity_scene_3d(gres, seed=7, device=DEV, noise=0.3)
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale = sc["dt"] / cell_vol / sc["rho"]
    base = torch.zeros_like(sc["lvol"])
    cases = []
    v = base.clone(); v[:] = 1.0; cases.append(("all one", v))
    cases.append(("all zero", base.clone()))
    v = base.clone(); v[8:16] = 1.0; v[::7, ::5, ::3] = -0.0; cases.append(("negative zeros", v))
    v = base.clone(); v[:] = 1.0; v[::5, ::7, ::9] = float(np.nextafter(dt == torch.float32 and np.float32(1) or 1.0, 2)); cases.append(("one plus ulp", v))
    g = torch.Generator(device=DEV).manual_seed(3)
    v = base.clone(); v[:] = 1.0
    idx = torch.randint(0, v.numel(), (200,), generator=g, device=DEV)
    v.view(-1)[idx] = 0.37; cases.append(("isolated odd samples in bulk", v))
    v = base.clone(); idx = torch.randint(0, v.numel(), (200,), generator=g, device=DEV)
    v.view(-1)[idx] = 0.61; cases.append(("isolated odd samples in air", v))
    for name, vol in cases:
        outs = []
        for comp in (True, False):
            eng = _engine(gres, dt, 1)
            eng.set_compress(comp)
            eng.setup(scale, 37.0, sc["sphi"], vol)
            d, dv = _direction(eng, sc, seed=11)
            vecs = [eng.new_vector()[0] for _ in range(4)]
            q = vecs[3]
            q.fill_(3.0)
            eng.bind(vecs[0], vecs[1], d, vecs[2], q)
            eng.phase_apply()
            torch.cuda.synchronize()
            outs.append(q.clone())
            if comp:
                census = eng.class_census()
        assert torch.equal(outs[0], outs[1]), (name, float((outs[0] - outs[1]).abs().max()))
        # bit for bit wherever the result is not a zero (a wave whose 64 vectors are all air skips the rows and stores +0
        # where the arithmetic 0 * d -+ 0 * d' ... would have left a zero of either sign)
        it = torch.int32 if dt == torch.float32 else torch.int64
        nz = outs[1] != 0
        assert torch.equal(outs[0].view(it)[nz], outs[1].view(it)[nz]), name
        vec = 4 if dt == torch.float32 else 2
        assert census == _torch_census(gres, vol.to(dt), vec), name


@pytest.mark.parametrize("dt", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_zero_class_covers_the_operands_taken_from_neighbouring_vectors(dt):
    """A ZERO vector's rows must be empty, not only its own samples zero: the last cell's u / v rows use EXZ / EYZ at z0+VEC and
    the first cell's w row C at z0-1, which the march takes from the NEIGHBOURING lanes' registers.  One isolated EXZ sample
    placed so that (i) it is the first sample of lane 0 of a wave and (ii) every vector of the wave before it has all its own
    samples zero (rows of 12 / 24 vectors: waves do not end on row boundaries): a classifier that looked at own samples
    only let that whole wave skip its rows and stored q = 0 for a face whose row is not empty."""
    require_default_engine("test_zero_class_covers_the_operands_taken_from_neighbouring_vectors")
    gres = (10, 12, 48)
    vec = 4 if dt == torch.float32 else 2
    nzv = gres[2] // vec
    row, zv = divmod(64, nzv)                  # vector 64 of the plane tile = lane 0 of the second wave
    x, y, z0 = 5, row + 1, zv * vec
    sc = scenes.viscosity_scene_3d(gres, seed=7, device=DEV, noise=0.3)
    sphi = torch.ones_like(sc["sphi"])         # no solids: every interior face is an unknown
    cell_vol = float(np.prod(np.array(sc["bound_size"]) / np.array(gres)))
    scale = sc["dt"] / cell_vol / sc["rho"]
    vol = torch.zeros_like(sc["lvol"])
    vol[2 * x, 2 * y + 1, 2 * z0] = 0.5        # class EXZ (x even, y odd, z even) at compact index (x, y, z0)
    outs = []
    for comp in (True, False):
        eng = _engine(gres, dt, 1)
        eng.set_compress(comp)
        eng.setup(scale, 37.0, sphi, vol)
        assert eng.apply_kernel() == "march"
        d, dv = _direction(eng, dict(sc, sphi=sphi), seed=11)
        vecs = [eng.new_vector()[0] for _ in range(4)]
        q = vecs[3]
        q.fill_(3.0)
        eng.bind(vecs[0], vecs[1], d, vecs[2], q)
        eng.phase_apply()
        torch.cuda.synchronize()
        outs.append(q.clone())
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    assert float(outs[1].abs().max()) > 0      # (the sample does reach some rows)


# ------------------------------------------------------------------ sparse lists of a single-domain solve (round 3) ----
@pytest.mark.parametrize("prec", ["fp64", "fp32"])
def test_sparse_lists_match_the_dense_loop(prec, monkeypatch):
    """round 3, single-domain solves from 2^21 unknowns: the r and d / x updates sweep the live 32-unknown chunks only and the
    loop's march launches visit only the busy (tile, plane) pairs (an all-air pair's q = +0 was stored by the solve's initial
    q = A x).  96^3 buckling-like scene (7 % liquid), then -- THROUGH THE SAME SOLVER, so that q, r, d hold the first
    solve's values where the liquid was -- the scene mirrored in x; against MFS_VISC_SPARSE=0: same iteration count (+-1),
    same history to rounding (the dot products group differently), same velocities."""
    require_default_engine("test_sparse_lists_match_the_dense_loop")
    import solver.ViscosityCGSolver3D as V
    gres = (96, 96, 96)
    sc = scenes.viscosity_scene_3d(gres, seed=5, device=DEV)
    flip = lambda t: t.flip(0).contiguous()  # noqa: E731
    runs = {}
    for sparse in ("1", "0"):
        monkeypatch.setenv("MFS_VISC_SPARSE", sparse)
        s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=prec, device=DEV)
        res = []
        for mirrored in (False, True):
            m = flip if mirrored else (lambda t: t)
            v = [m(sc["vx"]).clone() * (-1.0 if mirrored else 1.0), m(sc["vy"]).clone(), m(sc["vz"]).clone()]
            s.solve(sc["dt"], sc["mu"], sc["rho"], *v, m(sc["sphi"]), m(sc["sv"]), m(sc["lphi"]), m(sc["lvol"]))
            torch.cuda.synchronize()
            info = s._engine.sparse_info()
            assert (info["listed_pairs"] > 0 and 0 < info["live_chunks"] < info["chunks"] // 4) == (sparse == "1"), info
            res.append((s.iterations, np.asarray(s.history), [t.clone() for t in v]))
        runs[sparse] = res
    for (it_s, h_s, v_s), (it_d, h_d, v_d) in zip(runs["1"], runs["0"]):
        assert abs(it_s - it_d) <= 1, (it_s, it_d)
        n = min(len(h_s), len(h_d), 41)
        np.testing.assert_allclose(h_s[:n], h_d[:n], rtol=1e-9 if prec == "fp64" else 2e-4)
        for a, b in zip(v_s, v_d):
            assert float((a - b).abs().max()) <= (1e-8 if prec == "fp64" else 1e-3) * float(b.abs().max())
    # the mirrored solve is the mirror image of the first one (same operator, mirrored): its iteration count agrees
    assert abs(runs["1"][0][0] - runs["1"][1][0]) <= 2


@pytest.mark.parametrize("prec", ["fp64", "fp32"])
def test_sparse_lists_in_the_jacobi_loop(prec, monkeypatch):
    """the opt-in Jacobi loop takes the solve's lists as well (z = r / diag is 0 wherever r is; the stored z is cleared when the
    lists are built): 96^3 buckling-like scene, then the scene mirrored in x through the same solver, against MFS_VISC_SPARSE=0 --
    same iteration count (+-1), same history to rounding, same velocities."""
    require_default_engine("test_sparse_lists_in_the_jacobi_loop")
    import solver.ViscosityCGSolver3D as V
    gres = (96, 96, 96)
    sc = scenes.viscosity_scene_3d(gres, seed=5, device=DEV)
    flip = lambda t: t.flip(0).contiguous()  # noqa: E731
    runs = {}
    for sparse in ("1", "0"):
        monkeypatch.setenv("MFS_VISC_SPARSE", sparse)
        s = V.ViscosityCGSolver3D(gres, sc["bound_size"], precision=prec, device=DEV, jacobi=True)
        res = []
        for mirrored in (False, True):
            m = flip if mirrored else (lambda t: t)
            v = [m(sc["vx"]).clone() * (-1.0 if mirrored else 1.0), m(sc["vy"]).clone(), m(sc["vz"]).clone()]
            s.solve(sc["dt"], sc["mu"], sc["rho"], *v, m(sc["sphi"]), m(sc["sv"]), m(sc["lphi"]), m(sc["lvol"]))
            torch.cuda.synchronize()
            assert s._engine.loop_info()["jacobi"]
            info = s._engine.sparse_info()
            assert (info["listed_pairs"] > 0 and 0 < info["live_chunks"] < info["chunks"] // 4) == (sparse == "1"), info
            res.append((s.iterations, np.asarray(s.history), [t.clone() for t in v]))
        runs[sparse] = res
    for (it_s, h_s, v_s), (it_d, h_d, v_d) in zip(runs["1"], runs["0"]):
        assert abs(it_s - it_d) <= 1, (it_s, it_d)
        n = min(len(h_s), len(h_d), 41)
        np.testing.assert_allclose(h_s[:n], h_d[:n], rtol=1e-9 if prec == "fp64" else 2e-4)
        for a, b in zip(v_s, v_d):
            assert float((a - b).abs().max()) <= (1e-8 if prec == "fp64" else 1e-3) * float(b.abs().max())
