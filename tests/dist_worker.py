"""Worker side of the multi-process CPU tests of the slab-decomposed CG driver
(mfs/dist.py).  TEST INFRASTRUCTURE: on the GPU the driver's `ops` is the HIP
engine (mfs.pcg.PcgEngine); here a numpy stand-in with the same phase interface
is injected so that the partition, halo-exchange and all-reduce logic runs over
`gloo` without a GPU.  The stand-in computes with the oracle -- it is the checker
of the distributed logic, never a product path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "python-fluid-simulation_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from mfs import _lib  # noqa: E402
from mfs.dist import SlabCG, SlabPartition, SlabVCG  # noqa: E402
from oracle import mfs_oracle as O  # noqa: E402

S = _lib


class OracleSlabOps:
    """Same phase API and scalar-slot semantics as csrc/mfs_pcg.hip, on CPU tensors."""

    def __init__(self, lgres, lphi, wx, wy, wz, b):
        self.g = tuple(lgres)
        self.lphi, self.wx, self.wy, self.wz = lphi, wx, wy, wz
        self.scalars = torch.zeros(S.NSCALARS, dtype=torch.float64)
        t = lambda: torch.zeros(self.g, dtype=torch.float64)  # noqa: E731
        self.b, self.x, self.d, self.r, self.q = torch.as_tensor(b).clone(), t(), t(), t(), t()
        self.hist = []
        self._pdq = 0.0

    def _done(self):
        return self.scalars[S.S_DONE].item() != 0.0

    def _apply(self, v, out, xb, xe):
        xb, xe = max(xb, 1), min(xe, self.g[0] - 1)
        if xe <= xb:
            return
        sl = slice(xb - 1, xe + 1)
        sub = (xe - xb + 2, self.g[1], self.g[2])
        O.pressure_apply3d(sub, v.numpy()[sl], out.numpy()[sl], self.wx[xb - 1:xe + 2], self.wy[sl], self.wz[sl],
                           self.lphi[sl])

    def begin_local(self, tol):
        self.scalars.zero_()
        self.scalars[S.S_TOL2] = tol * tol
        self.x *= 0.0
        self._apply(self.x, self.q, 1, self.g[0] - 1)
        self.d.copy_(self.b - self.q)
        self.r.copy_(self.d)
        self.scalars[S.S_RR] = float((self.r * self.r).sum())

    def begin_finish(self):
        rr = self.scalars[S.S_RR].item()
        self.scalars[S.S_DELTA] = rr
        self.scalars[S.S_LASTRR] = rr
        self.hist = [rr]
        if rr < self.scalars[S.S_TOL2].item():
            self.scalars[S.S_DONE] = 1.0

    def phase_apply(self, xb, xe, first):
        if self._done():
            return
        if first:
            self._pdq = 0.0
        xb, xe = max(xb, 1), min(xe, self.g[0] - 1)
        self._apply(self.d, self.q, xb, xe)
        self._pdq += float((self.d[xb:xe] * self.q[xb:xe]).sum())

    def phase_reduce(self, which):
        if self._done():
            return
        if which == 0:
            self.scalars[S.S_DQ] = self._pdq
            self.scalars[S.S_DELTA] = self.scalars[S.S_RR].item()
        else:
            self.scalars[S.S_RR] = float((self.r * self.r).sum())

    def phase_update_xr(self):
        if self._done():
            return
        alpha = self.scalars[S.S_DELTA].item() / self.scalars[S.S_DQ].item()
        self.x += alpha * self.d
        self.r -= alpha * self.q

    def phase_update_r(self):
        if self._done():
            return
        alpha = self.scalars[S.S_DELTA].item() / self.scalars[S.S_DQ].item()
        self.r -= alpha * self.q

    def phase_update_x(self):
        if self._done():
            return
        alpha = self.scalars[S.S_DELTA].item() / self.scalars[S.S_DQ].item()
        self.x += alpha * self.d

    def phase_update_d(self):
        if self._done():
            return
        rr, delta = self.scalars[S.S_RR].item(), self.scalars[S.S_DELTA].item()
        dq = self.scalars[S.S_DQ].item()
        self.hist += [dq, rr]
        self.scalars[S.S_ITERS] += 1
        self.scalars[S.S_LASTRR] = rr
        self.scalars[S.S_ALPHA] = delta / dq
        if rr < self.scalars[S.S_TOL2].item():
            self.scalars[S.S_DONE] = 1.0
            return
        beta = rr / delta
        self.scalars[S.S_BETA] = beta
        self.d.copy_(self.r + beta * self.d)

    def iterate(self, n):
        for _ in range(n):
            self.phase_apply(1, self.g[0] - 1, True)
            self.phase_reduce(0)
            self.phase_update_xr()
            self.phase_reduce(1)
            self.phase_update_d()


class OracleSlabVOps:
    """Viscosity: same phase API and scalar-slot semantics as csrc/mfs_visc.hip (mfs_vcg3d_phase_*), on CPU tensors."""

    def __init__(self, lgres, scale, mu, sphi, vol, b, x):
        self.g = tuple(lgres)
        self.scale, self.mu, self.sphi, self.vol = scale, mu, sphi, vol
        self.scalars = torch.zeros(S.NSCALARS, dtype=torch.float64)
        L, Ny, Nz = self.g
        shp = [(L + 1, Ny, Nz), (L, Ny + 1, Nz), (L, Ny, Nz + 1)]
        z = lambda: [torch.zeros(s, dtype=torch.float64) for s in shp]  # noqa: E731
        self.b = [torch.as_tensor(a).clone() for a in b]
        self.x = [torch.as_tensor(a).clone() for a in x]
        self.d, self.r, self.q = z(), z(), z()
        self.hist = []
        self.skip_top_x = False

    def set_slab(self, skip_top_x):
        self.skip_top_x = bool(skip_top_x)

    def _done(self):
        return self.scalars[S.S_DONE].item() != 0.0

    def _apply(self, v, out):
        O.visc_apply3d(self.g, self.scale, self.mu, *[t.numpy() for t in v], *[t.numpy() for t in out], self.sphi, self.vol)
        if self.skip_top_x:
            out[0][self.g[0] - 1].zero_()      # the engine does not compute the neighbour's u plane at all

    def _dot(self, a, b):
        return float(sum((u * w).sum() for u, w in zip(a, b)))

    def begin_local(self, tol):
        self.scalars.zero_()
        self.scalars[S.S_TOL2] = tol * tol
        self._apply(self.x, self.q)
        for d, r, b, q in zip(self.d, self.r, self.b, self.q):
            d.copy_(b - q)
            r.copy_(d)
        self.scalars[S.S_RR] = self._dot(self.r, self.r)

    begin_finish = OracleSlabOps.begin_finish

    def phase_apply(self):
        if not self._done():
            self._apply(self.d, self.q)

    def phase_reduce(self, which):
        if self._done():
            return
        if which == 0:
            self.scalars[S.S_DQ] = self._dot(self.d, self.q)
            self.scalars[S.S_DELTA] = self.scalars[S.S_RR].item()
        else:
            self.scalars[S.S_RR] = self._dot(self.r, self.r)

    def phase_update_xr(self):
        if self._done():
            return
        alpha = self.scalars[S.S_DELTA].item() / self.scalars[S.S_DQ].item()
        for x, d, r, q in zip(self.x, self.d, self.r, self.q):
            x += alpha * d
            r -= alpha * q

    def phase_update_d(self):
        if self._done():
            return
        rr, delta = self.scalars[S.S_RR].item(), self.scalars[S.S_DELTA].item()
        dq = self.scalars[S.S_DQ].item()
        self.hist += [dq, rr]
        self.scalars[S.S_ITERS] += 1
        self.scalars[S.S_LASTRR] = rr
        self.scalars[S.S_ALPHA] = delta / dq
        if rr < self.scalars[S.S_TOL2].item():
            self.scalars[S.S_DONE] = 1.0
            return
        beta = rr / delta
        self.scalars[S.S_BETA] = beta
        for d, r in zip(self.d, self.r):
            d.copy_(r + beta * d)

    def poll(self):
        return dict(done=self._done(), iterations=int(self.scalars[S.S_ITERS].item()))


def run_viscosity(rank, world, port, path, tol, max_iter):
    """one rank of the slab viscosity solve on a golden scene: extrapolation with per-sweep ghost exchange, RHS, CG"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with np.load(path, allow_pickle=False) as z:
            g = {k: z[k] for k in z.files}
        gres = tuple(int(v) for v in g["gres"])
        part = SlabPartition(gres[0], world, rank)
        lo, hi = part.local_range
        L = hi - lo
        lg = (L, gres[1], gres[2])
        cell_vol = float(np.prod(g["bound_size"] / np.array(gres, dtype=np.float64)))
        scale = float(g["dt"]) / cell_vol / float(g["rho"])
        mu = float(g["mu"])
        sphi = g["sphi"][2 * lo:2 * hi + 1]
        vol = g["lvol"][2 * lo:2 * hi + 1] / (cell_vol * 0.125)
        x = [np.array(g["in_vx"][lo:hi + 1], dtype=np.float64), np.array(g["in_vy"][lo:hi], dtype=np.float64),
             np.array(g["in_vz"][lo:hi], dtype=np.float64)]
        zeros = lambda: [np.zeros_like(a) for a in x]  # noqa: E731
        ops = OracleSlabVOps(lg, scale, mu, sphi, vol, zeros(), zeros())
        cg = SlabVCG(ops, part, ops.d, dist if world > 1 else None)
        # extrapolation: one sweep at a time (the oracle's num_iter = 1 on explicit validity is not exposed, so the
        # sweep is restated here from its definition, solver/ViscosityCGSolver3D.py:8-39), ghosts after every sweep
        valids = [sphi[0::2, 1::2, 1::2] >= 0, sphi[1::2, 0::2, 1::2] >= 0, sphi[1::2, 1::2, 0::2] >= 0]
        for c in range(3):
            v, valid = torch.as_tensor(x[c]), torch.as_tensor(valids[c].astype(np.uint8))
            for _ in range(3):
                vn, mn = v.clone(), valid.clone()
                va, ma = v.numpy(), valid.numpy().astype(bool)
                I = (slice(1, -1),) * 3
                cnt = np.zeros(va[I].shape)
                acc = np.zeros(va[I].shape)
                for ax in range(3):
                    for sh in (slice(2, None), slice(0, -2)):
                        J = tuple(sh if a == ax else slice(1, -1) for a in range(3))
                        acc += np.where(ma[J], va[J], 0.0)
                        cnt += ma[J]
                upd = (~ma[I]) & (cnt > 0)
                vn.numpy()[I] = np.where(upd, acc / np.maximum(cnt, 1), va[I])
                mn.numpy()[I] = (ma[I] | upd).astype(np.uint8)
                cg.exchange([vn, mn])
                v, valid = vn, mn
            x[c] = v.numpy()
        b = zeros()
        O.visc_rhs3d(lg, scale, mu, x[0], x[1], x[2], sphi, None, vol, b[0], b[1], b[2])
        if part.right is not None:
            b[0][L - 1] = 0.0
        for a in b[1:]:
            a[0] = 0.0
            a[L - 1] = 0.0
        for dst, src in zip(ops.b, b):
            dst.copy_(torch.as_tensor(src))
        for dst, src in zip(ops.x, x):
            dst.copy_(torch.as_tensor(src))
        ok, it = cg.solve(tol, max_iter, check_every=4)
        np.savez(f"{path}.rank{rank}.npz", lo=lo, hi=hi, iters=it, done=int(ok), hist=np.array(ops.hist),
                 **{f"{n}_{c}": getattr(ops, n)[i].numpy() for n in "xbqr" for i, c in enumerate("xyz")},
                 **{f"e_{c}": x[i] for i, c in enumerate("xyz")})
    finally:
        dist.destroy_process_group()


def local_problem(gl, part):
    """slice the global scene arrays down to this rank's slab (planes [lo, hi))."""
    lo, hi = part.local_range
    ny, nz = gl["gres"][1], gl["gres"][2]
    lg = (hi - lo, ny, nz)
    loc = dict(gres=lg, lphi=gl["lphi"][lo:hi], wx=gl["wx"][lo:hi + 1], wy=gl["wy"][lo:hi], wz=gl["wz"][lo:hi],
               vx=gl["vx"][lo:hi + 1], vy=gl["vy"][lo:hi], vz=gl["vz"][lo:hi],
               sphi=gl["sphi"][2 * lo:2 * hi + 1], sv=gl["sv"][2 * lo:2 * hi + 1])
    b = np.zeros(lg)
    O.pressure_rhs3d(gl["cell_size"], lg, loc["vx"], loc["vy"], loc["vz"], loc["sphi"], loc["sv"], loc["lphi"], b,
                     loc["wx"], loc["wy"], loc["wz"])
    loc["b"] = b
    return loc


def run_lost_peer(rank, world, port, path, timeout_s):
    """fault injection for the COLLECTIVE loop: rank 1 joins the group and then never takes part in the solve.  Rank 0
    must get MfsTimeout (status MFS_E_TIMEOUT) out of begin() / iterate() within the bound -- not a blocked process."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MFS_COLLECTIVE_TIMEOUT_S"] = str(timeout_s)
    from mfs.dist import pg_timeout
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout())
    with np.load(path, allow_pickle=False) as z:
        gl = {k: z[k] for k in z.files}
    gl["gres"] = tuple(int(v) for v in gl["gres"])
    part = SlabPartition(gl["gres"][0], world, rank)
    if rank == 0:
        loc = local_problem(gl, part)
        ops = OracleSlabOps(loc["gres"], loc["lphi"], loc["wx"], loc["wy"], loc["wz"], loc["b"])
        cg = SlabCG(ops, part, ops.d, dist)
        t0 = time.perf_counter()
        try:
            cg.begin(1e-9)
            cg.iterate(4)
            outcome = "no error"
        except _lib.MfsError as exc:
            outcome = f"{type(exc).__name__}: {exc}"
        with open(f"{path}.rank0.txt", "w") as f:
            f.write(f"{time.perf_counter() - t0:.3f}\n{outcome}\n")
    else:
        time.sleep(float(timeout_s) + 3.0)      # alive (the connection stays open), but never in the solve
    os._exit(0)                                 # no collective teardown with a peer that has given up


def run_bands(rank, world, port, path, nx):
    """SlabBands (the particle-sharded time step's plane-band exchanges and particle migration) over gloo on CPU tensors:
    every rank fills a global-shaped field only from 'its own particles', reduces, fetches ghosts, migrates; the results
    are checked against the single-process answer by the test."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from mfs.dist import SlabBands, pg_timeout
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout())
    try:
        B = SlabBands(dist, None, nx)
        rng = np.random.default_rng(5)
        P = 4000
        cell = rng.integers(0, nx, size=P)                        # every rank draws the SAME particle set
        wgt = rng.standard_normal(P)
        ids = np.arange(P)
        mine = np.asarray(B.owner_of_cells(torch.as_tensor(cell)).numpy() == rank)
        out = {}
        for kind, n, scale in (("cell", nx, 1), ("xface", nx + 1, 1), ("doubled", 2 * nx + 1, 2)):
            for op in ("sum", "min"):
                f = torch.zeros((n, 3, 2), dtype=torch.float64) if op == "sum" else torch.full((n, 3, 2), 9.0, dtype=torch.float64)
                for c, w in zip(cell[mine], wgt[mine]):          # a scatter with reach 2 cells in x
                    for dxx in range(-2 * scale, 2 * scale + 1):
                        pl = min(max(c * scale + dxx, 0), n - 1)
                        if op == "sum":
                            f[pl] += w
                        else:
                            f[pl] = torch.minimum(f[pl], torch.full((3, 2), float(w), dtype=torch.float64))
                B.reduce([f], kind, 3, op)
                a, b = B.owned(kind)
                owned = f[a:b].clone()
                f2 = f.clone()
                f2[:a] = -777.0                                    # whatever sits outside the range must be replaced by ghosts
                f2[b:] = -777.0
                B.ghosts([f2], kind, 4)
                out[f"{kind}_{op}_owned"] = owned.numpy()
                out[f"{kind}_{op}_ghosted"] = f2.numpy()
                out[f"{kind}_own"] = np.array([a, b])
        # migration: every particle moves to a random new cell; fields travel with it
        newcell = np.random.default_rng(6).integers(0, nx, size=P)
        fields = [torch.as_tensor(wgt[mine]), torch.as_tensor(np.stack([wgt[mine], 2 * wgt[mine], 3 * wgt[mine]], 1)),
                  torch.as_tensor(ids[mine])]
        dest = B.owner_of_cells(torch.as_tensor(newcell[mine]))
        w2, v2, id2 = B.migrate(fields, dest)
        out["mig_ids"], out["mig_w"], out["mig_v"] = id2.numpy(), w2.numpy(), v2.numpy()
        out["mig_expected_owner"] = B.owner_of_cells(torch.as_tensor(newcell)).numpy()
        np.savez(f"{path}.rank{rank}.npz", cell=cell, wgt=wgt, bytes_moved=B.bytes_moved, **out)
    finally:
        dist.destroy_process_group()


def run(rank, world, port, path, tol, overlap, max_iter):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with np.load(path, allow_pickle=False) as z:
            gl = {k: z[k] for k in z.files}
        gl["gres"] = tuple(int(v) for v in gl["gres"])
        part = SlabPartition(gl["gres"][0], world, rank)
        loc = local_problem(gl, part)
        ops = OracleSlabOps(loc["gres"], loc["lphi"], loc["wx"], loc["wy"], loc["wz"], loc["b"])
        cg = SlabCG(ops, part, ops.d, dist if world > 1 else None, overlap=overlap)
        cg.begin(tol)
        it = 0
        while not ops._done() and it < max_iter:
            cg.iterate(1)
            it += 1
        cg.exchange(ops.x)     # not needed for correctness of owned planes; exercises the helper
        np.savez(f"{path}.rank{rank}.npz", x=ops.x.numpy(), b=loc["b"], hist=np.array(ops.hist),
                 iters=int(ops.scalars[S.S_ITERS].item()), done=int(ops._done()), lo=part.local_range[0],
                 hi=part.local_range[1], q=ops.q.numpy(), r=ops.r.numpy())
    finally:
        dist.destroy_process_group()
