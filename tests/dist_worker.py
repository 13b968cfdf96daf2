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
from mfs.dist import SlabCG, SlabPartition  # noqa: E402
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
