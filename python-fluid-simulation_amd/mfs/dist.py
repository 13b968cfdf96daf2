"""Slab-decomposed multi-GPU CG driver: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

New design -- the reference is single-GPU.  The grid is cut
into contiguous slabs along axis 0 (the slowest-varying array axis, so a halo
plane is one contiguous Ny*Nz block and needs no pack kernel; `north_star`
calls the slab axis "z").

Decomposition.  The reference operator never writes boundary cells
(solver/PressureCGSolver3D.py:55-57): only planes 1..Nx-2 are computed and planes
0 and Nx-1 are read-only neighbours.  The COMPUTED planes [1, Nx-1) are split
into `world` contiguous ranges [a_p, a_{p+1}); rank p holds local arrays over
global planes [a_p - 1, a_{p+1} + 1), i.e. its range plus one plane each side.
Local plane 0 / L-1 is then either the true global boundary plane (ranks 0 and
world-1) or a ghost copy of the neighbour's edge plane -- and in both cases the
single-domain kernels treat it exactly right by skipping it as "boundary".  So
every rank runs the UNMODIFIED single-GPU kernels on its local array; multi-GPU
adds only
  * per iteration: exchange of the two edge planes of `d` with the neighbours
    (interior planes are applied while the planes are in flight), and
  * two scalar all-reduces (d.q and r.r).
Two transports: "p2p" (default when a window passes its self-test) -- the loop runs
natively in the library and the planes / dot products move as xGMI stores issued by
the solver's kernels into HIP-IPC-mapped windows (mfs/p2p.py, csrc/mfs_pcg_slab.h);
"rccl" -- the phase-by-phase loop below with torch.distributed collectives (RCCL on
the GPUs, gloo in the CPU tests), which is also the fallback.
Ghost planes of b, r, q stay exactly 0 (no kernel writes them), so local dot
products never double count; ghost planes of x accumulate alpha*d_ghost, which is
the neighbour's own update, so the final velocity update finds p[x-1] in place.
"""
from __future__ import annotations

import datetime
import os
import time
import warnings

from . import _lib


def collective_timeout_s():
    """bound on every wait of the collective ("rccl") loops: MFS_COLLECTIVE_TIMEOUT_S, default 60 s"""
    return float(os.environ.get("MFS_COLLECTIVE_TIMEOUT_S", "60"))


def coll_device(dist, group, device):
    """where the operands of a small host-side collective (counts, a CFL maximum, gathered particle tables) must live:
    an "nccl" (RCCL) group has no CPU backend -- `dist.all_reduce(cpu_tensor)` raises "No backend type associated with
    device type cpu" -- so they go to the rank's GPU and come back with .cpu(); every other backend (gloo: CPU tests,
    one-GPU rehearsals) takes host tensors"""
    import torch
    if dist.get_backend(group) == "nccl":
        return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def pg_timeout():
    """`timeout=` for init_process_group in the launchers (bench.py, tools, test workers): the backend's own bound
    (gloo: every blocking call; nccl: the watchdog that aborts a communicator whose collective never completes)"""
    return datetime.timedelta(seconds=max(30.0, 2.0 * collective_timeout_s()))


class _BoundedCollectives:
    """Bounded waits for the collective loops of SlabCG / SlabVCG.  A lost peer must surface as MfsTimeout (status
    MFS_E_TIMEOUT, like the window loop's bounded spins), never as a process blocked in a collective:
      * host-blocking backends (gloo: tests, rehearsals): every collective runs async_op=True and is awaited with
        work.wait(timeout);
      * "nccl" (RCCL): a wait only orders the compute stream behind the collective, the host runs ahead -- so the
        bound sits where the host does block, at the poll: the stream is drained with an event polled against a
        deadline (`drain`), and the group's own timeout (pg_timeout) lets the watchdog abort the stuck communicator."""

    def _bc_init(self):
        self.timeout_s = collective_timeout_s()

    def _blocking_backend(self):
        return self.dist.get_backend(self.group) != "nccl"

    def _timed_out(self, what, exc=None):
        msg = (f"{what} failed with status {_lib.MFS_E_TIMEOUT}: collective wait timed out after {self.timeout_s:g} s "
               f"on rank {self.part.rank} of {self.part.world} (a peer rank is lost or stuck; MFS_COLLECTIVE_TIMEOUT_S)")
        if exc is not None:
            msg += f" [{type(exc).__name__}: {str(exc)[:200]}]"
        return _lib.MfsTimeout(msg)

    def _wait(self, work, what):
        if work is None:
            return
        if not self._blocking_backend():
            work.wait()                       # stream-ordered; bounded at drain()
            return
        try:
            ok = work.wait(timeout=datetime.timedelta(seconds=self.timeout_s))
        except RuntimeError as exc:           # gloo raises on timeout / on a peer that closed its connection
            raise self._timed_out(what, exc) from None
        if ok is False:
            raise self._timed_out(what)

    def _allreduce_bounded(self, t, what):
        self._wait(self.dist.all_reduce(t, group=self.group, async_op=True), what)

    def drain(self, what="CG loop"):
        """wait, bounded, until this rank's stream has executed everything enqueued so far"""
        sc = getattr(self.ops, "scalars", None)
        if sc is None or not getattr(sc, "is_cuda", False):
            return
        import torch
        ev = torch.cuda.Event()
        ev.record()
        deadline = time.monotonic() + self.timeout_s
        while not ev.query():
            if time.monotonic() > deadline:
                raise self._timed_out(what)
            time.sleep(2e-4)



class SlabPartition:
    def __init__(self, nx_global, world, rank):
        nx_global, world, rank = int(nx_global), int(world), int(rank)
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        if nx_global - 2 < world:
            raise ValueError(f"{nx_global} planes cannot be cut into {world} slabs")
        self.nx_global, self.world, self.rank = nx_global, world, rank
        cut = lambda p: 1 + (nx_global - 2) * p // world  # noqa: E731
        self.owned = (cut(rank), cut(rank + 1))                    # computed planes [a, b)
        self.local_range = (self.owned[0] - 1, self.owned[1] + 1)  # planes held locally
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None

    @property
    def local_planes(self):
        return self.local_range[1] - self.local_range[0]

    def global_cells(self, ggrid):
        return int(ggrid[0]) * int(ggrid[1]) * int(ggrid[2])


class SlabCG(_BoundedCollectives):
    """CG iterations over one slab.  `ops` is the engine (mfs.pcg.PcgEngine on the
    GPU): begin_local/begin_finish/phase_*/iterate plus a `scalars` tensor that the
    all-reduces act on.  `d` is the local direction vector (planes 0 and L-1 are
    the ghost / boundary planes).  `dist` is torch.distributed or None (1 rank)."""

    def __init__(self, ops, part, d, dist=None, group=None, overlap=True, force_multi=False, window=None, rccl=None):
        """window: a connected mfs.p2p.P2PWindow -> the native window loop ("p2p").  rccl: an mfs.rccl.RcclComm -> the
        native COLLECTIVE loop ("rccl", round 3): the window loop's launches, halo planes and dot products through RCCL
        between them, enqueued from C.  Neither: the phase-by-phase loop below with torch.distributed collectives."""
        self.ops, self.part, self.d, self.dist, self.group = ops, part, d, dist, group
        self.rccl = rccl
        if rccl is not None and window is None:
            from .rccl import LocalWindow
            window = LocalWindow(int(d[0].numel()) * d.element_size(), d.device)
        self.overlap = overlap
        self.L = int(d.shape[0])
        if self.L != part.local_planes:
            raise ValueError("d does not match the partition's local plane count")
        # force_multi: take the phase-by-phase path (with its collectives) even on one rank (tests)
        self.multi = dist is not None and (part.world > 1 or force_multi)
        # window: a connected, self-tested mfs.p2p.P2PWindow -> the loop runs natively inside the library,
        # halo planes and dot products moving as xGMI stores between the solver's own kernels
        # (csrc/mfs_pcg_slab.h); without one, the phase-by-phase loop below with RCCL collectives
        self.window = window if (window is not None and window.ok) else None
        self._attach()
        self._bc_init()
        self._p2p_active = None          # which loop the last begin() / solve() took (None: none yet)
        self.downgraded = ""             # why a given window is NOT being used (empty: it is, or none was given)

    def _attach(self):
        """point the engine at THIS driver's transport (several drivers may share one engine: bench.py's cross-checks)"""
        if self.window is None:
            return
        if self.rccl is not None:
            self.ops.attach_rccl(self.window, self.rccl)
        else:
            if hasattr(self.ops, "attach_rccl") and getattr(self.ops, "_rccl", None) is not None:
                self.ops.attach_rccl(None, None)
            self.ops.attach_p2p(self.window)

    @property
    def mode(self):
        """the transport of the loop that runs: after begin() / solve() what was taken, before it what would be"""
        p2p = self._p2p_active if self._p2p_active is not None else self._p2p()
        if p2p and self.rccl is not None:
            return "rccl"                 # the native collective loop
        return "p2p" if p2p else ("rccl" if (self.multi or (self.window is not None and self.dist is not None)) else "single")

    def _note_downgrade(self):
        if self.window is not None and not self.downgraded:
            self.downgraded = ("the engine as bound cannot run the window loop (needs the fused direction update, "
                               "the vector path, stencil variant 2, matching plane size): collective loop")
            warnings.warn(f"SlabCG rank {self.part.rank}: p2p window given but not usable -- {self.downgraded}",
                          RuntimeWarning, stacklevel=3)

    def _drop_jacobi(self):
        """The opt-in Jacobi preconditioner runs in the single-GPU loop and in the WINDOW slab loop; the collective loop is
        the reference's unpreconditioned CG (same solution, the reference's iteration count).  Say so once and switch the
        flag off instead of silently running phases that ignore it."""
        info = self.ops.loop_info() if hasattr(self.ops, "loop_info") else {}
        if info.get("jacobi"):
            warnings.warn("SlabCG: the collective slab loop has no Jacobi preconditioning (the window loop and the single-GPU "
                          "loop do) -- running the reference's unpreconditioned CG (mfs_pcg3d_set_jacobi switched off)",
                          RuntimeWarning, stacklevel=3)
            self.ops.set_jacobi(False)

    def _allreduce(self, slot):
        self._allreduce_bounded(self.ops.scalars[slot:slot + 1], f"all-reduce of CG scalar {slot}")

    def _halo_start(self):
        """start the exchange of d's two edge planes; returns the function that completes it.  RCCL moves the device
        planes themselves; any other backend (gloo in tests and rehearsals) gets host copies."""
        dist, d, p, L = self.dist, self.d, self.part, self.L
        staged = getattr(d, "is_cuda", False) and dist.get_backend(self.group) != "nccl"
        src = {k: d[k].cpu() for k in (0, 1, L - 2, L - 1)} if staged else d
        ops = []
        if p.left is not None:
            ops.append(dist.P2POp(dist.isend, src[1], p.left, self.group))
            ops.append(dist.P2POp(dist.irecv, src[0], p.left, self.group))
        if p.right is not None:
            ops.append(dist.P2POp(dist.isend, src[L - 2], p.right, self.group))
            ops.append(dist.P2POp(dist.irecv, src[L - 1], p.right, self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []

        def finish():
            for w in reqs:
                self._wait(w, "halo exchange of d")
            if staged:
                if p.left is not None:
                    d[0].copy_(src[0])
                if p.right is not None:
                    d[L - 1].copy_(src[L - 1])
        return finish

    def _p2p(self):
        """the window is usable for the engine as bound right now (else: the collective loop, on every rank alike --
        the conditions are properties of the grid and dtype, identical across ranks)"""
        return self.window is not None and self.ops.slab_supported()

    def begin(self, tol):
        self._attach()
        if self._p2p():
            self._p2p_active = True
            if self.rccl is not None:
                self._drop_jacobi()
            self.ops.slab_begin(tol)
            return
        self._p2p_active = False
        if self.window is not None and self.dist is not None:
            self._note_downgrade()
            self.multi = True                  # a window was given but cannot be used: the collective loop
        if not self.multi:
            if hasattr(self.ops, "begin"):
                self.ops.begin(tol)            # the engine's single-domain entry point (it also builds the solve's sparse lists)
            else:
                self.ops.begin_local(tol)
                self.ops.begin_finish()
            return
        self._drop_jacobi()
        self.ops.begin_local(tol)              # x = 0 everywhere, so q = A x needs no halo
        self._allreduce(_lib.S_RR)
        self.ops.begin_finish()

    def iterate(self, n):
        if getattr(self, "_p2p_active", False):
            self.ops.slab_iterate(n)
            return
        if not self.multi:
            self.ops.iterate(n)
            return
        L, ops = self.L, self.ops
        for _ in range(int(n)):
            halo_done = self._halo_start()
            if self.overlap and L > 4:
                ops.phase_apply(2, L - 2, True)      # planes that touch no ghost, while the halos fly
                halo_done()
                if hasattr(ops, "phase_apply2"):
                    ops.phase_apply2(1, 2, L - 2, L - 1, False)      # both edge planes, one launch
                else:
                    ops.phase_apply(1, 2, False)
                    ops.phase_apply(L - 2, L - 1, False)
            else:
                halo_done()
                ops.phase_apply(1, L - 1, True)
            ops.phase_reduce(0)
            self._allreduce(_lib.S_DQ)
            if self.overlap:
                # r first, so the r.r all-reduce is in flight (on the collective's own stream)
                # while x += alpha d runs on the compute stream
                ops.phase_update_r()
                ops.phase_reduce(1)
                work = self.dist.all_reduce(self.ops.scalars[_lib.S_RR:_lib.S_RR + 1], group=self.group,
                                            async_op=True)
                ops.phase_update_x()
                self._wait(work, "all-reduce of r.r")
            else:
                ops.phase_update_xr()
                ops.phase_reduce(1)
                self._allreduce(_lib.S_RR)
            ops.phase_update_d()

    def finish(self):
        """for callers of begin / iterate: settle what the device loop still owes (the deferred solution update,
        the direction vector's home buffer).  solve() does this itself."""
        if getattr(self, "_p2p_active", False) or not self.multi:
            self.ops.finish()

    def solve(self, tol, max_iter, check_every=32):
        """begin + iterate until the device-resident `done` flag or max_iter; returns (converged, iterations).
        COLLECTIVE.  The scalars every rank tests are bit-identical, so all ranks leave the loop together."""
        self._attach()
        if self._p2p() and self.rccl is None:
            self._p2p_active = True
            return self.ops.slab_solve(tol, max_iter, check_every)
        if self._p2p():      # native collective loop: the same batches, every host wait bounded (a lost peer stalls RCCL)
            self.begin(tol)
            self.drain()
            st = self.ops.poll()
            enq = 0
            while not st["done"] and enq < max_iter:
                n = min(int(check_every), int(max_iter) - enq)
                self.iterate(n)
                enq += n
                self.drain()
                st = self.ops.poll()
            self.ops.finish()
            return bool(st["done"]), int(st["iterations"])
        self._p2p_active = False
        if self.window is not None and self.dist is not None:
            self._note_downgrade()
            self.multi = True                  # a window was given but cannot be used: the collective loop
        if not self.multi:
            return self.ops.solve(tol, max_iter, check_every)
        self._drop_jacobi()
        self.begin(tol)
        self.drain()
        st = self.ops.poll()
        enq = 0
        while not st["done"] and enq < max_iter:
            n = min(int(check_every), int(max_iter) - enq)
            self.iterate(n)
            enq += n
            self.drain()
            st = self.ops.poll()
        return bool(st["done"]), int(st["iterations"])

    def exchange(self, t):
        """one-off halo exchange of another local field (e.g. x before a velocity update)."""
        if self.dist is None or self.part.world == 1:
            return
        dist, p, L = self.dist, self.part, self.L
        staged = getattr(t, "is_cuda", False) and dist.get_backend(self.group) != "nccl"
        if staged:
            # only RCCL moves device memory; any other backend (gloo in the tests) gets host copies of the
            # four planes involved.  The copy to the host waits for the kernels that produced them.
            src = {k: t[k].cpu() for k in (0, 1, L - 2, L - 1)}
        else:
            src = t
        ops = []
        if p.left is not None:
            ops.append(dist.P2POp(dist.isend, src[1], p.left, self.group))
            ops.append(dist.P2POp(dist.irecv, src[0], p.left, self.group))
        if p.right is not None:
            ops.append(dist.P2POp(dist.isend, src[L - 2], p.right, self.group))
            ops.append(dist.P2POp(dist.irecv, src[L - 1], p.right, self.group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            self._wait(w, "halo exchange")
        if staged:
            if p.left is not None:
                t[0].copy_(src[0])
            if p.right is not None:
                t[L - 1].copy_(src[L - 1])


def exchange_edge_planes(dist, group, part, fields, L, wait=None):
    """Halo exchange of several local fields at once: for each tensor (first axis = local plane index, planes 0 and
    L-1 ghost / boundary) plane 1 goes to the left neighbour's plane L'-1 and plane L-2 to the right neighbour's
    plane 0.  One batch of sends / receives for all fields.  RCCL moves device planes; any other backend (gloo
    in the tests) gets host copies."""
    if dist is None or part.world == 1:
        return
    staged = any(getattr(t, "is_cuda", False) for t in fields) and dist.get_backend(group) != "nccl"
    srcs = [({k: t[k].cpu() for k in (0, 1, L - 2, L - 1)} if staged else t) for t in fields]
    ops = []
    for src in srcs:
        if part.left is not None:
            ops.append(dist.P2POp(dist.isend, src[1], part.left, group))
            ops.append(dist.P2POp(dist.irecv, src[0], part.left, group))
        if part.right is not None:
            ops.append(dist.P2POp(dist.isend, src[L - 2], part.right, group))
            ops.append(dist.P2POp(dist.irecv, src[L - 1], part.right, group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        if wait is not None:
            wait(w, "halo exchange")           # bounded (SlabVCG / SlabCG._wait)
        else:
            w.wait()
    if staged:
        for t, src in zip(fields, srcs):
            if part.left is not None:
                t[0].copy_(src[0])
            if part.right is not None:
                t[L - 1].copy_(src[L - 1])


class SlabVCG(_BoundedCollectives):
    """The viscosity CG (three staggered components) over one x-slab: the same decomposition as SlabCG.

    Rank p holds cell planes [a_p - 1, a_{p+1} + 1) of the global grid (L planes), hence u (x-face) planes
    [a_p - 1, a_{p+1} + 1] (L + 1 of them) and v / w planes as cells.  In local indices, for all three
    components: plane 0 = ghost (left neighbour's plane L'-2) or global array boundary, planes [1, L-1) owned,
    plane L-1 = ghost (right neighbour's plane 1) -- except on the last rank, whose u plane L-1 is the real
    global face Nx-1 and is computed there (`set_slab(skip_top_x)` is off on that rank only); u plane L is never
    read.  The single-domain kernels run unmodified on the local arrays (they skip array-boundary faces exactly
    where the ghosts are); with b = 0 on the ghost planes, q and r stay 0 there, so the local dot products are
    the owned sums.  Per iteration: one halo exchange of d (3 components, one batch), two scalar all-reduces.
    `ops`: mfs.vcg.VcgEngine (or a stand-in with the same phase interface); `d_views`: the three component
    views of the bound direction vector."""

    def __init__(self, ops, part, d_views, dist=None, group=None, window=None):
        self.ops, self.part, self.dist, self.group = ops, part, dist, group
        self.d_views = list(d_views)
        self.L = int(self.d_views[1].shape[0])
        if self.L != part.local_planes or int(self.d_views[0].shape[0]) != self.L + 1:
            raise ValueError("d views do not match the partition's local plane count")
        self.multi = dist is not None and part.world > 1
        ops.set_slab(part.right is not None)
        # window: a connected, self-tested mfs.p2p.P2PWindow of ops.edge_plane_bytes() -> the halo planes and the two
        # dot products of the CG loop move as xGMI stores from kernels of the library (mfs_vcg3d_slab_*); the
        # collectives below remain for the extrapolation sweeps and as the fallback
        self.window = window if (window is not None and window.ok) else None
        if self.window is not None:
            ops.attach_p2p(self.window)
        self._bc_init()
        # the opt-in Jacobi preconditioning runs in the single-GPU loop and in the WINDOW slab loop; the collective loop's
        # phases are the reference's unpreconditioned CG -- say so instead of silently ignoring the flag
        if self.multi and self.window is None and hasattr(ops, "loop_info") and ops.loop_info().get("jacobi"):
            warnings.warn("SlabVCG: the collective slab loop has no Jacobi preconditioning (the window loop and the single-GPU "
                          "loop do) -- running the reference's unpreconditioned CG (mfs_vcg3d_set_jacobi switched off)",
                          RuntimeWarning, stacklevel=2)
            ops.set_jacobi(False)

    @property
    def mode(self):
        return "p2p" if self.window is not None else ("rccl" if self.multi else "single")

    def _allreduce(self, slot):
        if self.multi:
            self._allreduce_bounded(self.ops.scalars[slot:slot + 1], f"all-reduce of CG scalar {slot}")

    def exchange(self, fields):
        exchange_edge_planes(self.dist if self.multi else None, self.group, self.part, list(fields), self.L,
                             wait=self._wait)

    def begin(self, tol):
        if self.window is not None:
            self.ops.slab_begin(tol)
            return
        self.ops.begin_local(tol)              # q = A x on the local slab (x's ghosts are the caller's), d = r = b - q
        self._allreduce(_lib.S_RR)
        self.ops.begin_finish()

    def iterate(self, n):
        ops = self.ops
        if self.window is not None:
            ops.slab_iterate(n)
            return
        for _ in range(int(n)):
            self.exchange(self.d_views)
            ops.phase_apply()
            ops.phase_reduce(0)
            self._allreduce(_lib.S_DQ)
            ops.phase_update_xr()
            ops.phase_reduce(1)
            self._allreduce(_lib.S_RR)
            ops.phase_update_d()

    def solve(self, tol, max_iter, check_every=32):
        """COLLECTIVE; returns (converged, iterations).  Every rank tests the same all-reduced scalars."""
        self.begin(tol)
        bounded = self.multi and self.window is None      # the window loop bounds its own waits on the device
        if bounded:
            self.drain()
        st = self.ops.poll()
        enq = 0
        while not st["done"] and enq < max_iter:
            n = min(int(check_every), int(max_iter) - enq)
            self.iterate(n)
            enq += n
            if bounded:
                self.drain()
            st = self.ops.poll()
        return bool(st["done"]), int(st["iterations"])


class SlabBands:
    """Plane-band exchanges for fields kept as GLOBAL-shaped arrays of which every rank maintains only its own x-range
    (the particle-sharded time step, notebook_sim.ShardedNotebookSimulation): no whole-grid collective, only the
    planes near the cuts travel.

    Ownership: the cell planes [0, Nx) are cut at the slab partition's cuts (`SlabPartition.owned`; the two array-
    boundary planes 0 and Nx-1 go to the first / last rank).  An array's `kind` says how its axis 0 relates to cells:
    "cell" (Nx planes), "xface" (Nx+1: face i belongs to the owner of cell i, the last face to the last rank) or
    "doubled" (2Nx+1 nodes: nodes 2i, 2i+1 belong to the owner of cell i, the last node to the last rank).

      reduce(arrs, kind, reach, op): every rank has scattered its own particles' contributions, up to `reach` cells
          beyond its range; the contributions on planes another rank owns are sent there and combined (op "sum" | "min").
      ghosts(arrs, kind, width): every rank receives the owners' values of the planes up to `width` cells beyond its range.

    Bands may span more than one neighbour (thin slabs): a rank talks to every rank that owns planes of its band.
    P2P sends / receives of contiguous plane ranges (RCCL moves device memory; any other backend gets host copies);
    waits are bounded like every collective wait of this module."""

    def __init__(self, dist, group, nx, device=None):
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.nx = int(nx)
        self.part = SlabPartition(nx, self.world, self.rank)
        self.cuts = [0] + [SlabPartition(nx, self.world, r).owned[0] for r in range(1, self.world)] + [self.nx]
        self.timeout_s = collective_timeout_s()
        self.bytes_moved = 0
        self.device = device

    def _bounded(self, work, what):
        """wait for an async collective: host-blocking backends against the deadline; "nccl" orders the stream (the
        group's own timeout -- pg_timeout -- is the bound there, as for every RCCL wait of this module)"""
        if work is None:
            return
        if self.dist.get_backend(self.group) == "nccl":
            work.wait()
            return
        import datetime as _dt
        try:
            ok = work.wait(timeout=_dt.timedelta(seconds=self.timeout_s))
        except RuntimeError as exc:
            raise _lib.MfsTimeout(f"{what} failed with status {_lib.MFS_E_TIMEOUT}: wait timed out on rank "
                                  f"{self.rank} [{type(exc).__name__}: {str(exc)[:160]}]") from None
        if ok is False:
            raise _lib.MfsTimeout(f"{what} failed with status {_lib.MFS_E_TIMEOUT}: wait timed out on rank {self.rank}")

    def exchange_counts(self, counts, device=None):
        """all-gather of a small int64 vector (one per rank) -> [W, len] host tensor; operands on the collective device"""
        import torch
        cd = coll_device(self.dist, self.group, device if device is not None else self.device)
        mine = counts.to(device=cd, dtype=torch.int64).contiguous()
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        self._bounded(self.dist.all_gather(allc, mine, group=self.group, async_op=True), "band count exchange")
        return torch.stack([c.cpu() for c in allc], dim=0)

    def allreduce_scalar(self, value, op="max", device=None):
        """one double combined over the ranks (the CFL maximum of the sharded time step); returns a float"""
        import torch
        cd = coll_device(self.dist, self.group, device if device is not None else self.device)
        t = torch.tensor([float(value)], dtype=torch.float64, device=cd)
        rop = {"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op]
        self._bounded(self.dist.all_reduce(t, op=rop, group=self.group, async_op=True), "band scalar all-reduce")
        return float(t.cpu().item())

    def cells(self, r=None):
        r = self.rank if r is None else r
        return self.cuts[r], self.cuts[r + 1]

    def owned(self, kind, r=None):
        r = self.rank if r is None else r
        a, b = self.cells(r)
        last = r == self.world - 1
        if kind == "cell":
            return a, b
        if kind == "xface":
            return a, b + (1 if last else 0)
        if kind == "doubled":
            return 2 * a, 2 * b + (1 if last else 0)
        raise ValueError(kind)

    def _band(self, kind, width, r):
        """the planes up to `width` cells beyond rank r's range, as (lo band, hi band), clipped to the array"""
        a, b = self.owned(kind, r)
        w = width * (2 if kind == "doubled" else 1)
        n = {"cell": self.nx, "xface": self.nx + 1, "doubled": 2 * self.nx + 1}[kind]
        return (max(0, a - w), a), (b, min(n, b + w))

    @staticmethod
    def _cut(r0, r1):
        lo, hi = max(r0[0], r1[0]), min(r0[1], r1[1])
        return (lo, hi) if hi > lo else None

    def _plan(self, kind, width, mine_is_band):
        """[(peer, send range, recv range)]: mine_is_band -> I SEND my band planes the peer owns and RECEIVE the peer's
        band planes I own (reduce); else I send my owned planes in the peer's band and receive its owned planes in mine"""
        plan = []
        for q in range(self.world):
            if q == self.rank:
                continue
            send = recv = None
            for mb, qb in zip(self._band(kind, width, self.rank), self._band(kind, width, q)):
                if mine_is_band:
                    s, rcv = self._cut(mb, self.owned(kind, q)), self._cut(qb, self.owned(kind))
                else:
                    s, rcv = self._cut(self.owned(kind), qb), self._cut(self.owned(kind, q), mb)
                send = send or s
                recv = recv or rcv
            if send or recv:
                plan.append((q, send, recv))
        return plan

    def _run(self, ops_send, ops_recv):
        """ops_send: [(tensor view, peer)], ops_recv: [(buffer, peer)]; completes them all (bounded)"""
        dist = self.dist
        if not ops_send and not ops_recv:
            return
        staged = any(getattr(t, "is_cuda", False) for t, _ in ops_send + ops_recv) and dist.get_backend(self.group) != "nccl"
        sends = [(t.cpu() if staged else t.contiguous(), q) for t, q in ops_send]
        recvs = [((torch_empty_like_cpu(t) if staged else t), q, t) for t, q in ops_recv]
        ops = [dist.P2POp(dist.isend, t, q, self.group) for t, q in sends]
        ops += [dist.P2POp(dist.irecv, t, q, self.group) for t, q, _ in recvs]
        for w in dist.batch_isend_irecv(ops):
            self._bounded(w, "band exchange")
        if staged:
            for buf, _, dst in recvs:
                dst.copy_(buf)
        self.bytes_moved += sum(t.numel() * t.element_size() for t, _ in ops_send)

    def reduce(self, arrs, kind, reach, op="sum"):
        if self.world == 1:
            return
        plan = self._plan(kind, reach, True)
        sends, recvs, combine = [], [], []
        for arr in arrs:
            for q, s, r in plan:
                if s:
                    sends.append((arr[s[0]:s[1]], q))
                if r:
                    buf = arr.new_empty((r[1] - r[0],) + tuple(arr.shape[1:]))
                    recvs.append((buf, q))
                    combine.append((arr, r, buf))
        self._run(sends, recvs)
        for arr, r, buf in combine:
            dst = arr[r[0]:r[1]]
            if op == "sum":
                dst.add_(buf)
            elif op == "min":
                import torch
                torch.minimum(dst, buf, out=dst)
            else:
                raise ValueError(op)

    def ghosts(self, arrs, kind, width):
        if self.world == 1:
            return
        plan = self._plan(kind, width, False)
        sends, recvs = [], []
        for arr in arrs:
            for q, s, r in plan:
                if s:
                    sends.append((arr[s[0]:s[1]], q))
                if r:
                    recvs.append((arr[r[0]:r[1]], q))
        self._run(sends, recvs)

    def owner_of_cells(self, ix):
        """rank owning cell plane ix (tensor of int64 cell indices, clamped to the grid)"""
        import torch
        cuts = torch.as_tensor(self.cuts[1:-1], dtype=ix.dtype, device=ix.device)
        return torch.bucketize(ix.clamp(0, self.nx - 1), cuts, right=True)

    def migrate(self, fields, dest):
        """particles change owner: `fields` = list of per-particle tensors (same length P), `dest` = destination rank of
        each particle.  Returns the list of tensors after the exchange (kept particles first, then arrivals in rank
        order).  One count exchange + one packed payload per peer pair."""
        import torch
        dist, me, W = self.dist, self.rank, self.world
        if W == 1:
            return fields
        keep = dest == me
        counts = torch.bincount(dest, minlength=W).to(torch.int64)
        allc = self.exchange_counts(counts, device=fields[0].device)   # allc[q][r] = particles q sends to r
        counts = counts.cpu()
        widths = []
        for f in fields:
            w = 1
            for s_ in f.shape[1:]:
                w *= int(s_)
            widths.append(w)
        dev = fields[0].device
        packed = torch.cat([f.reshape(f.shape[0], w).to(torch.float64) for f, w in zip(fields, widths)], dim=1)   # ids ride as exact doubles
        sends, recvs, bufs = [], [], []
        for q in range(W):
            if q == me:
                continue
            ns, nr = int(counts[q]), int(allc[q][me])
            if ns:
                sends.append((packed[dest == q].contiguous(), q))
            if nr:
                b = torch.empty((nr, packed.shape[1]), dtype=torch.float64, device=dev)
                recvs.append((b, q))
                bufs.append(b)
        self._run(sends, recvs)
        out = torch.cat([packed[keep]] + bufs, dim=0) if bufs else packed[keep]
        res, o = [], 0
        for f, w in zip(fields, widths):
            col = out[:, o:o + w]
            o += w
            res.append(col.reshape((out.shape[0],) + tuple(f.shape[1:])).to(f.dtype).contiguous())
        return res


def torch_empty_like_cpu(t):
    import torch
    return torch.empty(tuple(t.shape), dtype=t.dtype, device="cpu")
