"""The notebook's scene containers and time step (3D_viscous_fluid_sim.ipynb code cells 9 and 11) on the
MI355X drop-ins: every stage of the loop body ipynb:4571-4667 -- advect, `sdf.project`, level set / fluid
volume, density solve, p2g, gravity, viscosity CG, pressure CG, extrapolate, boundary condition, g2p --
runs as HIP kernels behind the C ABI (SURVEY.md 8(f) ranks 1-4 around the two hot-path solvers).

This is the driver a user of the reference's notebook switches to: same containers (attribute names of
its `edict`s), same dtypes (bounds / biases float32, cell sizes float64, particle arrays float64, grid mass
and velocity float32, level sets float64), same call order with the `solver='apic'` branch (the U-Net
branch needs a checkpoint that is a remote download).  `NotebookSimulation`: single GPU;
`SlabNotebookSimulation`: one process per GPU, the two hot-path CG solves slab-decomposed (BASELINE config 5).
"""
import time
import types

import numpy as np
import torch

import notebook_kernels as K
from solver import sdf3D as sdf
from solver.CGSolverBuffer import CGSolverBuffer
from solver.DensityCGSolver3D import DensityCGSolver3D, SlabDensityCGSolver3D
from solver.PressureCGSolver3D import PressureCGSolver3D, SlabPressureCGSolver3D
from solver.ViscosityCGSolver3D import SlabViscosityCGSolver3D, ViscosityCGSolver3D

NS = types.SimpleNamespace


def grid_positions(res, bound_min, cell_size, bias, device):
    """get_grid_pos (code cell 9): bound_min + (float32 index + float32 bias) * cell_size, float64."""
    ax = [torch.arange(int(r), dtype=torch.float32, device=device) for r in res]
    idx = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1)
    b = torch.as_tensor(np.asarray(bias, np.float32), device=device)
    cs = torch.as_tensor(np.asarray(cell_size, np.float64), device=device)
    bm = torch.as_tensor(np.asarray(bound_min, np.float32), device=device).to(torch.float64)
    return (bm + (idx + b).to(torch.float64) * cs).contiguous()


class NotebookSimulation:
    """Containers of code cell 9 and the loop body of code cell 11.

    gres, gdx: cell grid and spacing; bound_min: float32 triple; rb_d: packed rigid bodies (solver.sdf3D);
    px: (P,3) float64 particle positions; pdx: particle spacing (mass = rho * pdx^3, volume = pdx^3)."""

    def __init__(self, gres, gdx, bound_min, rb_d, px, pdx, rho=1000.0, mu=1.0, dt=1.0 / 300.0, device="cuda",
                 precision=None, jacobi=False):
        """jacobi=True: the build's opt-in Jacobi preconditioning for the viscosity and pressure solves (NOT the reference's
        iterations -- same stopping rules, a fraction of the CG iterations; single-GPU class only)"""
        dev = torch.device(device)
        g = tuple(int(v) for v in gres)
        self.GRES, self.GDX, self.PDX, self.RHO, self.MU, self.DT = g, float(gdx), float(pdx), float(rho), float(mu), float(dt)
        self.device = dev
        bmin = np.asarray(bound_min, np.float32)
        bsz = (np.asarray(g, np.float64) * gdx).astype(np.float32)         # BOUND_SIZE is a float32 array
        self.BOUND_MIN, self.BOUND_SIZE = bmin, bsz
        self.rb_d = rb_d
        px = torch.as_tensor(px, dtype=torch.float64, device=dev).contiguous()
        n = px.shape[0]
        z3 = lambda: torch.zeros((n, 3), dtype=torch.float64, device=dev)  # noqa: E731
        self.particle = NS(num_particles=n, x=px, m=torch.full((n,), rho * pdx ** 3, dtype=torch.float64, device=dev),
                           v=z3(), cx=z3(), cy=z3(), cz=z3(), vol=pdx ** 3)
        eye = np.eye(3, dtype=np.int64)
        cs = bsz / np.asarray(g, np.int64)                                 # float32 / int64 -> float64, as in cupy

        def comp(a, bias):
            shape = tuple(np.asarray(g) + eye[a])
            f = lambda: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
            return NS(resolution=shape, bias=np.asarray(bias, np.float32), m=f(), v=f(), dv=f())
        self.grid = NS(resolution=g, bound_size=bsz, bound_min=bmin, cell_size=cs, x=comp(0, [0, .5, .5]),
                       y=comp(1, [.5, 0, .5]), z=comp(2, [.5, .5, 0]))
        dres = tuple(2 * v + 1 for v in g)
        dcs = bsz / (2 * np.asarray(g, np.int64))
        self.solid_levelset = NS(resolution=dres, bound_size=bsz, bound_min=bmin, cell_size=dcs,
                                 bias=np.zeros(3, np.float32),
                                 phi=torch.zeros(dres, dtype=torch.float64, device=dev),
                                 v=torch.zeros(dres + (3,), dtype=torch.float64, device=dev))
        self.solid_levelset.pos = grid_positions(dres, bmin, dcs, self.solid_levelset.bias, dev)
        sdf.evaluate(rb_d, self.solid_levelset.phi, self.solid_levelset.v, self.solid_levelset.pos)
        self.fluid_levelset = NS(resolution=g, bound_size=bsz, bound_min=bmin, cell_size=cs,
                                 phi=torch.zeros(g, dtype=torch.float64, device=dev))
        self.fluid_volume = NS(resolution=dres, bound_size=bsz, bound_min=bmin, cell_size=dcs,
                               vol=torch.zeros(dres, dtype=torch.float64, device=dev))
        self._precision = precision
        self._jacobi = bool(jacobi)
        self._make_solvers()
        self.current_time = 0.0
        self.iterations = 0

    def _make_solvers(self):
        """CGBuf / PressureSolver / DensitySolver / ViscositySolver of code cell 9 (ipynb:777-780)."""
        g, dev = self.GRES, self.device
        self.CGBuf = CGSolverBuffer(g, precision=self._precision, device=dev)
        self.PressureSolver = PressureCGSolver3D(self.CGBuf, g, self.GDX)
        self.DensitySolver = DensityCGSolver3D(self.CGBuf, g, self.BOUND_MIN, self.BOUND_SIZE)
        self.ViscositySolver = ViscosityCGSolver3D(g, self.BOUND_SIZE, precision=self._precision, device=dev)
        if getattr(self, "_jacobi", False):
            self.PressureSolver._engine.set_jacobi(True)
            self.ViscositySolver._engine.set_jacobi(True)
            self.DensitySolver._engine.set_jacobi(True)      # (the pressure engine with the density operator's -z tap)

    def _solve_grid(self, dt, tick, t):
        """the two hot-path solves of the loop body (ipynb:4623, 4648) on the grid velocities, in place"""
        g, sl, fl, fv = self.grid, self.solid_levelset, self.fluid_levelset, self.fluid_volume
        if self.MU > 0:
            self.ViscositySolver.solve(dt, self.MU, self.RHO, g.x.v, g.y.v, g.z.v, sl.phi, sl.v, fl.phi, fv.vol)
        t = tick("viscosity", t)
        ds = self.DensitySolver
        self.PressureSolver.solve(g.x.v, g.y.v, g.z.v, sl.phi, sl.v, fl.phi, wx=ds.wx, wy=ds.wy, wz=ds.wz)
        return tick("pressure", t)

    def step(self, duration_left=float("inf"), timings=None):
        """One pass of the loop body (ipynb:4571-4667, solver == 'apic').  Returns the dt it took."""
        p, g, sl, fl, fv = self.particle, self.grid, self.solid_levelset, self.fluid_levelset, self.fluid_volume

        def tick(name, t0):
            if timings is not None:
                torch.cuda.synchronize()
                timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()

        t = time.perf_counter()
        vmax = torch.sqrt((p.v ** 2).sum(dim=-1)).max().item() if p.num_particles else 0.0
        cfl_dt = self.GDX / max(1e-10, vmax)
        dt = min(self.DT, cfl_dt, duration_left)
        self.current_time += dt
        p.x += p.v * dt
        sdf.project(self.rb_d, p.x)
        t = tick("advect+project", t)
        K.compute_fluid_levelset(p, fl, self.GDX)
        K.compute_fluid_volume(p, fv, p.vol)
        t = tick("levelset+volume", t)
        self.DensitySolver.solve(self.RHO, dt, p.x, p.m, p.vol, g.x.v, g.y.v, g.z.v, sl.phi, sl.v, fl.phi, fv.vol)
        t = tick("density", t)
        K.compute_fluid_levelset(p, fl, self.GDX)
        K.compute_fluid_volume(p, fv, p.vol)
        t = tick("levelset+volume", t)
        for c in (g.x, g.y, g.z):
            c.m.zero_()
            c.v.zero_()
        K.p2g(p, g)
        g.y.v += -10 * dt                                                   # gravity
        t = tick("p2g", t)
        t = self._solve_grid(dt, tick, t)
        K.extrapolate(self.GRES, 2, g.x.v, g.y.v, g.z.v, g.x.m, g.y.m, g.z.m)
        K.apply_boundary_condition(g, sl, self.GDX)
        t = tick("extrapolate+bc", t)
        K.g2p(p, g)
        tick("g2p", t)
        self.iterations += 1
        return dt


class SlabNotebookSimulation(NotebookSimulation):
    """The same time step on N GPUs, one process per GPU (extension; BASELINE config 5 names 4 GPUs).

    What is sharded: the three CG loops -- 98 % of a 256^3 step on one GPU -- run slab-decomposed
    along x on this rank's planes (`SlabViscosityCGSolver3D`, `SlabPressureCGSolver3D`, `SlabDensityCGSolver3D`: halo
    planes and dot products over xGMI).  What is replicated: the particle stages and the small grid stages (advect,
    project, level set / volume, splat, p2g, extrapolate, boundary condition, g2p) -- every rank holds all particles
    and the full grids for them.  Their atomics make the replicas differ in the last bits, so rank 0's grid
    state is broadcast before the solves (one source of truth for every slab and its ghost planes), and the solved
    velocities are gathered back (each rank broadcasts its owned planes) for the replicated stages that follow.
    Amdahl: with the fraction f of a one-GPU step in the three loops the step takes (1 - f) + f / N; sharding the
    particles (migration between slabs) is the next step.  `step` is collective."""

    def __init__(self, *args, dist, group=None, transport="auto", **kw):
        self.dist, self.group, self._transport = dist, group, transport
        super().__init__(*args, **kw)      # (jacobi=True: honoured by the window slab loops; the collective loops drop it with a warning)

    def _make_solvers(self):
        from mfs.dist import SlabPartition
        g, dev, dist, group = self.GRES, self.device, self.dist, self.group
        self.CGBuf = CGSolverBuffer(g, precision=self._precision, device=dev)            # the density solve's global RHS
        self.DensitySolver = SlabDensityCGSolver3D(self.CGBuf, g, self.BOUND_MIN, self.BOUND_SIZE, dist, group,
                                                   transport=self._transport)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.part = SlabPartition(g[0], self.world, self.rank)
        lg = SlabPressureCGSolver3D.local_gres(g, self.world, self.rank)
        self._slab_buf = CGSolverBuffer(lg, precision=self._precision, device=dev)
        self.PressureSolver = SlabPressureCGSolver3D(self._slab_buf, g, self.GDX, dist, group, transport=self._transport)
        self.ViscositySolver = SlabViscosityCGSolver3D(g, self.BOUND_SIZE, dist, group, precision=self._precision, device=dev,
                                                       transport=self._transport, jacobi=True if self._jacobi else None)
        if self._jacobi:
            self.PressureSolver._engine.set_jacobi(True)
            self.DensitySolver._engine.set_jacobi(True)

    def _solve_grid(self, dt, tick, t):
        from mfs.dist import SlabPartition
        g, sl, fl, fv, ds = self.grid, self.solid_levelset, self.fluid_levelset, self.fluid_volume, self.DensitySolver
        dist, group = self.dist, self.group
        if self.world > 1:
            for a in (g.x.v, g.y.v, g.z.v, fl.phi, fv.vol, ds.wx, ds.wy, ds.wz):
                dist.broadcast(a, src=0, group=group)
        t = tick("broadcast", t)
        lo, hi = self.part.local_range
        vx, vy, vz = g.x.v[lo:hi + 1], g.y.v[lo:hi], g.z.v[lo:hi]            # views: the solves update the planes in place
        sphi, sv, lphi = sl.phi[2 * lo:2 * hi + 1], sl.v[2 * lo:2 * hi + 1], fl.phi[lo:hi]
        if self.MU > 0:
            self.ViscositySolver.solve(dt, self.MU, self.RHO, vx, vy, vz, sphi, sv, lphi, fv.vol[2 * lo:2 * hi + 1])
        t = tick("viscosity", t)
        self.PressureSolver.solve(vx, vy, vz, sphi, sv, lphi, wx=ds.wx[lo:hi + 1], wy=ds.wy[lo:hi], wz=ds.wz[lo:hi])
        t = tick("pressure", t)
        if self.world > 1:       # every rank's owned planes -> every rank (the last rank also owns the planes up to Nx-1)
            for r in range(self.world):
                a, b = SlabPartition(self.GRES[0], self.world, r).owned
                if r == self.world - 1:
                    b += 1
                for arr in (g.x.v, g.y.v, g.z.v):
                    dist.broadcast(arr[a:b], src=r, group=group)
        return tick("gather", t)

    def close(self):
        self.PressureSolver.close()
        self.DensitySolver.close()
        self.ViscositySolver.close()


class ShardedNotebookSimulation(SlabNotebookSimulation):
    """BASELINE config 5 with the PARTICLES sharded as well: one process per GPU, rank r owns the particles whose cell
    lies in its x-range (`mfs.dist.SlabBands`: the slab partition's cuts) and maintains every grid field on its own
    planes plus a band of ghost planes; no stage touches a whole grid over the wire.

      advect + project      own particles; those that left the range MIGRATE to their new owner (ids travel with them)
      level set / volume    atomic-min / trilinear splat of the own particles; contributions beyond the range go to their
                            owners (band reduce, min / sum), ghost planes come back from the owners
      density solve         `SlabDensityCGSolver3D.solve_sharded` (splat -> band reduce; CG slab-decomposed; displacement)
      p2g                   scatter of mass and momentum -> band reduce -> normalise -> ghosts
      viscosity, pressure   slab-decomposed CG on views of this rank's planes (as in SlabNotebookSimulation)
      extrapolate, BC, g2p  grid stencils on the range + ghost band (their reach, summed, is the band width), gather by
                            the own particles
    The constructor takes the FULL initial particle set (every rank builds the scene identically) and keeps the own
    share; `gather_particles()` assembles (id, x, v) over the ranks for inspection.  `step` is collective."""

    GHOST, REACH = 4, 3        # cells: ghost band maintained around the range; scatter reach (5^3 level-set stencil + slack)

    def __init__(self, *args, dist, group=None, transport="auto", **kw):
        super().__init__(*args, dist=dist, group=group, transport=transport, **kw)
        from mfs.dist import SlabBands
        self.bands = SlabBands(dist, group, self.GRES[0], device=self.device)
        p = self.particle
        p.id = torch.arange(p.num_particles, dtype=torch.int64, device=self.device)
        self.total_particles = p.num_particles
        own = self.bands.owner_of_cells(self._cell_x(p.x)) == self.rank      # every rank built the full set: keep the own share
        p.x, p.v, p.cx, p.cy, p.cz, p.m, p.id = (t[own].contiguous() for t in (p.x, p.v, p.cx, p.cy, p.cz, p.m, p.id))
        p.num_particles = int(p.x.shape[0])

    def set_particle_velocities(self, v_all):
        """initial velocities given for the FULL particle set (indexed by particle id)"""
        v_all = torch.as_tensor(v_all, dtype=torch.float64, device=self.device)
        self.particle.v.copy_(v_all[self.particle.id])

    def _cell_x(self, px):
        g = self.grid
        t = (px[:, 0].to(torch.float32) - float(g.bound_min[0])).to(torch.float64) / float(g.cell_size[0])
        return torch.floor(t).to(torch.int64)

    def _migrate(self):
        p = self.particle
        dest = self.bands.owner_of_cells(self._cell_x(p.x))
        fields = [p.x, p.v, p.cx, p.cy, p.cz, p.m, p.id]
        p.x, p.v, p.cx, p.cy, p.cz, p.m, p.id = self.bands.migrate(fields, dest)
        p.num_particles = int(p.x.shape[0])

    def gather_particles(self):
        """(id, x, v) of all particles, sorted by id, on every rank (inspection / tests: a whole-set collective)"""
        from mfs.dist import coll_device
        p, dist = self.particle, self.dist
        # operands on the group's collective device (an RCCL group has no CPU backend); the table comes back to the host
        cd = coll_device(dist, self.group, self.device)
        ns = self.bands.exchange_counts(torch.tensor([p.num_particles], dtype=torch.int64))[:, 0]
        mine = torch.cat([p.id.to(torch.float64)[:, None], p.x, p.v], dim=1).to(cd)
        parts = []
        for r in range(self.world):
            buf = mine if r == self.rank else torch.empty((int(ns[r]), 7), dtype=torch.float64, device=cd)
            dist.broadcast(buf, src=r, group=self.group)
            parts.append(buf.cpu())
        allp = torch.cat(parts, dim=0)
        allp = allp[torch.argsort(allp[:, 0])]
        return allp[:, 0].to(torch.int64), allp[:, 1:4], allp[:, 4:7], [int(v) for v in ns]

    def step(self, duration_left=float("inf"), timings=None):
        p, g, sl, fl, fv = self.particle, self.grid, self.solid_levelset, self.fluid_levelset, self.fluid_volume
        B, W, R, dist = self.bands, self.GHOST, self.REACH, self.dist

        def tick(name, t0):
            if timings is not None:
                torch.cuda.synchronize()
                timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()

        def levelset_and_volume():
            K.compute_fluid_levelset(p, fl, self.GDX)
            B.reduce([fl.phi], "cell", R, "min")
            B.ghosts([fl.phi], "cell", W)
            K.compute_fluid_volume(p, fv, p.vol)            # splat + clamp of the own particles' share ...
            B.reduce([fv.vol], "doubled", R, "sum")         # ... non-negative shares: clamp(sum of clamped) == clamp(sum)
            fv.vol.clamp_(max=float(np.prod(fv.cell_size)))
            B.ghosts([fv.vol], "doubled", W)

        t = time.perf_counter()
        vloc = torch.sqrt((p.v ** 2).sum(dim=-1)).max().item() if p.num_particles else 0.0
        vmax = B.allreduce_scalar(vloc, "max")          # operand on the collective device: RCCL has no CPU backend
        cfl_dt = self.GDX / max(1e-10, vmax)
        dt = min(self.DT, cfl_dt, duration_left)
        self.current_time += dt
        p.x += p.v * dt
        sdf.project(self.rb_d, p.x)
        self._migrate()
        t = tick("advect+project+migrate", t)
        levelset_and_volume()
        t = tick("levelset+volume", t)
        self.DensitySolver.solve_sharded(B, self.RHO, dt, p.x, p.m, p.vol, sl.phi, sl.v, fl.phi, fv.vol, reach=R, width=W)
        self._migrate()                                      # the displacement moved particles
        t = tick("density", t)
        levelset_and_volume()
        t = tick("levelset+volume", t)
        comps = (g.x, g.y, g.z)
        for c in comps:
            c.m.zero_()
            c.v.zero_()
        K.p2g_scatter(p, g)
        B.reduce([g.x.m, g.x.v], "xface", R, "sum")
        B.reduce([g.y.m, g.y.v, g.z.m, g.z.v], "cell", R, "sum")
        K.p2g_normalize(g)
        B.ghosts([g.x.m, g.x.v], "xface", W)
        B.ghosts([g.y.m, g.y.v, g.z.m, g.z.v], "cell", W)
        g.y.v += -10 * dt                                                   # gravity
        t = tick("p2g", t)
        # the two hot-path solves on views of this rank's planes (one ghost / boundary plane each side)
        ds = self.DensitySolver
        lo, hi = self.part.local_range
        vx, vy, vz = g.x.v[lo:hi + 1], g.y.v[lo:hi], g.z.v[lo:hi]
        sphi, sv, lphi = sl.phi[2 * lo:2 * hi + 1], sl.v[2 * lo:2 * hi + 1], fl.phi[lo:hi]
        if self.MU > 0:
            self.ViscositySolver.solve(dt, self.MU, self.RHO, vx, vy, vz, sphi, sv, lphi, fv.vol[2 * lo:2 * hi + 1])
            B.ghosts([g.x.v], "xface", W)                    # the pressure RHS reads the ghost planes' faces
            B.ghosts([g.y.v, g.z.v], "cell", W)
        t = tick("viscosity", t)
        self.PressureSolver.solve(vx, vy, vz, sphi, sv, lphi, wx=ds.wx[lo:hi + 1], wy=ds.wy[lo:hi], wz=ds.wz[lo:hi])
        B.ghosts([g.x.v], "xface", W)
        B.ghosts([g.y.v, g.z.v], "cell", W)
        t = tick("pressure", t)
        K.extrapolate(self.GRES, 2, g.x.v, g.y.v, g.z.v, g.x.m, g.y.m, g.z.m)
        K.apply_boundary_condition(g, sl, self.GDX)
        t = tick("extrapolate+bc", t)
        K.g2p(p, g)
        tick("g2p", t)
        self.iterations += 1
        return dt


def add_box(center, size, dx, rng, keep=None):
    """Particle seeding of code cell 9 (`add_box`): a jittered lattice of spacing dx filling a box."""
    center, size = np.asarray(center, np.float64), np.asarray(size, np.float64)
    dims = (size / dx).astype(np.int64)
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in dims]), axis=-1).astype(np.float32)
    pos = (center - 0.5 * size) + size * ((idx + 0.5) / dims)
    pos = pos.reshape(-1, 3)
    if keep is not None:
        pos = pos[keep(pos)]
    return pos + rng.standard_normal(pos.shape) * dx * 0.3
