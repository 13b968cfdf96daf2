/* mfs.h -- C ABI of libmfs_hip.so: the MI355X (gfx950) pressure / viscosity CG path.
 *
 * The reference (SSTDV-Project/python-fluid-simulation) has no FFI: its `solver/`
 * package is Python that launches numba-CUDA kernels and cupy expressions
 * directly.  This header is therefore the boundary a maintainer binds INSTEAD of
 * those launches: every entry point names the reference function it replaces
 * (file:line, relative to the reference repo root).  The Python classes in
 * python-fluid-simulation_amd/solver/ bind it with ctypes.
 *
 * Conventions
 *  - plain pointers + sizes only; every array pointer is a DEVICE pointer unless
 *    the parameter name ends in `_host`.  No allocation happens behind the ABI:
 *    workspaces are sized by a query and handed in by the caller.
 *  - arrays are C-order with the reference's axis order [x,y,z] (z contiguous):
 *    cell arrays (Nx,Ny,Nz); face arrays vx/wx (Nx+1,Ny,Nz), vy/wy (Nx,Ny+1,Nz),
 *    vz/wz (Nx,Ny,Nz+1); doubled grid sphi/lvol (2Nx+1,2Ny+1,2Nz+1), sv (...,3).
 *  - `*_dt` arguments are mfs_dtype codes describing the element type of the
 *    array group they follow.
 *  - every call is asynchronous on `stream` (a hipStream_t) unless documented
 *    otherwise; return value is an mfs_status.  Nothing throws.
 */
#ifndef MFS_H
#define MFS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFS_ABI_VERSION 3   /* = the round the entry-point set last grew in */

typedef enum { MFS_F32 = 0, MFS_F64 = 1 } mfs_dtype;

typedef enum {
  MFS_OK = 0,
  MFS_NOT_CONVERGED = 1,  /* reference: raise ValueError("Failed to converge!"), PressureCGSolver3D.py:222-223 */
  MFS_E_INVALID = -1,     /* bad argument (null pointer, bad dtype, bad size, misuse) */
  MFS_E_HIP = -2,         /* a HIP runtime call failed; see mfs_last_error() */
  MFS_E_NODEVICE = -3,    /* no gfx950 device visible */
  MFS_E_TIMEOUT = -4,     /* slab loop: a peer GPU did not answer within MFS_P2P_TIMEOUT_MS; the solve was stopped */
  MFS_E_ZERODIV = -5,     /* CG: d.q == 0 -- the reference's `alpha = delta / dq` raises ZeroDivisionError
                             (PressureCGSolver3D.py:211, ViscosityCGSolver3D.py:594); returned by the next poll / solve */
  MFS_E_NONFINITE = -6    /* CG: d.q or r.r is NaN / inf (poisoned input).  The reference would iterate to max_iter on
                             `nan < tol**2` and then raise ValueError("Failed to converge!"); the device loop stops at once */
} mfs_status;

typedef void* mfs_stream; /* hipStream_t */

int mfs_abi_version(void);
/* thread-local text of the last failure on this thread ("" if none) */
const char* mfs_last_error(void);
/* name of the device the current HIP context runs on, e.g. "gfx950:..."; host call */
int mfs_device_name(char* buf_host, size_t cap);

/* ------------------------------------------------------------------------- */
/* Solid fractions                                                           */
/* ------------------------------------------------------------------------- */
/* replaces compute_solid_frac -- solver/SolidFraction3D.py:6-32.
 * Writes w*[0:N] on every axis; the upper faces w*[N] are left untouched.      */
int mfs_solid_frac3d(const int64_t gres[3], const void* sphi, int sphi_dt,
                     void* wx, void* wy, void* wz, int w_dt, mfs_stream stream);
/* replaces compute_solid_frac -- solver/SolidFraction2D.py:6-26 */
int mfs_solid_frac2d(const int64_t gres[2], const void* sphi, int sphi_dt,
                     void* wx, void* wy, int w_dt, mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Pressure, 3D -- stateless kernels (the reference's module-level functions)  */
/* ------------------------------------------------------------------------- */
/* replaces initialize_solver -- solver/PressureCGSolver3D.py:6-50,155-159 */
int mfs_pressure_rhs3d(const int64_t gres[3], const double cell_size[3],
                       const void* vx, const void* vy, const void* vz, int v_dt,
                       const void* sv, int sv_dt, const void* lphi, int lphi_dt,
                       const void* wx, const void* wy, const void* wz, int w_dt,
                       void* b, int b_dt, mfs_stream stream);
/* replaces matvecmul -- solver/PressureCGSolver3D.py:52-130,161-165.
 * Ghost-fluid 7-point operator straight from lphi and the face weights; boundary
 * cells of `out` are not written.  v,out share `dt`.                            */
int mfs_pressure_apply3d(const int64_t gres[3], const void* v, void* out, int dt,
                         const void* wx, const void* wy, const void* wz, int w_dt,
                         const void* lphi, int lphi_dt, mfs_stream stream);
/* replaces apply_pressure -- solver/PressureCGSolver3D.py:132-153,167-171 (in place on vx,vy,vz) */
int mfs_pressure_update3d(const int64_t gres[3], const double cell_size[3],
                          void* vx, void* vy, void* vz, int v_dt, const void* pv, int pv_dt,
                          const void* wx, const void* wy, const void* wz, int w_dt,
                          const void* sv, int sv_dt, const void* lphi, int lphi_dt,
                          mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Pressure, 3D -- the CG engine (replaces the loop PressureCGSolver3D.py:198-223) */
/* ------------------------------------------------------------------------- */
typedef struct mfs_pcg3d mfs_pcg3d;

/* bytes of device workspace mfs_pcg3d_create needs for this grid / state dtype */
size_t mfs_pcg3d_workspace_bytes(const int64_t gres[3], int dt);
/* number of doubles in the residual-history buffer kept inside the workspace */
int64_t mfs_pcg3d_history_capacity(void);
/* host call. `workspace` must be 256-byte aligned device memory of at least
 * mfs_pcg3d_workspace_bytes(); it is zeroed asynchronously on `stream`.         */
int mfs_pcg3d_create(mfs_pcg3d** out_host, const int64_t gres[3], int dt,
                     void* workspace, size_t workspace_bytes, mfs_stream stream);
int mfs_pcg3d_destroy(mfs_pcg3d* h);
/* once per solve: fold lphi + face weights into the 4 coefficient arrays the
 * per-iteration stencil reads (diag, and fluid-masked lower-face weights).     */
int mfs_pcg3d_setup(mfs_pcg3d* h, const void* lphi, int lphi_dt,
                    const void* wx, const void* wy, const void* wz, int w_dt, mfs_stream stream);
/* out = A v on cell planes [x_begin, x_end) (clipped to the interior 1..Nx-2);
 * the per-iteration hot kernel.  Also leaves sum(v*out) partials in the handle. */
int mfs_pcg3d_apply(mfs_pcg3d* h, const void* v, void* out, int64_t x_begin, int64_t x_end,
                    mfs_stream stream);
/* CG vectors: caller-owned cell arrays of the handle's dtype (CGSolverBuffer.py:3-8 + solver.x) */
int mfs_pcg3d_bind(mfs_pcg3d* h, void* b, void* x, void* d, void* r, void* q);
/* x*=0; q=A x; d=r=b-q; delta0=sum r^2; done=(delta0<tol^2)   (PressureCGSolver3D.py:198-206) */
int mfs_pcg3d_begin(mfs_pcg3d* h, double tol, mfs_stream stream);
/* enqueue n CG iterations (lines 207-221); iterations after convergence are device-side no-ops */
int mfs_pcg3d_iterate(mfs_pcg3d* h, int64_t n, mfs_stream stream);
/* the two halves of ONE iteration of mfs_pcg3d_iterate's launch-per-phase loop (so a caller can bracket the stencil launch
 * with events): the stencil launch, then the x/r update and the direction update / its bookkeeping.  In the lean form
 * (mfs_pcg3d_set_lean) the bookkeeping of iteration j is done by the stencil launch of j + 1, or -- when a poll, history,
 * finish or another loop form comes first -- by a one-block launch those calls enqueue themselves. */
int mfs_pcg3d_native_apply(mfs_pcg3d* h, mfs_stream stream);
int mfs_pcg3d_native_finish(mfs_pcg3d* h, mfs_stream stream);
/* after mfs_pcg3d_iterate calls of one's own: settle what the fused loop forms still owe -- the deferred x update
 * (mfs_pcg3d_set_defer_x) and the direction vector parked in the engine's partner buffer -- so that x and d are the
 * reference's.  mfs_pcg3d_solve does this itself.  Host-synchronous; call once, when no more iterations follow. */
int mfs_pcg3d_finish(mfs_pcg3d* h, mfs_stream stream);
/* native fused loop: let `x += alpha d` ride in the NEXT iteration's stencil launch (which re-reads d from cache) so
 * that the x/r update kernel only streams r and q; same values; the last update is owed until mfs_pcg3d_finish /
 * the end of mfs_pcg3d_solve.  on < 0 = auto (default; env MFS_DEFER_X): only when the CG vectors exceed the
 * Infinity Cache.  A caller of mfs_pcg3d_iterate must therefore end with mfs_pcg3d_finish before reading x.   */
int mfs_pcg3d_set_defer_x(mfs_pcg3d* h, int on);
/* native fused loop without the deferred x update: mfs_pcg3d_iterate closes iteration j (convergence test :218, beta :220,
 * history) at the top of the stencil launch of iteration j + 1 -- every workgroup folds the r.r partials itself -- and
 * ends a batch with a one-block bookkeeping launch, instead of a reduction tail in the x/r update.  Same values, bit
 * for bit; shorter dependent chain per iteration (what bounds small grids).  on < 0 = auto (default; env MFS_LEAN). */
int mfs_pcg3d_set_lean(mfs_pcg3d* h, int on);
/* small grids (the reference notebook's 48x80x48 and the like): mfs_pcg3d_iterate runs a whole batch of iterations
 * (PressureCGSolver3D.py:207-221) as ONE launch of <= 64 resident workgroups that keep x, r, d, q in registers and
 * exchange the two dot products and the box faces through self-validating records in the workspace -- no kernel
 * boundary per reduction.  Same arithmetic per cell; the grouping of the dot products differs from the launch-per-phase
 * loops, so results agree with them to rounding.  on < 0 = auto (default; env MFS_RESIDENT): whenever the grid fits. */
int mfs_pcg3d_set_resident(mfs_pcg3d* h, int on);
/* synchronises `stream`, then reports the device-resident solver state. host call. */
int mfs_pcg3d_poll(mfs_pcg3d* h, mfs_stream stream, int64_t* iters_host, int* done_host,
                   double* delta_host, double* alpha_host, double* beta_host);
/* begin + iterate/poll until done or max_iter; MFS_OK or MFS_NOT_CONVERGED. host-synchronous.  check_every = iterations
 * enqueued between two looks at the convergence flag (the resident small-grid loop, which stops by itself inside a
 * batch, uses max(check_every, 128)). */
int mfs_pcg3d_solve(mfs_pcg3d* h, double tol, int64_t max_iter, int64_t check_every,
                    mfs_stream stream, int64_t* iters_host);
/* copies [delta0, dq1, delta1, dq2, delta2, ...] (first `cap` values) to the host; returns count or <0 */
int64_t mfs_pcg3d_history(mfs_pcg3d* h, double* out_host, int64_t cap, mfs_stream stream);

/* -- single phases of one iteration, for the slab-decomposed multi-GPU driver.
 * Order per iteration k:  [halo exchange of d] phase_apply(ranges...) ->
 * phase_reduce(0) -> [all-reduce scalars[DQ]] -> phase_update_xr ->
 * phase_reduce(1) -> [all-reduce scalars[RR]] -> phase_update_d
 * (or phase_update_r -> phase_reduce(1) -> [all-reduce RR || phase_update_x] -> phase_update_d). */
int mfs_pcg3d_phase_apply(mfs_pcg3d* h, int64_t x_begin, int64_t x_end, int first, mfs_stream stream);
/* two disjoint plane ranges in one launch (the driver's two edge planes) */
int mfs_pcg3d_phase_apply2(mfs_pcg3d* h, int64_t x_begin, int64_t x_end, int64_t x_begin2, int64_t x_end2,
                           int first, mfs_stream stream);
int mfs_pcg3d_phase_reduce(mfs_pcg3d* h, int which, mfs_stream stream);
int mfs_pcg3d_phase_update_xr(mfs_pcg3d* h, mfs_stream stream);
/* the same update as two kernels: `r -= a q` (+ r.r partials) and `x += a d`, so that the
 * driver can start the r.r all-reduce after the first and overlap it with the second */
int mfs_pcg3d_phase_update_r(mfs_pcg3d* h, mfs_stream stream);
int mfs_pcg3d_phase_update_x(mfs_pcg3d* h, mfs_stream stream);
int mfs_pcg3d_phase_update_d(mfs_pcg3d* h, mfs_stream stream);
/* begin, split around the one reduction it contains: begin_local -> [all-reduce RR] -> begin_finish */
int mfs_pcg3d_begin_local(mfs_pcg3d* h, double tol, mfs_stream stream);
int mfs_pcg3d_begin_finish(mfs_pcg3d* h, mfs_stream stream);
/* device pointer to the engine's double[MFS_PCG_NSCALARS] block and slot indices */
#define MFS_PCG_NSCALARS 16
#define MFS_PCG_S_DQ 0      /* sum d.q of the current iteration */
#define MFS_PCG_S_RR 1      /* sum r.r produced by the latest update (delta_new) */
#define MFS_PCG_S_DELTA 2   /* delta the current iteration started from */
#define MFS_PCG_S_TOL2 3
#define MFS_PCG_S_DONE 4
#define MFS_PCG_S_ITERS 5
#define MFS_PCG_S_ALPHA 6
#define MFS_PCG_S_BETA 7
#define MFS_PCG_S_LASTRR 8  /* r.r of the last completed iteration (what the reference keeps in self.delta) */
#define MFS_PCG_S_ERR 11    /* != 0: the device loop stopped itself: 1 / 2 a peer-to-peer wait of the slab loop timed out,
                               3 d.q == 0, 4 non-finite d.q or r.r */
#define MFS_PCG_S_LANE 13   /* diagnostics: 1 when the solve's listed launches mask dead vectors lane by lane (mfs_pcg3d_set_sparse) */
void* mfs_pcg3d_scalars(mfs_pcg3d* h);
/* performance knobs of the stencil kernel (results are identical for every setting):
 * variant 0 = direct loads, 1 = x-marching in registers, 2 = x-marching + LDS-staged
 * plane tiles (default); xchunk = cap on the planes of one march (0 = none);
 * blocks_per_cu = workgroups per CU the (tile, plane) work is cut into;
 * nontemporal > 0 marks the once-read coefficient streams, 0 never, < 0 = auto
 * (on when the six arrays of one apply exceed the Infinity Cache).                */
int mfs_pcg3d_tune(mfs_pcg3d* h, int variant, int xchunk, int blocks_per_cu, int nontemporal);
/* compressed coefficient access (default on): the per-iteration kernel reads the four
 * coefficient arrays only for z-vectors that are neither all-zero rows nor regular
 * interior rows (class byte per vector built by mfs_pcg3d_setup); results are bit-identical */
int mfs_pcg3d_set_compress(mfs_pcg3d* h, int on);
/* form of the native loop for the engine as bound: bit 0 = direction update fused into the stencil launch,
 * bit 1 = x update deferred into it too, bit 2 = Jacobi loop, bit 3 = resident small-grid loop (mfs_pcg3d_set_resident) */
int mfs_pcg3d_loop_info(mfs_pcg3d* h);
/* sparse lists of a single-domain solve (round 3; default on from 2^21 cells, env MFS_SPARSE / MFS_SPARSE_MIN): behind the
 * initial residual mfs_pcg3d_begin lists the 32-cell chunks holding a live z-vector (row not ZERO, or r, d != 0) and the
 * (tile, plane) pairs of the march holding one; the r update sweeps the listed chunks, the fused stencil launches visit the
 * listed pairs.  Dead vectors keep q = r = d = +0 and x unchanged, which is what the dense loop computes for them; the dot
 * products group differently (rounding).  The window / collective slab loops (mfs_pcg3d_slab_begin) build the same lists for
 * their owned planes and their interior launch, and so does the fused single-domain Jacobi loop (z = r / diag is 0 wherever r
 * is); begin_local / phase callers, the slab and the three-launch Jacobi loops stay dense.  Where more than
 * a quarter of the listed pairs' vectors are dead, the listed launches also mask those vectors' loads and stores lane by lane
 * (decided per solve on the device, picked up by the host with its first look at the scalar block: slot 13, diagnostics).
 * mfs_pcg3d_sparse_info (host-synchronous): out = {listed chunks, chunks, listed pairs, pairs}; zeros where a list is off. */
int mfs_pcg3d_set_sparse(mfs_pcg3d* h, int on);
int mfs_pcg3d_sparse_info(mfs_pcg3d* h, mfs_stream stream, int64_t out[4]);
/* OPT-IN Jacobi preconditioning of the native loop (default off; env MFS_JACOBI=1): delta = r.z with z = r / diag,
 * convergence test unchanged (r.r < tol^2).  Fused form (default where the fused direction update is available): the r
 * update stores z and closes the iteration, the next stencil launch forms d = z + beta d (2 launches per iteration);
 * otherwise z is formed inside the two vector phases (3 launches).  NOT the reference's algorithm -- the reference's CG
 * is unpreconditioned and its residual history cannot be matched with this on.  Runs in the single-GPU loop and in the
 * WINDOW slab loop (mfs_pcg3d_slab_*: z as the operand of the edge / interior direction updates, r.r and r.z all-reduced
 * in the tail of the r / z update); the phase API of the collective loop has no Jacobi form. */
int mfs_pcg3d_set_jacobi(mfs_pcg3d* h, int on);
/* fused direction update (default on, native loop only): `d = r + beta d` is formed inside the next
 * stencil launch instead of in a pass of its own; bit-identical; d ping-pongs with an engine buffer */
int mfs_pcg3d_set_fuse(mfs_pcg3d* h, int on);
/* planes of the operand stream the LDS march keeps in flight ahead of the plane it computes (1 or 2) */
int mfs_pcg3d_set_prefetch(mfs_pcg3d* h, int planes);

/* ------------------------------------------------------------------------- */
/* Pressure, 3D -- slab-decomposed CG over peer-to-peer windows (multi-GPU)     */
/* ------------------------------------------------------------------------- */
/* New design (the reference is single-GPU): one process per GPU, the grid cut into
 * x-slabs (mfs/dist.py).  A window is a block of uncached device memory per rank that
 * the other ranks of the node map through HIP IPC; inside the CG loop the edge planes of
 * the direction vector and the two dot products travel as plain xGMI stores issued by
 * the solver's own kernels (csrc/mfs_p2p.h, csrc/mfs_pcg_slab.h) -- no host round trip
 * and no collective call per iteration.  EXCEPTION to "nothing allocates behind the
 * ABI": the window is allocated here, because IPC-exportable uncached memory cannot be
 * carved from a caller's sub-allocator.                                             */
typedef struct mfs_p2p mfs_p2p;
/* size of the opaque IPC handle mfs_p2p_create writes (hipIpcMemHandle_t) */
size_t mfs_p2p_handle_bytes(void);
/* host call: allocate this rank's window for halo planes of `plane_bytes` (= Ny*Nz*sizeof
 * element) and write its IPC handle to handle_out_host (mfs_p2p_handle_bytes() bytes)  */
int mfs_p2p_create(mfs_p2p** out_host, int rank, int world, size_t plane_bytes, void* handle_out_host);
/* host call: map the peers' windows; handles_host = world handles in rank order (all-gathered by the caller) */
int mfs_p2p_connect(mfs_p2p* p, const void* handles_host);
/* COLLECTIVE, host-synchronous: every rank sends a patterned plane to its neighbours, takes part
 * in one all-reduce through the windows and verifies what it received.  ok_host = 1 only if
 * everything arrived intact within the time limit; detail_host (4 words, optional): ok, payload
 * vectors missing or wrong, all-reduce timed out, all-reduce sum as float bits                  */
int mfs_p2p_selftest(mfs_p2p* p, int round, mfs_stream stream, int* ok_host, unsigned* detail_host);
/* alloc_kind: 1 = hipDeviceMallocUncached, 2 = hipDeviceMallocFinegrained */
int mfs_p2p_info(mfs_p2p* p, int* alloc_kind_host, size_t* window_bytes_host);
/* host call; every rank must have finished using the windows (barrier first) */
int mfs_p2p_destroy(mfs_p2p* p);

/* let the engine run its slab loop over this window (null detaches) */
int mfs_pcg3d_attach_p2p(mfs_pcg3d* h, mfs_p2p* p);
/* the slab loop: begin / iterate / solve with the semantics of mfs_pcg3d_begin / _iterate /
 * _solve on the GLOBAL grid; COLLECTIVE -- every rank of the window calls them in step.
 * The engine's grid is this rank's slab incl. one ghost / boundary plane each side.
 * A peer that does not answer within MFS_P2P_TIMEOUT_MS (default 10000) stops the solve:
 * the next poll / solve returns MFS_E_TIMEOUT.                                          */
/* 1 if the slab loop can run on this engine (window attached, CG vectors bound and 16-byte aligned, Nz a multiple
 * of the 16-byte vector length, stencil variant 2, pressure operator); else the caller uses the collective loop */
/* Collective transport for the slab loop (round 3): the same launches as the window loop (edge d, fused interior march,
 * edge apply + d.q, r update + r.r) with NO in-kernel exchange -- the two edge planes travel by ncclSend / ncclRecv on the
 * solver's second stream beside the interior launch, each dot product is ONE ncclAllReduce on the device scalar block, all
 * enqueued from C (no host synchronisation, no Python inside a batch).  RCCL is resolved at run time from `lib_path` (the
 * librccl the process already maps).  mfs_rccl_unique_id on rank 0 -> every rank -> mfs_rccl_create (collective); then
 * mfs_pcg3d_attach_rccl with the rank's OWN one-rank window (mfs_p2p_create(&w, 0, 1, ...)); mfs_pcg3d_slab_begin / _iterate /
 * _solve run the loop.  The reference is single-GPU; this is the fallback where HIP-IPC windows fail. */
typedef struct mfs_rccl mfs_rccl;
int mfs_rccl_unique_id_bytes(void);
int mfs_rccl_unique_id(const char* lib_path, void* id_out);
int mfs_rccl_create(mfs_rccl** out_host, const char* lib_path, const void* id, int rank, int world);
int mfs_rccl_destroy(mfs_rccl* r);
int mfs_pcg3d_attach_rccl(mfs_pcg3d* h, mfs_p2p* own_window, mfs_rccl* comm);
int mfs_pcg3d_slab_supported(mfs_pcg3d* h);
/* slab loop tuning: send the edge planes from a second HIP stream so that the xGMI stores overlap the interior
 * stencil launch (costs two cross-stream event hops per iteration; default: on for planes >= 1 MiB of granules) */
int mfs_pcg3d_slab_set_aux(mfs_pcg3d* h, int on);
int mfs_pcg3d_slab_begin(mfs_pcg3d* h, double tol, mfs_stream stream);
int mfs_pcg3d_slab_iterate(mfs_pcg3d* h, int64_t n, mfs_stream stream);
int mfs_pcg3d_slab_solve(mfs_pcg3d* h, double tol, int64_t max_iter, int64_t check_every,
                         mfs_stream stream, int64_t* iters_host);

/* ------------------------------------------------------------------------- */
/* Viscosity, 3D -- stateless kernels (the reference's module-level functions) */
/* ------------------------------------------------------------------------- */
/* replaces extrapolate -- solver/ViscosityCGSolver3D.py:8-39,472-502 (in place on vx,vy,vz;
 * `workspace` holds the ping-pong copies and validity masks)                     */
size_t mfs_visc_extrapolate3d_workspace_bytes(const int64_t gres[3], int v_dt);
int mfs_visc_extrapolate3d(const int64_t gres[3], int num_iter, void* vx, void* vy, void* vz, int v_dt,
                           const void* sphi, int sphi_dt, void* workspace, size_t workspace_bytes,
                           mfs_stream stream);
/* the two pieces of extrapolate, for callers that act between sweeps (slab decomposition: the ghost planes of
 * values and validity travel after every sweep).  comp = 0/1/2 (vx/vy/vz); valid = one byte per face:
 * `valid = sphi(face) >= 0` (:479-481), then one Jacobi sweep old -> new (:8-39, the loop body of :483-502) */
int mfs_visc_valid3d(const int64_t gres[3], int comp, const void* sphi, int sphi_dt, unsigned char* valid,
                     mfs_stream stream);
int mfs_visc_extrapolate_sweep3d(const int64_t gres[3], int comp, const void* v_in, void* v_out, int v_dt,
                                 const unsigned char* valid_in, unsigned char* valid_out, mfs_stream stream);
/* replaces initialize_solver -- solver/ViscosityCGSolver3D.py:41-246,504-513.
 * `vol` is the doubled-grid fluid volume already divided by cell_vol/8 (self.vol, :568) */
int mfs_visc_rhs3d(const int64_t gres[3], double scale, double mu,
                   const void* vx, const void* vy, const void* vz, int v_dt,
                   const void* sphi, int sphi_dt, const void* vol, int vol_dt,
                   void* b_x, void* b_y, void* b_z, int b_dt, mfs_stream stream);
/* replaces matvecmul -- solver/ViscosityCGSolver3D.py:248-456,515-524 */
int mfs_visc_apply3d(const int64_t gres[3], double scale, double mu,
                     const void* vx, const void* vy, const void* vz, int v_dt,
                     void* out_x, void* out_y, void* out_z, int out_dt,
                     const void* sphi, int sphi_dt, const void* vol, int vol_dt, mfs_stream stream);
/* replaces apply_viscosity -- solver/ViscosityCGSolver3D.py:458-470,526-530 (in place on vx,vy,vz) */
int mfs_visc_writeback3d(const int64_t gres[3], void* vx, void* vy, void* vz, int v_dt,
                         const void* out_x, const void* out_y, const void* out_z, int out_dt,
                         const void* sphi, int sphi_dt, mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Viscosity, 3D -- the CG engine (replaces the loop ViscosityCGSolver3D.py:575-612) */
/* ------------------------------------------------------------------------- */
/* The five CG vectors are FLAT arrays of mfs_vcg3d_dofs() elements laid out
 * [ x-faces (Nx+1,Ny,Nz) | y-faces (Nx,Ny+1,Nz) | z-faces (Nx,Ny,Nz+1) ]; the
 * reference's per-component arrays (x_x, x_y, x_z, ...) are views into them.       */
typedef struct mfs_vcg3d mfs_vcg3d;
size_t mfs_vcg3d_workspace_bytes(const int64_t gres[3], int dt);
int64_t mfs_vcg3d_dofs(const int64_t gres[3]);
int mfs_vcg3d_create(mfs_vcg3d** out_host, const int64_t gres[3], int dt,
                     void* workspace, size_t workspace_bytes, mfs_stream stream);
int mfs_vcg3d_destroy(mfs_vcg3d* h);
/* once per solve: de-interleave the doubled-grid `vol` into its 7 parity classes and
 * `sphi >= 0` at the 3 face classes into byte masks (unit stride for the iteration) */
int mfs_vcg3d_setup(mfs_vcg3d* h, double scale, double mu, const void* sphi, int sphi_dt,
                    const void* vol, int vol_dt, mfs_stream stream);
/* out = A v on flat vectors; the per-iteration kernel(s) */
int mfs_vcg3d_apply(mfs_vcg3d* h, const void* v, void* out, mfs_stream stream);
int mfs_vcg3d_bind(mfs_vcg3d* h, void* b, void* x, void* d, void* r, void* q);
/* q=A x (x = extrapolated velocity, NOT zeroed); d=r=b-q; delta0   (:575-587) */
int mfs_vcg3d_begin(mfs_vcg3d* h, double tol, mfs_stream stream);
/* n iterations of the loop :588-610: stencil launch, r update, direction + x update.  Opt-in form (mfs_vcg3d_set_fuse,
 * where the x-marching kernel serves the engine -- mfs_vcg3d_loop_info bit 0): 2 launches per iteration -- the stencil launch of iteration j also forms d_j = r + beta d_{j-1} (:609-610) and
 * performs x += alpha d_{j-1} (:595-597) for the faces it owns, the r update's last block closes the iteration (:604-608).
 * x then lags one update behind and d_j may sit in an engine buffer until mfs_vcg3d_finish / the end of mfs_vcg3d_solve. */
int mfs_vcg3d_iterate(mfs_vcg3d* h, int64_t n, mfs_stream stream);
/* settles what mfs_vcg3d_iterate's fused loop owes (last x update; d_iters = r + beta d unless converged, brought home to
 * the bound `d`): state as after the reference's loop :588-610.  Host-synchronous; begin again before iterating further. */
int mfs_vcg3d_finish(mfs_vcg3d* h, mfs_stream stream);
/* 1 / 0: fused direction + x update in mfs_vcg3d_iterate / solve (default 0 -- measured slower, DESIGN.md section 4;
 * env MFS_VISC_FUSE); results bit-identical */
int mfs_vcg3d_set_fuse(mfs_vcg3d* h, int on);
/* compressed class access of the x-marching kernel (default on; env MFS_VISC_COMPRESS): per z-vector and plane a class
 * (built by mfs_vcg3d_setup, stored in spare bits of the packed mask bytes) says whether every volume sample the
 * vector's step loads is 0 (air), 1 (bulk liquid) or anything else; only the last kind reads the seven class arrays
 * (lvol de-interleaved, ViscosityCGSolver3D.py:248-456 reads it at 7 of 8 doubled-grid parities).  Bit-identical. */
int mfs_vcg3d_set_compress(mfs_vcg3d* h, int on);
/* census of those classes after mfs_vcg3d_setup: counts_host[0] = z-vectors whose samples are all 0, [1] = all 1,
 * [2] = the rest (the only ones that read the class arrays).  Diagnostic, host-synchronous. */
int mfs_vcg3d_class_census(mfs_vcg3d* h, int64_t counts_host[3], mfs_stream stream);
/* sparse lists of a single-domain solve (round 3; default on from 2^21 unknowns, env MFS_VISC_SPARSE / MFS_VISC_SPARSE_MIN): the
 * r and d / x updates sweep the 32-unknown chunks holding a face whose row is not empty (or r, d != 0), and -- with the
 * compressed class access -- the march launches of mfs_vcg3d_iterate visit only the (tile, plane) pairs that are not all air
 * (their q = +0 was stored by the solve's initial q = A x; nothing else writes it).  Values as in the dense loop; the dot
 * products group differently.  The window slab loop and the single-domain Jacobi loop take the lists too; phase callers, the
 * opt-in fused loop and the slab Jacobi loop stay dense.
 * mfs_vcg3d_sparse_info (host-synchronous): out = {listed chunks, chunks, listed pairs, pairs} of the solve begun last. */
int mfs_vcg3d_set_sparse(mfs_vcg3d* h, int on);
int mfs_vcg3d_sparse_info(mfs_vcg3d* h, mfs_stream stream, int64_t out[4]);
/* bit 0: mfs_vcg3d_iterate will run the fused 2-launch loop for the engine as bound and set up; bit 1: the small-problem
 * loop -- the r update (:592-601), the r.r reduction, the test / bookkeeping (:604-608) and the x / direction updates
 * (:595-597, :609-610) in ONE launch whose resident workgroups exchange their partial sums (csrc/mfs_cg_core.h
 * k_update_rdx): 2 launches per iteration, results equal to the three-launch loop's up to the grouping of r.r */
int mfs_vcg3d_loop_info(mfs_vcg3d* h);
/* OPT-IN Jacobi preconditioning of the viscosity loop (default off; env MFS_VISC_JACOBI=1): z = r / diag with the operator's
 * own diagonal (vol_c + scale mu (...), :268 / :338 / :408; built once per solve), delta = r.z, convergence test unchanged
 * (r.r < tol^2).  NOT the reference's iteration (ViscosityCGSolver3D.py:575-612 is unpreconditioned): another residual
 * history, the same solution to the tolerance, far fewer iterations where partly filled cells make the diagonal span orders
 * of magnitude.  Single GPU (mfs_vcg3d_begin / iterate / solve) and the window slab loop (mfs_vcg3d_slab_begin / slab_iterate:
 * r.z all-reduced as a third episode); bit 2 of mfs_vcg3d_loop_info. */
int mfs_vcg3d_set_jacobi(mfs_vcg3d* h, int on);
/* 1 / 0: allow that small-problem loop (default 1; env MFS_RDX).  A launch that is not fully resident (shared GPU) times
 * out without having written anything; the next poll switches the engine to the three-launch loop for good. */
int mfs_vcg3d_set_merged(mfs_vcg3d* h, int on);
/* 1 / 0: allow the RESIDENT small-grid loop (default on where the grid qualifies; env MFS_VISC_RESIDENT): the loop
 * ViscosityCGSolver3D.py:588-610 as ONE launch per mfs_vcg3d_iterate batch -- x, r, d, q of every face in the registers of
 * up to 128 resident workgroups, d with its halo in LDS, dot products and halo faces exchanged as self-validating records
 * (csrc/mfs_vcg_resident.h).  Same arithmetic per element; the dot products group differently, so histories agree with the
 * launch-per-phase loop to rounding.  A launch that is not fully resident (shared GPU) times out at its first dot product
 * without having written anything; the next poll switches the engine to the launch-per-phase loop for good.
 * Bit 3 of mfs_vcg3d_loop_info. */
int mfs_vcg3d_set_resident(mfs_vcg3d* h, int on);
int mfs_vcg3d_poll(mfs_vcg3d* h, mfs_stream stream, int64_t* iters_host, int* done_host,
                   double* delta_host, double* alpha_host, double* beta_host);
int mfs_vcg3d_solve(mfs_vcg3d* h, double tol, int64_t max_iter, int64_t check_every,
                    mfs_stream stream, int64_t* iters_host);
int64_t mfs_vcg3d_history(mfs_vcg3d* h, double* out_host, int64_t cap, mfs_stream stream);
/* Which kernel the CG applies (matvecmul_{x,y,z}_kernel, ViscosityCGSolver3D.py:248-456, on the direction vector)
 * take for the engine as bound: 2 = x-marching 16-byte-vector kernel (csrc/mfs_vcg_march.h; needs Nz % 4 == 0 (fp32) /
 * Nz % 2 == 0 (fp64) and 16-byte aligned vectors), 1 = LDS-tiled one-cell-per-lane kernel (MFS_VISC_TILED=1),
 * 0 = one-cell-per-lane direct-load kernel.  All three give bit-identical results.                              */
int mfs_vcg3d_apply_kernel(mfs_vcg3d* h);
/* Slab decomposition along x (the build's extension; host driver mfs/dist.py:SlabVCG): one
 * iteration = halo exchange of d's edge planes (3 components), phase_apply, phase_reduce(0), all-reduce of
 * scalars[MFS_PCG_S_DQ], phase_update_xr, phase_reduce(1), all-reduce of scalars[MFS_PCG_S_RR], phase_update_d
 * -- the same phases and scalar block (device pointer: mfs_vcg3d_scalars) as mfs_pcg3d_phase_*.
 * set_slab(skip_top_x = 1) on every rank but the last: its last u plane is the right neighbour's.       */
int mfs_vcg3d_set_slab(mfs_vcg3d* h, int skip_top_x);
void* mfs_vcg3d_scalars(mfs_vcg3d* h);
int mfs_vcg3d_begin_local(mfs_vcg3d* h, double tol, mfs_stream stream);
int mfs_vcg3d_begin_finish(mfs_vcg3d* h, mfs_stream stream);
int mfs_vcg3d_phase_apply(mfs_vcg3d* h, mfs_stream stream);
int mfs_vcg3d_phase_reduce(mfs_vcg3d* h, int which, mfs_stream stream);
int mfs_vcg3d_phase_update_xr(mfs_vcg3d* h, mfs_stream stream);
int mfs_vcg3d_phase_update_d(mfs_vcg3d* h, mfs_stream stream);
/* The same loop with the halo planes and the two dot products moving through a peer-to-peer window (mfs_p2p_*, below /
 * above): attach a connected window created with plane_bytes = (Ny*Nz + (Ny+1)*Nz + Ny*(Nz+1)) * sizeof(element);
 * slab_begin / slab_iterate then enqueue whole iterations without any host-side exchange.  Poll as usual.   */
int mfs_vcg3d_attach_p2p(mfs_vcg3d* h, mfs_p2p* window);
int mfs_vcg3d_slab_begin(mfs_vcg3d* h, double tol, mfs_stream stream);
int mfs_vcg3d_slab_iterate(mfs_vcg3d* h, int64_t n, mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Density solver, 3D -- reference solver/DensityCGSolver3D.py */
/* ------------------------------------------------------------------------- */
/* The CG loop of DensityCGSolver3D.solve (:318-345) runs on the pressure engine: call
 * mfs_pcg3d_setup_density instead of mfs_pcg3d_setup, then bind / solve as for pressure.
 * Its operator (matvecmul_kernel :118-207) is the pressure stencil with diag counting 1 per
 * fluid neighbour and the -z tap weighted by wz[x,y,z+1] (kept as written, :184).            */
int mfs_pcg3d_setup_density(mfs_pcg3d* h, const void* lphi, int lphi_dt,
                            const void* wx, const void* wy, const void* wz, int w_dt, mfs_stream stream);
/* replaces initialize_density -- :8-36,255-260: scatters particle mass pm[p] and the (scalar) particle
 * volume to the 8 surrounding cell centres with fp atomics; px is (P,3) row-major; gm, gvol share g_dt */
int mfs_density_splat3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                        const void* px, int px_dt, const void* pm, int pm_dt, double pvol, int64_t num_particles,
                        void* gm, void* gvol, int g_dt, mfs_stream stream);
/* replaces fix_volume -- :38-86,262-269 (in place on gvol; the reference's `lvol` argument is unused there) */
int mfs_density_fix_volume3d(const int64_t gres[3], const double cell_size[3], void* gvol, int g_dt,
                             const void* sphi, int sphi_dt, const void* lphi, int lphi_dt,
                             const void* wx, const void* wy, const void* wz, int w_dt, mfs_stream stream);
/* replaces initialize_solver -- :88-116,271-277 */
int mfs_density_rhs3d(const int64_t gres[3], double rho0, double dt, const double cell_size[3],
                      const void* gm, const void* gvol, int g_dt, const void* lphi, int lphi_dt,
                      const void* wx, const void* wy, const void* wz, int w_dt, void* b, int b_dt, mfs_stream stream);
/* replaces matvecmul -- :118-207,279-283 (stateless; boundary cells of `out` are not written) */
int mfs_density_apply3d(const int64_t gres[3], const void* v, void* out, int dt,
                        const void* wx, const void* wy, const void* wz, int w_dt,
                        const void* lphi, int lphi_dt, mfs_stream stream);
/* replaces compute_displacement -- :209-222,285-289: face displacements from the solved field pv */
int mfs_density_displacement3d(const int64_t gres[3], double dt, const double cell_size[3],
                               void* dx, void* dy, void* dz, int d_dt, const void* pv, int pv_dt,
                               const void* lphi, int lphi_dt, mfs_stream stream);
/* replaces apply_displacement -- :224-253,291-296: px[:, axis] += trilinear sample of the face array d
 * (shape dshape, samples at (index + grid_bias) * cell_size + bound_min)                              */
int mfs_density_advect3d(void* px, int px_dt, int64_t num_particles, const void* d, int d_dt,
                         const int64_t dshape[3], const double bound_min[3], const double cell_size[3],
                         const double grid_bias[3], int axis, mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Pressure, 2D (BASELINE config 1) -- reference solver/PressureCGSolver2D.py    */
/* ------------------------------------------------------------------------- */
/* replaces initialize_solver -- solver/PressureCGSolver2D.py:6-44,122-126 */
int mfs_pressure_rhs2d(const int64_t gres[2], const double cell_size[2],
                       const void* vx, const void* vy, int v_dt, const void* sv, int sv_dt,
                       const void* lphi, int lphi_dt, const void* wx, const void* wy, int w_dt,
                       void* b, int b_dt, mfs_stream stream);
/* replaces matvecmul -- solver/PressureCGSolver2D.py:46-100,128-132 */
int mfs_pressure_apply2d(const int64_t gres[2], const void* v, void* out, int dt,
                         const void* wx, const void* wy, int w_dt, const void* lphi, int lphi_dt,
                         mfs_stream stream);
/* replaces apply_pressure -- solver/PressureCGSolver2D.py:102-120,134-138 */
int mfs_pressure_update2d(const int64_t gres[2], const double cell_size[2], void* vx, void* vy, int v_dt,
                          const void* pv, int pv_dt, const void* wx, const void* wy, int w_dt,
                          const void* sv, int sv_dt, const void* lphi, int lphi_dt, mfs_stream stream);
/* the CG loop solver/PressureCGSolver2D.py:159-177 (no error on non-convergence in the reference:
 * mfs_pcg2d_solve returns MFS_NOT_CONVERGED and the caller carries on)                             */
typedef struct mfs_pcg2d mfs_pcg2d;
size_t mfs_pcg2d_workspace_bytes(const int64_t gres[2], int dt);
int mfs_pcg2d_create(mfs_pcg2d** out_host, const int64_t gres[2], int dt,
                     void* workspace, size_t workspace_bytes, mfs_stream stream);
int mfs_pcg2d_destroy(mfs_pcg2d* h);
int mfs_pcg2d_setup(mfs_pcg2d* h, const void* lphi, int lphi_dt, const void* wx, const void* wy, int w_dt);
int mfs_pcg2d_bind(mfs_pcg2d* h, void* b, void* x, void* d, void* r, void* q);
int mfs_pcg2d_solve(mfs_pcg2d* h, double tol, int64_t max_iter, int64_t check_every,
                    mfs_stream stream, int64_t* iters_host);
int mfs_pcg2d_poll(mfs_pcg2d* h, mfs_stream stream, int64_t* iters_host, int* done_host,
                   double* delta_host, double* alpha_host, double* beta_host);
int64_t mfs_pcg2d_history(mfs_pcg2d* h, double* out_host, int64_t cap, mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Notebook grid kernels that bracket the two solves    */
/* ------------------------------------------------------------------------- */
/* replaces `extrapolate(gres, num_iter, vx, vy, vz, mx, my, mz)` -- 3D_viscous_fluid_sim.ipynb code cell 7
 * (call site ipynb:4652): validity = grid mass > 0.  Workspace as for mfs_visc_extrapolate3d.          */
int mfs_grid_extrapolate3d(const int64_t gres[3], int num_iter, void* vx, void* vy, void* vz, int v_dt,
                           const void* mx, const void* my, const void* mz, int m_dt,
                           void* workspace, size_t workspace_bytes, mfs_stream stream);
/* replaces the three kernels of `apply_boundary_condition(g, solid, dx)` -- code cell 5 (call site
 * ipynb:4655): writes the free-slip corrections dv_x, dv_y, dv_z (the caller adds them to the velocities) */
int mfs_grid_boundary_condition3d(const int64_t gres[3], const void* gvx, const void* gvy, const void* gvz, int v_dt,
                                  const void* gmx, const void* gmy, const void* gmz, int m_dt,
                                  const void* sphi, int sphi_dt, const void* sv, int sv_dt, double dx,
                                  void* dvx, void* dvy, void* dvz, int dv_dt, mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Notebook particle <-> grid transfers                  */
/* ------------------------------------------------------------------------- */
/* Particle arrays: px, pv, pca are (P,3) row-major, pm is (P).  The reference kernels keep float32
 * locals whatever the array dtypes; `bound_min` and `grid_bias` are therefore used at float32 precision
 * and `cell_size` at float64 -- the notebook's container dtypes (code cell 9) -- so that base indices and
 * weights equal the reference's.  Scatters use fp atomics (order unspecified, as in the reference).     */
/* replaces p2g_particle -- 3D_viscous_fluid_sim.ipynb code cell 2: APIC scatter of mass and momentum of velocity
 * component `axis` to its face array (shape gres + e_axis; indices are clamped to gres - 1 as in the reference) */
int mfs_p2g_scatter3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                      const double grid_bias[3], int axis, const void* px, int px_dt, const void* pm, int pm_dt,
                      const void* pv, int pv_dt, const void* pca, int pca_dt, int64_t num_particles,
                      void* gm, void* gv, int g_dt, mfs_stream stream);
/* Tile-sorted forms of the three scatters (the build's own; same results up to the order of the fp atomics, which the
 * reference leaves unspecified too).  mfs_particle_tile_sort3d buckets the particles by the tile of 8^3 CELLS their cell
 * (floor((x - bound_min) / cell_size), clamped to gres) lies in: perm[num_particles] = particle indices tile by tile,
 * tile_start[mfs_particle_tiles3d(gres) + 1] = segment starts; work = 2 * tiles + num_particles int32 of scratch.  The
 * *_tiled entry points take that order and run one workgroup per tile with the tile's nodes in LDS (csrc/mfs_particles.hip);
 * they stay correct when particles have moved since the sort (contributions outside the staged nodes take the global
 * atomic), only slower.  p2g_particle / compute_fls_kernel / compute_fluid_volume_kernel: ipynb code cells 2, 4, 6. */
int64_t mfs_particle_tiles3d(const int64_t gres[3]);
int mfs_particle_tile_sort3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3], const void* px,
                             int px_dt, int64_t num_particles, int32_t* perm, int32_t* tile_start, int32_t* work,
                             mfs_stream stream);
int mfs_p2g_scatter3d_tiled(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                            const double grid_bias[3], int axis, const void* px, int px_dt, const void* pm, int pm_dt,
                            const void* pv, int pv_dt, const void* pca, int pca_dt, int64_t num_particles,
                            const int32_t* perm, const int32_t* tile_start, void* gm, void* gv, int g_dt, mfs_stream stream);
int mfs_fluid_levelset3d_tiled(const int64_t gres[3], const double bound_min[3], const double cell_size[3], double radius,
                               const void* px, int px_dt, int64_t num_particles, const int32_t* perm,
                               const int32_t* tile_start, void* phi, int phi_dt, mfs_stream stream);
/* initialize_density_kernel (solver/DensityCGSolver3D.py:8-36) on the same tile order */
int mfs_density_splat3d_tiled(const int64_t gres[3], const double bound_min[3], const double cell_size[3], const void* px,
                              int px_dt, const void* pm, int pm_dt, double pvol, int64_t num_particles, const int32_t* perm,
                              const int32_t* tile_start, void* gm, void* gvol, int g_dt, mfs_stream stream);
int mfs_fluid_volume3d_tiled(const int64_t vres[3], const double bound_min[3], const double cell_size[3], const void* px,
                             int px_dt, double pvol, int64_t num_particles, const int32_t* perm, const int32_t* tile_start,
                             void* gvol, int g_dt, mfs_stream stream);
/* replaces p2g_grid -- code cell 2: gv /= gm wherever gm > 0 (count = elements of the face array) */
int mfs_p2g_normalize3d(int64_t count, const void* gm, void* gv, int g_dt, mfs_stream stream);
/* replaces g2p_particle -- code cell 3: pv[:, axis] and the affine row pca[:, :] from the face array gv */
int mfs_g2p_gather3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                     const double grid_bias[3], int axis, const void* px, int px_dt, void* pv, int pv_dt,
                     void* pca, int pca_dt, int64_t num_particles, const void* gv, int g_dt, mfs_stream stream);
/* replaces compute_fls_kernel -- code cell 4: phi = min(phi, |cell centre - x| - radius) over the 5^3 cells around
 * each particle (atomic min).  The caller pre-fills phi (the notebook: `ls.phi[:] = gdx * 3`).               */
int mfs_fluid_levelset3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3], double radius,
                         const void* px, int px_dt, int64_t num_particles, void* phi, int phi_dt, mfs_stream stream);
/* replaces compute_fluid_volume_kernel + constrain_fluid_volume_kernel -- code cell 6: trilinear splat of the
 * particle volume onto the nodes of the array `gvol` (shape vres, spacing cell_size), then min(., cell volume).
 * The caller zeroes gvol first (the notebook: `fv.vol[:] = 0.0`).                                            */
int mfs_fluid_volume3d(const int64_t vres[3], const double bound_min[3], const double cell_size[3],
                       const void* px, int px_dt, double pvol, int64_t num_particles, void* gvol, int g_dt,
                       mfs_stream stream);

/* ------------------------------------------------------------------------- */
/* Rigid-body signed distances -- reference solver/sdf3D.py */
/* ------------------------------------------------------------------------- */
/* rb_d: num_bodies x 10 x 4 float64 (generate_rb :277-305): row 0 = [type code, parameters] (code // 2:
 * 0 sphere, 1 box, 2 cylinder; odd = flipped), rows 1-4 translation, rows 5-8 rotation, row 9 velocity.
 * position / vel are (P,3) row-major.                                                                   */
/* replaces evaluate_kernel -- solver/sdf3D.py:218-239 (the caller zeroes vel first, as evaluate() :266 does) */
int mfs_sdf_evaluate3d(const void* rb_d, int64_t num_bodies, const void* position, int pos_dt, int64_t num_positions,
                       void* sd, int sd_dt, void* vel, int vel_dt, mfs_stream stream);
/* replaces project_kernel -- solver/sdf3D.py:241-258 (in place on position) */
int mfs_sdf_project3d(const void* rb_d, int64_t num_bodies, void* position, int pos_dt, int64_t num_positions,
                      mfs_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MFS_H */
