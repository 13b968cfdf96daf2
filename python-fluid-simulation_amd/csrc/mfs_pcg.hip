// mfs_pcg.hip -- the per-iteration hot path of PressureCGSolver3D on gfx950.
//
// Replaces the loop solver/PressureCGSolver3D.py:198-223 (1 numba launch + ~12
// cupy kernels + 3 host syncs per iteration in the reference) by five launches
// per iteration with no host sync: the stencil kernel of mfs_pcg_apply.h plus the
// solver-independent phases of mfs_cg_core.h (see both headers).
//
// The stencil reads 4 solver-owned coefficient arrays built once per solve by
// k_pcg_setup from (lphi, wx, wy, wz):  diag (the reference's `diag`, ghost-fluid
// terms included, accumulated in its order) and cx,cy,cz = the face weight
// between a cell and its LOWER neighbour on that axis if both are fluid, else 0.
// All four are cell-shaped (rows of Nz, 16-byte aligned when Nz%VEC==0), unlike
// the caller's wz whose rows of Nz+1 break vector alignment.  Same 6 scalars per
// cell as the reference formulation (v, lphi, wx, wy, wz -> out).
#include "mfs_cg_core.h"
#include "mfs_pcg_apply.h"
#include "mfs_pcg_slab.h"
#include "mfs_rccl.h"
#include "mfs_pcg_resident.h"

namespace mfs {

// ---------------------------------------------------------------- setup -----
// diag / masked lower-face weights from lphi and w (PressureCGSolver3D.py:59-126).
// DENSITY: the operator of solver/DensityCGSolver3D.py:118-207 instead -- same off-diagonal weights except
// the -z tap, which reads wz[x,y,z+1] (:184; kept as written) -> cz2; diag counts 1 per fluid neighbour and
// 1/theta per non-fluid one (:135-199) instead of the face weights.
template <typename T, bool DENSITY>
__global__ void __launch_bounds__(kBlock)
k_pcg_setup(int Nx, int Ny, int Nz, const void* lphi, int ldt, const void* wx, const void* wy, const void* wz,
            int wdt, T* __restrict__ diag, T* __restrict__ cx, T* __restrict__ cy, T* __restrict__ cz,
            T* __restrict__ cz2) {
  const int64_t n = (int64_t)Nx * Ny * Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % Nz), y = (int)((i / Nz) % Ny), x = (int)(i / ((int64_t)Nz * Ny));
  const int64_t sx = (int64_t)Ny * Nz, sy = Nz;
  const double phi = ldx(lphi, ldt, i);
  const bool fl = phi < 0;
  // lower-face weights (face between this cell and the one below it on the axis)
  const double wxm = x > 0 ? ldx(wx, wdt, i) : 0.0;                                   // wx[x,y,z]
  const double wym = y > 0 ? ldx(wy, wdt, ((int64_t)x * (Ny + 1) + y) * Nz + z) : 0.0;  // wy[x,y,z]
  const double wzm = z > 0 ? ldx(wz, wdt, ((int64_t)x * Ny + y) * (Nz + 1) + z) : 0.0;  // wz[x,y,z]
  const double pxm = x > 0 ? ldx(lphi, ldt, i - sx) : 1.0;
  const double pym = y > 0 ? ldx(lphi, ldt, i - sy) : 1.0;
  const double pzm = z > 0 ? ldx(lphi, ldt, i - 1) : 1.0;
  cx[i] = (T)((fl && pxm < 0) ? wxm : 0.0);
  cy[i] = (T)((fl && pym < 0) ? wym : 0.0);
  cz[i] = (T)((fl && pzm < 0) ? wzm : 0.0);
  if (DENSITY) cz2[i] = (T)((fl && pzm < 0 && z > 0) ? ldx(wz, wdt, ((int64_t)x * Ny + y) * (Nz + 1) + z + 1) : 0.0);
  double dg = 0.0;
  const bool interior = x > 0 && x < Nx - 1 && y > 0 && y < Ny - 1 && z > 0 && z < Nz - 1;
  if (interior && fl) {
    auto acc = [&](double nphi, double w) {
      if (DENSITY) w = 1.0;
      if (nphi < 0) dg += w;
      else dg += w / fmin(1.0, fmax(0.01, phi / (phi - nphi)));
    };
    acc(ldx(lphi, ldt, i + sx), ldx(wx, wdt, i + sx));                                    // +x  wx[x+1]
    acc(pxm, wxm);                                                                        // -x
    acc(ldx(lphi, ldt, i + sy), ldx(wy, wdt, ((int64_t)x * (Ny + 1) + y + 1) * Nz + z));  // +y
    acc(pym, wym);                                                                        // -y
    acc(ldx(lphi, ldt, i + 1), ldx(wz, wdt, ((int64_t)x * Ny + y) * (Nz + 1) + z + 1));   // +z
    acc(pzm, wzm);                                                                        // -z
  }
  diag[i] = (T)dg;
}

// (the three-launch Jacobi kernels -- jac_z, k_jac_init, k_jac_begin_finish, k_jac_update_xr, k_jac_update_d -- live in
// mfs_cg_core.h: the viscosity engine uses them too)

// (k_jac_update_rz, jac_book, JacSlab: mfs_cg_core.h -- the viscosity engine's Jacobi loop uses them too)

}  // namespace mfs

using namespace mfs;

struct mfs_pcg3d {
  int Nx, Ny, Nz, dt;
  int64_t n;
  CgCore c;
  char* ws;
  size_t ws_bytes;
  void *diag, *cx, *cy, *cz;
  bool x_owed;                 // the loop has deferred an x update that pcg_home_d still has to apply
  bool book_pending;           // lean loop: the last enqueued iteration's bookkeeping has not been launched yet
  bool slab_loop;              // the running solve is the slab loop (x lives on the owned planes [1, Nx-1) only)
  int defer_x;                 // 1: native fused loop lets x += alpha d ride in the NEXT stencil launch (mfs_pcg3d_finish owes the last one)
  int64_t last_iters;          // iterations of this engine's previous converged solve (sizes the first batch of the next one)
  int jacobi;                  // 1: opt-in Jacobi-preconditioned loop (mfs_pcg3d_set_jacobi); NOT the reference's CG
  double* part_rz;             // partial sums of r.z (Jacobi loop)
  void* zb;                    // z = r / diag as stored by the fused Jacobi loop's r update (operand of the next stencil launch)
  void* cz2;                   // asym only: weight of the -z tap (the density operator, DensityCGSolver3D.py:184)
  int asym;                    // 1: set up by mfs_pcg3d_setup_density
  void* d2;                    // ping-pong partner of the bound d (fused direction update)
  int pd;                      // prefetch depth (planes) of the operand stream in the LDS march: 1 or 2
  int fuse;                    // 1: native loop folds d = r + beta d into the stencil launch
  int lean;                    // mfs_pcg3d_iterate closes iteration j in the stencil launch of j + 1 (no reduction tail in the update): 1 / 0 / -1 auto
  unsigned char* cls;          // class byte per z-vector (compressed coefficient access)
  int resident;                // small grids: the CG loop as one resident launch per batch (mfs_pcg_resident.h): 1 / 0 / -1 auto
  int res_w;                   // its number of workgroups (MFS_RES_W, default 64)
  ResPlan res;                 // decomposition (res.ok false: the grid does not qualify)
  u64 *res_ar, *res_mirror;    // granule table of the dot products / mirror of the box faces of r (workspace)
  unsigned res_epoch;          // episode tags handed out so far (monotonic over the engine's life)
  // read once at creation (never per batch): record stride, the two spin bounds (wall clock, 100 MHz), fault injection
  int res_rec_stride, res_drop_wg;
  unsigned long long res_timeout_ticks, res_first_timeout_ticks;
  int compress;                // 1: per-iteration kernel skips the coefficient arrays of ZERO / REGULAR vectors
  int grid_apply, cus;
  int variant, xchunk, nt, bpc, nt_auto;   // apply-kernel tuning (mfs_pcg3d_tune)
  bool vec_ok;
  bool is_setup;
  mfs_p2p* p2p;                // peer-to-peer window of the slab loop (mfs_pcg3d_attach_p2p), or null
  // slab loop: the edge planes of d are formed and stored into the neighbours' windows on a SECOND stream, so that
  // the xGMI stores (whose completion the producing kernel has to wait for) overlap the interior stencil launch
  hipStream_t aux;
  hipEvent_t ev_main, ev_aux;
  const int *skip_items, *skip_runrem, *skip_count;   // sparse work list of the fused stencil launches (null: dense), per solve
  int skip_xb, skip_xe;        // ... built for the launch over the planes [skip_xb, skip_xe)
  bool lane_mask;              // ... and whether its launches mask dead lanes (from the scalar block of the latest poll)
  int* skip_ws;                // tile flags | items | runrem | count
  int sparse_vec;              // 1 (default; MFS_SPARSE): the r update of a single-domain solve sweeps live chunks only (LiveMap)
  int64_t sparse_min;          // ... from this many cells on
  int* live_ws;                // flags | list | count
  mfs_rccl* rccl;              // collective transport of the slab loop (mfs_pcg3d_attach_rccl): the attached window is then this
                               // rank's OWN 1-rank window, halo planes and dot products travel through RCCL between the launches
  int use_aux;
};

// Coefficient arrays are staggered by an odd number of 4 KiB pages so that the six
// streams of one tile do not all start on the same HBM channel/bank phase.
static size_t coef_stride(int64_t n, size_t elt) { return align_up((size_t)n * elt, 4096) + 4096 * 3 + 256; }

// xdef: deferred x update rides along.  book: the launch also closes the previous iteration (BookArgs in mfs_pcg_apply.h)
struct FuseArgs { const void* r; const void* d_old; void* d_new; void* xdef = nullptr; bool book = false; };

template <typename T, int VEC>
static int launch_apply_v(mfs_pcg3d* h, const T* v, T* out, int xb, int xe, int xb2, int xe2, double* partial,
                          const double* done, hipStream_t st, int* grid_out, const FuseArgs* fz = nullptr) {
  const T *dg = (const T*)h->diag, *cx = (const T*)h->cx, *cy = (const T*)h->cy, *cz = (const T*)h->cz;
  const T* cz2 = (const T*)h->cz2;                 // the density operator's -z weights (asym only)
  const bool asym = h->asym != 0;
  const int nzv = h->Nz / VEC;
  const int64_t ipp = (int64_t)(h->Ny - 2) * nzv;
  int variant = asym ? std::max(1, h->variant) : h->variant;
  const size_t lds = 2 * ((size_t)kApplyBlock * VEC + 2 * (size_t)h->Nz) * sizeof(T);
  if (variant >= 2 && lds > 64 * 1024) variant = 1;          // absurdly long rows: skip the LDS image
  const int xchunk = std::max(0, h->xchunk);   // 0 = no cap on the length of one march
  ApplyArgs a{h->Nx, h->Ny, h->Nz, xb, xe, xchunk, xb2, xe2};
  // sparse work list (built behind the initial residual of a single-domain solve): the fused launches of the loop visit
  // only (tile, plane) pairs that compute anything.  The launch's range must be the list's: all computed planes, one range.
  if (fz && h->skip_items && h->compress != 0 && VEC > 1 && xb == h->skip_xb && xe == h->skip_xe && xe2 == xb2 && variant >= 2) {
    a.items = h->skip_items; a.runrem = h->skip_runrem; a.count = h->skip_count;
  }
  const bool lmask = a.items != nullptr && h->lane_mask && VEC > 1;
  const int np = (xe - xb) + (xe2 - xb2);
  if (variant == 0) {
    const int64_t items = (int64_t)np * ipp;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->grid_apply, (items + kApplyBlock - 1) / kApplyBlock));
    hipLaunchKernelGGL((k_pcg_apply_direct<T, VEC>), dim3(grid), dim3(kApplyBlock), 0, st, v, out, dg, cx, cy, cz, a,
                       partial, done);
    *grid_out = grid;
  } else {
    const int64_t tiles = (ipp + kApplyBlock - 1) / kApplyBlock;
    const int64_t total = tiles * np;                 // (tile, plane) pairs, cut into `grid` equal segments
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(kMaxPartials, h->cus * h->bpc), total));
    // Nontemporal loads for the once-read coefficient streams pay off only when the
    // apply's working set (6 arrays) cannot sit in the 256 MiB Infinity Cache anyway;
    // below that, default caching lets the next iteration hit on-die.  nt < 0 = auto.
    // bit 0: diag, cz   bit 1: cx   bit 2: cy   -- the once-per-iteration coefficient streams.
    int nt = h->nt < 0 ? ((6.0 * (double)h->n * sizeof(T) > 200e6) ? h->nt_auto : 0) : (h->nt & 7);
    const bool comp = h->compress != 0 && VEC > 1;
    const int pd = (h->pd >= 2 && !asym) ? 2 : 1;
    // <LDS, NT, COMP, FUSE, PD, ASYM> with the fused operands (null unless fz)
#define MFS_GO(LDSF, NTV, CMP, FUS, PDV, ASY)                                                                          \
    hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, LDSF, NTV, CMP, FUS, PDV, ASY>), dim3(grid), dim3(kApplyBlock),      \
                       LDSF ? lds : 0, st, v, out, dg, cx, cy, cz, h->cls, a, partial, done,                          \
                       (const T*)(fz ? fz->r : nullptr), (const T*)(fz ? fz->d_old : nullptr),                         \
                       (T*)(fz ? fz->d_new : nullptr), (const double*)(fz ? h->c.scal + S_BETA : nullptr), cz2)
#define MFS_GO_NT_CMP(FUS, PDV, ASY)                                                   \
    do {                                                                               \
      if (comp) { if (nt) MFS_GO(true, 7, true, FUS, PDV, ASY); else MFS_GO(true, 0, true, FUS, PDV, ASY); }   \
      else      { if (nt) MFS_GO(true, 7, false, FUS, PDV, ASY); else MFS_GO(true, 0, false, FUS, PDV, ASY); } \
    } while (0)
    if (variant == 1) {                               // marching without the LDS image (fallback / A-B)
      MFS_REQUIRE(fz == nullptr, "the fused direction update needs the LDS march");
      if (asym) { if (nt) MFS_GO(false, 1, false, false, 1, true); else MFS_GO(false, 0, false, false, 1, true); }
      else      { if (nt) MFS_GO(false, 1, false, false, 1, false); else MFS_GO(false, 0, false, false, 1, false); }
    } else if (fz && fz->book) {
      // fused direction update + the bookkeeping of the previous iteration (PD 1), with or without the deferred x update
      const BookArgs bk{h->c.scal, h->c.hist, kHistCap, h->c.part_rr, h->c.n_part_rr, (int)((h->c.iter_enq - 1) & 1)};
#define MFS_GO_BM(NTV, CMP, ASY, XDF, LMK)                                                                            \
      hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, true, NTV, CMP, true, 1, ASY, XDF, true, LMK>), dim3(grid),        \
                         dim3(kApplyBlock), lds, st, v, out, dg, cx, cy, cz, h->cls, a, partial, done,                 \
                         (const T*)fz->r, (const T*)fz->d_old, (T*)fz->d_new, (const double*)nullptr, cz2,             \
                         (T*)fz->xdef, (const double*)nullptr, bk)
      // (LMASK only with compressed access, and only when the launch carries the work list)
#define MFS_GO_B(NTV, CMP, ASY, XDF)                                                                                   \
      do { if (CMP && lmask) MFS_GO_BM(NTV, CMP, ASY, XDF, CMP); else MFS_GO_BM(NTV, CMP, ASY, XDF, false); } while (0)
      if (fz->xdef && asym) {      // (the density operator's -z tap and the deferred x update are independent of each other)
        if (comp) { if (nt) MFS_GO_B(7, true, true, true); else MFS_GO_B(0, true, true, true); }
        else      { if (nt) MFS_GO_B(7, false, true, true); else MFS_GO_B(0, false, true, true); }
      } else if (fz->xdef) {
        if (comp) { if (nt) MFS_GO_B(7, true, false, true); else MFS_GO_B(0, true, false, true); }
        else      { if (nt) MFS_GO_B(7, false, false, true); else MFS_GO_B(0, false, false, true); }
      } else if (asym) {
        if (comp) { if (nt) MFS_GO_B(7, true, true, false); else MFS_GO_B(0, true, true, false); }
        else      { if (nt) MFS_GO_B(7, false, true, false); else MFS_GO_B(0, false, true, false); }
      } else {
        if (comp) { if (nt) MFS_GO_B(7, true, false, false); else MFS_GO_B(0, true, false, false); }
        else      { if (nt) MFS_GO_B(7, false, false, false); else MFS_GO_B(0, false, false, false); }
      }
#undef MFS_GO_B
#undef MFS_GO_BM
    } else if (fz && fz->xdef) {
      // fused direction update + the previous iteration's x update (PD 1 only); symmetric or density operator
#define MFS_GO_XM(NTV, CMP, ASY, LMK)                                                                                  \
      hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, true, NTV, CMP, true, 1, ASY, true, false, LMK>), dim3(grid), dim3(kApplyBlock), \
                         lds, st, v, out, dg, cx, cy, cz, h->cls, a, partial, done, (const T*)fz->r, (const T*)fz->d_old, \
                         (T*)fz->d_new, (const double*)(h->c.scal + S_BETA), cz2, (T*)fz->xdef,                          \
                         (const double*)(h->c.scal + S_ALPHA))
#define MFS_GO_X(NTV, CMP, ASY)                                                                                        \
      do { if (CMP && lmask) MFS_GO_XM(NTV, CMP, ASY, CMP); else MFS_GO_XM(NTV, CMP, ASY, false); } while (0)
      if (asym) {
        if (comp) { if (nt) MFS_GO_X(7, true, true); else MFS_GO_X(0, true, true); }
        else      { if (nt) MFS_GO_X(7, false, true); else MFS_GO_X(0, false, true); }
      } else {
        if (comp) { if (nt) MFS_GO_X(7, true, false); else MFS_GO_X(0, true, false); }
        else      { if (nt) MFS_GO_X(7, false, false); else MFS_GO_X(0, false, false); }
      }
#undef MFS_GO_X
#undef MFS_GO_XM
    } else if (asym) {
      if (fz) MFS_GO_NT_CMP(true, 1, true); else MFS_GO_NT_CMP(false, 1, true);
    } else if (fz) {
      if (pd == 2) MFS_GO_NT_CMP(true, 2, false); else MFS_GO_NT_CMP(true, 1, false);
    } else if (comp || nt == 0 || nt == 7) {
      if (pd == 2) MFS_GO_NT_CMP(false, 2, false); else MFS_GO_NT_CMP(false, 1, false);
    } else {                                          // dense access with a partial nontemporal mask (A-B of the hints)
      switch (nt) {
        case 1: MFS_GO(true, 1, false, false, 1, false); break;
        case 3: MFS_GO(true, 3, false, false, 1, false); break;
        case 5: MFS_GO(true, 5, false, false, 1, false); break;
        default: MFS_GO(true, 7, false, false, 1, false); break;
      }
    }
#undef MFS_GO_NT_CMP
#undef MFS_GO
    *grid_out = grid;
  }
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

template <typename T>
static int launch_apply(mfs_pcg3d* h, const void* v, void* out, int xb, int xe, int xb2, int xe2, double* partial,
                        int use_done, hipStream_t st, int* grid_out, const FuseArgs* fz = nullptr) {
  if (xe < xb) xe = xb;
  if (xe2 < xb2) xe2 = xb2;
  if ((xe - xb) + (xe2 - xb2) <= 0) { *grid_out = 0; return MFS_OK; }
  constexpr int VEC = VecOf<T>::N;
  const bool vec = h->vec_ok && ((uintptr_t)v % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const double* done = use_done ? h->c.scal + S_DONE : nullptr;
  if (vec) return launch_apply_v<T, VEC>(h, (const T*)v, (T*)out, xb, xe, xb2, xe2, partial, done, st, grid_out, fz);
  MFS_REQUIRE(fz == nullptr, "fused direction update needs the vector path");
  return launch_apply_v<T, 1>(h, (const T*)v, (T*)out, xb, xe, xb2, xe2, partial, done, st, grid_out);
}

static int apply_dispatch(mfs_pcg3d* h, const void* v, void* out, int64_t xb, int64_t xe, double* partial,
                          int use_done, hipStream_t st, int* grid_out, int64_t xb2 = 0, int64_t xe2 = 0,
                          const FuseArgs* fz = nullptr) {
  const int lo = 1, hi = h->Nx - 1;
  auto clip = [&](int64_t v_) { return (int)std::max<int64_t>(lo, std::min<int64_t>(hi, v_)); };
  if (h->Ny < 3 || h->Nz < 3) { *grid_out = 0; return MFS_OK; }
  const int b = clip(xb), e = clip(xe), b2 = clip(xb2), e2 = xe2 > xb2 ? clip(xe2) : b2;
  return h->dt == MFS_F32 ? launch_apply<float>(h, v, out, b, e, b2, e2, partial, use_done, st, grid_out, fz)
                          : launch_apply<double>(h, v, out, b, e, b2, e2, partial, use_done, st, grid_out, fz);
}

// (tile, plane) pairs of the marching kernel on this grid: tiles of kApplyBlock z-vectors per interior plane x Nx
static int64_t skip_pairs(const int64_t gres[3], int dt) {
  const int64_t vec = dt == MFS_F32 ? 4 : 2;
  return ((gres[1] * (gres[2] / vec + 1) + kApplyBlock - 1) / kApplyBlock + 1) * (gres[0] + 1);
}
static size_t skip_ws_bytes(const int64_t gres[3], int dt) {
  return align_up((size_t)(3 * skip_pairs(gres, dt) + 64 + core_compact_scratch_ints(skip_pairs(gres, dt))) * sizeof(int), 4096);
}

extern "C" {

size_t mfs_pcg3d_workspace_bytes(const int64_t gres[3], int dt) {
  if (!gres || !dtype_ok(dt)) return 0;
  const int64_t n = gres[0] * gres[1] * gres[2];
  return core_ws_bytes() + 7 * coef_stride(n, dtype_size(dt)) + 4096 + align_up((size_t)n, 4096) +
         align_up((size_t)kMaxPartials * 8, 4096) + res_ws_bytes(n, dtype_size(dt)) + core_live_ws_bytes(n) +
         skip_ws_bytes(gres, dt);
}

int64_t mfs_pcg3d_history_capacity(void) { return kHistCap; }

int mfs_pcg3d_create(mfs_pcg3d** out, const int64_t gres[3], int dt, void* workspace, size_t workspace_bytes,
                     mfs_stream stream) {
  MFS_REQUIRE(out && gres && workspace, "null argument");
  MFS_REQUIRE(dtype_ok(dt), "dtype");
  for (int a = 0; a < 3; ++a) MFS_REQUIRE(gres[a] >= 1 && gres[a] <= 4096, "grid resolution out of range [1,4096]");
  MFS_REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
  MFS_REQUIRE(workspace_bytes >= mfs_pcg3d_workspace_bytes(gres, dt), "workspace too small");
  mfs_pcg3d* h = new mfs_pcg3d();
  h->Nx = (int)gres[0]; h->Ny = (int)gres[1]; h->Nz = (int)gres[2]; h->dt = dt;
  h->n = gres[0] * gres[1] * gres[2];
  if (int e = core_init(h->c, dt, h->n)) { delete h; return e; }
  h->ws = (char*)workspace; h->ws_bytes = workspace_bytes;
  char* p = core_carve(h->c, h->ws);
  p = (char*)align_up((uintptr_t)p, 4096);
  const size_t cs = coef_stride(h->n, h->c.elt);
  h->diag = p; h->cx = p + cs; h->cy = p + 2 * cs; h->cz = p + 3 * cs;
  h->d2 = p + 4 * cs;
  h->cz2 = p + 5 * cs;
  h->asym = 0;
  h->zb = p + 6 * cs;
  h->cls = (unsigned char*)(p + 7 * cs);
  h->part_rz = (double*)(p + 7 * cs + align_up((size_t)h->n, 4096));
  h->resident = env_int("MFS_RESIDENT", -1);
  h->res_w = std::max(1, std::min(kResMaxW, env_int("MFS_RES_W", 64)));
  h->res_rec_stride = std::max(2, std::min(kResRecStrideMax, env_int("MFS_RES_REC_STRIDE", 64) & ~1));   // u64 words between records (512 B: one record per memory line pair; 16-byte stride costs 0.4 us per iteration)
  h->res_timeout_ticks = (unsigned long long)std::max(1, env_int("MFS_RES_TIMEOUT_MS", 2000)) * 100000ull;
  h->res_first_timeout_ticks = (unsigned long long)std::max(1, env_int("MFS_RES_FIRST_TIMEOUT_MS", 250)) * 100000ull;
  h->res_drop_wg = env_int("MFS_RES_TEST_DROP_WG", -1);      // tests only
  h->res = ResPlan{};
  h->res_ar = nullptr; h->res_mirror = nullptr; h->res_epoch = 0;
  if (res_ws_bytes(h->n, h->c.elt) > 0) {
    char* rp = (char*)h->part_rz + align_up((size_t)kMaxPartials * 8, 4096);
    h->res_ar = (u64*)rp;
    h->res_mirror = (u64*)(rp + align_up((size_t)kResRing * kResMaxW * kResRecStrideMax * 8, 4096));
    h->res = res_plan(h->Nx, h->Ny, h->Nz, dt == MFS_F32 ? 4 : 2, h->c.elt, h->res_w);
  }
  h->live_ws = (int*)((char*)h->part_rz + align_up((size_t)kMaxPartials * 8, 4096) + res_ws_bytes(h->n, h->c.elt));
  h->skip_ws = (int*)((char*)h->live_ws + core_live_ws_bytes(h->n));
  h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
  h->lane_mask = false;
  h->sparse_vec = env_int("MFS_SPARSE", 1);
  h->sparse_min = (int64_t)env_int("MFS_SPARSE_MIN", 1 << 21);
  h->jacobi = env_int("MFS_JACOBI", 0);
  h->last_iters = 0;
  h->defer_x = env_int("MFS_DEFER_X", -1);
  h->x_owed = false;
  h->book_pending = false;
  h->lean = env_int("MFS_LEAN", -1);
  h->slab_loop = false;
  h->fuse = env_int("MFS_FUSE_D", 1);
  h->pd = env_int("MFS_APPLY_PD", 1);
  h->compress = env_int("MFS_APPLY_COMPRESS", 1);
  const int vec = dt == MFS_F32 ? 4 : 2;
  h->vec_ok = (h->Nz % vec) == 0 && h->Nz >= 2 * vec;
  h->cus = h->c.cus;
  h->variant = env_int("MFS_APPLY_VARIANT", 2);
  h->xchunk = env_int("MFS_APPLY_XCHUNK", 0);
  h->nt = env_int("MFS_APPLY_NT", -1);
  h->nt_auto = env_int("MFS_APPLY_NT_AUTO", 7);
  // workgroups per CU the march's work is cut into: 2 since round 3 (same-process A/B on the bench workload, tools/pd_probe.py,
  // 2 vs 3: timed loop 87.3 vs 89.7 us, dense plain apply 73.5 vs 83.8, fp64 timed loop 170.8 vs 172.8; 1 and 4 lose everywhere)
  h->bpc = env_int("MFS_APPLY_BLOCKS_PER_CU", 2);
  h->grid_apply = std::min(kMaxPartials, h->cus * 8);
  h->is_setup = false;
  h->p2p = nullptr;
  h->aux = nullptr; h->ev_main = nullptr; h->ev_aux = nullptr;
  h->rccl = nullptr;
  h->use_aux = env_int("MFS_SLAB_AUX_STREAM", -1);   // -1: decided at attach time from the plane size
  if (hipMemsetAsync(workspace, 0, mfs_pcg3d_workspace_bytes(gres, dt), (hipStream_t)stream) != hipSuccess) {
    set_error("hipMemsetAsync(workspace) failed");
    core_free(h->c);
    delete h;
    return MFS_E_HIP;
  }
  *out = h;
  return MFS_OK;
}

int mfs_pcg3d_destroy(mfs_pcg3d* h) {
  if (!h) return MFS_OK;
  if (h->ev_main) (void)hipEventDestroy(h->ev_main);
  if (h->ev_aux) (void)hipEventDestroy(h->ev_aux);
  if (h->aux) (void)hipStreamDestroy(h->aux);
  core_free(h->c);
  delete h;
  return MFS_OK;
}

static int pcg_setup_impl(mfs_pcg3d* h, const void* lphi, int lphi_dt, const void* wx, const void* wy, const void* wz,
                          int w_dt, bool density, hipStream_t st) {
  MFS_REQUIRE(h && lphi && wx && wy && wz, "null argument");
  MFS_REQUIRE(dtype_ok(lphi_dt) && dtype_ok(w_dt), "dtype");
  const int grid = cdiv(h->n, kBlock);
#define MFS_SETUP(TT, DEN)                                                                                         \
  hipLaunchKernelGGL((k_pcg_setup<TT, DEN>), dim3(grid), dim3(kBlock), 0, st, h->Nx, h->Ny, h->Nz, lphi, lphi_dt, wx, \
                     wy, wz, w_dt, (TT*)h->diag, (TT*)h->cx, (TT*)h->cy, (TT*)h->cz, (TT*)h->cz2)
  if (h->dt == MFS_F32) { if (density) MFS_SETUP(float, true); else MFS_SETUP(float, false); }
  else                  { if (density) MFS_SETUP(double, true); else MFS_SETUP(double, false); }
#undef MFS_SETUP
  MFS_LAUNCH_CHECK();
  h->asym = density ? 1 : 0;
  if (h->vec_ok) {   // class byte per z-vector for the compressed coefficient access
    if (h->dt == MFS_F32) {
      const int64_t nvec = h->n / 4;
      hipLaunchKernelGGL((k_pcg_classify<float, 4>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, (const float*)h->diag,
                         (const float*)h->cx, (const float*)h->cy, (const float*)h->cz, h->Nx, h->Ny, h->Nz, h->cls,
                         density ? (const float*)h->cz2 : (const float*)nullptr);
    } else {
      const int64_t nvec = h->n / 2;
      hipLaunchKernelGGL((k_pcg_classify<double, 2>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, (const double*)h->diag,
                         (const double*)h->cx, (const double*)h->cy, (const double*)h->cz, h->Nx, h->Ny, h->Nz, h->cls,
                         density ? (const double*)h->cz2 : (const double*)nullptr);
    }
    MFS_LAUNCH_CHECK();
  }
  h->is_setup = true;
  h->c.live = LiveMap{nullptr, nullptr, 0};      // a solve's sparse lists belong to the operator it began with
  h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
  return MFS_OK;
}

int mfs_pcg3d_setup(mfs_pcg3d* h, const void* lphi, int lphi_dt, const void* wx, const void* wy, const void* wz,
                    int w_dt, mfs_stream stream) {
  return pcg_setup_impl(h, lphi, lphi_dt, wx, wy, wz, w_dt, false, (hipStream_t)stream);
}

int mfs_pcg3d_setup_density(mfs_pcg3d* h, const void* lphi, int lphi_dt, const void* wx, const void* wy,
                            const void* wz, int w_dt, mfs_stream stream) {
  return pcg_setup_impl(h, lphi, lphi_dt, wx, wy, wz, w_dt, true, (hipStream_t)stream);
}

int mfs_pcg3d_apply(mfs_pcg3d* h, const void* v, void* out, int64_t x_begin, int64_t x_end, mfs_stream stream) {
  MFS_REQUIRE(h && v && out, "null argument");
  MFS_REQUIRE(v != out, "apply cannot run in place");
  MFS_REQUIRE(h->is_setup, "mfs_pcg3d_setup has not been called");
  int grid = 0;
  if (int e = apply_dispatch(h, v, out, x_begin, x_end, h->c.part_dq, 0, (hipStream_t)stream, &grid)) return e;
  h->c.n_part_dq = grid;
  return MFS_OK;
}

int mfs_pcg3d_bind(mfs_pcg3d* h, void* b, void* x, void* d, void* r, void* q) {
  MFS_REQUIRE(h, "null handle");
  h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
  return core_bind(h->c, b, x, d, r, q);
}

void* mfs_pcg3d_scalars(mfs_pcg3d* h) { return h ? h->c.scal : nullptr; }

int mfs_pcg3d_tune(mfs_pcg3d* h, int variant, int xchunk, int blocks_per_cu, int nontemporal) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(variant >= 0 && variant <= 2, "variant: 0 direct, 1 march, 2 march+LDS");
  MFS_REQUIRE(xchunk >= 0 && blocks_per_cu >= 1, "xchunk must be >= 0 (0 = no cap), blocks_per_cu >= 1");
  h->variant = variant; h->xchunk = xchunk; h->bpc = blocks_per_cu;
  h->nt = nontemporal < 0 ? -1 : (nontemporal & 7);
  if (nontemporal >= 0) {
    h->c.nt_x = (nontemporal >> 3) & 1;
    h->c.rev_xr = (nontemporal >> 4) & 1;   // experiment bits: sweep direction of the vector phases
    h->c.rev_d = (nontemporal >> 5) & 1;
  }
  return MFS_OK;
}

int mfs_pcg3d_set_prefetch(mfs_pcg3d* h, int planes) {
  MFS_REQUIRE(h && (planes == 1 || planes == 2), "prefetch depth must be 1 or 2");
  h->pd = planes;
  return MFS_OK;
}

int mfs_pcg3d_set_fuse(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->fuse = on != 0;
  return MFS_OK;
}

int mfs_pcg3d_set_defer_x(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->defer_x = on < 0 ? -1 : (on != 0);
  return MFS_OK;
}

int mfs_pcg3d_set_resident(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->resident = on < 0 ? -1 : (on != 0);
  return MFS_OK;
}

int mfs_pcg3d_set_lean(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->lean = on < 0 ? -1 : (on != 0);
  return MFS_OK;
}

int mfs_pcg3d_set_jacobi(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  if (h->jacobi != (on != 0)) h->last_iters = 0;      // another iteration: the previous solve predicts nothing
  h->jacobi = on != 0;
  return MFS_OK;
}

int mfs_pcg3d_set_compress(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->compress = on != 0;
  return MFS_OK;
}

int mfs_pcg3d_phase_apply(mfs_pcg3d* h, int64_t x_begin, int64_t x_end, int first, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.d && h->is_setup, "engine not bound / set up");
  if (first) h->c.n_part_dq = 0;
  MFS_REQUIRE(h->c.n_part_dq + std::max(h->grid_apply, h->cus * h->bpc) <= kMaxPartials,
              "too many apply ranges in one iteration");
  int grid = 0;
  if (int e = apply_dispatch(h, h->c.d, h->c.q, x_begin, x_end, h->c.part_dq + h->c.n_part_dq, 1,
                             (hipStream_t)stream, &grid))
    return e;
  h->c.n_part_dq += grid;
  return MFS_OK;
}

int mfs_pcg3d_phase_apply2(mfs_pcg3d* h, int64_t x_begin, int64_t x_end, int64_t x_begin2, int64_t x_end2, int first,
                           mfs_stream stream) {
  MFS_REQUIRE(h && h->c.d && h->is_setup, "engine not bound / set up");
  MFS_REQUIRE(x_end <= x_begin2 || x_end2 <= x_begin, "the two plane ranges must not overlap");
  if (first) h->c.n_part_dq = 0;
  MFS_REQUIRE(h->c.n_part_dq + std::max(h->grid_apply, h->cus * h->bpc) <= kMaxPartials,
              "too many apply ranges in one iteration");
  int grid = 0;
  if (int e = apply_dispatch(h, h->c.d, h->c.q, x_begin, x_end, h->c.part_dq + h->c.n_part_dq, 1,
                             (hipStream_t)stream, &grid, x_begin2, x_end2))
    return e;
  h->c.n_part_dq += grid;
  return MFS_OK;
}

int mfs_pcg3d_phase_reduce(mfs_pcg3d* h, int which, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(which == 0 || which == 1, "which must be 0 (d.q) or 1 (r.r)");
  return core_reduce(h->c, which, 1, (hipStream_t)stream);
}

int mfs_pcg3d_phase_update_xr(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_update_xr(h->c, false, (hipStream_t)stream);
}

int mfs_pcg3d_phase_update_r(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_update_xr(h->c, false, (hipStream_t)stream, 1);
}

int mfs_pcg3d_phase_update_x(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_update_xr(h->c, false, (hipStream_t)stream, 2);
}

int mfs_pcg3d_phase_update_d(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_update_d(h->c, false, (hipStream_t)stream);
}

int mfs_pcg3d_begin_local(mfs_pcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  h->x_owed = false;
  h->book_pending = false;
  h->slab_loop = false;
  h->c.live = LiveMap{nullptr, nullptr, 0};      // (slab loops and callers that drive the phases themselves sweep every chunk)
  h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (int e = core_begin_pre(h->c, tol, true, st)) return e;            // self.x *= 0.0  (:198)
  int grid = 0;
  if (int e = apply_dispatch(h, h->c.x, h->c.q, 1, h->Nx - 1, h->c.part_dq, 0, st, &grid)) return e;  // q = A x (:201)
  return core_begin_post(h->c, st);
}

int mfs_pcg3d_begin_finish(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_begin_finish(h->c, (hipStream_t)stream);
}

}  // extern "C"

static bool native_fuse_ok(const mfs_pcg3d* h);
static bool jac_fuse_ok(const mfs_pcg3d* h);

// ---- live chunks of the cell vectors (mfs_cg_core.h LiveMap): a z-vector is DEAD when every computed cell of it is a
// ZERO row (not fluid: k_pcg_classify) and r = d = 0 there at the start of the loop -- q, r and d then stay exactly 0 for the
// whole solve.  The fused stencil launch still sweeps every vector (it owns the direction and x updates); the r update,
// a pure streaming kernel, sweeps the live chunks only.  Single-domain loops; built behind the initial residual.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_pcg_live_flags(unsigned char* __restrict__ cls, const T* __restrict__ r, const T* __restrict__ d, int64_t n, int* __restrict__ flags,
                 int x_first, int Ny, int nzv, int xb, int xe, int* __restrict__ tflags, int* __restrict__ nlive) {
  // cls, r, d start at plane x_first of the arrays; n elements from there; the march's planes are [xb, xe)
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i0 = iv * VEC;
  bool live = false, counted = false;
  if (i0 < n) {
    const unsigned char c = cls[iv];
    const bool zero_row = c == kClsZero || c == kClsDead;      // (a mark of the engine's previous solve is re-decided here)
    live = !zero_row;
#pragma unroll
    for (int j = 0; j < VEC; ++j) live = live || r[i0 + j] != (T)0 || d[i0 + j] != (T)0;
    if (zero_row) cls[iv] = live ? kClsZero : kClsDead;        // lane-level: the listed launches move nothing for a dead vector
    if (live && tflags) {      // ... and the march's (tile, plane) pair of this vector (interior vectors only: the others are never computed)
      const int zv = (int)(iv % nzv), y = (int)((iv / nzv) % Ny), x = x_first + (int)(iv / ((int64_t)nzv * Ny));
      if (x >= xb && x < xe && y >= 1 && y <= Ny - 2) {
        tflags[(int64_t)(((y - 1) * nzv + zv) / kApplyBlock) * (xe - xb) + (x - xb)] = 1;
        counted = true;
      }
    }
  }
  if (nlive) {      // live vectors inside the march's pairs (one atomic per wave): what fraction of the listed pairs' lanes is dead?
    const unsigned long long m = __builtin_amdgcn_ballot_w64(counted);
    if (m != 0 && (threadIdx.x & 63) == 0) atomicAdd(nlive, __popcll(m));
  }
  if (live) flags[i0 / kLiveChunk] = 1;      // (a vector never straddles a chunk: both are multiples of VEC unknowns)
}

// lane-level masking of the listed launches (mfs_pcg_apply.h MASK) on when more than a quarter of the listed pairs' vectors are dead
static __global__ void k_pcg_lane_mask_flag(const int* __restrict__ sc, double* __restrict__ scal) {
  if (threadIdx.x == 0) scal[S_LANE] = ((long long)sc[1] * 4 < (long long)sc[0] * kApplyBlock * 3) ? 1.0 : 0.0;
}

// slab: the window / collective slab loops -- the vector phases cover the owned planes [1, Nx-1), the fused interior launch
// the planes [2, Nx-2) that touch no ghost (the edge planes have their own dense launches).  A vector's liveness is a
// property of its own rank: a ZERO row gives q = 0 whatever its neighbours -- ghost planes included -- hold.
static int pcg_build_live(mfs_pcg3d* h, hipStream_t st, bool slab = false) {
  h->c.live = LiveMap{nullptr, nullptr, 0};
  h->c.live_off = 0; h->c.live_cnt = h->n;
  h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
  // (the opt-in Jacobi loop: its fused single-domain form only -- z = r / diag is 0 wherever r is, so a dead vector stays dead)
  if (!h->sparse_vec || !h->compress || !h->vec_ok || !core_vec_ok(h->c) || h->n < h->sparse_min) return MFS_OK;
  if (h->jacobi && (slab || !jac_fuse_ok(h))) return MFS_OK;
  const int vec = h->dt == MFS_F32 ? 4 : 2;
  const int64_t plane_elems = (int64_t)h->Ny * h->Nz;
  if (slab && h->Nx < 3) return MFS_OK;
  const int x_first = slab ? 1 : 0;
  const int64_t off = slab ? plane_elems : 0, cnt = slab ? plane_elems * (h->Nx - 2) : h->n;
  const int nchunks = (int)((cnt + kLiveChunk - 1) / kLiveChunk);
  int* flags = h->live_ws;
  int* list = flags + nchunks;
  int* count = list + nchunks;
  MFS_HIP_TRY(hipMemsetAsync(flags, 0, (size_t)nchunks * sizeof(int), st));
  // the fused stencil launches' work list: (tile, plane) pairs with a live vector, tile-major
  const int xb = slab ? 2 : 1, xe = slab ? h->Nx - 2 : h->Nx - 1;
  const int nzv = h->Nz / vec, np = xe - xb;
  const int64_t ipp = (int64_t)(h->Ny - 2) * nzv;
  const int tiles = (int)((ipp + kApplyBlock - 1) / kApplyBlock);
  const int64_t gr[3] = {h->Nx, h->Ny, h->Nz};
  const int64_t npairs = (int64_t)tiles * np;
  const bool skip = h->Nx >= 3 && h->Ny >= 3 && np > 0 && npairs > 0 && npairs <= skip_pairs(gr, h->dt) && npairs < 0x7fffffff &&
                    (slab || native_fuse_ok(h) || jac_fuse_ok(h));
  int* tflags = h->skip_ws;
  int* items = tflags + npairs;
  int* runrem = items + npairs;
  int* scount = runrem + npairs;
  if (skip) MFS_HIP_TRY(hipMemsetAsync(tflags, 0, (size_t)npairs * sizeof(int), st));
  if (skip) MFS_HIP_TRY(hipMemsetAsync(scount, 0, 64 * sizeof(int), st));      // [0] listed pairs, [1] live vectors in them
  const int64_t nvec = cnt / vec;
  if (h->dt == MFS_F32)
    hipLaunchKernelGGL((k_pcg_live_flags<float, 4>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, h->cls + off / vec, (const float*)h->c.r + off,
                       (const float*)h->c.d + off, cnt, flags, x_first, h->Ny, nzv, xb, xe, skip ? tflags : (int*)nullptr, skip ? scount + 1 : (int*)nullptr);
  else
    hipLaunchKernelGGL((k_pcg_live_flags<double, 2>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, h->cls + off / vec, (const double*)h->c.r + off,
                       (const double*)h->c.d + off, cnt, flags, x_first, h->Ny, nzv, xb, xe, skip ? tflags : (int*)nullptr, skip ? scount + 1 : (int*)nullptr);
  if (int e = core_compact_flags<int>(flags, nchunks, list, count, count + 64, st)) return e;
  int shift = 0;
  while ((1 << shift) < kLiveChunk / vec) ++shift;
  h->c.live = LiveMap{list, count, shift};
  h->c.live_off = off; h->c.live_cnt = cnt;
  if (skip) {
    if (int e = core_compact_flags<int>(tflags, (int)npairs, items, scount, scount + 64, st)) return e;
    hipLaunchKernelGGL(k_list_runs, dim3(cdiv(npairs, 256)), dim3(256), 0, st, items, scount, np, runrem);
    hipLaunchKernelGGL(k_pcg_lane_mask_flag, dim3(1), dim3(64), 0, st, (const int*)scount, h->c.scal);
    MFS_LAUNCH_CHECK();
    // the partner buffer of the direction vector must be 0 wherever the loop never writes it (a previous solve's liquid)
    MFS_HIP_TRY(hipMemsetAsync(h->d2, 0, (size_t)h->n * h->c.elt, st));
    // ... and so must the Jacobi loop's stored z (the r / z update sweeps the live chunks only; the stencil launch reads z of
    // every vector of a listed pair)
    if (h->jacobi) MFS_HIP_TRY(hipMemsetAsync(h->zb, 0, (size_t)h->n * h->c.elt, st));
    h->skip_items = items; h->skip_runrem = runrem; h->skip_count = scount;
    h->skip_xb = xb; h->skip_xe = xe;
  }
  return MFS_OK;
}

static int jac_begin(mfs_pcg3d* h, double tol, hipStream_t st) {
  if (int e = core_begin_pre(h->c, tol, true, st)) return e;
  int grid = 0;
  if (int e = apply_dispatch(h, h->c.x, h->c.q, 1, h->Nx - 1, h->c.part_dq, 0, st, &grid)) return e;
  const int g2 = std::max(1, (int)std::min<int64_t>(h->c.grid_vec, (h->n + kBlock - 1) / kBlock));
  if (h->dt == MFS_F32)
    hipLaunchKernelGGL((k_jac_init<float>), dim3(g2), dim3(kBlock), 0, st, (const float*)h->c.b, (const float*)h->c.q,
                       (const float*)h->diag, (float*)h->c.d, (float*)h->c.r, h->n, h->c.part_rr, h->part_rz);
  else
    hipLaunchKernelGGL((k_jac_init<double>), dim3(g2), dim3(kBlock), 0, st, (const double*)h->c.b, (const double*)h->c.q,
                       (const double*)h->diag, (double*)h->c.d, (double*)h->c.r, h->n, h->c.part_rr, h->part_rz);
  MFS_LAUNCH_CHECK();
  h->c.n_part_rr = g2;
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st, h->c.part_rr, g2, h->c.scal, (int)S_RR, 0);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st, h->part_rz, g2, h->c.scal, (int)S_RZ, 0);
  hipLaunchKernelGGL(k_jac_begin_finish, dim3(1), dim3(64), 0, st, h->c.scal, h->c.hist);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

template <typename T, int VEC>
static int jac_iteration(mfs_pcg3d* h, hipStream_t st) {
  if (int e = mfs_pcg3d_phase_apply(h, 1, h->Nx - 1, 1, st)) return e;      // q = A d, d.q partials
  const bool vec = VEC > 1;
  const int grid = core_vec_grid(h->c, vec);
  const int par = (int)(h->c.iter_enq & 1);
  hipLaunchKernelGGL((k_jac_update_xr<T, VEC>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)h->c.d, (T*)h->c.r,
                     (const T*)h->c.q, (const T*)h->diag, h->n, h->c.scal, h->c.part_rr, h->part_rz, par, h->c.part_dq,
                     h->c.n_part_dq);
  hipLaunchKernelGGL((k_jac_update_d<T, VEC>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.d, (const T*)h->c.r,
                     (const T*)h->diag, h->n, h->c.scal, h->c.hist, kHistCap, par, h->c.part_rr, h->part_rz, grid);
  MFS_LAUNCH_CHECK();
  h->c.n_part_rr = grid;
  ++h->c.iter_enq;
  return MFS_OK;
}

extern "C" {

int mfs_pcg3d_begin(mfs_pcg3d* h, double tol, mfs_stream stream) {
  if (h && h->jacobi) {
    MFS_REQUIRE(h->c.x && h->is_setup, "engine not bound / set up");
    h->x_owed = false;
    h->book_pending = false;
    h->slab_loop = false;
    h->c.live = LiveMap{nullptr, nullptr, 0};
    h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
    if (int e = jac_begin(h, tol, (hipStream_t)stream)) return e;
    return pcg_build_live(h, (hipStream_t)stream);
  }
  if (int e = mfs_pcg3d_begin_local(h, tol, stream)) return e;
  if (int e = mfs_pcg3d_begin_finish(h, stream)) return e;
  return pcg_build_live(h, (hipStream_t)stream);      // single-domain solve: the r update sweeps live chunks only
}

static bool native_fuse_ok(const mfs_pcg3d* h);
static bool jac_fuse_ok(const mfs_pcg3d* h);
static bool resident_ok(const mfs_pcg3d* h);
// deferred x update: fused native loop (symmetric or density operator), prefetch depth 1, x 16-byte aligned
static bool xdef_ok(const mfs_pcg3d* h) {
  // auto (< 0): on once the CG vectors no longer fit the Infinity Cache (256^3 fp32: 134.7 -> 130.2 us/iteration,
  // fp64: 295 -> 276); below that the extra streams cost the march more than the update kernel saves (128^3: +4 %)
  const bool on = h->defer_x < 0 ? (5.0 * (double)h->n * h->c.elt > 200e6) : (h->defer_x != 0);
  return on && (native_fuse_ok(h) || jac_fuse_ok(h)) && h->pd < 2 && ((uintptr_t)h->c.x % 16 == 0);
}

// the fused stencil launch can serve the engine as bound (LDS march, whole 16-byte vectors, aligned CG vectors)
static bool fuse_shape_ok(const mfs_pcg3d* h) {
  const bool vec_in = h->vec_ok && ((uintptr_t)h->c.d % 16 == 0) && ((uintptr_t)h->c.q % 16 == 0) &&
                      ((uintptr_t)h->c.r % 16 == 0);
  return h->fuse != 0 && h->variant == 2 && vec_in && h->Ny >= 3 && h->Nz >= 3 && h->Nx >= 3;
}
static bool native_fuse_ok(const mfs_pcg3d* h) { return fuse_shape_ok(h) && !h->jacobi; }
// ... and the opt-in Jacobi loop in its fused form (same kernel, operand z = r / diag)
static bool jac_fuse_ok(const mfs_pcg3d* h) { return fuse_shape_ok(h) && h->jacobi && h->pd < 2; }

// one iteration of the fused Jacobi loop: stencil launch (iteration 0: on d_0 = z_0 as begin left it; afterwards forming
// d_j = z_j + beta d_{j-1} into the other buffer of {bound d, d2}, the deferred x update riding along), r / z update with
// the two dot products' partials, one-block bookkeeping
extern "C++" {
template <typename T, int VEC>
static int jac_iteration_fused(mfs_pcg3d* h, hipStream_t st) {
  const int64_t j = h->c.iter_enq;
  void* d_cur = (j & 1) ? h->d2 : h->c.d;
  void* d_prev = (j & 1) ? h->c.d : h->d2;
  const bool xdef = xdef_ok(h);
  int grid = 0, e;
  if (j == 0) {
    if ((e = apply_dispatch(h, d_cur, h->c.q, 1, h->Nx - 1, h->c.part_dq, 1, st, &grid))) return e;
  } else {
    FuseArgs fz{h->zb, d_prev, d_cur};
    if (xdef) fz.xdef = h->c.x;
    if ((e = apply_dispatch(h, d_cur, h->c.q, 1, h->Nx - 1, h->c.part_dq, 1, st, &grid, 0, 0, &fz))) return e;
  }
  h->c.n_part_dq = grid;
  const int g = core_vec_grid(h->c, true);
  const int par = (int)(j & 1);
  const unsigned char* cls = h->compress != 0 ? h->cls : nullptr;      // class bytes are valid whenever the march uses them
  const LiveMap lm = (h->c.live.list && h->c.live_off == 0 && (h->c.live_cnt < 0 || h->c.live_cnt == h->n)) ? h->c.live : LiveMap{nullptr, nullptr, 0};
  if (xdef)
    hipLaunchKernelGGL((k_jac_update_rz<T, VEC, false>), dim3(g), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)d_cur, (T*)h->c.r,
                       (const T*)h->c.q, (const T*)h->diag, (T*)h->zb, h->n, h->c.scal, h->c.part_rr, h->part_rz, par,
                       h->c.part_dq, h->c.n_part_dq, cls, h->c.hist, kHistCap, h->c.tickets, JacSlab{}, lm);
  else
    hipLaunchKernelGGL((k_jac_update_rz<T, VEC, true>), dim3(g), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)d_cur, (T*)h->c.r,
                       (const T*)h->c.q, (const T*)h->diag, (T*)h->zb, h->n, h->c.scal, h->c.part_rr, h->part_rz, par,
                       h->c.part_dq, h->c.n_part_dq, cls, h->c.hist, kHistCap, h->c.tickets, JacSlab{}, lm);
  MFS_LAUNCH_CHECK();
  h->c.n_part_rr = g;
  if (xdef) h->x_owed = true;
  ++h->c.iter_enq;
  return MFS_OK;
}
}  // extern "C++"

// the tail-less form of the fused loop (see BookArgs)
static bool lean_ok(const mfs_pcg3d* h) {
  return h->lean != 0 && native_fuse_ok(h) && h->pd < 2;
}

// The lean loop leaves iteration j open (its r.r partials written, its bookkeeping not) until the stencil launch of
// j + 1 closes it.  Whoever needs the scalar block complete -- a poll, the end of a batch, a switch of loop form --
// closes it with the one-block bookkeeping launch instead.
static int pcg_close_pending(mfs_pcg3d* h, hipStream_t st) {
  if (!h->book_pending) return MFS_OK;
  h->book_pending = false;
  hipLaunchKernelGGL(k_cg_book, dim3(1), dim3(kBlock), 0, st, h->c.scal, h->c.hist, kHistCap,
                     (int)((h->c.iter_enq - 1) & 1), h->c.part_rr, h->c.n_part_rr);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

// the stencil launch of native iteration j = iter_enq: plain for j = 0, else with d_j = r + beta d_{j-1}
// formed on the fly into the other buffer of the pair {bound d, d2}; beta from the scalar block, or -- lean loop, iteration
// j - 1 still open -- from the r.r partials, the launch closing j - 1 itself (BookArgs)
int mfs_pcg3d_native_apply(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (!lean_ok(h) && (e = pcg_close_pending(h, st))) return e;
  if (!native_fuse_ok(h)) return mfs_pcg3d_phase_apply(h, 1, h->Nx - 1, 1, stream);
  const int64_t j = h->c.iter_enq;
  void* d_cur = (j & 1) ? h->d2 : h->c.d;
  void* d_prev = (j & 1) ? h->c.d : h->d2;
  int grid = 0;
  if (j == 0) {
    if ((e = apply_dispatch(h, d_cur, h->c.q, 1, h->Nx - 1, h->c.part_dq, 1, st, &grid))) return e;
  } else {
    FuseArgs fz{h->c.r, d_prev, d_cur};
    if (xdef_ok(h)) fz.xdef = h->c.x;
    fz.book = h->book_pending;
    h->book_pending = false;
    if ((e = apply_dispatch(h, d_cur, h->c.q, 1, h->Nx - 1, h->c.part_dq, 1, st, &grid, 0, 0, &fz))) return e;
  }
  h->c.n_part_dq = grid;
  return MFS_OK;
}

// the rest of native iteration j: x/r update (d.q folded in) and -- unless the lean loop leaves that to the next
// stencil launch -- the direction update or its bookkeeping
int mfs_pcg3d_native_finish(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (!native_fuse_ok(h)) {
    if ((e = core_update_xr(h->c, true, st))) return e;
    return core_update_d(h->c, true, st);
  }
  const int64_t j = h->c.iter_enq;
  void* d_cur = (j & 1) ? h->d2 : h->c.d;
  const bool xdef = xdef_ok(h);
  if (xdef) h->x_owed = true;                    // r only: x += alpha d rides in the next stencil launch
  if (lean_ok(h)) {
    if ((e = core_update_xr(h->c, true, st, xdef ? 1 : 0, d_cur))) return e;
    ++h->c.iter_enq;
    h->book_pending = true;
    return MFS_OK;
  }
  // the last block of the update closes the iteration (convergence test, history, beta): 2 launches per iteration
  if (xdef) {
    XrTail tl{1, h->c.hist, kHistCap, nullptr, 0, 0};
    if ((e = core_update_xr(h->c, true, st, 1, d_cur, 0, -1, &tl, nullptr))) return e;
    ++h->c.iter_enq;
    return MFS_OK;
  }
  return core_update_xr_close(h->c, true, st, d_cur, 1);
}

// n iterations as  A U | A* U | ... | A* U | B : A = fused stencil launch (beta from the scalar block), A* = stencil launch
// that first closes the iteration before it, U = x/r update without a tail, B = one-block bookkeeping for the last
// iteration of the batch -- so that the scalar block is complete whenever the host looks.
static int pcg_iterate_lean(mfs_pcg3d* h, int64_t n, hipStream_t st) {
  for (int64_t i = 0; i < n; ++i) {
    int e;
    if ((e = mfs_pcg3d_native_apply(h, (mfs_stream)st))) return e;
    if ((e = mfs_pcg3d_native_finish(h, (mfs_stream)st))) return e;
  }
  return pcg_close_pending(h, st);
}

// the resident loop (mfs_pcg_resident.h): a grid that fits W workgroups' registers, the fused loop's preconditions
// (aligned vectors, compressed coefficient classes), no deferred x update pending
static bool resident_ok(const mfs_pcg3d* h) {
  return h->resident != 0 && h->res.ok && h->res_ar && (native_fuse_ok(h) || (jac_fuse_ok(h) && core_vec_ok(h->c))) &&
         !h->slab_loop && h->defer_x <= 0 && !h->x_owed && ((uintptr_t)h->c.x % 16 == 0);
}

extern "C++" {
template <typename T, int VEC>
static int pcg_launch_resident(mfs_pcg3d* h, const ResArgs& a, hipStream_t st) {
  const ResPlan& p = h->res;
#define MFS_RES_ONE(KVV, ASY, JCB)                                                                                \
  do {                                                                                                            \
    MFS_HIP_TRY(hipFuncSetAttribute((const void*)k_pcg_resident<T, VEC, KVV, ASY, JCB>,                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));                     \
    hipLaunchKernelGGL((k_pcg_resident<T, VEC, KVV, ASY, JCB>), dim3(p.W), dim3(kResBlock), p.lds, st, a);        \
  } while (0)
#define MFS_RES(KVV)                                                                                              \
  do {                                                                                                            \
    if (h->jacobi) { if (h->asym) MFS_RES_ONE(KVV, true, true); else MFS_RES_ONE(KVV, false, true); }             \
    else           { if (h->asym) MFS_RES_ONE(KVV, true, false); else MFS_RES_ONE(KVV, false, false); }           \
  } while (0)
  if (p.kv <= 1) MFS_RES(1); else if (p.kv == 2) MFS_RES(2); else if (p.kv == 3) MFS_RES(3); else MFS_RES(4);
#undef MFS_RES
#undef MFS_RES_ONE
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}
}  // extern "C++"

static int pcg_iterate_resident(mfs_pcg3d* h, int64_t n, hipStream_t st) {
  while (n > 0) {
    const int nb = (int)std::min<int64_t>(n, 1 << 20);
    const unsigned per_it = h->jacobi ? 3u : 2u;                   // episode tags per iteration (Jacobi: d.q, r.r, r.z)
    if (h->res_epoch > 0xf0000000u - per_it * (unsigned)nb) {      // tags about to wrap: start over on clean tables
      MFS_HIP_TRY(hipMemsetAsync(h->res_ar, 0, res_ws_bytes(h->n, h->c.elt), st));
      h->res_epoch = 0;
    }
    ResArgs a{};
    a.x = h->c.x; a.r = h->c.r; a.q = h->c.q; a.dbuf[0] = h->c.d; a.dbuf[1] = h->d2; a.zb = h->zb;
    a.diag = h->diag; a.cx = h->cx; a.cy = h->cy; a.cz = h->cz; a.cz2 = h->cz2; a.cls = h->cls;
    a.Nx = h->Nx; a.Ny = h->Ny; a.Nz = h->Nz; a.Px = h->res.Px; a.Py = h->res.Py; a.bxm = h->res.bxm; a.bym = h->res.bym;
    a.scal = h->c.scal; a.hist = h->c.hist; a.hist_cap = kHistCap;
    a.j0 = h->c.iter_enq; a.n_iter = nb;
    a.ar = h->res_ar; a.mirror = h->res_mirror;
    a.rec_stride = h->res_rec_stride;
    a.tag0 = h->res_epoch + 1u;
    a.timeout_ticks = h->res_timeout_ticks;
    a.first_timeout_ticks = h->res_first_timeout_ticks;
    a.test_drop_wg = h->res_drop_wg;
    int e = h->dt == MFS_F32 ? pcg_launch_resident<float, 4>(h, a, st) : pcg_launch_resident<double, 2>(h, a, st);
    if (e) return e;
    h->res_epoch += per_it * (unsigned)nb;
    h->c.iter_enq += nb;
    n -= nb;
  }
  return MFS_OK;
}

int mfs_pcg3d_iterate(mfs_pcg3d* h, int64_t n, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  if (h->jacobi || resident_ok(h) || !lean_ok(h))
    if (int e = pcg_close_pending(h, (hipStream_t)stream)) return e;
  if (resident_ok(h)) return pcg_iterate_resident(h, n, (hipStream_t)stream);      // (plain or Jacobi: template flag JAC)
  if (!h->jacobi && lean_ok(h)) return pcg_iterate_lean(h, n, (hipStream_t)stream);
  if (h->jacobi && jac_fuse_ok(h) && core_vec_ok(h->c)) {
    for (int64_t i = 0; i < n; ++i) {
      const int e = h->dt == MFS_F32 ? jac_iteration_fused<float, 4>(h, (hipStream_t)stream)
                                     : jac_iteration_fused<double, 2>(h, (hipStream_t)stream);
      if (e) return e;
    }
    return MFS_OK;
  }
  if (h->jacobi) {
    const bool vec = core_vec_ok(h->c) && ((uintptr_t)h->diag % 16 == 0);
    for (int64_t i = 0; i < n; ++i) {
      int e;
      if (h->dt == MFS_F32) e = vec ? jac_iteration<float, 4>(h, (hipStream_t)stream) : jac_iteration<float, 1>(h, (hipStream_t)stream);
      else e = vec ? jac_iteration<double, 2>(h, (hipStream_t)stream) : jac_iteration<double, 1>(h, (hipStream_t)stream);
      if (e) return e;
    }
    return MFS_OK;
  }
  for (int64_t i = 0; i < n; ++i) {   // 3 launches per iteration: the dots are folded into their consumers
    int e;
    if ((e = mfs_pcg3d_native_apply(h, stream))) return e;
    if ((e = mfs_pcg3d_native_finish(h, stream))) return e;
  }
  return MFS_OK;
}

// after a fused native loop the reference's `d` (d of the last completed iteration) may sit in the
// engine's partner buffer: bring it home to the bound array (iteration count known from a poll)
static int pcg_home_d(mfs_pcg3d* h, int64_t iters, bool converged, hipStream_t st) {
  const bool jac = jac_fuse_ok(h) && core_vec_ok(h->c);
  if (!(native_fuse_ok(h) || jac) || iters < 1 || !h->c.d) return MFS_OK;
  void* cur = ((iters - 1) & 1) ? h->d2 : h->c.d;             // holds d_{iters-1}
  const void* rsrc = jac ? h->zb : h->c.r;                    // fused Jacobi loop: d = z + beta d
  if (h->x_owed) {                                            // owed: x += alpha_{iters-1} d_{iters-1}
    h->x_owed = false;
    const int grid = core_vec_grid(h->c, true);
    const int64_t plane = (int64_t)h->Ny * h->Nz;
    const int64_t off = h->slab_loop ? plane : 0, cnt = h->slab_loop ? plane * (h->Nx - 2) : h->n;   // ghost planes are not ours
    // (the solve's live chunks, when they were built over exactly this range: d is 0 everywhere else)
    const LiveMap lm = (h->c.live.list && off == h->c.live_off && cnt == (h->c.live_cnt < 0 ? h->n : h->c.live_cnt)) ? h->c.live : LiveMap{nullptr, nullptr, 0};
    if (h->dt == MFS_F32) hipLaunchKernelGGL((k_x_axpy<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)h->c.x + off, (const float*)cur + off, cnt, h->c.scal, lm);
    else hipLaunchKernelGGL((k_x_axpy<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)h->c.x + off, (const double*)cur + off, cnt, h->c.scal, lm);
    MFS_LAUNCH_CHECK();
  }
  if (!converged) {                                           // owed: d_iters = r + beta d_{iters-1}
    const bool vec = ((uintptr_t)cur % 16 == 0) && ((uintptr_t)rsrc % 16 == 0);
    const int grid = core_vec_grid(h->c, vec);
    const LiveMap lm = (vec && !jac && h->c.live.list && h->c.live_off == 0 && (h->c.live_cnt < 0 || h->c.live_cnt == h->n)) ? h->c.live : LiveMap{nullptr, nullptr, 0};
    if (h->dt == MFS_F32) {
      if (vec) hipLaunchKernelGGL((k_d_axpy<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)cur, (const float*)rsrc, h->n, h->c.scal, lm);
      else hipLaunchKernelGGL((k_d_axpy<float, 1>), dim3(grid), dim3(kBlock), 0, st, (float*)cur, (const float*)rsrc, h->n, h->c.scal);
    } else {
      if (vec) hipLaunchKernelGGL((k_d_axpy<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)cur, (const double*)rsrc, h->n, h->c.scal, lm);
      else hipLaunchKernelGGL((k_d_axpy<double, 1>), dim3(grid), dim3(kBlock), 0, st, (double*)cur, (const double*)rsrc, h->n, h->c.scal);
    }
    MFS_LAUNCH_CHECK();
  }
  if (cur != h->c.d) {
    const bool whole = h->c.live.list && !jac && h->c.live_off == 0 && (h->c.live_cnt < 0 || h->c.live_cnt == h->n) && core_vec_ok(h->c) &&
                       ((uintptr_t)cur % 16 == 0);
    if (whole) {      // (both buffers hold 0 outside the solve's live chunks)
      const int grid = core_vec_grid(h->c, true);
      if (h->dt == MFS_F32) hipLaunchKernelGGL((k_copy_live<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)h->c.d, (const float*)cur, h->n, h->c.live);
      else hipLaunchKernelGGL((k_copy_live<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)h->c.d, (const double*)cur, h->n, h->c.live);
      MFS_LAUNCH_CHECK();
    } else {
      MFS_HIP_TRY(hipMemcpyAsync(h->c.d, cur, (size_t)h->n * h->c.elt, hipMemcpyDeviceToDevice, st));
    }
  }
  return MFS_OK;
}

int mfs_pcg3d_poll(mfs_pcg3d* h, mfs_stream stream, int64_t* iters, int* done, double* delta, double* alpha,
                   double* beta) {
  MFS_REQUIRE(h, "null handle");
  hipStream_t st = (hipStream_t)stream;
  if (int e = pcg_close_pending(h, st)) return e;
  // The resident loop did not get its workgroups together (a shared GPU): that launch -- and every launch queued behind
  // it -- wrote nothing, so the state is the one the scalar block describes.  Clear the flag, switch this engine to the
  // launch-per-phase loop for good, and report the iterations that did complete: the caller (mfs_pcg3d_solve does) goes on
  // from there.
  MFS_HIP_TRY(hipMemcpyAsync(h->c.pinned, h->c.scal, MFS_PCG_NSCALARS * sizeof(double), hipMemcpyDeviceToHost, st));
  MFS_HIP_TRY(hipStreamSynchronize(st));
  bool fresh = true;
  if ((int)h->c.pinned[S_ERR] == kErrNotResident) {
    MFS_HIP_TRY(hipMemsetAsync(h->c.scal + S_ERR, 0, sizeof(double), st));
    MFS_HIP_TRY(hipMemsetAsync(h->c.scal + S_DONE, 0, sizeof(double), st));
    h->resident = 0;
    h->c.iter_enq = (int64_t)h->c.pinned[S_ITERS];
    fresh = false;
  }
  // lane-level masking of the listed launches (mfs_pcg_apply.h LMASK): this solve's decision arrives with the scalar block;
  // launches enqueued from here on -- and the next solve's first batch -- take it (the class marks are per solve either way)
  h->lane_mask = h->c.pinned[S_LANE] != 0.0;
  return core_poll(h->c, st, iters, done, delta, alpha, beta, fresh);
}

// what the native loop will do for the engine as bound: bit 0 fused direction update, bit 1 deferred x update, bit 2 Jacobi,
// bit 3 resident small-grid loop
int mfs_pcg3d_loop_info(mfs_pcg3d* h) {
  if (!h) return 0;
  if (!h->c.x) return h->jacobi ? 4 : 0;          // not bound yet: only the mode is known
  const bool res = resident_ok(h);
  const bool jf = jac_fuse_ok(h) && core_vec_ok(h->c);
  return ((native_fuse_ok(h) || jf) ? 1 : 0) | (!res && xdef_ok(h) ? 2 : 0) | (h->jacobi ? 4 : 0) | (res ? 8 : 0);
}

int mfs_pcg3d_set_sparse(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->sparse_vec = on ? 1 : 0;
  return MFS_OK;
}

int mfs_pcg3d_sparse_info(mfs_pcg3d* h, mfs_stream stream, int64_t out[4]) {
  MFS_REQUIRE(h && out, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  hipStream_t st = (hipStream_t)stream;
  int v = 0;
  if (h->c.live.count) {
    MFS_HIP_TRY(hipMemcpyAsync(&v, h->c.live.count, sizeof(int), hipMemcpyDeviceToHost, st));
    MFS_HIP_TRY(hipStreamSynchronize(st));
    out[0] = v;
    out[1] = ((h->c.live_cnt < 0 ? h->n : h->c.live_cnt) + kLiveChunk - 1) / kLiveChunk;
  }
  if (h->skip_count) {
    MFS_HIP_TRY(hipMemcpyAsync(&v, h->skip_count, sizeof(int), hipMemcpyDeviceToHost, st));
    MFS_HIP_TRY(hipStreamSynchronize(st));
    const int vec = h->dt == MFS_F32 ? 4 : 2;
    const int64_t ipp = (int64_t)(h->Ny - 2) * (h->Nz / vec);
    out[2] = v;
    out[3] = ((ipp + kApplyBlock - 1) / kApplyBlock) * (h->skip_xe - h->skip_xb);
  }
  return MFS_OK;
}

// for callers that drive begin / iterate themselves: settles what the loop forms owe (the deferred x update, the
// direction vector parked in the engine's partner buffer).  Host-synchronous (it needs the iteration count).
int mfs_pcg3d_finish(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  int64_t iters = 0;
  int done = 0;
  if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  return pcg_home_d(h, iters, done != 0, (hipStream_t)stream);
}

int mfs_pcg3d_solve(mfs_pcg3d* h, double tol, int64_t max_iter, int64_t check_every, mfs_stream stream,
                    int64_t* iters_host) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(max_iter >= 0 && check_every >= 1, "max_iter / check_every");
  if (int e = mfs_pcg3d_begin(h, tol, stream)) return e;
  int64_t enq = 0, iters = 0;
  int done = 0;
  // (no look at the scalar block before the loop: launches queued behind a problem that starts converged return at their top)
  bool first = true;
  while (!done && enq < max_iter) {
    // the resident loop stops by itself inside a batch (one launch), so a longer batch costs nothing but saves the
    // launch, the reload of the state and the look (a stream synchronisation: ~25 us, three or four iterations' worth): at
    // least 1024 iterations per look there -- the notebook's solves (300 .. 600 iterations) then cost one look each
    const int64_t every = resident_ok(h) ? std::max<int64_t>(check_every, 1024) : check_every;
    int64_t n = std::min(every, max_iter - enq);
    // launch-per-phase loops, SHORT solves (up to 4 x check_every iterations, where a look at the scalar block costs as much as
    // half a dozen iterations): the first batch is sized by this engine's previous solve -- consecutive time steps need about
    // the same number of iterations -- so that the solve costs one look, not one per `check_every`.  (Long solves keep the
    // fixed batches: a prediction that overshoots by hundreds of iterations costs more no-op launches than the looks it saves.)
    if (first && h->last_iters > 0 && h->last_iters <= 4 * check_every && !resident_ok(h))
      n = std::min<int64_t>(max_iter - enq, std::min<int64_t>(h->last_iters + h->last_iters / 8 + 2, h->last_iters + 256));
    first = false;
    if (int e = mfs_pcg3d_iterate(h, n, stream)) return e;
    if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
    enq = h->c.iter_enq;        // = enq + n, unless the poll has just taken a batch back (resident loop not resident)
  }
  if (max_iter == 0)       // (nothing iterated: the scalar block as begin left it)
    if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  if (done) h->last_iters = iters;
  if (int e = pcg_home_d(h, iters, done != 0, (hipStream_t)stream)) return e;
  if (iters_host) *iters_host = iters;
  return done ? MFS_OK : MFS_NOT_CONVERGED;
}

int64_t mfs_pcg3d_history(mfs_pcg3d* h, double* out_host, int64_t cap, mfs_stream stream) {
  if (!h) { set_error("mfs_pcg3d_history: null handle"); return MFS_E_INVALID; }
  if (int e = pcg_close_pending(h, (hipStream_t)stream)) return e;
  return core_history(h->c, out_host, cap, (hipStream_t)stream);
}

}  // extern "C"

// ----------------------------------------------------------------------------------------------
// Slab loop over peer-to-peer windows (mfs_pcg_slab.h): one rank of a grid cut into x-slabs.
// ----------------------------------------------------------------------------------------------
static bool slab_ok(const mfs_pcg3d* h) {
  // (the opt-in Jacobi loop too, in its fused form: the same kernels with z = r / diag as the operand of the direction update)
  return h->p2p && h->p2p->connected && (native_fuse_ok(h) || (jac_fuse_ok(h) && core_vec_ok(h->c))) &&
         (size_t)h->Ny * h->Nz * h->c.elt == h->p2p->plane_bytes;
}

static unsigned slab_ar_tag(const mfs_p2p* p, int64_t episode) {
  return 0x80000000u | ((p->epoch & 0x7ffu) << 20) | (unsigned)(episode & 0xfffff);
}

template <typename T>
static int slab_iteration(mfs_pcg3d* h, hipStream_t st) {
  constexpr int VEC = VecOf<T>::N;
  mfs_p2p* p = h->p2p;
  const int64_t j = h->c.iter_enq;
  const int par = (int)(j & 1), L = h->Nx;
  T* d_cur = (T*)((j & 1) ? h->d2 : h->c.d);
  T* d_prev = (T*)((j & 1) ? h->c.d : h->d2);
  const SlabEdge e = slab_edges(L, p->rank, p->world);
  const int64_t plane_elems = (int64_t)h->Ny * h->Nz;
  const unsigned halo_tag = 0x80000000u | ((p->epoch & 0x7ffu) << 20) | (unsigned)((j + 1) & 0xfffff);
  int e_;
  // Jacobi (opt-in): the operand of the direction update is z = r / diag as stored by the previous r / z update, and an
  // iteration has three all-reduce episodes (d.q, r.r, r.z) instead of two; begin used episodes 0 and 1
  const bool jac = h->jacobi != 0;
  const T* rsrc = (const T*)(jac ? h->zb : h->c.r);
  const int64_t ep_dq = jac ? 3 * j + 2 : 2 * j + 1, ep_rr = jac ? 3 * j + 3 : 2 * j + 2, ep_rz = 3 * j + 4;
  // deferred x update (as in the native loop): x += alpha_{j-1} d_{j-1} rides in this iteration's edge / interior launches
  const bool xdef = xdef_ok(h) && L - 2 > 2;
  // 1. edge planes of d_j: local + into the neighbours' windows -- on the second stream, behind everything
  //    the main stream has done so far (beta, r of the previous iteration)
  mfs_rccl* const rc = h->rccl;      // collective transport: RCCL moves the planes and sums the dot products between the launches
  if (rc) MFS_REQUIRE(!jac, "the collective slab loop has no Jacobi form");
  const bool aux = (rc != nullptr || h->use_aux) && h->aux && e.np > 0 && L - 2 > 2;
  hipStream_t se = aux ? h->aux : st;
  if (aux) {
    MFS_HIP_TRY(hipEventRecord(h->ev_main, st));
    MFS_HIP_TRY(hipStreamWaitEvent(h->aux, h->ev_main, 0));
  }
  if (e.np > 0) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(4 * h->cus, (plane_elems / VEC + kBlock - 1) / kBlock));
    if (j == 0)
      hipLaunchKernelGGL((k_slab_edge_d<T, VEC, true>), dim3(grid), dim3(kBlock), 0, se, (const T*)nullptr, (const T*)d_cur,
                         (T*)nullptr, plane_elems, e, h->c.scal, p->dev, par, halo_tag, (T*)nullptr);
    else
      hipLaunchKernelGGL((k_slab_edge_d<T, VEC, false>), dim3(grid), dim3(kBlock), 0, se, rsrc, (const T*)d_prev,
                         d_cur, plane_elems, e, h->c.scal, p->dev, par, halo_tag, xdef ? (T*)h->c.x : (T*)nullptr);
    MFS_LAUNCH_CHECK();
  }
  if (rc && e.np > 0)      // ... and to / from the neighbours: one send / recv group behind the edge kernel
    if ((e_ = rccl_halo(rc, (char*)d_cur, (size_t)plane_elems * sizeof(T), L, se))) return e_;
  if (aux) MFS_HIP_TRY(hipEventRecord(h->ev_aux, h->aux));
  // 2. planes that touch no ghost, while the edge planes travel
  int n_part = 0;
  if (L - 2 > 2) {
    int grid = 0;
    if (j == 0) {
      if ((e_ = apply_dispatch(h, d_cur, h->c.q, 2, L - 2, h->c.part_dq, 1, st, &grid))) return e_;
    } else {
      FuseArgs fz{rsrc, d_prev, d_cur};
      if (xdef) fz.xdef = h->c.x;
      if ((e_ = apply_dispatch(h, d_cur, h->c.q, 2, L - 2, h->c.part_dq, 1, st, &grid, 0, 0, &fz))) return e_;
    }
    n_part = grid;
  }
  // 3. edge planes of q (ghost operands from the own window); needs this rank's own edge planes of d_j too
  if (aux) MFS_HIP_TRY(hipStreamWaitEvent(st, h->ev_aux, 0));
  if (e.np > 0) {
    const int64_t items = (int64_t)e.np * (h->Ny - 2) * (h->Nz / VEC);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (items + kApplyBlock - 1) / kApplyBlock));
    hipLaunchKernelGGL((k_slab_edge_apply<T, VEC>), dim3(grid), dim3(kApplyBlock), 0, st, (const T*)d_cur, (T*)h->c.q,
                       (const T*)h->diag, (const T*)h->cx, (const T*)h->cy, (const T*)h->cz,
                       (const T*)(h->asym ? h->cz2 : h->cz), L, h->Ny, h->Nz, e,
                       h->c.part_dq, n_part, h->c.scal, p->dev, par, halo_tag, h->c.tickets + kTicketWords,
                       (int)(ep_dq & (kArRing - 1)), slab_ar_tag(p, ep_dq));
    MFS_LAUNCH_CHECK();
    n_part += grid;
    if (rc && (e_ = rccl_sum(rc, h->c.scal + S_DQ, 1, st))) return e_;      // the edge launch left THIS rank's d.q there
  }
  h->c.n_part_dq = n_part;
  if (jac) {
    // 4'. r / z update over the owned planes; its last block: r.r and r.z over all ranks + bookkeeping
    const int64_t off = plane_elems, cnt = plane_elems * (L - 2);
    const int g = std::max(1, (int)std::min<int64_t>(core_vec_grid(h->c, true), (cnt / VEC + kBlock - 1) / kBlock));
    const unsigned char* cls = h->compress != 0 ? h->cls + off / VEC : nullptr;
    JacSlab sl{1, p->dev, {(int)(ep_rr & (kArRing - 1)), (int)(ep_rz & (kArRing - 1))}, {slab_ar_tag(p, ep_rr), slab_ar_tag(p, ep_rz)}};
    if (xdef) {
      h->x_owed = true;
      hipLaunchKernelGGL((k_jac_update_rz<T, VEC, false>), dim3(g), dim3(kBlock), 0, st, (T*)h->c.x + off, (const T*)d_cur + off,
                         (T*)h->c.r + off, (const T*)h->c.q + off, (const T*)h->diag + off, (T*)h->zb + off, cnt, h->c.scal,
                         h->c.part_rr, h->part_rz, par, h->c.part_dq, 0, cls, h->c.hist, kHistCap, h->c.tickets, sl);
    } else {
      hipLaunchKernelGGL((k_jac_update_rz<T, VEC, true>), dim3(g), dim3(kBlock), 0, st, (T*)h->c.x + off, (const T*)d_cur + off,
                         (T*)h->c.r + off, (const T*)h->c.q + off, (const T*)h->diag + off, (T*)h->zb + off, cnt, h->c.scal,
                         h->c.part_rr, h->part_rz, par, h->c.part_dq, 0, cls, h->c.hist, kHistCap, h->c.tickets, sl);
    }
    MFS_LAUNCH_CHECK();
    h->c.n_part_rr = g;
    ++h->c.iter_enq;
    return MFS_OK;
  }
  // 4. x, r update (r only when the x update is deferred); its last block: r.r over all ranks + bookkeeping
  if (rc) {      // r.r: the rank's total into the scalar block, ONE all-reduce, then the one-block bookkeeping
    if (xdef) h->x_owed = true;
    XrTail tl{3, h->c.hist, kHistCap, nullptr, 0, 0};
    if ((e_ = core_update_xr(h->c, false, st, xdef ? 1 : 0, d_cur, plane_elems, plane_elems * (L - 2), &tl, &p->dev))) return e_;
    if ((e_ = rccl_sum(rc, h->c.scal + S_RR, 1, st))) return e_;
    hipLaunchKernelGGL(k_cg_book, dim3(1), dim3(kBlock), 0, st, h->c.scal, h->c.hist, kHistCap, par, h->c.part_rr, 0);
    MFS_LAUNCH_CHECK();
    ++h->c.iter_enq;
    return MFS_OK;
  }
  if (xdef) {
    h->x_owed = true;
    XrTail tl{2, h->c.hist, kHistCap, nullptr, (int)((2 * j + 2) & (kArRing - 1)), slab_ar_tag(p, 2 * j + 2)};
    if ((e_ = core_update_xr(h->c, false, st, 1, d_cur, plane_elems, plane_elems * (L - 2), &tl, &p->dev))) return e_;
    ++h->c.iter_enq;
    return MFS_OK;
  }
  return core_update_xr_close(h->c, false, st, d_cur, 2, plane_elems, plane_elems * (L - 2), &p->dev,
                              (int)((2 * j + 2) & (kArRing - 1)), slab_ar_tag(p, 2 * j + 2));
}

extern "C" {

int mfs_pcg3d_attach_p2p(mfs_pcg3d* h, mfs_p2p* p) {
  MFS_REQUIRE(h, "null handle");
  if (p) {
    MFS_REQUIRE(p->connected, "mfs_p2p_connect has not been called");
    MFS_REQUIRE((size_t)h->Ny * h->Nz * h->c.elt == p->plane_bytes, "window plane size != Ny*Nz*sizeof(element)");
  }
  h->p2p = p;
  if (p) {
    // second stream for the edge-plane sends: the two cross-stream event hops cost ~16 us per iteration (measured on
    // one GPU), the xGMI drain they hide grows with the plane -- default on from 1 MiB of granules per plane
    // (bench.py calibrates the choice on the machine it runs on; MFS_SLAB_AUX_STREAM / mfs_pcg3d_slab_set_aux override)
    if (h->use_aux < 0) h->use_aux = (p->world > 1 && 2 * p->plane_bytes >= (1u << 20)) ? 1 : 0;
    return mfs_pcg3d_slab_set_aux(h, h->use_aux);
  }
  return MFS_OK;
}

// collective transport for the slab loop: `p` must be this rank's OWN one-rank window (mfs_p2p_create(rank 0, world 1)),
// `rc` the communicator over the slab ranks.  The loop then runs the window loop's launches with no in-kernel exchange:
// edge planes by ncclSend / ncclRecv on the second stream beside the interior launch, each dot product one ncclAllReduce.
int mfs_pcg3d_attach_rccl(mfs_pcg3d* h, mfs_p2p* p, mfs_rccl* rc) {
  MFS_REQUIRE(h, "null handle");
  if (!rc) { h->rccl = nullptr; return mfs_pcg3d_attach_p2p(h, nullptr); }
  MFS_REQUIRE(p && p->connected && p->world == 1, "the collective loop takes this rank's own one-rank window");
  MFS_REQUIRE(rc->comm, "communicator not initialised");
  if (int e = mfs_pcg3d_attach_p2p(h, p)) return e;
  h->rccl = rc;
  h->use_aux = 1;
  if (!h->aux) {
    MFS_HIP_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    MFS_HIP_TRY(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    MFS_HIP_TRY(hipEventCreateWithFlags(&h->ev_aux, hipEventDisableTiming));
  }
  return MFS_OK;
}

int mfs_rccl_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

// rank 0 calls this and hands the bytes to every rank (torch.distributed broadcast); `lib_path`: the librccl the process carries
int mfs_rccl_unique_id(const char* lib_path, void* id_out) {
  MFS_REQUIRE(lib_path && id_out, "null argument");
  mfs_rccl tmp;
  if (int e = rccl_load(&tmp, lib_path)) return e;
  ncclUniqueId id;
  MFS_RCCL_TRY(&tmp, tmp.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return MFS_OK;
}

// COLLECTIVE over the `world` ranks: every rank calls with the same id (the calling thread's current device is the rank's GPU)
int mfs_rccl_create(mfs_rccl** out, const char* lib_path, const void* id, int rank, int world) {
  MFS_REQUIRE(out && lib_path && id && world >= 1 && rank >= 0 && rank < world, "arguments");
  mfs_rccl* r = new mfs_rccl();
  if (int e = rccl_load(r, lib_path)) { delete r; return e; }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  r->rank = rank; r->world = world;
  const ncclResult_t rc = r->CommInitRank(&r->comm, world, uid, rank);
  if (rc != ncclSuccess) {
    set_error("ncclCommInitRank failed: %s", r->GetErrorString(rc));
    delete r;
    return MFS_E_HIP;
  }
  *out = r;
  return MFS_OK;
}

int mfs_rccl_destroy(mfs_rccl* r) {
  if (!r) return MFS_OK;
  if (r->comm) (void)r->CommDestroy(r->comm);
  delete r;
  return MFS_OK;
}

// 1 if the slab loop can run on this engine with the attached window (vector path, LDS march, bound aligned vectors)
int mfs_pcg3d_slab_supported(mfs_pcg3d* h) {
  if (!h || !h->c.x) return 0;
  return slab_ok(h) ? 1 : 0;
}

int mfs_pcg3d_slab_set_aux(mfs_pcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->use_aux = on != 0;
  if (h->use_aux && h->p2p && !h->aux) {
    MFS_HIP_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    MFS_HIP_TRY(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    MFS_HIP_TRY(hipEventCreateWithFlags(&h->ev_aux, hipEventDisableTiming));
  }
  return MFS_OK;
}

int mfs_pcg3d_slab_begin(mfs_pcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  MFS_REQUIRE(slab_ok(h), "slab loop needs an attached window, the vector path (Nz % 4 (fp32) / 2 (fp64) == 0, 16-byte aligned CG vectors) and stencil variant 2");
  hipStream_t st = (hipStream_t)stream;
  h->x_owed = false;
  h->book_pending = false;
  h->slab_loop = true;
  h->c.live = LiveMap{nullptr, nullptr, 0};
  h->skip_items = nullptr; h->skip_runrem = nullptr; h->skip_count = nullptr;
  ++h->p2p->epoch;
  if (int e = core_begin_pre(h->c, tol, true, st)) return e;            // self.x *= 0.0  (:198)
  int grid = 0;
  if (int e = apply_dispatch(h, h->c.x, h->c.q, 1, h->Nx - 1, h->c.part_dq, 0, st, &grid)) return e;  // q = A x (:201)
  MFS_REQUIRE(!(h->rccl && h->jacobi), "the collective slab loop has no Jacobi form (mfs_pcg3d_set_jacobi(0) first)");
  if (h->jacobi) {     // r = b - q, d = z = r / diag, both dot products over all ranks (episodes 0 and 1), delta0 = r.z
    const int g2 = std::max(1, (int)std::min<int64_t>(h->c.grid_vec, (h->n + kBlock - 1) / kBlock));
    if (h->dt == MFS_F32)
      hipLaunchKernelGGL((k_jac_init<float>), dim3(g2), dim3(kBlock), 0, st, (const float*)h->c.b, (const float*)h->c.q,
                         (const float*)h->diag, (float*)h->c.d, (float*)h->c.r, h->n, h->c.part_rr, h->part_rz);
    else
      hipLaunchKernelGGL((k_jac_init<double>), dim3(g2), dim3(kBlock), 0, st, (const double*)h->c.b, (const double*)h->c.q,
                         (const double*)h->diag, (double*)h->c.d, (double*)h->c.r, h->n, h->c.part_rr, h->part_rz);
    h->c.n_part_rr = g2;
    hipLaunchKernelGGL(k_slab_allreduce_rr, dim3(1), dim3(kBlock), 0, st, h->c.part_rr, g2, h->c.scal, h->p2p->dev, 0,
                       slab_ar_tag(h->p2p, 0), (int)S_RR);
    hipLaunchKernelGGL(k_slab_allreduce_rr, dim3(1), dim3(kBlock), 0, st, h->part_rz, g2, h->c.scal, h->p2p->dev, 1,
                       slab_ar_tag(h->p2p, 1), (int)S_RZ);
    hipLaunchKernelGGL(k_jac_begin_finish, dim3(1), dim3(64), 0, st, h->c.scal, h->c.hist);
    MFS_LAUNCH_CHECK();
    return MFS_OK;
  }
  if (int e = core_begin_post(h->c, st, false)) return e;             // d = r = b - q, partials of r.r
  hipLaunchKernelGGL(k_slab_allreduce_rr, dim3(1), dim3(kBlock), 0, st, h->c.part_rr, h->c.n_part_rr, h->c.scal,
                     h->p2p->dev, 0, slab_ar_tag(h->p2p, 0));
  MFS_LAUNCH_CHECK();
  if (h->rccl) { if (int e = rccl_sum(h->rccl, h->c.scal + S_RR, 1, st)) return e; }      // (the window is this rank's own)
  if (int e = core_begin_finish(h->c, st)) return e;
  return pcg_build_live(h, st, true);
}

int mfs_pcg3d_slab_iterate(mfs_pcg3d* h, int64_t n, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  MFS_REQUIRE(slab_ok(h), "slab loop not available for this engine (see mfs_pcg3d_slab_begin)");
  for (int64_t i = 0; i < n; ++i) {
    const int e = h->dt == MFS_F32 ? slab_iteration<float>(h, (hipStream_t)stream)
                                   : slab_iteration<double>(h, (hipStream_t)stream);
    if (e) return e;
  }
  return MFS_OK;
}

int mfs_pcg3d_slab_solve(mfs_pcg3d* h, double tol, int64_t max_iter, int64_t check_every, mfs_stream stream,
                         int64_t* iters_host) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(max_iter >= 0 && check_every >= 1, "max_iter / check_every");
  if (int e = mfs_pcg3d_slab_begin(h, tol, stream)) return e;
  int64_t enq = 0, iters = 0;
  int done = 0;
  if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  while (!done && enq < max_iter) {       // every rank sees the same `done` at the same iteration: same loop everywhere
    const int64_t n = std::min(check_every, max_iter - enq);
    if (int e = mfs_pcg3d_slab_iterate(h, n, stream)) return e;
    enq += n;
    if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  }
  if (int e = pcg_home_d(h, iters, done != 0, (hipStream_t)stream)) return e;
  if (iters_host) *iters_host = iters;
  return done ? MFS_OK : MFS_NOT_CONVERGED;
}

}  // extern "C"
