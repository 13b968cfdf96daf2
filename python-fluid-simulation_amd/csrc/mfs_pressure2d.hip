// mfs_pressure2d.hip -- PressureCGSolver2D on gfx950 (BASELINE config 1: plumbing case).
//
// Reference: solver/PressureCGSolver2D.py.  The grids are tiny (64^2), so these
// are plain one-thread-per-cell kernels in the reference's accumulation order; the
// CG loop is the shared device-resident core (mfs_cg_core.h) with the 5-point
// ghost-fluid operator applied straight from (lphi, wx, wy).
#include "mfs_cg_core.h"

namespace mfs {

__device__ __forceinline__ double edge_in_fraction2(double l, double r) {  // SolidFractionCommon.py:4-16
  const bool li = l < 0, ri = r < 0;
  if (li && ri) return 1.0;
  if (!li && !ri) return 0.0;
  const double diff = -fabs(l - r);
  return li ? l / diff : r / diff;
}

struct Grid2 {
  int Nx, Ny;
  __device__ int64_t c(int x, int y) const { return (int64_t)x * Ny + y; }
  __device__ int64_t fx(int x, int y) const { return (int64_t)x * Ny + y; }
  __device__ int64_t fy(int x, int y) const { return (int64_t)x * (Ny + 1) + y; }
  __device__ int64_t dg(int i, int j) const { return (int64_t)i * (2 * Ny + 1) + j; }
};

// solver/PressureCGSolver2D.py:6-44
__global__ void __launch_bounds__(256)
k_pressure_rhs2d(Grid2 g, double csx, double csy, const void* vx, const void* vy, int vdt, const void* sv, int svdt,
                 const void* lphi, int ldt, const void* wx, const void* wy, int wdt, void* b, int bdt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)g.Nx * g.Ny) return;
  const int y = (int)(i % g.Ny), x = (int)(i / g.Ny);
  if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1) return;
  if (!(ldx(lphi, ldt, i) < 0)) { stx(b, bdt, i, 0.0); return; }
  double bv = 0.0, w;
  w = ldx(wx, wdt, g.fx(x + 1, y));
  bv += w * ldx(vx, vdt, g.fx(x + 1, y)) / csx;
  if (w < 1) bv -= w * ldx(sv, svdt, 2 * g.dg(2 * x + 2, 2 * y + 1) + 0) / csx;
  w = ldx(wx, wdt, g.fx(x, y));
  bv -= w * ldx(vx, vdt, g.fx(x, y)) / csx;
  if (w < 1) bv += w * ldx(sv, svdt, 2 * g.dg(2 * x, 2 * y + 1) + 0) / csx;
  w = ldx(wy, wdt, g.fy(x, y + 1));
  bv += w * ldx(vy, vdt, g.fy(x, y + 1)) / csy;
  if (w < 1) bv -= w * ldx(sv, svdt, 2 * g.dg(2 * x + 1, 2 * y + 2) + 1) / csy;
  w = ldx(wy, wdt, g.fy(x, y));
  bv -= w * ldx(vy, vdt, g.fy(x, y)) / csy;
  if (w < 1) bv += w * ldx(sv, svdt, 2 * g.dg(2 * x + 1, 2 * y) + 1) / csy;
  stx(b, bdt, i, bv);
}

// solver/PressureCGSolver2D.py:46-100; also leaves per-block partials of v.out
__global__ void __launch_bounds__(256)
k_pressure_apply2d(Grid2 g, const void* v, void* out, int dt, const void* wx, const void* wy, int wdt,
                   const void* lphi, int ldt, double* partial, const double* done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  const int64_t n = (int64_t)g.Nx * g.Ny;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i % g.Ny), x = (int)(i / g.Ny);
    if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1) continue;
    const double phi = ldx(lphi, ldt, i);
    if (!(phi < 0)) { stx(out, dt, i, 0.0); continue; }
    double val = 0.0, diag = 0.0;
    auto tap = [&](int64_t nb, double w) {
      const double nphi = ldx(lphi, ldt, nb);
      if (nphi < 0) { val -= w * ldx(v, dt, nb); diag += w; }
      else          { diag += w / fmin(1.0, fmax(0.01, phi / (phi - nphi))); }
    };
    tap(i + g.Ny, ldx(wx, wdt, g.fx(x + 1, y)));
    tap(i - g.Ny, ldx(wx, wdt, g.fx(x, y)));
    tap(i + 1, ldx(wy, wdt, g.fy(x, y + 1)));
    tap(i - 1, ldx(wy, wdt, g.fy(x, y)));
    const double vc = ldx(v, dt, i);
    val += diag * vc;
    stx(out, dt, i, val);
    acc += vc * (dt == MFS_F32 ? (double)(float)val : val);
  }
  if (partial) {
    const double tot = block_sum<256>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
  }
}

// solver/PressureCGSolver2D.py:102-120
__global__ void __launch_bounds__(256)
k_pressure_update2d(Grid2 g, double csx, double csy, void* vx, void* vy, int vdt, const void* pv, int pdt,
                    const void* wx, const void* wy, int wdt, const void* sv, int svdt, const void* lphi, int ldt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)g.Nx * g.Ny) return;
  const int y = (int)(i % g.Ny), x = (int)(i / g.Ny);
  if (x == 0 || y == 0) return;
  const double pc = ldx(lphi, ldt, i), p = ldx(pv, pdt, i);
  auto axis = [&](void* vel, int64_t fi, int64_t nb, const void* w, int64_t svi, double c) {
    const double pm = ldx(lphi, ldt, nb);
    if (pc < 0 || pm < 0) {
      const double th = fmin(1.0, fmax(0.01, edge_in_fraction2(pc, pm)));
      double nv = ldx(vel, vdt, fi);
      nv += (p - ldx(pv, pdt, nb)) * c / th;
      const double ww = ldx(w, wdt, fi);
      nv = ww * nv + (1 - ww) * ldx(sv, svdt, svi);
      stx(vel, vdt, fi, nv);
    }
  };
  axis(vx, g.fx(x, y), i - g.Ny, wx, 2 * g.dg(2 * x, 2 * y + 1) + 0, csx);
  axis(vy, g.fy(x, y), i - 1, wy, 2 * g.dg(2 * x + 1, 2 * y) + 1, csy);
}

}  // namespace mfs

using namespace mfs;

struct mfs_pcg2d {
  Grid2 g;
  int dt;
  CgCore c;
  const void *lphi, *wx, *wy;
  int ldt, wdt;
  int grid;
};

static int check_gres2(const int64_t gres[2]) {
  MFS_REQUIRE(gres != nullptr, "gres is null");
  MFS_REQUIRE(gres[0] >= 1 && gres[1] >= 1 && gres[0] <= 65536 && gres[1] <= 65536, "grid resolution out of range");
  return MFS_OK;
}

static int apply2d(mfs_pcg2d* h, const void* v, void* out, bool use_done, hipStream_t st) {
  hipLaunchKernelGGL(k_pressure_apply2d, dim3(h->grid), dim3(256), 0, st, h->g, v, out, h->dt, h->wx, h->wy, h->wdt,
                     h->lphi, h->ldt, h->c.part_dq, use_done ? h->c.scal + S_DONE : nullptr);
  MFS_LAUNCH_CHECK();
  h->c.n_part_dq = h->grid;
  return MFS_OK;
}

extern "C" {

int mfs_pressure_rhs2d(const int64_t gres[2], const double cell_size[2], const void* vx, const void* vy, int v_dt,
                       const void* sv, int sv_dt, const void* lphi, int lphi_dt, const void* wx, const void* wy,
                       int w_dt, void* b, int b_dt, mfs_stream stream) {
  if (int e = check_gres2(gres)) return e;
  MFS_REQUIRE(cell_size && vx && vy && sv && lphi && wx && wy && b, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(sv_dt) && dtype_ok(lphi_dt) && dtype_ok(w_dt) && dtype_ok(b_dt), "dtype");
  Grid2 g{(int)gres[0], (int)gres[1]};
  hipLaunchKernelGGL(k_pressure_rhs2d, dim3(cdiv(gres[0] * gres[1], 256)), dim3(256), 0, (hipStream_t)stream, g,
                     cell_size[0], cell_size[1], vx, vy, v_dt, sv, sv_dt, lphi, lphi_dt, wx, wy, w_dt, b, b_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pressure_apply2d(const int64_t gres[2], const void* v, void* out, int dt, const void* wx, const void* wy,
                         int w_dt, const void* lphi, int lphi_dt, mfs_stream stream) {
  if (int e = check_gres2(gres)) return e;
  MFS_REQUIRE(v && out && wx && wy && lphi && v != out, "null / aliased array");
  MFS_REQUIRE(dtype_ok(dt) && dtype_ok(w_dt) && dtype_ok(lphi_dt), "dtype");
  Grid2 g{(int)gres[0], (int)gres[1]};
  hipLaunchKernelGGL(k_pressure_apply2d, dim3(cdiv(gres[0] * gres[1], 256)), dim3(256), 0, (hipStream_t)stream, g, v,
                     out, dt, wx, wy, w_dt, lphi, lphi_dt, (double*)nullptr, (const double*)nullptr);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pressure_update2d(const int64_t gres[2], const double cell_size[2], void* vx, void* vy, int v_dt,
                          const void* pv, int pv_dt, const void* wx, const void* wy, int w_dt, const void* sv,
                          int sv_dt, const void* lphi, int lphi_dt, mfs_stream stream) {
  if (int e = check_gres2(gres)) return e;
  MFS_REQUIRE(cell_size && vx && vy && pv && wx && wy && sv && lphi, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(pv_dt) && dtype_ok(w_dt) && dtype_ok(sv_dt) && dtype_ok(lphi_dt), "dtype");
  Grid2 g{(int)gres[0], (int)gres[1]};
  hipLaunchKernelGGL(k_pressure_update2d, dim3(cdiv(gres[0] * gres[1], 256)), dim3(256), 0, (hipStream_t)stream, g,
                     cell_size[0], cell_size[1], vx, vy, v_dt, pv, pv_dt, wx, wy, w_dt, sv, sv_dt, lphi, lphi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

size_t mfs_pcg2d_workspace_bytes(const int64_t gres[2], int dt) {
  if (!gres || !dtype_ok(dt)) return 0;
  return core_ws_bytes() + 256;
}

int mfs_pcg2d_create(mfs_pcg2d** out, const int64_t gres[2], int dt, void* workspace, size_t workspace_bytes,
                     mfs_stream stream) {
  MFS_REQUIRE(out && workspace, "null argument");
  if (int e = check_gres2(gres)) return e;
  MFS_REQUIRE(dtype_ok(dt), "dtype");
  MFS_REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
  MFS_REQUIRE(workspace_bytes >= mfs_pcg2d_workspace_bytes(gres, dt), "workspace too small");
  mfs_pcg2d* h = new mfs_pcg2d();
  h->g = Grid2{(int)gres[0], (int)gres[1]};
  h->dt = dt;
  if (int e = core_init(h->c, dt, gres[0] * gres[1])) { delete h; return e; }
  core_carve(h->c, (char*)workspace);
  h->lphi = h->wx = h->wy = nullptr;
  h->grid = std::max(1, std::min(h->c.grid_vec, cdiv(gres[0] * gres[1], 256)));
  if (hipMemsetAsync(workspace, 0, mfs_pcg2d_workspace_bytes(gres, dt), (hipStream_t)stream) != hipSuccess) {
    set_error("hipMemsetAsync(workspace) failed");
    core_free(h->c);
    delete h;
    return MFS_E_HIP;
  }
  *out = h;
  return MFS_OK;
}

int mfs_pcg2d_destroy(mfs_pcg2d* h) {
  if (!h) return MFS_OK;
  core_free(h->c);
  delete h;
  return MFS_OK;
}

int mfs_pcg2d_setup(mfs_pcg2d* h, const void* lphi, int lphi_dt, const void* wx, const void* wy, int w_dt) {
  MFS_REQUIRE(h && lphi && wx && wy, "null argument");
  MFS_REQUIRE(dtype_ok(lphi_dt) && dtype_ok(w_dt), "dtype");
  h->lphi = lphi; h->ldt = lphi_dt; h->wx = wx; h->wy = wy; h->wdt = w_dt;
  return MFS_OK;
}

int mfs_pcg2d_bind(mfs_pcg2d* h, void* b, void* x, void* d, void* r, void* q) {
  MFS_REQUIRE(h, "null handle");
  return core_bind(h->c, b, x, d, r, q);
}

int mfs_pcg2d_poll(mfs_pcg2d* h, mfs_stream stream, int64_t* iters, int* done, double* delta, double* alpha,
                   double* beta) {
  MFS_REQUIRE(h, "null handle");
  return core_poll(h->c, (hipStream_t)stream, iters, done, delta, alpha, beta);
}

// solver/PressureCGSolver2D.py:159-177.  Returns MFS_NOT_CONVERGED after max_iter
// iterations; the 2D reference then continues silently (no raise, Q3) -- the caller decides.
int mfs_pcg2d_solve(mfs_pcg2d* h, double tol, int64_t max_iter, int64_t check_every, mfs_stream stream,
                    int64_t* iters_host) {
  MFS_REQUIRE(h && h->c.x && h->lphi, "engine not bound / set up");
  MFS_REQUIRE(max_iter >= 0 && check_every >= 1, "max_iter / check_every");
  hipStream_t st = (hipStream_t)stream;
  int e;
  if ((e = core_begin_pre(h->c, tol, true, st))) return e;
  if ((e = apply2d(h, h->c.x, h->c.q, false, st))) return e;
  if ((e = core_begin_post(h->c, st))) return e;
  if ((e = core_begin_finish(h->c, st))) return e;
  int64_t enq = 0, iters = 0;
  int done = 0;
  if ((e = core_poll(h->c, st, &iters, &done, nullptr, nullptr, nullptr))) return e;
  while (!done && enq < max_iter) {
    const int64_t n = std::min(check_every, max_iter - enq);
    for (int64_t i = 0; i < n; ++i) {
      if ((e = apply2d(h, h->c.d, h->c.q, true, st))) return e;
      if ((e = core_update_xr(h->c, true, st))) return e;
      if ((e = core_update_d(h->c, true, st))) return e;
    }
    enq += n;
    if ((e = core_poll(h->c, st, &iters, &done, nullptr, nullptr, nullptr))) return e;
  }
  if (iters_host) *iters_host = iters;
  return done ? MFS_OK : MFS_NOT_CONVERGED;
}

int64_t mfs_pcg2d_history(mfs_pcg2d* h, double* out_host, int64_t cap, mfs_stream stream) {
  if (!h) { set_error("mfs_pcg2d_history: null handle"); return MFS_E_INVALID; }
  return core_history(h->c, out_host, cap, (hipStream_t)stream);
}

}  // extern "C"
