// mfs_pcg.hip -- the per-iteration hot path of PressureCGSolver3D on gfx950.
//
// Replaces the loop solver/PressureCGSolver3D.py:198-223 (1 numba launch + ~12
// cupy kernels + 3 host syncs per iteration in the reference) by five launches
// per iteration with no host sync:
//
//   k_pcg_apply      q = A d  and per-block partials of d.q        6 scalars/cell
//   k_reduce(DQ)     1 block: partials -> scalars[DQ]
//   k_update_xr      x += a d ; r -= a q ; partials of r.r         6 scalars/cell
//   k_reduce(RR)     1 block: partials -> scalars[RR]
//   k_update_d       convergence test, history, d = r + b d        3 scalars/cell
//
// alpha, beta, delta, the iteration count and a `done` flag live in device
// memory; once `done` is set every later kernel is a no-op, so the iteration
// count and the final state equal the reference's even though the host only
// looks every `check_every` iterations.  Reductions are deterministic: fixed
// shuffle tree per wave, waves in order, blocks in order -- no float atomics.
//
// Storage dtype T is fp32 or fp64; ALL arithmetic (stencil, axpys, dots) is
// fp64 in registers -- the path is HBM-bound, so fp64 math is
// free and fp32 mode differs from the fp64 reference only by storage rounding.
//
// The stencil reads 4 solver-owned coefficient arrays built once per solve by
// k_pcg_setup from (lphi, wx, wy, wz):  diag (the reference's `diag`, ghost-fluid
// terms included, accumulated in its order) and cx,cy,cz = the face weight
// between a cell and its LOWER neighbour on that axis if both are fluid, else 0.
// All four are cell-shaped (rows of Nz, 16-byte aligned when Nz%VEC==0), unlike
// the caller's wz whose rows of Nz+1 break vector alignment.  Same 6 scalars per
// cell as the reference formulation (v, lphi, wx, wy, wz -> out).
#include <stdlib.h>

#include <algorithm>

#include "mfs_common.h"
#include "mfs_pcg_apply.h"

namespace mfs {

constexpr int kBlock = 256;
constexpr int kMaxPartials = 8192;
constexpr int64_t kHistCap = 16384;

enum { S_DQ = MFS_PCG_S_DQ, S_RR = MFS_PCG_S_RR, S_DELTA = MFS_PCG_S_DELTA, S_TOL2 = MFS_PCG_S_TOL2,
       S_DONE = MFS_PCG_S_DONE, S_ITERS = MFS_PCG_S_ITERS, S_ALPHA = MFS_PCG_S_ALPHA, S_BETA = MFS_PCG_S_BETA,
       S_LASTRR = MFS_PCG_S_LASTRR };

// ---------------------------------------------------------------- setup -----
// diag / masked lower-face weights from lphi and w (PressureCGSolver3D.py:59-126).
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_pcg_setup(int Nx, int Ny, int Nz, const void* lphi, int ldt, const void* wx, const void* wy, const void* wz,
            int wdt, T* __restrict__ diag, T* __restrict__ cx, T* __restrict__ cy, T* __restrict__ cz) {
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
  double dg = 0.0;
  const bool interior = x > 0 && x < Nx - 1 && y > 0 && y < Ny - 1 && z > 0 && z < Nz - 1;
  if (interior && fl) {
    auto acc = [&](double nphi, double w) {
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

template <typename T, int VEC>
__device__ __forceinline__ Vec<T, VEC> ldv(const T* p) { return *reinterpret_cast<const Vec<T, VEC>*>(p); }

// ---------------------------------------------------------- vector phases ---
template <typename T, int VEC, typename F>
__device__ __forceinline__ void for_each_vec(int64_t n, F&& f) {
  // f(i, lanes): process elements [i, i+lanes)
  const int64_t nv = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nv; k += stride) f(k * VEC, true);
  // scalar tail (n % VEC elements) handled by the first threads of block 0
  const int64_t tail = n - nv * VEC;
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < tail) f(nv * VEC + threadIdx.x, false);
}

// d = b - q ; r = d ; partial sum r^2          (PressureCGSolver3D.py:202-204)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_cg_init(const T* __restrict__ b, const T* __restrict__ q, T* __restrict__ d, T* __restrict__ r, int64_t n,
          double* __restrict__ partial) {
  double acc = 0.0;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      const Vec<T, VEC> bv = ldv<T, VEC>(b + i), qv = ldv<T, VEC>(q + i);
      Vec<T, VEC> dv;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dv.v[j] = (T)((double)bv.v[j] - (double)qv.v[j]);
        acc += (double)dv.v[j] * (double)dv.v[j];
      }
      *reinterpret_cast<Vec<T, VEC>*>(d + i) = dv;
      *reinterpret_cast<Vec<T, VEC>*>(r + i) = dv;
    } else {
      const T dv = (T)((double)b[i] - (double)q[i]);
      d[i] = dv; r[i] = dv;
      acc += (double)dv * (double)dv;
    }
  });
  const double tot = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// alpha = delta / dq ; x += alpha d ; r -= alpha q ; partial sum r^2   (:211-216)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_update_xr(T* __restrict__ x, const T* __restrict__ d, T* __restrict__ r, const T* __restrict__ q, int64_t n,
            const double* __restrict__ scal, double* __restrict__ partial) {
  if (scal[S_DONE] != 0.0) return;
  const double alpha = scal[S_DELTA] / scal[S_DQ];
  double acc = 0.0;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      Vec<T, VEC> xv = ldv<T, VEC>(x + i), rv = ldv<T, VEC>(r + i);
      const Vec<T, VEC> dv = ldv<T, VEC>(d + i), qv = ldv<T, VEC>(q + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xv.v[j] = (T)((double)xv.v[j] + alpha * (double)dv.v[j]);
        rv.v[j] = (T)((double)rv.v[j] - alpha * (double)qv.v[j]);
        acc += (double)rv.v[j] * (double)rv.v[j];
      }
      *reinterpret_cast<Vec<T, VEC>*>(x + i) = xv;
      *reinterpret_cast<Vec<T, VEC>*>(r + i) = rv;
    } else {
      const T xn = (T)((double)x[i] + alpha * (double)d[i]);
      const T rn = (T)((double)r[i] - alpha * (double)q[i]);
      x[i] = xn; r[i] = rn;
      acc += (double)rn * (double)rn;
    }
  });
  const double tot = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// convergence test (:218), bookkeeping, beta (:220), d = r + beta d (:221)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_update_d(T* __restrict__ d, const T* __restrict__ r, int64_t n, double* __restrict__ scal,
           double* __restrict__ hist, int64_t hist_cap) {
  if (scal[S_DONE] != 0.0) return;
  const double rr = scal[S_RR], delta = scal[S_DELTA], tol2 = scal[S_TOL2];
  const bool conv = rr < tol2;
  const double beta = rr / delta;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double dq = scal[S_DQ];
    const int64_t it = (int64_t)scal[S_ITERS];
    if (2 * it + 2 < hist_cap) { hist[2 * it + 1] = dq; hist[2 * it + 2] = rr; }
    scal[S_ITERS] = (double)(it + 1);
    scal[S_LASTRR] = rr;
    scal[S_ALPHA] = delta / dq;
    if (conv) scal[S_DONE] = 1.0; else scal[S_BETA] = beta;
  }
  if (conv) return;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      Vec<T, VEC> dv = ldv<T, VEC>(d + i);
      const Vec<T, VEC> rv = ldv<T, VEC>(r + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dv.v[j] = (T)((double)rv.v[j] + beta * (double)dv.v[j]);
      *reinterpret_cast<Vec<T, VEC>*>(d + i) = dv;
    } else {
      d[i] = (T)((double)r[i] + beta * (double)d[i]);
    }
  });
}

// x *= 0.0 (:198) -- a multiply, not a memset, so NaN/inf survive as in the reference.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_scale0(T* __restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = (T)((double)x[i] * 0.0);
}

// one block: partials[0..count) -> scal[which]; fixed order => deterministic.
__global__ void __launch_bounds__(kBlock)
k_reduce(const double* __restrict__ partial, int count, double* __restrict__ scal, int which, int check_done) {
  if (check_done && scal[S_DONE] != 0.0) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
  const double tot = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) {
    scal[which] = tot;
    if (which == S_DQ) scal[S_DELTA] = scal[S_RR];  // the iteration that starts here begins from the latest r.r
  }
}

__global__ void k_begin_init(double* scal, double tol2) {
  if (threadIdx.x == 0) {
    for (int i = 0; i < MFS_PCG_NSCALARS; ++i) scal[i] = 0.0;
    scal[S_TOL2] = tol2;
  }
}

__global__ void k_begin_finish(double* scal, double* hist) {
  if (threadIdx.x == 0) {
    const double rr = scal[S_RR];
    scal[S_DELTA] = rr;
    scal[S_LASTRR] = rr;
    hist[0] = rr;
    if (rr < scal[S_TOL2]) scal[S_DONE] = 1.0;  // `if not self.delta < tol ** 2` (:206)
  }
}

}  // namespace mfs

using namespace mfs;

struct mfs_pcg3d {
  int Nx, Ny, Nz, dt;
  int64_t n;
  size_t elt;
  char* ws;
  size_t ws_bytes;
  double *scal, *hist, *part_dq, *part_rr;
  void *diag, *cx, *cy, *cz;
  void *b, *x, *d, *r, *q;
  int n_part_dq, n_part_rr;
  int grid_apply, grid_vec, cus;
  int variant, xchunk, nt, bpc;   // apply-kernel tuning (mfs_pcg3d_tune)
  bool vec_ok;
  bool is_setup;
  double* pinned;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Coefficient arrays are staggered by an odd number of 4 KiB pages so that the six
// streams of one tile do not all start on the same HBM channel/bank phase.
static size_t coef_stride(int64_t n, size_t elt) { return align_up((size_t)n * elt, 4096) + 4096 * 3 + 256; }

static int env_int(const char* name, int defv) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : defv;
}

template <typename T, int VEC>
static int launch_apply_v(mfs_pcg3d* h, const T* v, T* out, int xb, int xe, double* partial, const double* done,
                          hipStream_t st, int* grid_out) {
  const T *dg = (const T*)h->diag, *cx = (const T*)h->cx, *cy = (const T*)h->cy, *cz = (const T*)h->cz;
  const int nzv = h->Nz / VEC;
  const int64_t ipp = (int64_t)(h->Ny - 2) * nzv;
  int variant = h->variant;
  const size_t lds = 2 * ((size_t)kApplyBlock * VEC + 2 * (size_t)h->Nz) * sizeof(T);
  if (variant == 2 && lds > 64 * 1024) variant = 1;          // absurdly long rows: skip the LDS image
  const int xchunk = std::max(0, h->xchunk);   // 0 = no cap on the length of one march
  ApplyArgs a{h->Nx, h->Ny, h->Nz, xb, xe, xchunk};
  if (variant == 0) {
    const int64_t items = (int64_t)(xe - xb) * ipp;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->grid_apply, (items + kApplyBlock - 1) / kApplyBlock));
    hipLaunchKernelGGL((k_pcg_apply_direct<T, VEC>), dim3(grid), dim3(kApplyBlock), 0, st, v, out, dg, cx, cy, cz, a,
                       partial, done);
    *grid_out = grid;
  } else {
    const int64_t tiles = (ipp + kApplyBlock - 1) / kApplyBlock;
    const int64_t total = tiles * (xe - xb);                 // (tile, plane) pairs, cut into `grid` equal segments
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(kMaxPartials, h->cus * h->bpc), total));
    // Nontemporal loads for the once-read coefficient streams pay off only when the
    // apply's working set (6 arrays) cannot sit in the 256 MiB Infinity Cache anyway;
    // below that, default caching lets the next iteration hit on-die.  nt < 0 = auto.
    const bool nt = h->nt < 0 ? (6.0 * (double)h->n * sizeof(T) > 200e6) : (h->nt != 0);
    if (variant == 2) {
      if (nt) hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, true, 1>), dim3(grid), dim3(kApplyBlock), lds, st, v, out, dg, cx, cy, cz, a, partial, done);
      else    hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, true, 0>), dim3(grid), dim3(kApplyBlock), lds, st, v, out, dg, cx, cy, cz, a, partial, done);
    } else {
      if (nt) hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, false, 1>), dim3(grid), dim3(kApplyBlock), 0, st, v, out, dg, cx, cy, cz, a, partial, done);
      else    hipLaunchKernelGGL((k_pcg_apply_march<T, VEC, false, 0>), dim3(grid), dim3(kApplyBlock), 0, st, v, out, dg, cx, cy, cz, a, partial, done);
    }
    *grid_out = grid;
  }
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

template <typename T>
static int launch_apply(mfs_pcg3d* h, const void* v, void* out, int xb, int xe, double* partial, int use_done,
                        hipStream_t st, int* grid_out) {
  if (xe - xb <= 0) { *grid_out = 0; return MFS_OK; }
  constexpr int VEC = VecOf<T>::N;
  const bool vec = h->vec_ok && ((uintptr_t)v % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const double* done = use_done ? h->scal + S_DONE : nullptr;
  if (vec) return launch_apply_v<T, VEC>(h, (const T*)v, (T*)out, xb, xe, partial, done, st, grid_out);
  return launch_apply_v<T, 1>(h, (const T*)v, (T*)out, xb, xe, partial, done, st, grid_out);
}

static int apply_dispatch(mfs_pcg3d* h, const void* v, void* out, int64_t xb, int64_t xe, double* partial,
                          int use_done, hipStream_t st, int* grid_out) {
  const int b = (int)std::max<int64_t>(xb, 1), e = (int)std::min<int64_t>(xe, h->Nx - 1);
  if (h->Ny < 3 || h->Nz < 3) { *grid_out = 0; return MFS_OK; }
  return h->dt == MFS_F32 ? launch_apply<float>(h, v, out, b, e, partial, use_done, st, grid_out)
                          : launch_apply<double>(h, v, out, b, e, partial, use_done, st, grid_out);
}

#define DISPATCH_T(h, CALL_F32, CALL_F64) \
  do { if ((h)->dt == MFS_F32) { CALL_F32; } else { CALL_F64; } } while (0)

static bool vec_flat_ok(const mfs_pcg3d* h) {
  auto al = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  return al(h->b) && al(h->x) && al(h->d) && al(h->r) && al(h->q);
}

extern "C" {

size_t mfs_pcg3d_workspace_bytes(const int64_t gres[3], int dt) {
  if (!gres || !dtype_ok(dt)) return 0;
  const int64_t n = gres[0] * gres[1] * gres[2];
  size_t tot = 256;                                  // scalars
  tot += align_up((size_t)kHistCap * 8, 256);        // history
  tot += 2 * align_up((size_t)kMaxPartials * 8, 256);  // partials
  tot += 4 * coef_stride(n, dtype_size(dt)) + 4096;
  return tot;
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
  h->elt = dtype_size(dt);
  h->ws = (char*)workspace; h->ws_bytes = workspace_bytes;
  char* p = h->ws;
  h->scal = (double*)p; p += 256;
  h->hist = (double*)p; p += align_up((size_t)kHistCap * 8, 256);
  h->part_dq = (double*)p; p += align_up((size_t)kMaxPartials * 8, 256);
  h->part_rr = (double*)p; p += align_up((size_t)kMaxPartials * 8, 256);
  p = (char*)align_up((uintptr_t)p, 4096);
  const size_t cs = coef_stride(h->n, h->elt);
  h->diag = p; h->cx = p + cs; h->cy = p + 2 * cs; h->cz = p + 3 * cs;
  h->b = h->x = h->d = h->r = h->q = nullptr;
  h->n_part_dq = h->n_part_rr = 0;
  const int vec = dt == MFS_F32 ? 4 : 2;
  h->vec_ok = (h->Nz % vec) == 0 && h->Nz >= 2 * vec;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  }
  h->cus = cus;
  h->variant = env_int("MFS_APPLY_VARIANT", 2);
  h->xchunk = env_int("MFS_APPLY_XCHUNK", 0);
  h->nt = env_int("MFS_APPLY_NT", -1);
  h->bpc = env_int("MFS_APPLY_BLOCKS_PER_CU", 2);
  h->grid_apply = std::min(kMaxPartials, cus * 8);
  h->grid_vec = std::min(kMaxPartials, cus * env_int("MFS_VEC_BLOCKS_PER_CU", 8));
  h->is_setup = false;
  h->pinned = nullptr;
  if (hipHostMalloc((void**)&h->pinned, MFS_PCG_NSCALARS * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc for the poll buffer failed");
    delete h;
    return MFS_E_HIP;
  }
  if (hipMemsetAsync(workspace, 0, mfs_pcg3d_workspace_bytes(gres, dt), (hipStream_t)stream) != hipSuccess) {
    set_error("hipMemsetAsync(workspace) failed");
    (void)hipHostFree(h->pinned);
    delete h;
    return MFS_E_HIP;
  }
  *out = h;
  return MFS_OK;
}

int mfs_pcg3d_destroy(mfs_pcg3d* h) {
  if (!h) return MFS_OK;
  if (h->pinned) (void)hipHostFree(h->pinned);
  delete h;
  return MFS_OK;
}

int mfs_pcg3d_setup(mfs_pcg3d* h, const void* lphi, int lphi_dt, const void* wx, const void* wy, const void* wz,
                    int w_dt, mfs_stream stream) {
  MFS_REQUIRE(h && lphi && wx && wy && wz, "null argument");
  MFS_REQUIRE(dtype_ok(lphi_dt) && dtype_ok(w_dt), "dtype");
  const int grid = cdiv(h->n, kBlock);
  DISPATCH_T(h,
             hipLaunchKernelGGL((k_pcg_setup<float>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, h->Nx, h->Ny,
                                h->Nz, lphi, lphi_dt, wx, wy, wz, w_dt, (float*)h->diag, (float*)h->cx, (float*)h->cy,
                                (float*)h->cz),
             hipLaunchKernelGGL((k_pcg_setup<double>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, h->Nx, h->Ny,
                                h->Nz, lphi, lphi_dt, wx, wy, wz, w_dt, (double*)h->diag, (double*)h->cx,
                                (double*)h->cy, (double*)h->cz));
  MFS_LAUNCH_CHECK();
  h->is_setup = true;
  return MFS_OK;
}

int mfs_pcg3d_apply(mfs_pcg3d* h, const void* v, void* out, int64_t x_begin, int64_t x_end, mfs_stream stream) {
  MFS_REQUIRE(h && v && out, "null argument");
  MFS_REQUIRE(v != out, "apply cannot run in place");
  MFS_REQUIRE(h->is_setup, "mfs_pcg3d_setup has not been called");
  int grid = 0;
  if (int e = apply_dispatch(h, v, out, x_begin, x_end, h->part_dq, 0, (hipStream_t)stream, &grid)) return e;
  h->n_part_dq = grid;
  return MFS_OK;
}

int mfs_pcg3d_bind(mfs_pcg3d* h, void* b, void* x, void* d, void* r, void* q) {
  MFS_REQUIRE(h && b && x && d && r && q, "null argument");
  void* a[5] = {b, x, d, r, q};
  for (int i = 0; i < 5; ++i) {
    MFS_REQUIRE(((uintptr_t)a[i] % h->elt) == 0, "CG vector not aligned to its element size");
    for (int j = i + 1; j < 5; ++j) MFS_REQUIRE(a[i] != a[j], "CG vectors must be distinct arrays");
  }
  h->b = b; h->x = x; h->d = d; h->r = r; h->q = q;
  return MFS_OK;
}

void* mfs_pcg3d_scalars(mfs_pcg3d* h) { return h ? h->scal : nullptr; }

int mfs_pcg3d_tune(mfs_pcg3d* h, int variant, int xchunk, int blocks_per_cu, int nontemporal) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(variant >= 0 && variant <= 2, "variant must be 0 (direct), 1 (march) or 2 (march + LDS)");
  MFS_REQUIRE(xchunk >= 0 && blocks_per_cu >= 1, "xchunk must be >= 0 (0 = auto), blocks_per_cu >= 1");
  h->variant = variant; h->xchunk = xchunk; h->nt = nontemporal; h->bpc = blocks_per_cu;
  return MFS_OK;
}

int mfs_pcg3d_phase_apply(mfs_pcg3d* h, int64_t x_begin, int64_t x_end, int first, mfs_stream stream) {
  MFS_REQUIRE(h && h->d && h->is_setup, "engine not bound / set up");
  if (first) h->n_part_dq = 0;
  MFS_REQUIRE(h->n_part_dq + std::max(h->grid_apply, h->cus * h->bpc) <= kMaxPartials, "too many apply ranges in one iteration");
  int grid = 0;
  if (int e = apply_dispatch(h, h->d, h->q, x_begin, x_end, h->part_dq + h->n_part_dq, 1, (hipStream_t)stream, &grid))
    return e;
  h->n_part_dq += grid;
  return MFS_OK;
}

int mfs_pcg3d_phase_reduce(mfs_pcg3d* h, int which, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(which == 0 || which == 1, "which must be 0 (d.q) or 1 (r.r)");
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, which == 0 ? h->part_dq : h->part_rr,
                     which == 0 ? h->n_part_dq : h->n_part_rr, h->scal, which == 0 ? S_DQ : S_RR, 1);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pcg3d_phase_update_xr(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h && h->x, "engine not bound");
  const bool vec = vec_flat_ok(h);
  const int grid = std::max(1, (int)std::min<int64_t>(h->grid_vec, (h->n / (vec ? (h->dt == MFS_F32 ? 4 : 2) : 1) + kBlock - 1) / kBlock));
  hipStream_t st = (hipStream_t)stream;
  if (h->dt == MFS_F32) {
    if (vec) hipLaunchKernelGGL((k_update_xr<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)h->x, (const float*)h->d, (float*)h->r, (const float*)h->q, h->n, h->scal, h->part_rr);
    else hipLaunchKernelGGL((k_update_xr<float, 1>), dim3(grid), dim3(kBlock), 0, st, (float*)h->x, (const float*)h->d, (float*)h->r, (const float*)h->q, h->n, h->scal, h->part_rr);
  } else {
    if (vec) hipLaunchKernelGGL((k_update_xr<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)h->x, (const double*)h->d, (double*)h->r, (const double*)h->q, h->n, h->scal, h->part_rr);
    else hipLaunchKernelGGL((k_update_xr<double, 1>), dim3(grid), dim3(kBlock), 0, st, (double*)h->x, (const double*)h->d, (double*)h->r, (const double*)h->q, h->n, h->scal, h->part_rr);
  }
  MFS_LAUNCH_CHECK();
  h->n_part_rr = grid;
  return MFS_OK;
}

int mfs_pcg3d_phase_update_d(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h && h->d, "engine not bound");
  const bool vec = vec_flat_ok(h);
  const int grid = std::max(1, (int)std::min<int64_t>(h->grid_vec, (h->n / (vec ? (h->dt == MFS_F32 ? 4 : 2) : 1) + kBlock - 1) / kBlock));
  hipStream_t st = (hipStream_t)stream;
  if (h->dt == MFS_F32) {
    if (vec) hipLaunchKernelGGL((k_update_d<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)h->d, (const float*)h->r, h->n, h->scal, h->hist, kHistCap);
    else hipLaunchKernelGGL((k_update_d<float, 1>), dim3(grid), dim3(kBlock), 0, st, (float*)h->d, (const float*)h->r, h->n, h->scal, h->hist, kHistCap);
  } else {
    if (vec) hipLaunchKernelGGL((k_update_d<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)h->d, (const double*)h->r, h->n, h->scal, h->hist, kHistCap);
    else hipLaunchKernelGGL((k_update_d<double, 1>), dim3(grid), dim3(kBlock), 0, st, (double*)h->d, (const double*)h->r, h->n, h->scal, h->hist, kHistCap);
  }
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pcg3d_begin_local(mfs_pcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_begin_init, dim3(1), dim3(64), 0, st, h->scal, tol * tol);
  MFS_LAUNCH_CHECK();
  const int gs = std::max(1, (int)std::min<int64_t>(h->grid_vec, (h->n + kBlock - 1) / kBlock));
  DISPATCH_T(h, hipLaunchKernelGGL((k_scale0<float>), dim3(gs), dim3(kBlock), 0, st, (float*)h->x, h->n),
             hipLaunchKernelGGL((k_scale0<double>), dim3(gs), dim3(kBlock), 0, st, (double*)h->x, h->n));
  MFS_LAUNCH_CHECK();
  int grid = 0;
  if (int e = apply_dispatch(h, h->x, h->q, 1, h->Nx - 1, h->part_dq, 0, st, &grid)) return e;  // q = A x  (:201)
  const bool vec = vec_flat_ok(h);
  const int g2 = std::max(1, (int)std::min<int64_t>(h->grid_vec, (h->n / (vec ? (h->dt == MFS_F32 ? 4 : 2) : 1) + kBlock - 1) / kBlock));
  if (h->dt == MFS_F32) {
    if (vec) hipLaunchKernelGGL((k_cg_init<float, 4>), dim3(g2), dim3(kBlock), 0, st, (const float*)h->b, (const float*)h->q, (float*)h->d, (float*)h->r, h->n, h->part_rr);
    else hipLaunchKernelGGL((k_cg_init<float, 1>), dim3(g2), dim3(kBlock), 0, st, (const float*)h->b, (const float*)h->q, (float*)h->d, (float*)h->r, h->n, h->part_rr);
  } else {
    if (vec) hipLaunchKernelGGL((k_cg_init<double, 2>), dim3(g2), dim3(kBlock), 0, st, (const double*)h->b, (const double*)h->q, (double*)h->d, (double*)h->r, h->n, h->part_rr);
    else hipLaunchKernelGGL((k_cg_init<double, 1>), dim3(g2), dim3(kBlock), 0, st, (const double*)h->b, (const double*)h->q, (double*)h->d, (double*)h->r, h->n, h->part_rr);
  }
  MFS_LAUNCH_CHECK();
  h->n_part_rr = g2;
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st, h->part_rr, h->n_part_rr, h->scal, S_RR, 0);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pcg3d_begin_finish(mfs_pcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  hipLaunchKernelGGL(k_begin_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, h->scal, h->hist);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pcg3d_begin(mfs_pcg3d* h, double tol, mfs_stream stream) {
  if (int e = mfs_pcg3d_begin_local(h, tol, stream)) return e;
  return mfs_pcg3d_begin_finish(h, stream);
}

int mfs_pcg3d_iterate(mfs_pcg3d* h, int64_t n, mfs_stream stream) {
  MFS_REQUIRE(h && h->x && h->is_setup, "engine not bound / set up");
  for (int64_t i = 0; i < n; ++i) {
    int e;
    if ((e = mfs_pcg3d_phase_apply(h, 1, h->Nx - 1, 1, stream))) return e;
    if ((e = mfs_pcg3d_phase_reduce(h, 0, stream))) return e;
    if ((e = mfs_pcg3d_phase_update_xr(h, stream))) return e;
    if ((e = mfs_pcg3d_phase_reduce(h, 1, stream))) return e;
    if ((e = mfs_pcg3d_phase_update_d(h, stream))) return e;
  }
  return MFS_OK;
}

int mfs_pcg3d_poll(mfs_pcg3d* h, mfs_stream stream, int64_t* iters, int* done, double* delta, double* alpha,
                   double* beta) {
  MFS_REQUIRE(h, "null handle");
  hipStream_t st = (hipStream_t)stream;
  MFS_HIP_TRY(hipMemcpyAsync(h->pinned, h->scal, MFS_PCG_NSCALARS * sizeof(double), hipMemcpyDeviceToHost, st));
  MFS_HIP_TRY(hipStreamSynchronize(st));
  if (iters) *iters = (int64_t)h->pinned[S_ITERS];
  if (done) *done = h->pinned[S_DONE] != 0.0;
  if (delta) *delta = h->pinned[S_LASTRR];
  if (alpha) *alpha = h->pinned[S_ALPHA];
  if (beta) *beta = h->pinned[S_BETA];
  return MFS_OK;
}

int mfs_pcg3d_solve(mfs_pcg3d* h, double tol, int64_t max_iter, int64_t check_every, mfs_stream stream,
                    int64_t* iters_host) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(max_iter >= 0 && check_every >= 1, "max_iter / check_every");
  if (int e = mfs_pcg3d_begin(h, tol, stream)) return e;
  int64_t enq = 0, iters = 0;
  int done = 0;
  if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  while (!done && enq < max_iter) {
    const int64_t n = std::min(check_every, max_iter - enq);
    if (int e = mfs_pcg3d_iterate(h, n, stream)) return e;
    enq += n;
    if (int e = mfs_pcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  }
  if (iters_host) *iters_host = iters;
  return done ? MFS_OK : MFS_NOT_CONVERGED;
}

int64_t mfs_pcg3d_history(mfs_pcg3d* h, double* out_host, int64_t cap, mfs_stream stream) {
  if (!h || !out_host || cap < 0) { set_error("mfs_pcg3d_history: bad argument"); return MFS_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(h->pinned, h->scal, MFS_PCG_NSCALARS * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { set_error("history: scalar readback failed"); return MFS_E_HIP; }
  int64_t cnt = std::min<int64_t>(2 * (int64_t)h->pinned[S_ITERS] + 1, kHistCap);
  cnt = std::min(cnt, cap);
  if (cnt > 0) {
    if (hipMemcpyAsync(out_host, h->hist, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { set_error("history: copy failed"); return MFS_E_HIP; }
  }
  return cnt;
}

}  // extern "C"
