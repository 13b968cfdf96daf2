// mfs_pressure.hip -- once-per-solve pressure kernels and solid fractions (gfx950).
//
// These run once per solve (not per CG iteration), so they are written for
// coalescing and exact parity, not for the roofline: one thread per cell, the
// fastest thread index on the contiguous z axis (the reference maps it to the
// slowest axis, solver/PressureCGSolver3D.py:54), fp64 arithmetic in the
// reference's accumulation order regardless of the storage dtype.
#include <stdarg.h>

#include "mfs_common.h"

namespace mfs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// -------------------------------------------------------------- fractions ---
// solver/SolidFractionCommon.py:4-16
__device__ __forceinline__ double edge_in_fraction(double l, double r) {
  const bool li = l < 0, ri = r < 0;
  if (li && ri) return 1.0;
  if (!li && !ri) return 0.0;
  const double diff = -fabs(l - r);
  return li ? l / diff : r / diff;
}

__device__ __forceinline__ double pick3(int i, double a, double b, double c) {
  return i == 0 ? a : (i == 1 ? b : c);
}

// solver/SolidFractionCommon.py:18-50, branch for branch (as
// written the result is 1 iff all three vertices are inside, else 0).
__device__ __forceinline__ double tri_in_fraction(double v0, double v1, double v2) {
  const bool i0 = v0 < 0, i1 = v1 < 0, i2 = v2 < 0;
  const int cnt = (int)i0 + (int)i1 + (int)i2;
  if (cnt == 3) return 1.0;
  if (cnt == 2) {
    int out_v = 0;
    if (i0) { out_v = 1; if (i1) out_v = 2; }
    return 1.0 - edge_in_fraction(pick3((out_v + 1) % 3, v0, v1, v2), pick3((out_v + 2) % 3, v0, v1, v2));
  }
  if (cnt == 1) {
    int in_v = 0;
    if (!i0) { in_v = 1; if (!i1) in_v = 2; }
    return edge_in_fraction(pick3((in_v + 1) % 3, v0, v1, v2), pick3((in_v + 2) % 3, v0, v1, v2));
  }
  return 0.0;
}

// solver/SolidFractionCommon.py:52-60
__device__ __forceinline__ double face_in_fraction(double bl, double br, double tl, double tr) {
  const double ce = 0.25 * (bl + br + tl + tr);
  return 0.25 * (tri_in_fraction(bl, br, ce) + tri_in_fraction(br, tr, ce) + tri_in_fraction(tr, tl, ce) +
                 tri_in_fraction(tl, bl, ce));
}

// solver/SolidFraction3D.py:6-26
__global__ void __launch_bounds__(256) k_solid_frac3d(int Nx, int Ny, int Nz, const void* sphi, int sdt, void* wx,
                                                      void* wy, void* wz, int wdt) {
  const int64_t n = (int64_t)Nx * Ny * Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % Nz), y = (int)((i / Nz) % Ny), x = (int)(i / ((int64_t)Nz * Ny));
  const int64_t Sy = 2 * Nz + 1, Sx = (int64_t)(2 * Ny + 1) * Sy;
  auto S = [&](int ox, int oy, int oz) { return ldx(sphi, sdt, (2 * x + ox) * Sx + (2 * y + oy) * Sy + (2 * z + oz)); };
  const double blb = S(0, 0, 0), brb = S(2, 0, 0), tlb = S(0, 2, 0), trb = S(2, 2, 0);
  const double blf = S(0, 0, 2), brf = S(2, 0, 2), tlf = S(0, 2, 2);
  stx(wx, wdt, ((int64_t)x * Ny + y) * Nz + z, 1.0 - face_in_fraction(tlb, blb, tlf, blf));        // :22
  stx(wy, wdt, ((int64_t)x * (Ny + 1) + y) * Nz + z, 1.0 - face_in_fraction(brb, blb, brf, blf));  // :24
  stx(wz, wdt, ((int64_t)x * Ny + y) * (Nz + 1) + z, 1.0 - face_in_fraction(trb, tlb, brb, blb));  // :26
}

// solver/SolidFraction2D.py:6-20.  Cell (x,y) with x<Nx-1, y<Ny-1 writes its four
// faces; a face shared by two such cells receives the same value from both
// (same two nodes), so concurrent duplicate stores are benign.
__global__ void __launch_bounds__(256) k_solid_frac2d(int Nx, int Ny, const void* sphi, int sdt, void* wx, void* wy,
                                                      int wdt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Nx * Ny) return;
  const int y = (int)(i % Ny), x = (int)(i / Ny);
  if (x >= Nx - 1 || y >= Ny - 1) return;
  const int64_t Sx = 2 * Ny + 1;
  const double bl = ldx(sphi, sdt, (2 * x) * Sx + 2 * y), br = ldx(sphi, sdt, (2 * x + 2) * Sx + 2 * y);
  const double tl = ldx(sphi, sdt, (2 * x) * Sx + 2 * y + 2), tr = ldx(sphi, sdt, (2 * x + 2) * Sx + 2 * y + 2);
  stx(wx, wdt, (int64_t)(x + 1) * Ny + y, 1.0 - edge_in_fraction(tr, br));
  stx(wx, wdt, (int64_t)x * Ny + y, 1.0 - edge_in_fraction(tl, bl));
  stx(wy, wdt, (int64_t)x * (Ny + 1) + y + 1, 1.0 - edge_in_fraction(tr, tl));
  stx(wy, wdt, (int64_t)x * (Ny + 1) + y, 1.0 - edge_in_fraction(br, bl));
}

// ------------------------------------------------------------ pressure 3D ---
struct Grid3 {
  int Nx, Ny, Nz;
  __device__ __forceinline__ int64_t c(int x, int y, int z) const { return ((int64_t)x * Ny + y) * Nz + z; }
  __device__ __forceinline__ int64_t fx(int x, int y, int z) const { return ((int64_t)x * Ny + y) * Nz + z; }
  __device__ __forceinline__ int64_t fy(int x, int y, int z) const { return ((int64_t)x * (Ny + 1) + y) * Nz + z; }
  __device__ __forceinline__ int64_t fz(int x, int y, int z) const { return ((int64_t)x * Ny + y) * (Nz + 1) + z; }
  // doubled grid (2N+1)^3 and its 3-vector variant
  __device__ __forceinline__ int64_t dg(int i, int j, int k) const {
    return ((int64_t)i * (2 * Ny + 1) + j) * (2 * Nz + 1) + k;
  }
};

struct Cs3 { double x, y, z; };

// solver/PressureCGSolver3D.py:6-50
__global__ void __launch_bounds__(256)
k_pressure_rhs3d(Grid3 g, Cs3 cs, const void* vx, const void* vy, const void* vz, int vdt, const void* sv, int svdt,
                 const void* lphi, int ldt, const void* wx, const void* wy, const void* wz, int wdt, void* b, int bdt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1 || z == 0 || z >= g.Nz - 1) return;  // boundary untouched
  if (!(ldx(lphi, ldt, i) < 0)) { stx(b, bdt, i, 0.0); return; }
  double bv = 0.0, w;
  w = ldx(wx, wdt, g.fx(x + 1, y, z));
  bv += w * ldx(vx, vdt, g.fx(x + 1, y, z)) / cs.x;
  if (w < 1) bv -= w * ldx(sv, svdt, 3 * g.dg(2 * x + 2, 2 * y + 1, 2 * z + 1) + 0) / cs.x;
  w = ldx(wx, wdt, g.fx(x, y, z));
  bv -= w * ldx(vx, vdt, g.fx(x, y, z)) / cs.x;
  if (w < 1) bv += w * ldx(sv, svdt, 3 * g.dg(2 * x, 2 * y + 1, 2 * z + 1) + 0) / cs.x;
  w = ldx(wy, wdt, g.fy(x, y + 1, z));
  bv += w * ldx(vy, vdt, g.fy(x, y + 1, z)) / cs.y;
  if (w < 1) bv -= w * ldx(sv, svdt, 3 * g.dg(2 * x + 1, 2 * y + 2, 2 * z + 1) + 1) / cs.y;
  w = ldx(wy, wdt, g.fy(x, y, z));
  bv -= w * ldx(vy, vdt, g.fy(x, y, z)) / cs.y;
  if (w < 1) bv += w * ldx(sv, svdt, 3 * g.dg(2 * x + 1, 2 * y, 2 * z + 1) + 1) / cs.y;
  w = ldx(wz, wdt, g.fz(x, y, z + 1));
  bv += w * ldx(vz, vdt, g.fz(x, y, z + 1)) / cs.z;
  if (w < 1) bv -= w * ldx(sv, svdt, 3 * g.dg(2 * x + 1, 2 * y + 1, 2 * z + 2) + 2) / cs.z;
  w = ldx(wz, wdt, g.fz(x, y, z));
  bv -= w * ldx(vz, vdt, g.fz(x, y, z)) / cs.z;
  if (w < 1) bv += w * ldx(sv, svdt, 3 * g.dg(2 * x + 1, 2 * y + 1, 2 * z) + 2) / cs.z;
  stx(b, bdt, i, bv);
}

__device__ __forceinline__ double gf_theta(double phi, double nphi) {
  return fmin(1.0, fmax(0.01, phi / (phi - nphi)));  // PressureCGSolver3D.py:75
}

// solver/PressureCGSolver3D.py:52-130 -- the operator straight from lphi and w.
__global__ void __launch_bounds__(256)
k_pressure_apply3d(Grid3 g, const void* v, void* out, int dt, const void* wx, const void* wy, const void* wz, int wdt,
                   const void* lphi, int ldt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1 || z == 0 || z >= g.Nz - 1) return;
  const double phi = ldx(lphi, ldt, i);
  if (!(phi < 0)) { stx(out, dt, i, 0.0); return; }
  double val = 0.0, diag = 0.0;
  auto tap = [&](int64_t nb, double w) {
    const double nphi = ldx(lphi, ldt, nb);
    if (nphi < 0) { val -= w * ldx(v, dt, nb); diag += w; }
    else          { diag += w / gf_theta(phi, nphi); }
  };
  const int64_t sx = (int64_t)g.Ny * g.Nz, sy = g.Nz;
  tap(i + sx, ldx(wx, wdt, g.fx(x + 1, y, z)));
  tap(i - sx, ldx(wx, wdt, g.fx(x, y, z)));
  tap(i + sy, ldx(wy, wdt, g.fy(x, y + 1, z)));
  tap(i - sy, ldx(wy, wdt, g.fy(x, y, z)));
  tap(i + 1, ldx(wz, wdt, g.fz(x, y, z + 1)));
  tap(i - 1, ldx(wz, wdt, g.fz(x, y, z)));
  val += diag * ldx(v, dt, i);
  stx(out, dt, i, val);
}

// solver/PressureCGSolver3D.py:132-153 -- x,y,z in [1, N-1], in place.
__global__ void __launch_bounds__(256)
k_pressure_update3d(Grid3 g, Cs3 cs, void* vx, void* vy, void* vz, int vdt, const void* pv, int pdt, const void* wx,
                    const void* wy, const void* wz, int wdt, const void* sv, int svdt, const void* lphi, int ldt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || y == 0 || z == 0) return;
  const double pc = ldx(lphi, ldt, i), p = ldx(pv, pdt, i);
  const int64_t sx = (int64_t)g.Ny * g.Nz, sy = g.Nz;
  auto axis = [&](void* vel, int64_t fi, int64_t nb, const void* w, int64_t svi, double c) {
    const double pm = ldx(lphi, ldt, nb);
    if (pc < 0 || pm < 0) {
      const double th = fmin(1.0, fmax(0.01, edge_in_fraction(pc, pm)));
      double nv = ldx(vel, vdt, fi) + (p - ldx(pv, pdt, nb)) * c / th;
      const double ww = ldx(w, wdt, fi);
      nv = ww * nv + (1 - ww) * ldx(sv, svdt, svi);
      stx(vel, vdt, fi, nv);
    }
  };
  axis(vx, g.fx(x, y, z), i - sx, wx, 3 * g.dg(2 * x, 2 * y + 1, 2 * z + 1) + 0, cs.x);
  axis(vy, g.fy(x, y, z), i - sy, wy, 3 * g.dg(2 * x + 1, 2 * y, 2 * z + 1) + 1, cs.y);
  axis(vz, g.fz(x, y, z), i - 1, wz, 3 * g.dg(2 * x + 1, 2 * y + 1, 2 * z) + 2, cs.z);
}

static int check_gres3(const int64_t gres[3]) {
  MFS_REQUIRE(gres != nullptr, "gres is null");
  for (int a = 0; a < 3; ++a) MFS_REQUIRE(gres[a] >= 1 && gres[a] <= 4096, "grid resolution out of range [1,4096]");
  return MFS_OK;
}

}  // namespace mfs

using namespace mfs;

extern "C" {

int mfs_abi_version(void) { return MFS_ABI_VERSION; }
const char* mfs_last_error(void) { return g_err; }

int mfs_device_name(char* buf, size_t cap) {
  MFS_REQUIRE(buf && cap > 0, "buffer");
  int dev = 0;
  MFS_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  MFS_HIP_TRY(hipGetDeviceProperties(&p, dev));
  snprintf(buf, cap, "%s|%s|CUs=%d", p.gcnArchName, p.name, p.multiProcessorCount);
  return MFS_OK;
}

int mfs_solid_frac3d(const int64_t gres[3], const void* sphi, int sphi_dt, void* wx, void* wy, void* wz, int w_dt,
                     mfs_stream stream) {
  if (int e = check_gres3(gres)) return e;
  MFS_REQUIRE(sphi && wx && wy && wz, "null array");
  MFS_REQUIRE(dtype_ok(sphi_dt) && dtype_ok(w_dt), "dtype");
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_solid_frac3d, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (int)gres[0],
                     (int)gres[1], (int)gres[2], sphi, sphi_dt, wx, wy, wz, w_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_solid_frac2d(const int64_t gres[2], const void* sphi, int sphi_dt, void* wx, void* wy, int w_dt,
                     mfs_stream stream) {
  MFS_REQUIRE(gres && gres[0] >= 1 && gres[1] >= 1 && gres[0] <= 65536 && gres[1] <= 65536, "gres");
  MFS_REQUIRE(sphi && wx && wy, "null array");
  MFS_REQUIRE(dtype_ok(sphi_dt) && dtype_ok(w_dt), "dtype");
  const int64_t n = gres[0] * gres[1];
  hipLaunchKernelGGL(k_solid_frac2d, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (int)gres[0],
                     (int)gres[1], sphi, sphi_dt, wx, wy, w_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pressure_rhs3d(const int64_t gres[3], const double cell_size[3], const void* vx, const void* vy,
                       const void* vz, int v_dt, const void* sv, int sv_dt, const void* lphi, int lphi_dt,
                       const void* wx, const void* wy, const void* wz, int w_dt, void* b, int b_dt,
                       mfs_stream stream) {
  if (int e = check_gres3(gres)) return e;
  MFS_REQUIRE(cell_size && vx && vy && vz && sv && lphi && wx && wy && wz && b, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(sv_dt) && dtype_ok(lphi_dt) && dtype_ok(w_dt) && dtype_ok(b_dt), "dtype");
  Grid3 g{(int)gres[0], (int)gres[1], (int)gres[2]};
  Cs3 cs{cell_size[0], cell_size[1], cell_size[2]};
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_pressure_rhs3d, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, cs, vx, vy, vz,
                     v_dt, sv, sv_dt, lphi, lphi_dt, wx, wy, wz, w_dt, b, b_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pressure_apply3d(const int64_t gres[3], const void* v, void* out, int dt, const void* wx, const void* wy,
                         const void* wz, int w_dt, const void* lphi, int lphi_dt, mfs_stream stream) {
  if (int e = check_gres3(gres)) return e;
  MFS_REQUIRE(v && out && wx && wy && wz && lphi, "null array");
  MFS_REQUIRE(v != out, "apply cannot run in place");
  MFS_REQUIRE(dtype_ok(dt) && dtype_ok(w_dt) && dtype_ok(lphi_dt), "dtype");
  Grid3 g{(int)gres[0], (int)gres[1], (int)gres[2]};
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_pressure_apply3d, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, v, out, dt, wx,
                     wy, wz, w_dt, lphi, lphi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_pressure_update3d(const int64_t gres[3], const double cell_size[3], void* vx, void* vy, void* vz, int v_dt,
                          const void* pv, int pv_dt, const void* wx, const void* wy, const void* wz, int w_dt,
                          const void* sv, int sv_dt, const void* lphi, int lphi_dt, mfs_stream stream) {
  if (int e = check_gres3(gres)) return e;
  MFS_REQUIRE(cell_size && vx && vy && vz && pv && wx && wy && wz && sv && lphi, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(pv_dt) && dtype_ok(w_dt) && dtype_ok(sv_dt) && dtype_ok(lphi_dt), "dtype");
  Grid3 g{(int)gres[0], (int)gres[1], (int)gres[2]};
  Cs3 cs{cell_size[0], cell_size[1], cell_size[2]};
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_pressure_update3d, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, cs, vx, vy, vz,
                     v_dt, pv, pv_dt, wx, wy, wz, w_dt, sv, sv_dt, lphi, lphi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

}  // extern "C"
