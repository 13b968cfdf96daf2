// mfs_density.hip -- the once-per-solve kernels of DensityCGSolver3D on gfx950.
//
// Reference: solver/DensityCGSolver3D.py.  The solver's CG loop runs on the pressure engine
// (mfs_pcg.hip, mfs_pcg3d_setup_density): its operator (:118-207) is the pressure stencil with
// a different diagonal and one asymmetric tap.  Here: the particle splat (:8-36), fix_volume
// (:38-86), the right-hand side (:88-116), the stateless operator apply (the module-level
// matvecmul :327-331), compute_displacement (:209-222) and the particle gather
// apply_displacement (:224-253).  One thread per cell / particle, fastest index on the
// contiguous axis, fp64 arithmetic in the reference's order whatever the storage dtype.
#include <math.h>

#include "mfs_common.h"

// No FMA contraction in this file: base indices, float32-rounded grid positions and weights must round where
// the reference's separate multiply and add round (a contracted a*b+c flips a float32 rounding now and then).
#pragma clang fp contract(off)

namespace mfs {

struct DGrid {
  int Nx, Ny, Nz;
  __device__ __forceinline__ int64_t fx(int x, int y, int z) const { return ((int64_t)x * Ny + y) * Nz + z; }
  __device__ __forceinline__ int64_t fy(int x, int y, int z) const { return ((int64_t)x * (Ny + 1) + y) * Nz + z; }
  __device__ __forceinline__ int64_t fz(int x, int y, int z) const { return ((int64_t)x * Ny + y) * (Nz + 1) + z; }
  __device__ __forceinline__ int64_t dg(int i, int j, int k) const {
    return ((int64_t)i * (2 * Ny + 1) + j) * (2 * Nz + 1) + k;
  }
};
struct D3 { double v[3]; };

// solver/SolidFractionCommon.py:4-16
__device__ __forceinline__ double d_edge_in_fraction(double l, double r) {
  const bool li = l < 0, ri = r < 0;
  if (li && ri) return 1.0;
  if (!li && !ri) return 0.0;
  const double diff = -fabs(l - r);
  return li ? l / diff : r / diff;
}

__device__ __forceinline__ void atomic_addx(void* p, int dt, int64_t i, double v) {
  if (dt == MFS_F32) atomicAdd((float*)p + i, (float)v); else atomicAdd((double*)p + i, v);
}

// trilinear stencil of a particle on a grid whose samples sit at (index + bias) * cell_size + bound_min:
// base index gi and the |gx - x| / cell_size weights, exactly as :17-22 / :235-239
__device__ __forceinline__ void particle_cell(const void* px, int pdt, int64_t P, D3 bmin, D3 cs, D3 bias, long long gi[3],
                                              double w[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double x = ldx(px, pdt, 3 * P + d);
    gi[d] = (long long)floor((x - bmin.v[d]) / cs.v[d] - bias.v[d]);
    const double gx = ((double)gi[d] + bias.v[d]) * cs.v[d] + bmin.v[d];
    w[d] = fabs(gx - x) / cs.v[d];
  }
}

__device__ __forceinline__ double corner_weight(int i, double w) { return (double)i + (i ? -1.0 : 1.0) * (1.0 - w); }

// initialize_density_kernel :8-36 -- scatter particle mass and volume to the 8 surrounding cell centres
__global__ void __launch_bounds__(256)
k_density_splat(DGrid g, D3 bmin, D3 cs, const void* px, int pxdt, const void* pm, int pmdt, double pvol, int64_t P,
                void* gm, void* gvol, int gdt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const double m = ldx(pm, pmdt, p);
  long long gi[3];
  double w[3];
  particle_cell(px, pxdt, p, bmin, cs, D3{{0.5, 0.5, 0.5}}, gi, w);
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz) {
        const int cx = (int)max(0LL, min((long long)g.Nx - 1, gi[0] + ix));
        const int cy = (int)max(0LL, min((long long)g.Ny - 1, gi[1] + iy));
        const int cz = (int)max(0LL, min((long long)g.Nz - 1, gi[2] + iz));
        const double weight = corner_weight(ix, w[0]) * corner_weight(iy, w[1]) * corner_weight(iz, w[2]);
        const int64_t c = g.fx(cx, cy, cz);
        atomic_addx(gm, gdt, c, weight * m);
        atomic_addx(gvol, gdt, c, weight * pvol);
      }
}

// the same scatter, one workgroup per tile of 8^3 cells of a tile-sorted particle order (mfs_particle_tile_sort3d,
// csrc/mfs_particles.hip): the tile's cell centres [c0 - 1, c0 + 8] per axis staged in LDS, one global atomic per touched
// cell and tile; contributions outside the staged box (a particle that moved since the sort) take the global atomic
__global__ void __launch_bounds__(256)
k_density_splat_tiled(DGrid g, D3 bmin, D3 cs, int t1, int t2, const void* px, int pxdt, const void* pm, int pmdt, double pvol,
                      const int* __restrict__ perm, const int* __restrict__ tstart, void* gm, void* gvol, int gdt) {
  constexpr int TB = 8, E = TB + 2;
  __shared__ double lm[E * E * E], lv[E * E * E];
  const int tile = blockIdx.x;
  const int a = tstart[tile], b = tstart[tile + 1];
  if (a == b) return;
  const int tz = tile % t2, ty = (tile / t2) % t1, tx = tile / (t2 * t1);
  const int o[3] = {tx * TB - 1, ty * TB - 1, tz * TB - 1};
  for (int l = threadIdx.x; l < E * E * E; l += 256) { lm[l] = 0.0; lv[l] = 0.0; }
  __syncthreads();
  for (int i = a + threadIdx.x; i < b; i += 256) {
    const int64_t p = perm[i];
    const double m = ldx(pm, pmdt, p);
    long long gi[3];
    double w[3];
    particle_cell(px, pxdt, p, bmin, cs, D3{{0.5, 0.5, 0.5}}, gi, w);
    for (int ix = 0; ix < 2; ++ix)
      for (int iy = 0; iy < 2; ++iy)
        for (int iz = 0; iz < 2; ++iz) {
          const int cx = (int)max(0LL, min((long long)g.Nx - 1, gi[0] + ix));
          const int cy = (int)max(0LL, min((long long)g.Ny - 1, gi[1] + iy));
          const int cz = (int)max(0LL, min((long long)g.Nz - 1, gi[2] + iz));
          const double weight = corner_weight(ix, w[0]) * corner_weight(iy, w[1]) * corner_weight(iz, w[2]);
          const int lx = cx - o[0], ly = cy - o[1], lz = cz - o[2];
          if ((unsigned)lx < (unsigned)E && (unsigned)ly < (unsigned)E && (unsigned)lz < (unsigned)E) {
            const int l = (lx * E + ly) * E + lz;
            atomicAdd(&lm[l], weight * m);
            atomicAdd(&lv[l], weight * pvol);
          } else {
            const int64_t c = g.fx(cx, cy, cz);
            atomic_addx(gm, gdt, c, weight * m);
            atomic_addx(gvol, gdt, c, weight * pvol);
          }
        }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < E * E * E; l += 256) {
    const double vm = lm[l], vv = lv[l];
    if (vm == 0.0 && vv == 0.0) continue;
    const int lz = l % E, ly = (l / E) % E, lx = l / (E * E);
    const int64_t c = g.fx(o[0] + lx, o[1] + ly, o[2] + lz);
    atomic_addx(gm, gdt, c, vm);
    atomic_addx(gvol, gdt, c, vv);
  }
}

__device__ __forceinline__ double nonsolid_frac(const DGrid& g, const void* wx, const void* wy, const void* wz, int wdt,
                                                int x, int y, int z) {
  return (ldx(wx, wdt, g.fx(x, y, z)) + ldx(wx, wdt, g.fx(x + 1, y, z)) + ldx(wy, wdt, g.fy(x, y, z)) +
          ldx(wy, wdt, g.fy(x, y + 1, z)) + ldx(wz, wdt, g.fz(x, y, z)) + ldx(wz, wdt, g.fz(x, y, z + 1))) / 6;
}

// fix_volume_kernel :38-86 (interior cells, in place on gvol)
__global__ void __launch_bounds__(256)
k_density_fix_volume(DGrid g, double cvol, double dx, void* gvol, int gdt, const void* sphi, int sdt, const void* lphi,
                     int ldt, const void* wx, const void* wy, const void* wz, int wdt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1 || z == 0 || z >= g.Nz - 1) return;
  double fluid_vol = ldx(gvol, gdt, i);
  const bool near_solid = ldx(sphi, sdt, g.dg(2 * x + 1, 2 * y + 1, 2 * z + 1)) < dx;
  const int64_t sx = (int64_t)g.Ny * g.Nz, sy = g.Nz;
  const bool internal = ldx(lphi, ldt, i) < 0 && ldx(lphi, ldt, i + sx) < 0 && ldx(lphi, ldt, i - sx) < 0 &&
                        ldx(lphi, ldt, i + sy) < 0 && ldx(lphi, ldt, i - sy) < 0 && ldx(lphi, ldt, i + 1) < 0 &&
                        ldx(lphi, ldt, i - 1) < 0;
  if (internal && !near_solid) fluid_vol = cvol;
  stx(gvol, gdt, i, fmin(fluid_vol, cvol * nonsolid_frac(g, wx, wy, wz, wdt, x, y, z)));
}

// initialize_solver_kernel :88-116 -- b = (1 - clamp(density / rho0, 0.5, 1.5)) / dt in fluid cells
__global__ void __launch_bounds__(256)
k_density_rhs(DGrid g, double rho0, double cvol, double dt, const void* gm, const void* gvol, int gdt, const void* lphi,
              int ldt, const void* wx, const void* wy, const void* wz, int wdt, void* b, int bdt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1 || z == 0 || z >= g.Nz - 1) return;
  if (ldx(lphi, ldt, i) >= 0) { stx(b, bdt, i, 0.0); return; }
  const double solid_vol = (1 - nonsolid_frac(g, wx, wy, wz, wdt, x, y, z)) * cvol;
  const double solid_mass = rho0 * solid_vol;
  const double cell_mass = ldx(gm, gdt, i) + solid_mass;
  const double cell_vol = ldx(gvol, gdt, i) + solid_vol;
  double density_frac = cell_mass / fmax(cell_vol, 1e-10) / rho0;
  if (cell_mass < 1e-10) density_frac = 1;
  density_frac = fmax(0.5, fmin(1.5, density_frac));
  stx(b, bdt, i, (1 - density_frac) / dt);
}

// matvecmul_kernel :118-207 straight from lphi and w (the module-level function; the CG loop uses the engine).
// Kept as written: diag counts 1 per fluid neighbour, and the -z tap reads wz[x,y,z+1] (:184).
__global__ void __launch_bounds__(256)
k_density_apply(DGrid g, const void* v, void* out, int dt, const void* wx, const void* wy, const void* wz, int wdt,
                const void* lphi, int ldt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || x >= g.Nx - 1 || y == 0 || y >= g.Ny - 1 || z == 0 || z >= g.Nz - 1) return;
  const double phi = ldx(lphi, ldt, i);
  if (phi >= 0) { stx(out, dt, i, 0.0); return; }
  double val = 0.0, diag = 0.0;
  auto tap = [&](int64_t nb, double w) {
    const double nphi = ldx(lphi, ldt, nb);
    if (nphi < 0) { val -= w * ldx(v, dt, nb); diag += 1; }
    else          { diag += 1 / fmin(1.0, fmax(0.01, phi / (phi - nphi))); }
  };
  const int64_t sx = (int64_t)g.Ny * g.Nz, sy = g.Nz;
  tap(i + sx, ldx(wx, wdt, g.fx(x + 1, y, z)));
  tap(i - sx, ldx(wx, wdt, g.fx(x, y, z)));
  tap(i + sy, ldx(wy, wdt, g.fy(x, y + 1, z)));
  tap(i - sy, ldx(wy, wdt, g.fy(x, y, z)));
  tap(i + 1, ldx(wz, wdt, g.fz(x, y, z + 1)));
  tap(i - 1, ldx(wz, wdt, g.fz(x, y, z + 1)));
  val += diag * ldx(v, dt, i);
  stx(out, dt, i, val);
}

// compute_displacement_kernel :209-222 -- x,y,z in [1, N-1]
__global__ void __launch_bounds__(256)
k_density_displacement(DGrid g, double dt, D3 cs, void* dx, void* dy, void* dz, int ddt, const void* pv, int pdt,
                       const void* lphi, int ldt) {
  const int64_t n = (int64_t)g.Nx * g.Ny * g.Nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.Nz), y = (int)((i / g.Nz) % g.Ny), x = (int)(i / ((int64_t)g.Nz * g.Ny));
  if (x == 0 || y == 0 || z == 0) return;
  const int64_t sx = (int64_t)g.Ny * g.Nz, sy = g.Nz;
  const double pc = ldx(lphi, ldt, i), p = ldx(pv, pdt, i);
  const double phix = fmin(1.0, fmax(0.01, d_edge_in_fraction(pc, ldx(lphi, ldt, i - sx))));
  const double phiy = fmin(1.0, fmax(0.01, d_edge_in_fraction(pc, ldx(lphi, ldt, i - sy))));
  const double phiz = fmin(1.0, fmax(0.01, d_edge_in_fraction(pc, ldx(lphi, ldt, i - 1))));
  stx(dx, ddt, g.fx(x, y, z), (p - ldx(pv, pdt, i - sx)) * dt * cs.v[0] / phix);
  stx(dy, ddt, g.fy(x, y, z), (p - ldx(pv, pdt, i - sy)) * dt * cs.v[1] / phiy);
  stx(dz, ddt, g.fz(x, y, z), (p - ldx(pv, pdt, i - 1)) * dt * cs.v[2] / phiz);
}

// apply_displacement_kernel :224-253 -- px[P, axis] += trilinear sample of the face array `d` (shape s0,s1,s2)
__global__ void __launch_bounds__(256)
k_density_advect(void* px, int pxdt, int64_t P, const void* d, int ddt, int s0, int s1, int s2, D3 bmin, D3 cs, D3 bias,
                 int axis) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  long long gi[3];
  double w[3];
  particle_cell(px, pxdt, p, bmin, cs, bias, gi, w);
  double pos = ldx(px, pxdt, 3 * p + axis);
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz) {
        const int cx = (int)max(0LL, min((long long)s0 - 1, gi[0] + ix));
        const int cy = (int)max(0LL, min((long long)s1 - 1, gi[1] + iy));
        const int cz = (int)max(0LL, min((long long)s2 - 1, gi[2] + iz));
        const double weight = corner_weight(ix, w[0]) * corner_weight(iy, w[1]) * corner_weight(iz, w[2]);
        const double add = weight * ldx(d, ddt, ((int64_t)cx * s1 + cy) * s2 + cz);
        // the reference accumulates into the array element itself: with an fp32 position array every
        // partial sum is rounded to fp32
        pos = pxdt == MFS_F32 ? (double)(float)(pos + add) : pos + add;
      }
  stx(px, pxdt, 3 * p + axis, pos);
}

static int check_g(const int64_t gres[3]) {
  MFS_REQUIRE(gres != nullptr, "gres is null");
  for (int a = 0; a < 3; ++a) MFS_REQUIRE(gres[a] >= 1 && gres[a] <= 4096, "grid resolution out of range [1,4096]");
  return MFS_OK;
}

}  // namespace mfs

using namespace mfs;

extern "C" {

int mfs_density_splat3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3], const void* px,
                        int px_dt, const void* pm, int pm_dt, double pvol, int64_t num_particles, void* gm, void* gvol,
                        int g_dt, mfs_stream stream) {
  if (int e = check_g(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && gm && gvol, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || (px && pm)), "particle arrays");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(pm_dt) && dtype_ok(g_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  DGrid g{(int)gres[0], (int)gres[1], (int)gres[2]};
  hipLaunchKernelGGL(k_density_splat, dim3(cdiv(num_particles, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     D3{{bound_min[0], bound_min[1], bound_min[2]}}, D3{{cell_size[0], cell_size[1], cell_size[2]}}, px,
                     px_dt, pm, pm_dt, pvol, num_particles, gm, gvol, g_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_density_splat3d_tiled(const int64_t gres[3], const double bound_min[3], const double cell_size[3], const void* px,
                              int px_dt, const void* pm, int pm_dt, double pvol, int64_t num_particles, const int32_t* perm,
                              const int32_t* tile_start, void* gm, void* gvol, int g_dt, mfs_stream stream) {
  if (int e = check_g(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && gm && gvol && perm && tile_start, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || (px && pm)), "particle arrays");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(pm_dt) && dtype_ok(g_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  DGrid g{(int)gres[0], (int)gres[1], (int)gres[2]};
  const int t0 = (int)((gres[0] + 7) / 8), t1 = (int)((gres[1] + 7) / 8), t2 = (int)((gres[2] + 7) / 8);
  hipLaunchKernelGGL(k_density_splat_tiled, dim3(t0 * t1 * t2), dim3(256), 0, (hipStream_t)stream, g,
                     D3{{bound_min[0], bound_min[1], bound_min[2]}}, D3{{cell_size[0], cell_size[1], cell_size[2]}}, t1, t2, px,
                     px_dt, pm, pm_dt, pvol, perm, tile_start, gm, gvol, g_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_density_fix_volume3d(const int64_t gres[3], const double cell_size[3], void* gvol, int g_dt, const void* sphi,
                             int sphi_dt, const void* lphi, int lphi_dt, const void* wx, const void* wy, const void* wz,
                             int w_dt, mfs_stream stream) {
  if (int e = check_g(gres)) return e;
  MFS_REQUIRE(cell_size && gvol && sphi && lphi && wx && wy && wz, "null argument");
  MFS_REQUIRE(dtype_ok(g_dt) && dtype_ok(sphi_dt) && dtype_ok(lphi_dt) && dtype_ok(w_dt), "dtype");
  DGrid g{(int)gres[0], (int)gres[1], (int)gres[2]};
  const double cvol = cell_size[0] * cell_size[1] * cell_size[2];                     // cp.prod(cell_size) :303
  const double dx = std::min(cell_size[0], std::min(cell_size[1], cell_size[2]));     // cp.min(cell_size) :304
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_density_fix_volume, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, cvol, dx, gvol, g_dt,
                     sphi, sphi_dt, lphi, lphi_dt, wx, wy, wz, w_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_density_rhs3d(const int64_t gres[3], double rho0, double dt, const double cell_size[3], const void* gm,
                      const void* gvol, int g_dt, const void* lphi, int lphi_dt, const void* wx, const void* wy,
                      const void* wz, int w_dt, void* b, int b_dt, mfs_stream stream) {
  if (int e = check_g(gres)) return e;
  MFS_REQUIRE(cell_size && gm && gvol && lphi && wx && wy && wz && b, "null argument");
  MFS_REQUIRE(dtype_ok(g_dt) && dtype_ok(lphi_dt) && dtype_ok(w_dt) && dtype_ok(b_dt), "dtype");
  DGrid g{(int)gres[0], (int)gres[1], (int)gres[2]};
  const double cvol = cell_size[0] * cell_size[1] * cell_size[2];
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_density_rhs, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, rho0, cvol, dt, gm, gvol,
                     g_dt, lphi, lphi_dt, wx, wy, wz, w_dt, b, b_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_density_apply3d(const int64_t gres[3], const void* v, void* out, int dt, const void* wx, const void* wy,
                        const void* wz, int w_dt, const void* lphi, int lphi_dt, mfs_stream stream) {
  if (int e = check_g(gres)) return e;
  MFS_REQUIRE(v && out && wx && wy && wz && lphi, "null argument");
  MFS_REQUIRE(v != out, "apply cannot run in place");
  MFS_REQUIRE(dtype_ok(dt) && dtype_ok(w_dt) && dtype_ok(lphi_dt), "dtype");
  DGrid g{(int)gres[0], (int)gres[1], (int)gres[2]};
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_density_apply, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, v, out, dt, wx, wy, wz,
                     w_dt, lphi, lphi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_density_displacement3d(const int64_t gres[3], double dt, const double cell_size[3], void* dx, void* dy, void* dz,
                               int d_dt, const void* pv, int pv_dt, const void* lphi, int lphi_dt, mfs_stream stream) {
  if (int e = check_g(gres)) return e;
  MFS_REQUIRE(cell_size && dx && dy && dz && pv && lphi, "null argument");
  MFS_REQUIRE(dtype_ok(d_dt) && dtype_ok(pv_dt) && dtype_ok(lphi_dt), "dtype");
  DGrid g{(int)gres[0], (int)gres[1], (int)gres[2]};
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_density_displacement, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, dt,
                     D3{{cell_size[0], cell_size[1], cell_size[2]}}, dx, dy, dz, d_dt, pv, pv_dt, lphi, lphi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_density_advect3d(void* px, int px_dt, int64_t num_particles, const void* d, int d_dt, const int64_t dshape[3],
                         const double bound_min[3], const double cell_size[3], const double grid_bias[3], int axis,
                         mfs_stream stream) {
  MFS_REQUIRE(d && dshape && bound_min && cell_size && grid_bias, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || px), "particle array");
  MFS_REQUIRE(axis >= 0 && axis < 3, "axis");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(d_dt), "dtype");
  for (int a = 0; a < 3; ++a) MFS_REQUIRE(dshape[a] >= 1 && dshape[a] <= 4097, "array shape");
  if (num_particles == 0) return MFS_OK;
  hipLaunchKernelGGL(k_density_advect, dim3(cdiv(num_particles, 256)), dim3(256), 0, (hipStream_t)stream, px, px_dt,
                     num_particles, d, d_dt, (int)dshape[0], (int)dshape[1], (int)dshape[2],
                     D3{{bound_min[0], bound_min[1], bound_min[2]}}, D3{{cell_size[0], cell_size[1], cell_size[2]}},
                     D3{{grid_bias[0], grid_bias[1], grid_bias[2]}}, axis);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

}  // extern "C"
