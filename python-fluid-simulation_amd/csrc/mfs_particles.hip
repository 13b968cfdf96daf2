// mfs_particles.hip -- the notebook's particle <-> grid transfers on gfx950.
//
// Reference: 3D_viscous_fluid_sim.ipynb code cells 2 (p2g), 3 (g2p), 4 (compute_fluid_levelset) and
// 6 (compute_fluid_volume).  One thread per particle; scatters use hardware fp atomics (the reference
// uses cuda.atomic.add / atomic.min, whose order is unspecified too).  The reference kernels keep
// float32 LOCALS (x, gx, disp, w) whatever the arrays' dtypes; the helpers below reproduce numba's
// typing of those mixed expressions for the notebook's containers -- bound_min and the grid biases
// float32, cell sizes float64 -- so that base indices and weights are the reference's, bit for bit.
#include <math.h>

#include "mfs_common.h"

// No FMA contraction in this file: base indices, float32-rounded grid positions and weights must round where
// the reference's separate multiply and add round (a contracted a*b+c flips a float32 rounding now and then).
#pragma clang fp contract(off)

namespace mfs {

struct PGrid {            // clamp extents (the `gres` argument of the kernel) and the target array's shape
  int N[3];
  int s1, s2;             // rows / row length of the target array
  __device__ __forceinline__ int64_t at(int x, int y, int z) const { return ((int64_t)x * s1 + y) * s2 + z; }
};
struct PGeom {
  float bmin[3];          // bound_min at float32 (the notebook's BOUND_MIN is a float32 array)
  double cs[3];           // cell_size (float64: float32 / int64 in cupy)
  double off[3];          // sample position offset: the float32 grid bias, or 0.5 / 0 for the level set / volume
  int has_bias;           // 1: `... / cell_size - grid_bias` (p2g, g2p); 0: no bias term in the index (cells 4, 6)
};

// x (float32), gi = floor(...), gx (float32) exactly as the kernels compute them
__device__ __forceinline__ void nb_cell(const void* px, int pdt, int64_t P, const PGeom& g, float x[3], long long gi[3],
                                        float gx[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    x[d] = (float)ldx(px, pdt, 3 * P + d);
    double t = (double)(x[d] - g.bmin[d]) / g.cs[d];               // float32 difference, float64 quotient
    if (g.has_bias) t -= g.off[d];
    gi[d] = (long long)floor(t);
    gx[d] = (float)(((double)gi[d] + g.off[d]) * g.cs[d] + (double)g.bmin[d]);
  }
}

__device__ __forceinline__ void atomic_add_t(void* p, int dt, int64_t i, double v) {
  if (dt == MFS_F32) atomicAdd((float*)p + i, (float)v); else atomicAdd((double*)p + i, v);
}

// min of a field cell and a candidate, atomically.  Two forms, the same bits (a minimum does not depend on the order of
// its candidates); a plain read first in both: once the field has settled most candidates lose without touching the atomic unit.
//  * native: ONE integer atomic -- IEEE values order like signed integers when they are >= 0 and like reversed unsigned
//    integers when they are < 0, so a non-negative candidate is a signed integer min and a negative one an unsigned integer
//    max on the same bits (either holds against a stored value of either sign: a negative value has the top bit set, i.e.
//    is below every non-negative one as a signed integer and above it as an unsigned one);
//  * cas: the compare-and-swap loop.
// Same-box A/B of k_fluid_levelset (twice per time step): 85 k particles 160.7 (cas) vs 109.9 us (native); 2.1 M particles level
// set + volume 2.96 vs 2.72 ms; 16.8 M particles 14.04 vs 17.35 ms -- the fire-and-forget atomics of the native form pile up in
// L2 at that size.  The kernel picks by particle count.
__device__ __forceinline__ void atomic_min_t(void* p, int dt, int64_t i, double v, bool cas) {
  if (dt == MFS_F32) {
    const float fv = (float)v;
    if (!cas) {
      if (((const float*)p)[i] <= fv) return;
      if (fv >= 0.f) atomicMin((int*)p + i, __float_as_int(fv));
      else atomicMax((unsigned*)p + i, __float_as_uint(fv));
      return;
    }
    int* a = (int*)p + i;
    int old = *a, assumed;
    do {
      assumed = old;
      if (__int_as_float(assumed) <= fv) break;
      old = atomicCAS(a, assumed, __float_as_int(fv));
    } while (assumed != old);
  } else {
    if (!cas) {
      if (((const double*)p)[i] <= v) return;
      if (v >= 0.0) atomicMin((long long*)p + i, __double_as_longlong(v));
      else atomicMax((unsigned long long*)p + i, (unsigned long long)__double_as_longlong(v));
      return;
    }
    unsigned long long* a = (unsigned long long*)p + i;
    unsigned long long old = *a, assumed;
    do {
      assumed = old;
      if (__longlong_as_double((long long)assumed) <= v) break;
      old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    } while (assumed != old);
  }
}

__device__ __forceinline__ int clampi(long long v, int n) { return (int)max(0LL, min((long long)n - 1, v)); }

// p2g_particle (code cell 2)
__global__ void __launch_bounds__(256)
k_p2g_scatter(PGrid g, PGeom geo, int axis, const void* px, int pxdt, const void* pm, int pmdt, const void* pv, int pvdt,
              const void* pca, int pcdt, int64_t P, void* gm, void* gv, int gdt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const double m = ldx(pm, pmdt, p);
  float x[3], gx[3], disp[3], w[3];
  long long gi[3];
  nb_cell(px, pxdt, p, geo, x, gi, gx);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    disp[d] = gx[d] - x[d];
    w[d] = (float)((double)fabsf(disp[d]) / geo.cs[d]);
  }
  const float va = (float)ldx(pv, pvdt, 3 * p + axis);
  const double c0 = ldx(pca, pcdt, 3 * p), c1 = ldx(pca, pcdt, 3 * p + 1), c2 = ldx(pca, pcdt, 3 * p + 2);
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz) {
        const int cx = clampi(gi[0] + ix, g.N[0]), cy = clampi(gi[1] + iy, g.N[1]), cz = clampi(gi[2] + iz, g.N[2]);
        const double wx = ix + (ix ? -1.0 : 1.0) * (1 - (double)w[0]);
        const double wy = iy + (iy ? -1.0 : 1.0) * (1 - (double)w[1]);
        const double wz = iz + (iz ? -1.0 : 1.0) * (1 - (double)w[2]);
        const double cv = ((double)disp[0] + ix * geo.cs[0]) * c0 + ((double)disp[1] + iy * geo.cs[1]) * c1 +
                          ((double)disp[2] + iz * geo.cs[2]) * c2;
        const double weight = wx * wy * wz;
        const int64_t c = g.at(cx, cy, cz);
        atomic_add_t(gm, gdt, c, weight * m);
        atomic_add_t(gv, gdt, c, weight * m * ((double)va + cv));
      }
}

// p2g_grid (code cell 2): gv /= gm where mass landed, in the arrays' own precision
__global__ void __launch_bounds__(256) k_p2g_normalize(int64_t n, const void* gm, void* gv, int gdt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (gdt == MFS_F32) {
    const float m = ((const float*)gm)[i];
    if (m > 0) ((float*)gv)[i] = ((float*)gv)[i] / m;
  } else {
    const double m = ((const double*)gm)[i];
    if (m > 0) ((double*)gv)[i] = ((double*)gv)[i] / m;
  }
}

// g2p_particle (code cell 3): pv[P, axis] and the affine row pca[P, :] are accumulated INTO the array
// elements in the reference, so with float32 particle arrays every partial sum is rounded to float32.
__global__ void __launch_bounds__(256)
k_g2p_gather(PGrid g, PGeom geo, int axis, const void* px, int pxdt, void* pv, int pvdt, void* pca, int pcdt, int64_t P,
             const void* gv, int gdt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float x[3], gx[3], w[3];
  long long gi[3];
  nb_cell(px, pxdt, p, geo, x, gi, gx);
#pragma unroll
  for (int d = 0; d < 3; ++d) w[d] = (float)((double)fabsf(gx[d] - x[d]) / geo.cs[d]);
  double vel = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
  auto acc = [](double s, double t, int dt) { return dt == MFS_F32 ? (double)(float)(s + t) : s + t; };
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz) {
        const int cx = clampi(gi[0] + ix, g.N[0]), cy = clampi(gi[1] + iy, g.N[1]), cz = clampi(gi[2] + iz, g.N[2]);
        const double wx = 1 - ix + (2 * ix - 1) * (double)w[0];
        const double wy = 1 - iy + (2 * iy - 1) * (double)w[1];
        const double wz = 1 - iz + (2 * iz - 1) * (double)w[2];
        const double gval = ldx(gv, gdt, g.at(cx, cy, cz));
        vel = acc(vel, wx * wy * wz * gval, pvdt);
        a0 = acc(a0, (2 * ix - 1) * wy * wz * gval / geo.cs[0], pcdt);
        a1 = acc(a1, wx * (2 * iy - 1) * wz * gval / geo.cs[1], pcdt);
        a2 = acc(a2, wx * wy * (2 * iz - 1) * gval / geo.cs[2], pcdt);
      }
  stx(pv, pvdt, 3 * p + axis, vel);
  stx(pca, pcdt, 3 * p, a0);
  stx(pca, pcdt, 3 * p + 1, a1);
  stx(pca, pcdt, 3 * p + 2, a2);
}

// compute_fls_kernel (code cell 4): phi = min(phi, |cell centre - x| - r) over the 5^3 cells around the particle
__global__ void __launch_bounds__(256)
k_fluid_levelset(PGrid g, PGeom geo, double r, const void* px, int pxdt, int64_t P, void* phi, int phidt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float x[3], gx[3];
  long long gi[3];
  nb_cell(px, pxdt, p, geo, x, gi, gx);
  const bool cas = P > ((int64_t)8 << 20);      // see atomic_min_t
  for (int dx = -2; dx <= 2; ++dx)
    for (int dy = -2; dy <= 2; ++dy)
      for (int dz = -2; dz <= 2; ++dz) {
        const int ii[3] = {clampi(gi[0] + dx, g.N[0]), clampi(gi[1] + dy, g.N[1]), clampi(gi[2] + dz, g.N[2])};
        double n = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float gip = (float)(((double)ii[d] + 0.5) * geo.cs[d] + (double)geo.bmin[d] - (double)x[d]);
          n += (double)(gip * gip);                              // float32 product, float64 sum (norm())
        }
        atomic_min_t(phi, phidt, g.at(ii[0], ii[1], ii[2]), sqrt(n) - r, cas);
      }
}

// compute_fluid_volume_kernel (code cell 6)
__global__ void __launch_bounds__(256)
k_fluid_volume_splat(PGrid g, PGeom geo, const void* px, int pxdt, double pvol, int64_t P, void* gvol, int gdt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float x[3], gx[3], w[3];
  long long gi[3];
  nb_cell(px, pxdt, p, geo, x, gi, gx);
#pragma unroll
  for (int d = 0; d < 3; ++d) w[d] = (float)((double)fabsf(gx[d] - x[d]) / geo.cs[d]);
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz) {
        const int cx = clampi(gi[0] + ix, g.N[0]), cy = clampi(gi[1] + iy, g.N[1]), cz = clampi(gi[2] + iz, g.N[2]);
        const double weight = (ix + (ix ? -1.0 : 1.0) * (1 - (double)w[0])) * (iy + (iy ? -1.0 : 1.0) * (1 - (double)w[1])) *
                              (iz + (iz ? -1.0 : 1.0) * (1 - (double)w[2]));
        atomic_add_t(gvol, gdt, g.at(cx, cy, cz), weight * pvol);
      }
}

// constrain_fluid_volume_kernel (code cell 6)
__global__ void __launch_bounds__(256) k_fluid_volume_constrain(int64_t n, void* gvol, int gdt, double cell_vol) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  stx(gvol, gdt, i, fmin(ldx(gvol, gdt, i), cell_vol));
}

// ------------------------------------------------------------------ tile-sorted scatters (round 3) -----------------
// At millions of particles the scatters above are bound by their global atomics (16 per particle and axis for p2g, up to
// 125 min candidates for the level set, 8 for the volume splat): 16.8 M particles, p2g 6.6 ms per axis, level set 4.0 ms.
// The particles are therefore bucketed once per position update by the TILE of 8^3 cells their cell lies in
// (counting sort on the tile id: `perm` lists particle indices tile by tile -- the arrays themselves stay where they are,
// so particle i stays particle i for every caller), and the scatters run one workgroup per tile with the tile's nodes
// (plus the one or two layers its particles can reach) in LDS: the atomics become LDS atomics, and each touched node
// goes to memory ONCE per tile instead of once per particle and corner.  A contribution that falls outside the staged
// nodes -- a particle that moved since the sort -- takes the global atomic as before: the result does not depend on the
// sort being current, only the speed does.  Same per-particle arithmetic (nb_cell, the weights); sums in another order,
// like every run of the atomics.
constexpr int kTB = 8;                    // tile edge, cells

struct PTiles { int t[3]; };              // tiles per axis
__device__ __forceinline__ int tile_of(const PTiles& pt, int cx, int cy, int cz) {
  return ((cx / kTB) * pt.t[1] + cy / kTB) * pt.t[2] + cz / kTB;
}

// A workgroup's 256 consecutive particles lie in a handful of tiles (particle order follows space, roughly): their tile
// ids go through a small LDS table first -- one LDS atomic per particle, then ONE global atomic per (workgroup, tile) --
// instead of 16.8 M global atomics on a few thousand hot counters (7.0 ms for the sort at that size).
constexpr int kTileTab = 128;                  // table entries (open addressing); a workgroup with more tiles overflows to memory
__device__ __forceinline__ int tile_slot(int* keys, int tile) {
  unsigned h = ((unsigned)tile * 2654435761u) >> 25;             // 7 bits
  for (int probe = 0; probe < kTileTab; ++probe, h = (h + 1) & (kTileTab - 1)) {
    const int seen = atomicCAS(keys + h, -1, tile);
    if (seen == -1 || seen == tile) return (int)h;
  }
  return -1;
}

// pass 1: tile id of every particle's cell (cell = floor((x - bound_min) / cell_size), clamped) and the tile histogram
__global__ void __launch_bounds__(256)
k_tile_count(PGrid g, PGeom geo, PTiles pt, const void* px, int pxdt, int64_t P, int* __restrict__ key, int* __restrict__ count) {
  __shared__ int keys[kTileTab], cnt[kTileTab];
  if (threadIdx.x < kTileTab) { keys[threadIdx.x] = -1; cnt[threadIdx.x] = 0; }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) {
    float x[3], gx[3];
    long long gi[3];
    nb_cell(px, pxdt, p, geo, x, gi, gx);
    const int t = tile_of(pt, clampi(gi[0], g.N[0]), clampi(gi[1], g.N[1]), clampi(gi[2], g.N[2]));
    key[p] = t;
    const int s = tile_slot(keys, t);
    if (s >= 0) atomicAdd(cnt + s, 1); else atomicAdd(count + t, 1);
  }
  __syncthreads();
  if (threadIdx.x < kTileTab && keys[threadIdx.x] >= 0) atomicAdd(count + keys[threadIdx.x], cnt[threadIdx.x]);
}

// pass 2: exclusive scan of the histogram (ONE block; n = tiles <= a few 100 k) -> start[0 .. n], cursor = start
__global__ void __launch_bounds__(1024)
k_tile_scan(const int* __restrict__ count, int n, int* __restrict__ start, int* __restrict__ cursor) {
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024, i0 = min(n, t * chunk), i1 = min(n, i0 + chunk);
  int sum = 0;
  for (int i = i0; i < i1; ++i) sum += count[i];
  __shared__ int s_pre[1024];
  s_pre[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? s_pre[t - o] : 0;
    __syncthreads();
    s_pre[t] += v;
    __syncthreads();
  }
  int run = s_pre[t] - sum;
  for (int i = i0; i < i1; ++i) { start[i] = run; cursor[i] = run; run += count[i]; }
  if (t == 1023) start[n] = s_pre[1023];
}

// pass 3: particle index into its tile's segment -- a rank inside the workgroup's share of the tile (LDS), the share's
// place in the segment by one global atomic per (workgroup, tile)
__global__ void __launch_bounds__(256)
k_tile_fill(const int* __restrict__ key, int64_t P, int* __restrict__ cursor, int* __restrict__ perm) {
  __shared__ int keys[kTileTab], cnt[kTileTab], base[kTileTab];
  if (threadIdx.x < kTileTab) { keys[threadIdx.x] = -1; cnt[threadIdx.x] = 0; }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int s = -2, rank = 0, t = 0;
  if (p < P) {
    t = key[p];
    s = tile_slot(keys, t);
    if (s >= 0) rank = atomicAdd(cnt + s, 1);
  }
  __syncthreads();
  if (threadIdx.x < kTileTab && keys[threadIdx.x] >= 0) base[threadIdx.x] = atomicAdd(cursor + keys[threadIdx.x], cnt[threadIdx.x]);
  __syncthreads();
  if (s >= 0) perm[base[s] + rank] = (int)p;
  else if (s == -1) perm[atomicAdd(cursor + t, 1)] = (int)p;          // table overflow: straight to memory
}

// local index of global node (cx, cy, cz) in a staged box with origin o and edge E; -1 if outside
template <int E>
__device__ __forceinline__ int local_of(int cx, int cy, int cz, const int o[3]) {
  const int lx = cx - o[0], ly = cy - o[1], lz = cz - o[2];
  return ((unsigned)lx < (unsigned)E && (unsigned)ly < (unsigned)E && (unsigned)lz < (unsigned)E) ? (lx * E + ly) * E + lz : -1;
}

// p2g_particle, one workgroup per tile: nodes [c0 - 1, c0 + kTB] per axis staged (base index = cell - 1 or cell)
__global__ void __launch_bounds__(256)
k_p2g_scatter_tiled(PGrid g, PGeom geo, PTiles pt, int axis, const void* px, int pxdt, const void* pm, int pmdt, const void* pv,
                    int pvdt, const void* pca, int pcdt, const int* __restrict__ perm, const int* __restrict__ tstart, void* gm,
                    void* gv, int gdt) {
  constexpr int E = kTB + 2;
  __shared__ double lm[E * E * E], lmv[E * E * E];
  const int tile = blockIdx.x;
  const int a = tstart[tile], b = tstart[tile + 1];
  if (a == b) return;
  const int tz = tile % pt.t[2], ty = (tile / pt.t[2]) % pt.t[1], tx = tile / (pt.t[2] * pt.t[1]);
  const int o[3] = {tx * kTB - 1, ty * kTB - 1, tz * kTB - 1};
  for (int l = threadIdx.x; l < E * E * E; l += 256) { lm[l] = 0.0; lmv[l] = 0.0; }
  __syncthreads();
  for (int i = a + threadIdx.x; i < b; i += 256) {
    const int64_t p = perm[i];
    const double m = ldx(pm, pmdt, p);
    float x[3], gx[3], disp[3], w[3];
    long long gi[3];
    nb_cell(px, pxdt, p, geo, x, gi, gx);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      disp[d] = gx[d] - x[d];
      w[d] = (float)((double)fabsf(disp[d]) / geo.cs[d]);
    }
    const float va = (float)ldx(pv, pvdt, 3 * p + axis);
    const double c0 = ldx(pca, pcdt, 3 * p), c1 = ldx(pca, pcdt, 3 * p + 1), c2 = ldx(pca, pcdt, 3 * p + 2);
    for (int ix = 0; ix < 2; ++ix)
      for (int iy = 0; iy < 2; ++iy)
        for (int iz = 0; iz < 2; ++iz) {
          const int cx = clampi(gi[0] + ix, g.N[0]), cy = clampi(gi[1] + iy, g.N[1]), cz = clampi(gi[2] + iz, g.N[2]);
          const double wx = ix + (ix ? -1.0 : 1.0) * (1 - (double)w[0]);
          const double wy = iy + (iy ? -1.0 : 1.0) * (1 - (double)w[1]);
          const double wz = iz + (iz ? -1.0 : 1.0) * (1 - (double)w[2]);
          const double cv = ((double)disp[0] + ix * geo.cs[0]) * c0 + ((double)disp[1] + iy * geo.cs[1]) * c1 +
                            ((double)disp[2] + iz * geo.cs[2]) * c2;
          const double weight = wx * wy * wz;
          const int l = local_of<E>(cx, cy, cz, o);
          if (l >= 0) {
            atomicAdd(&lm[l], weight * m);
            atomicAdd(&lmv[l], weight * m * ((double)va + cv));
          } else {                      // moved out of the tile's reach since the sort: straight to memory
            const int64_t c = g.at(cx, cy, cz);
            atomic_add_t(gm, gdt, c, weight * m);
            atomic_add_t(gv, gdt, c, weight * m * ((double)va + cv));
          }
        }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < E * E * E; l += 256) {
    const double vm = lm[l], vmv = lmv[l];
    if (vm == 0.0 && vmv == 0.0) continue;
    const int lz = l % E, ly = (l / E) % E, lx = l / (E * E);
    const int64_t c = g.at(o[0] + lx, o[1] + ly, o[2] + lz);       // a touched node is inside the array (indices were clamped)
    atomic_add_t(gm, gdt, c, vm);
    atomic_add_t(gv, gdt, c, vmv);
  }
}

// compute_fls_kernel, one workgroup per tile: cells [c0 - 2, c0 + kTB + 1] per axis staged
__global__ void __launch_bounds__(256)
k_fluid_levelset_tiled(PGrid g, PGeom geo, PTiles pt, double r, const void* px, int pxdt, const int* __restrict__ perm,
                       const int* __restrict__ tstart, void* phi, int phidt) {
  constexpr int E = kTB + 4;
  __shared__ double lp[E * E * E];
  const int tile = blockIdx.x;
  const int a = tstart[tile], b = tstart[tile + 1];
  if (a == b) return;
  const int tz = tile % pt.t[2], ty = (tile / pt.t[2]) % pt.t[1], tx = tile / (pt.t[2] * pt.t[1]);
  const int o[3] = {tx * kTB - 2, ty * kTB - 2, tz * kTB - 2};
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  for (int l = threadIdx.x; l < E * E * E; l += 256) lp[l] = kInf;
  __syncthreads();
  for (int i = a + threadIdx.x; i < b; i += 256) {
    const int64_t p = perm[i];
    float x[3], gx[3];
    long long gi[3];
    nb_cell(px, pxdt, p, geo, x, gi, gx);
    // per axis, once: the five clamped cell indices, their offsets in the staged box (-1: outside) and the squared
    // distance terms -- float32 product widened, exactly the reference's `gip * gip` -- so that the 125-candidate loop is
    // three adds, an LDS read and a compare (the index arithmetic used to be most of it)
    int ci[3][5], lo[3][5];
    double sq[3][5];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        ci[d][k] = clampi(gi[d] + k - 2, g.N[d]);
        const int l1 = ci[d][k] - o[d];
        lo[d][k] = (unsigned)l1 < (unsigned)E ? l1 * (d == 0 ? E * E : (d == 1 ? E : 1)) : -1;
        const float gip = (float)(((double)ci[d][k] + 0.5) * geo.cs[d] + (double)geo.bmin[d] - (double)x[d]);
        sq[d][k] = (double)(gip * gip);
      }
#pragma unroll
    for (int a0 = 0; a0 < 5; ++a0)
#pragma unroll
      for (int a1 = 0; a1 < 5; ++a1) {
        const double nxy = sq[0][a0] + sq[1][a1];                  // ((0 + x^2) + y^2) + z^2: the reference's order
        const int lxy = (lo[0][a0] | lo[1][a1]) < 0 ? -1 : lo[0][a0] + lo[1][a1];
#pragma unroll
        for (int a2 = 0; a2 < 5; ++a2) {
          const double n = nxy + sq[2][a2];
          const int l = (lxy | lo[2][a2]) < 0 ? -1 : lxy + lo[2][a2];
          if (l >= 0) {
            // the staged table holds the minimum of the SQUARED distance: n -> sqrt(n) - r is monotone (a correctly rounded
            // square root and a rounded subtraction of a constant both are), so sqrt(min n) - r, formed once per node below,
            // IS min(sqrt(n) - r) bit for bit -- and the 125 candidates of a particle cost an add, an LDS read and a
            // compare each instead of an fp64 square root.  n >= +0: its bit pattern orders like the value.
            if (n < lp[l]) atomicMin(reinterpret_cast<unsigned long long*>(&lp[l]), (unsigned long long)__double_as_longlong(n));
          } else {
            atomic_min_t(phi, phidt, g.at(ci[0][a0], ci[1][a1], ci[2][a2]), sqrt(n) - r, false);
          }
        }
      }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < E * E * E; l += 256) {
    const double v = lp[l];
    if (v == kInf) continue;
    const int lz = l % E, ly = (l / E) % E, lx = l / (E * E);
    atomic_min_t(phi, phidt, g.at(o[0] + lx, o[1] + ly, o[2] + lz), sqrt(v) - r, false);
  }
}

// compute_fluid_volume_kernel on the doubled grid, one workgroup per tile of CELLS: nodes [2 c0, 2 c0 + 2 kTB] staged
__global__ void __launch_bounds__(256)
k_fluid_volume_splat_tiled(PGrid g, PGeom geo, PTiles pt, const void* px, int pxdt, double pvol, const int* __restrict__ perm,
                           const int* __restrict__ tstart, void* gvol, int gdt) {
  constexpr int E = 2 * kTB + 1;
  __shared__ double lv[E * E * E];
  const int tile = blockIdx.x;
  const int a = tstart[tile], b = tstart[tile + 1];
  if (a == b) return;
  const int tz = tile % pt.t[2], ty = (tile / pt.t[2]) % pt.t[1], tx = tile / (pt.t[2] * pt.t[1]);
  const int o[3] = {2 * tx * kTB, 2 * ty * kTB, 2 * tz * kTB};
  for (int l = threadIdx.x; l < E * E * E; l += 256) lv[l] = 0.0;
  __syncthreads();
  for (int i = a + threadIdx.x; i < b; i += 256) {
    const int64_t p = perm[i];
    float x[3], gx[3], w[3];
    long long gi[3];
    nb_cell(px, pxdt, p, geo, x, gi, gx);
#pragma unroll
    for (int d = 0; d < 3; ++d) w[d] = (float)((double)fabsf(gx[d] - x[d]) / geo.cs[d]);
    for (int ix = 0; ix < 2; ++ix)
      for (int iy = 0; iy < 2; ++iy)
        for (int iz = 0; iz < 2; ++iz) {
          const int cx = clampi(gi[0] + ix, g.N[0]), cy = clampi(gi[1] + iy, g.N[1]), cz = clampi(gi[2] + iz, g.N[2]);
          const double weight = (ix + (ix ? -1.0 : 1.0) * (1 - (double)w[0])) * (iy + (iy ? -1.0 : 1.0) * (1 - (double)w[1])) *
                                (iz + (iz ? -1.0 : 1.0) * (1 - (double)w[2]));
          const int l = local_of<E>(cx, cy, cz, o);
          if (l >= 0) atomicAdd(&lv[l], weight * pvol);
          else atomic_add_t(gvol, gdt, g.at(cx, cy, cz), weight * pvol);
        }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < E * E * E; l += 256) {
    const double v = lv[l];
    if (v == 0.0) continue;
    const int lz = l % E, ly = (l / E) % E, lx = l / (E * E);
    atomic_add_t(gvol, gdt, g.at(o[0] + lx, o[1] + ly, o[2] + lz), v);
  }
}

static PTiles make_tiles(const int64_t gres[3]) {
  PTiles t;
  for (int a = 0; a < 3; ++a) t.t[a] = (int)((gres[a] + kTB - 1) / kTB);
  return t;
}

static int check_shape(const int64_t s[3]) {
  MFS_REQUIRE(s != nullptr, "shape is null");
  for (int a = 0; a < 3; ++a) MFS_REQUIRE(s[a] >= 1 && s[a] <= 8193, "array extent out of range [1,8193]");
  return MFS_OK;
}

static PGeom make_geom(const double bmin[3], const double cs[3], const double off[3], int has_bias) {
  PGeom g;
  for (int d = 0; d < 3; ++d) { g.bmin[d] = (float)bmin[d]; g.cs[d] = cs[d]; g.off[d] = (double)(float)off[d]; }
  g.has_bias = has_bias;
  return g;
}

}  // namespace mfs

using namespace mfs;

extern "C" {

int mfs_p2g_scatter3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                      const double grid_bias[3], int axis, const void* px, int px_dt, const void* pm, int pm_dt,
                      const void* pv, int pv_dt, const void* pca, int pca_dt, int64_t num_particles, void* gm, void* gv,
                      int g_dt, mfs_stream stream) {
  if (int e = check_shape(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && grid_bias && gm && gv, "null argument");
  MFS_REQUIRE(axis >= 0 && axis < 3, "axis");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || (px && pm && pv && pca)), "particle arrays");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(pm_dt) && dtype_ok(pv_dt) && dtype_ok(pca_dt) && dtype_ok(g_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  PGrid g{{(int)gres[0], (int)gres[1], (int)gres[2]}, (int)gres[1] + (axis == 1), (int)gres[2] + (axis == 2)};
  hipLaunchKernelGGL(k_p2g_scatter, dim3(cdiv(num_particles, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     make_geom(bound_min, cell_size, grid_bias, 1), axis, px, px_dt, pm, pm_dt, pv, pv_dt, pca, pca_dt,
                     num_particles, gm, gv, g_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_p2g_normalize3d(int64_t count, const void* gm, void* gv, int g_dt, mfs_stream stream) {
  MFS_REQUIRE(count >= 0 && gm && gv, "arguments");
  MFS_REQUIRE(dtype_ok(g_dt), "dtype");
  if (count == 0) return MFS_OK;
  hipLaunchKernelGGL(k_p2g_normalize, dim3(cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, count, gm, gv, g_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_g2p_gather3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                     const double grid_bias[3], int axis, const void* px, int px_dt, void* pv, int pv_dt, void* pca,
                     int pca_dt, int64_t num_particles, const void* gv, int g_dt, mfs_stream stream) {
  if (int e = check_shape(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && grid_bias && gv, "null argument");
  MFS_REQUIRE(axis >= 0 && axis < 3, "axis");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || (px && pv && pca)), "particle arrays");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(pv_dt) && dtype_ok(pca_dt) && dtype_ok(g_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  PGrid g{{(int)gres[0], (int)gres[1], (int)gres[2]}, (int)gres[1] + (axis == 1), (int)gres[2] + (axis == 2)};
  hipLaunchKernelGGL(k_g2p_gather, dim3(cdiv(num_particles, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     make_geom(bound_min, cell_size, grid_bias, 1), axis, px, px_dt, pv, pv_dt, pca, pca_dt,
                     num_particles, gv, g_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_fluid_levelset3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3], double radius,
                         const void* px, int px_dt, int64_t num_particles, void* phi, int phi_dt, mfs_stream stream) {
  if (int e = check_shape(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && phi, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || px), "particle array");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(phi_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  PGrid g{{(int)gres[0], (int)gres[1], (int)gres[2]}, (int)gres[1], (int)gres[2]};
  const double half[3] = {0.5, 0.5, 0.5};
  hipLaunchKernelGGL(k_fluid_levelset, dim3(cdiv(num_particles, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     make_geom(bound_min, cell_size, half, 0), radius, px, px_dt, num_particles, phi, phi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_fluid_volume3d(const int64_t vres[3], const double bound_min[3], const double cell_size[3], const void* px,
                       int px_dt, double pvol, int64_t num_particles, void* gvol, int g_dt, mfs_stream stream) {
  if (int e = check_shape(vres)) return e;
  MFS_REQUIRE(bound_min && cell_size && gvol, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || px), "particle array");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(g_dt), "dtype");
  PGrid g{{(int)vres[0], (int)vres[1], (int)vres[2]}, (int)vres[1], (int)vres[2]};
  const double zero[3] = {0.0, 0.0, 0.0};
  if (num_particles > 0)
    hipLaunchKernelGGL(k_fluid_volume_splat, dim3(cdiv(num_particles, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       make_geom(bound_min, cell_size, zero, 0), px, px_dt, pvol, num_particles, gvol, g_dt);
  const int64_t n = vres[0] * vres[1] * vres[2];
  const double cell_vol = cell_size[0] * cell_size[1] * cell_size[2];          // cp.prod(fv.cell_size)
  hipLaunchKernelGGL(k_fluid_volume_constrain, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, gvol, g_dt,
                     cell_vol);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int64_t mfs_particle_tiles3d(const int64_t gres[3]) {
  if (!gres) return 0;
  const PTiles t = make_tiles(gres);
  return (int64_t)t.t[0] * t.t[1] * t.t[2];
}

int mfs_particle_tile_sort3d(const int64_t gres[3], const double bound_min[3], const double cell_size[3], const void* px,
                             int px_dt, int64_t num_particles, int32_t* perm, int32_t* tile_start, int32_t* work,
                             mfs_stream stream) {
  if (int e = check_shape(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && perm && tile_start && work, "null argument");
  MFS_REQUIRE(num_particles >= 0 && num_particles < 0x7fffffff && (num_particles == 0 || px), "particle array");
  MFS_REQUIRE(dtype_ok(px_dt), "dtype");
  const PTiles pt = make_tiles(gres);
  const int64_t nt = (int64_t)pt.t[0] * pt.t[1] * pt.t[2];
  MFS_REQUIRE(nt < 0x7fffffff, "too many tiles");
  hipStream_t st = (hipStream_t)stream;
  int* count = work;                       // [nt]
  int* cursor = work + nt;                 // [nt]
  int* key = work + 2 * nt;                // [P]
  MFS_HIP_TRY(hipMemsetAsync(count, 0, (size_t)nt * sizeof(int), st));
  PGrid g{{(int)gres[0], (int)gres[1], (int)gres[2]}, (int)gres[1], (int)gres[2]};
  const double zero[3] = {0.0, 0.0, 0.0};
  if (num_particles > 0)
    hipLaunchKernelGGL(k_tile_count, dim3(cdiv(num_particles, 256)), dim3(256), 0, st, g, make_geom(bound_min, cell_size, zero, 0),
                       pt, px, px_dt, num_particles, key, count);
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, count, (int)nt, tile_start, cursor);
  if (num_particles > 0)
    hipLaunchKernelGGL(k_tile_fill, dim3(cdiv(num_particles, 256)), dim3(256), 0, st, key, num_particles, cursor, perm);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_p2g_scatter3d_tiled(const int64_t gres[3], const double bound_min[3], const double cell_size[3],
                            const double grid_bias[3], int axis, const void* px, int px_dt, const void* pm, int pm_dt,
                            const void* pv, int pv_dt, const void* pca, int pca_dt, int64_t num_particles,
                            const int32_t* perm, const int32_t* tile_start, void* gm, void* gv, int g_dt, mfs_stream stream) {
  if (int e = check_shape(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && grid_bias && gm && gv && perm && tile_start, "null argument");
  MFS_REQUIRE(axis >= 0 && axis < 3, "axis");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || (px && pm && pv && pca)), "particle arrays");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(pm_dt) && dtype_ok(pv_dt) && dtype_ok(pca_dt) && dtype_ok(g_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  PGrid g{{(int)gres[0], (int)gres[1], (int)gres[2]}, (int)gres[1] + (axis == 1), (int)gres[2] + (axis == 2)};
  const PTiles pt = make_tiles(gres);
  hipLaunchKernelGGL(k_p2g_scatter_tiled, dim3(pt.t[0] * pt.t[1] * pt.t[2]), dim3(256), 0, (hipStream_t)stream, g,
                     make_geom(bound_min, cell_size, grid_bias, 1), pt, axis, px, px_dt, pm, pm_dt, pv, pv_dt, pca, pca_dt, perm,
                     tile_start, gm, gv, g_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_fluid_levelset3d_tiled(const int64_t gres[3], const double bound_min[3], const double cell_size[3], double radius,
                               const void* px, int px_dt, int64_t num_particles, const int32_t* perm,
                               const int32_t* tile_start, void* phi, int phi_dt, mfs_stream stream) {
  if (int e = check_shape(gres)) return e;
  MFS_REQUIRE(bound_min && cell_size && phi && perm && tile_start, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || px), "particle array");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(phi_dt), "dtype");
  if (num_particles == 0) return MFS_OK;
  PGrid g{{(int)gres[0], (int)gres[1], (int)gres[2]}, (int)gres[1], (int)gres[2]};
  const double half[3] = {0.5, 0.5, 0.5};
  const PTiles pt = make_tiles(gres);
  hipLaunchKernelGGL(k_fluid_levelset_tiled, dim3(pt.t[0] * pt.t[1] * pt.t[2]), dim3(256), 0, (hipStream_t)stream, g,
                     make_geom(bound_min, cell_size, half, 0), pt, radius, px, px_dt, perm, tile_start, phi, phi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_fluid_volume3d_tiled(const int64_t vres[3], const double bound_min[3], const double cell_size[3], const void* px,
                             int px_dt, double pvol, int64_t num_particles, const int32_t* perm, const int32_t* tile_start,
                             void* gvol, int g_dt, mfs_stream stream) {
  if (int e = check_shape(vres)) return e;
  MFS_REQUIRE(bound_min && cell_size && gvol && perm && tile_start, "null argument");
  MFS_REQUIRE(num_particles >= 0 && (num_particles == 0 || px), "particle array");
  MFS_REQUIRE(dtype_ok(px_dt) && dtype_ok(g_dt), "dtype");
  MFS_REQUIRE(vres[0] % 2 == 1 && vres[1] % 2 == 1 && vres[2] % 2 == 1, "vres must be the doubled grid 2 * gres + 1");
  PGrid g{{(int)vres[0], (int)vres[1], (int)vres[2]}, (int)vres[1], (int)vres[2]};
  const double zero[3] = {0.0, 0.0, 0.0};
  const int64_t gres[3] = {(vres[0] - 1) / 2, (vres[1] - 1) / 2, (vres[2] - 1) / 2};       // the tiles are tiles of CELLS
  const PTiles pt = make_tiles(gres);
  if (num_particles > 0)
    hipLaunchKernelGGL(k_fluid_volume_splat_tiled, dim3(pt.t[0] * pt.t[1] * pt.t[2]), dim3(256), 0, (hipStream_t)stream, g,
                       make_geom(bound_min, cell_size, zero, 0), pt, px, px_dt, pvol, perm, tile_start, gvol, g_dt);
  const int64_t n = vres[0] * vres[1] * vres[2];
  const double cell_vol = cell_size[0] * cell_size[1] * cell_size[2];
  hipLaunchKernelGGL(k_fluid_volume_constrain, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, gvol, g_dt, cell_vol);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

}  // extern "C"
