// mfs_vcg_resident.h -- the whole viscosity CG loop of a SMALL grid in one launch per batch (gfx950), round 3.
//
// The reference's own scene (3D_viscous_fluid_sim.ipynb: 48 x 80 x 48 cells, 553 k unknowns) spends most of its time step
// in the loop solver/ViscosityCGSolver3D.py:588-610, and at that size an iteration is nothing but synchronisation: the
// launch-per-phase loop pays two kernel boundaries and a reduction tail per iteration (20.8 us measured).  This kernel is
// the viscosity counterpart of mfs_pcg_resident.h and keeps that design:
//
//   * W (<= 128) workgroups of 512 threads stay resident for up to `n_iter` iterations.  Workgroup (px, py) owns the box
//     [xa, xb) x [ya, yb) of cell columns, x in [1, Nx-1], y in [1, Ny-1] (the two extra column rows carry the u faces of
//     plane Nx-1 and the v faces of row Ny-1: the "slabs" of the launch-per-phase kernels), all of z.  An item = one cell
//     (x, y, z) with its THREE faces u, v, w; a thread holds x, r, d, q of KC cells' faces in registers for the whole launch.
//   * the box's d -- three component images with one halo layer in x and y, corners included (the operator couples
//     u[x, y] with v[x-1, y+1]) -- lives in LDS; the rows read their 27 velocity samples there.  The liquid-volume classes
//     are read from memory (L2-resident at these sizes) by the same sampler as the one-cell-per-lane kernel.
//   * rows: vcg_row_n<AXIS, false, 1> -- the SAME code as every other form of the operator, so q is bit-identical.
//   * dot products: workgroup sum -> a self-validating 16-byte record per workgroup -> every workgroup polls all W records
//     and adds them in one fixed order (mfs_pcg_resident.h's scheme, two polls per lane for W > 64): no counter, no fence.
//   * halos: after the r update every workgroup publishes r of its box-face cells as tagged granules (by global DOF index);
//     while the r.r records travel the neighbours' arrive, and each workgroup forms its halo's d_new itself
//     (d = r + beta d, k_update_d's arithmetic) -- one exchange per iteration.
//   * bookkeeping (test :605, history, alpha, beta) as k_update_rdx does it, by workgroup 0.
// Every wait is bounded.  A launch whose workgroups are not all resident times out at its FIRST dot product, before
// anything has been written: kErrNotResident, and mfs_vcg3d_poll switches the engine to the launch-per-phase loop.
//
// Arithmetic per element is that of the launch-per-phase loop (rows, r -= alpha q, x += alpha d, d = r + beta d: the same
// expressions); only the grouping of the two dot products differs, so histories agree to rounding, not bit for bit
// (tests/test_viscosity_resident_gpu.py states the tolerances).  State enters and leaves through the bound arrays and the
// scalar block exactly as after mfs_vcg3d_iterate's other forms (d updated at the end of every non-converged iteration).
#pragma once

namespace mfs {

constexpr int kVResBlock = 512;
constexpr int kVResMaxW = 256;
constexpr int kVResRing = 4;
constexpr int64_t kVResMaxDofs = 1 << 20;            // grids beyond this never qualify (mirror sizing)
constexpr size_t kVResLdsMax = 150 * 1024;
constexpr int kVResRecStride = 64;                    // u64 words between two workgroups' records (512 bytes: one per line pair, as in the pressure loop)
constexpr int kVResHB = 4;                            // halo requests a thread keeps in flight
constexpr int kVResHK = 8;                            // halo elements per thread, at most (plans with more do not qualify)

// measurement build only (tools/build_variant.sh NAME -DMFS_VRES_STAMP): workgroup 0 accumulates the wall-clock span of
// every phase of the loop and prints the averages when the launch ends
#ifdef MFS_VRES_STAMP
#define VRES_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const u64 t_ = wall_clock64(); st_acc[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define VRES_STAMP(i) do { } while (0)
#endif

struct VResArgs {
  void *x, *r, *q, *d;                                // flat [u | v | w] vectors
  Compact c;
  double k1, k2;
  int64_t off[3];
  int Px, Py, bxm, bym;
  double* scal;
  double* hist;
  int64_t hist_cap;
  int n_iter;
  u64* ar;                                            // [kVResRing][kVResMaxW] records
  u64* mirror;                                        // [2][n * Gran<T>::N] granules: r of the box faces, by flat DOF index
  unsigned tag0;                                      // first episode tag of this launch (2 per iteration)
  u64 timeout_ticks, first_timeout_ticks;
  int test_drop_wg;
};

typedef unsigned long long vres_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vres_u64x2 vres_load2(const u64* p) {
  vres_u64x2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void vres_store2(u64* p, u64 a, u64 b) {
  vres_u64x2 v = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u64 vres_load1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vres_store1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// dot product, first half: workgroup sum (fixed order) -> this workgroup's record
__device__ __forceinline__ void vres_allreduce_begin(double v, u64* ar, unsigned tag) {
  __shared__ double s_w[kVResBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  v = wave_sum(v);
  if (lane == 0) s_w[wave] = v;
  MFS_VISC_LDS_BARRIER();
  if (wave == 0 && lane == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kVResBlock / kWave; ++w) t += s_w[w];
    u64* tab = ar + ((size_t)(tag % kVResRing) * kVResMaxW + blockIdx.x) * kVResRecStride;
    const u64 bits = (u64)__double_as_longlong(t);
    vres_store2(tab, ((u64)tag << 32) | (bits & 0xffffffffull), ((u64)tag << 32) | (bits >> 32));
  }
}

// second half: the first wave polls the W records (up to two per lane) and adds them in one fixed order; every
// workgroup ends with the bit-identical total.  *ok false on timeout.
__device__ __forceinline__ double vres_allreduce_end(u64* ar, int W, unsigned tag, u64 timeout_ticks, bool* ok) {
  __shared__ double s_tot;
  __shared__ int s_ok;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (wave == 0) {
    constexpr int M = kVResMaxW / kWave;
    static_assert(M == 4, "the batched poll below names four request registers");
    double c[M];
    bool good = true;
    // all of this lane's records are requested at once, ONE wait per round (four sequential polls were four L2 round
    // trips: most of the reduction's latency); rounds repeat, for the records still missing, until every lane has its four
    vres_u64x2 w[M];
    bool got[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { w[m][0] = 0; w[m][1] = 0; got[m] = m * kWave + lane >= W; c[m] = 0.0; }
    u64 t0 = 0;
    for (int round = 0;; ++round) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (!got[m]) {
          const u64* g = ar + ((size_t)(tag % kVResRing) * kVResMaxW + m * kWave + lane) * kVResRecStride;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[m]) : "v"(g) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])::"memory");
      bool all = true;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (!got[m] && (w[m][0] >> 32) == tag && (w[m][1] >> 32) == tag) {
          got[m] = true;
          c[m] = __longlong_as_double((long long)((w[m][1] << 32) | (w[m][0] & 0xffffffffull)));
        }
        all = all && got[m];
      }
      if (__all(all)) break;
      if (round == 64) t0 = wall_clock64();              // the first rounds read no clock (itself a memory-latency operation)
      if (round > 64) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_ballot_w64(wall_clock64() - t0 > timeout_ticks) != 0) { good = all; break; }      // (wave-uniform exit)
      }
    }
    good = __all(good);
    double tot = 0.0;
#pragma unroll
    for (int m = 0; m < kVResMaxW / kWave; ++m) tot += wave_sum(c[m]);     // records in index order, one fixed tree
    if (lane == 0) { s_tot = tot; s_ok = good ? 1 : 0; }
  }
  MFS_VISC_LDS_BARRIER();
  *ok = s_ok != 0;
  return s_tot;
}

// one value as tagged granule(s), and back
template <typename T>
__device__ __forceinline__ void vres_publish(u64* buf, int64_t elem, T v, unsigned tag) {
  const u64 t = (u64)tag << 32;
  if (Gran<T>::N == 1) {
    vres_store1(buf + elem, t | (u64)__float_as_uint((float)v));
  } else {
    const u64 bits = (u64)__double_as_longlong((double)v);
    vres_store2(buf + 2 * elem, t | (bits & 0xffffffffull), t | (bits >> 32));
  }
}
template <typename T>
__device__ __forceinline__ bool vres_try(const u64* buf, int64_t elem, unsigned tag, T* out) {
  if (Gran<T>::N == 1) {
    const u64 w = vres_load1(buf + elem);
    if ((unsigned)(w >> 32) != tag) return false;
    *out = (T)__uint_as_float((unsigned)w);
    return true;
  }
  const vres_u64x2 w = vres_load2(buf + 2 * elem);
  if ((unsigned)(w[0] >> 32) != tag || (unsigned)(w[1] >> 32) != tag) return false;
  *out = (T)__longlong_as_double((long long)((w[1] << 32) | (w[0] & 0xffffffffull)));
  return true;
}
template <typename T>
__device__ __forceinline__ bool vres_fetch(const u64* buf, int64_t elem, unsigned tag, u64 timeout_ticks, T* out) {
  if (vres_try<T>(buf, elem, tag, out)) return true;
  const u64 t0 = wall_clock64();
  for (;;) {
    __builtin_amdgcn_s_sleep(1);
    if (vres_try<T>(buf, elem, tag, out)) return true;
    if (wall_clock64() - t0 > timeout_ticks) return false;
  }
}

// geometry shared by the kernel's index helpers
struct VResGeom {
  int Nx, Ny, Nz, PZ;          // PZ = Nz + 1: image row pitch (the w component has Nz + 1 faces per column)
  int xa, ya, bx, by;          // this workgroup's box of columns (extents 0 if empty)
  int pitch_y, lsx, isz;       // image: (lx + 1) * lsx + (ly + 1) * PZ + z, one component = isz elements
  __device__ __forceinline__ int s1(int c) const { return Ny + (c == 1 ? 1 : 0); }
  __device__ __forceinline__ int s2(int c) const { return Nz + (c == 2 ? 1 : 0); }
  __device__ __forceinline__ bool exists(int c, int gx, int gy, int z) const {
    return gx >= 0 && gx < Nx + (c == 0 ? 1 : 0) && gy >= 0 && gy < s1(c) && z >= 0 && z < s2(c);
  }
  // interior face of component c: computed by the operator, updated by the loop (everything else stays what it was)
  __device__ __forceinline__ bool active(int c, int gx, int gy, int z) const {
    return gx >= 1 && gx <= Nx - (c == 0 ? 1 : 2) && gy >= 1 && gy <= Ny - (c == 1 ? 1 : 2) && z >= 1 && z <= Nz - (c == 2 ? 1 : 2);
  }
  __device__ __forceinline__ int64_t fidx(int c, int gx, int gy, int z) const { return ((int64_t)gx * s1(c) + gy) * s2(c) + z; }
  __device__ __forceinline__ int lofs(int lx, int ly, int z) const { return (lx + 1) * lsx + (ly + 1) * PZ + z; }
};

// the operands of one cell's rows: velocities from the LDS images, volume classes from memory
template <typename T>
struct VResSampler {
  const T* img;                // component c at img + c * isz
  int isz, lsx, PZ, lofs;
  const Compact& c;
  int gx, gy, z;
  __device__ __forceinline__ double vol(int p, int ox, int oy, int oz) const {
    return (double)((const T*)c.vol[p])[c.idx(gx + ox, gy + oy, z + oz)];
  }
  __device__ __forceinline__ double vel(int comp, int dx, int dy, int dz) const {
    return (double)img[comp * isz + lofs + dx * lsx + dy * PZ + dz];
  }
  __device__ __forceinline__ bool tap_ok(int, int, int, int) const { return true; }
};

// the 16 distinct volume samples the three rows of ONE cell read (class p at compact offset (ox, oy, oz)), in the order
// VResClsSampler::vol resolves them
__device__ constexpr int kVResCls[16][4] = {
    {3, 0, 0, 0}, {5, 0, 0, 0}, {6, 0, 0, 0}, {7, 0, 0, 0}, {7, -1, 0, 0}, {7, 0, -1, 0}, {7, 0, 0, -1}, {1, 0, 0, 0},
    {1, 1, 0, 0}, {1, 0, 1, 0}, {2, 0, 0, 0}, {2, 1, 0, 0}, {2, 0, 0, 1}, {4, 0, 0, 0}, {4, 0, 1, 0}, {4, 0, 0, 1}};

// CREG: the same operands with the volume samples held in REGISTERS for the whole launch (they are constants of the
// solve): the rows then touch no memory but LDS -- 16 samples per cell, so only where KC cells' worth fit the file
template <typename T>
struct VResClsSampler {
  const T* img;
  int isz, lsx, PZ, lofs;
  const T (&cl)[16];
  __device__ __forceinline__ double vol(int p, int ox, int oy, int oz) const {
    const int key = p * 27 + (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1);
    switch (key) {
      case 3 * 27 + 13: return (double)cl[0];
      case 5 * 27 + 13: return (double)cl[1];
      case 6 * 27 + 13: return (double)cl[2];
      case 7 * 27 + 13: return (double)cl[3];
      case 7 * 27 + 0 * 9 + 1 * 3 + 1: return (double)cl[4];
      case 7 * 27 + 1 * 9 + 0 * 3 + 1: return (double)cl[5];
      case 7 * 27 + 1 * 9 + 1 * 3 + 0: return (double)cl[6];
      case 1 * 27 + 13: return (double)cl[7];
      case 1 * 27 + 2 * 9 + 1 * 3 + 1: return (double)cl[8];
      case 1 * 27 + 1 * 9 + 2 * 3 + 1: return (double)cl[9];
      case 2 * 27 + 13: return (double)cl[10];
      case 2 * 27 + 2 * 9 + 1 * 3 + 1: return (double)cl[11];
      case 2 * 27 + 1 * 9 + 1 * 3 + 2: return (double)cl[12];
      case 4 * 27 + 13: return (double)cl[13];
      case 4 * 27 + 1 * 9 + 2 * 3 + 1: return (double)cl[14];
      case 4 * 27 + 1 * 9 + 1 * 3 + 2: return (double)cl[15];
    }
    __builtin_trap();            // a sample the table does not hold: the tap table and kVResCls disagree
  }
  __device__ __forceinline__ double vel(int comp, int dx, int dy, int dz) const {
    return (double)img[comp * isz + lofs + dx * lsx + dy * PZ + dz];
  }
  __device__ __forceinline__ bool tap_ok(int, int, int, int) const { return true; }
};

template <typename T, int KC, bool CREG = false>
__global__ void __launch_bounds__(kVResBlock, 2)
k_vcg_resident(VResArgs a) {
  double* const scal = a.scal;
  if (__hip_atomic_load(scal + S_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) return;   // uniform over the grid
  if ((int)blockIdx.x == a.test_drop_wg) return;
  extern __shared__ __align__(16) unsigned char vres_smem[];
  T* const img = reinterpret_cast<T*>(vres_smem);
  __shared__ int s_fail;
  const int tid = threadIdx.x, W = gridDim.x;
  VResGeom g;
  g.Nx = a.c.N[0]; g.Ny = a.c.N[1]; g.Nz = a.c.N[2]; g.PZ = g.Nz + 1;
  {
    const int px = blockIdx.x / a.Py, py = blockIdx.x % a.Py;
    g.xa = 1 + px * a.bxm; g.ya = 1 + py * a.bym;
    g.bx = max(min(g.xa + a.bxm, g.Nx) - g.xa, 0);
    g.by = max(min(g.ya + a.bym, g.Ny) - g.ya, 0);
    if (g.bx == 0 || g.by == 0) { g.bx = 0; g.by = 0; }
  }
  g.pitch_y = a.bym + 2; g.lsx = g.pitch_y * g.PZ; g.isz = (a.bxm + 2) * g.lsx;
  T* const X = (T*)a.x; T* const R = (T*)a.r; T* const Q = (T*)a.q; T* const D = (T*)a.d;
  const int Nz = g.Nz;
  const int items = g.bx * g.by * Nz;

  // ---- the images: d of the box and its halo ring (faces that do not exist: 0)
  for (int e = tid; e < 3 * g.isz; e += kVResBlock) {
    const int c = e / g.isz, rem = e - c * g.isz;
    const int lx = rem / g.lsx - 1, r2 = rem % g.lsx, ly = r2 / g.PZ - 1, z = r2 % g.PZ;
    const int gx = g.xa + lx, gy = g.ya + ly;
    T v = (T)0;
    if (items > 0 && lx <= g.bx && ly <= g.by && g.exists(c, gx, gy, z)) v = D[a.off[c] + g.fidx(c, gx, gy, z)];
    img[e] = v;
  }
  // ---- this thread's cells
  unsigned flags[KC];       // bits 0-2: face u / v / w active; 3-5: its mask bit (not solid); 6: box-face cell (publish)
  int cell[KC];             // lx | ly << 8 | z << 16
  T xs[KC][3], rs[KC][3], ds[KC][3], qs[KC][3];
  T cl[CREG ? KC : 1][16];     // CREG: the cells' volume samples
  if (tid == 0) s_fail = 0;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int item = tid + k * kVResBlock;
    const bool act = item < items;
    const int it_ = act ? item : 0;
    const int col = it_ / Nz, z = it_ - col * Nz;
    const int lx = g.by > 0 ? col / g.by : 0, ly = g.by > 0 ? col - lx * g.by : 0;
    const int gx = g.xa + lx, gy = g.ya + ly;
    cell[k] = lx | (ly << 8) | (z << 16);
    unsigned f = 0;
    if (act) {
      const unsigned m = a.c.msk[a.c.idx(gx, gy, z)];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (g.active(c, gx, gy, z)) f |= 1u << c;
        if ((m >> c) & 1u) f |= 8u << c;
      }
      if (lx == 0 || lx == g.bx - 1 || ly == 0 || ly == g.by - 1) f |= 64u;
    }
    flags[k] = f;
    if constexpr (CREG) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        cl[k][i] = (T)0;
        if (f & 7u) cl[k][i] = ((const T*)a.c.vol[kVResCls[i][0]])[a.c.idx(gx + kVResCls[i][1], gy + kVResCls[i][2], z + kVResCls[i][3])];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xs[k][c] = (T)0; rs[k][c] = (T)0; ds[k][c] = (T)0; qs[k][c] = (T)0;
      if (f & (1u << c)) {
        const int64_t i = a.off[c] + g.fidx(c, gx, gy, z);
        xs[k][c] = X[i]; rs[k][c] = R[i]; ds[k][c] = D[i];
      }
    }
  }
  // ---- the halo ring: (bx + 2)(by + 2) - bx by columns x Nz x 3 components
  const int ring = items > 0 ? 2 * (g.by + 2) + 2 * g.bx : 0;
  const int nh = 3 * ring * Nz;
  auto halo_of = [&](int h, int* c_out, int* gx_out, int* gy_out, int* z_out, int* l_out) {
    const int c = h / (ring * Nz), rem = h - c * ring * Nz;
    const int colh = rem / Nz, z = rem - colh * Nz;
    int lx, ly;
    if (colh < 2 * (g.by + 2)) { const bool hi = colh >= g.by + 2; ly = (hi ? colh - (g.by + 2) : colh) - 1; lx = hi ? g.bx : -1; }
    else { const int c2 = colh - 2 * (g.by + 2); const bool hi = c2 >= g.bx; lx = hi ? c2 - g.bx : c2; ly = hi ? g.by : -1; }
    *c_out = c; *gx_out = g.xa + lx; *gy_out = g.ya + ly; *z_out = z;
    *l_out = c * g.isz + g.lofs(lx, ly, z);
  };
  // this thread's halo elements (h = tid + j * 512), decoded ONCE: image offset and flat DOF index (-1: a face nobody
  // updates -- outside the arrays or on their boundary: the image keeps what the set-up put there)
  int hl[kVResHK], hm[kVResHK];
#pragma unroll
  for (int j = 0; j < kVResHK; ++j) {
    const int h = tid + j * kVResBlock;
    hl[j] = 0; hm[j] = -1;
    if (h < nh) {
      int c, gx, gy, z, l;
      halo_of(h, &c, &gx, &gy, &z, &l);
      hl[j] = l;
      if (g.active(c, gx, gy, z)) hm[j] = (int)(a.off[c] + g.fidx(c, gx, gy, z));
    }
  }
  double delta = scal[S_RING + (int)((int64_t)scal[S_ITERS] & 1)];
  const double tol2 = scal[S_TOL2];
  const int64_t it0 = (int64_t)scal[S_ITERS];
  __syncthreads();

#ifdef MFS_VRES_STAMP
  u64 st_acc[10] = {}, st_last = wall_clock64();
  int st_n = 0;
#endif
  bool ran = false, d_current = true;       // d_current: the registers' d is the loop state's d (false only while it is owed)
  for (int it = 0; it < a.n_iter; ++it) {
#ifdef MFS_VRES_STAMP
    ++st_n;
#endif
    VRES_STAMP(9);
    const int par = (int)((it0 + it) & 1);
    const unsigned tag = a.tag0 + 2u * (unsigned)it;
    // ---- q = A d, d.q          (:589-592)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const unsigned f = flags[k];
      if (!(f & 7u)) continue;
      // the cell's coordinates through an opaque move: everything derived from them (16 class addresses, 27 image offsets
      // per cell) is loop-invariant, and hoisted out of the iteration loop it costs ~55 registers per cell -- recompute
      int ck = cell[k];
      asm volatile("" : "+v"(ck));
      const int lx = ck & 255, ly = (ck >> 8) & 255, z = ck >> 16;
      double own;
      if constexpr (CREG) {
        const VResClsSampler<T> smp{img, g.isz, g.lsx, g.PZ, g.lofs(lx, ly, z), cl[k]};
        if (f & 1u) { qs[k][0] = (T)vcg_row_s<0, false>(smp, a.k1, a.k2, (f & 8u) != 0, own); acc += own * (double)qs[k][0]; }
        if (f & 2u) { qs[k][1] = (T)vcg_row_s<1, false>(smp, a.k1, a.k2, (f & 16u) != 0, own); acc += own * (double)qs[k][1]; }
        if (f & 4u) { qs[k][2] = (T)vcg_row_s<2, false>(smp, a.k1, a.k2, (f & 32u) != 0, own); acc += own * (double)qs[k][2]; }
      } else {
        const VResSampler<T> smp{img, g.isz, g.lsx, g.PZ, g.lofs(lx, ly, z), a.c, g.xa + lx, g.ya + ly, z};
        if (f & 1u) { qs[k][0] = (T)vcg_row_s<0, false>(smp, a.k1, a.k2, (f & 8u) != 0, own); acc += own * (double)qs[k][0]; }
        if (f & 2u) { qs[k][1] = (T)vcg_row_s<1, false>(smp, a.k1, a.k2, (f & 16u) != 0, own); acc += own * (double)qs[k][1]; }
        if (f & 4u) { qs[k][2] = (T)vcg_row_s<2, false>(smp, a.k1, a.k2, (f & 32u) != 0, own); acc += own * (double)qs[k][2]; }
      }
    }
    bool ok;
    VRES_STAMP(0);
    vres_allreduce_begin(acc, a.ar, tag);
    VRES_STAMP(1);
    const double dq = vres_allreduce_end(a.ar, W, tag, it == 0 ? a.first_timeout_ticks : a.timeout_ticks, &ok);
    VRES_STAMP(2);
    if (!ok) { if (tid == 0) slab_fail(scal, it == 0 ? kErrNotResident : kErrArTimeout); ran = false; break; }
    ran = true;
    // ---- r -= alpha q ; r.r ; box faces of r -> mirror ; x += alpha d while the records travel          (:594-604)
    const double alpha = delta / dq;
    u64* const mir = a.mirror + (size_t)par * (size_t)(a.off[2] + (int64_t)g.Nx * g.Ny * (g.Nz + 1)) * Gran<T>::N;
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (!(flags[k] & (1u << c))) continue;
        rs[k][c] = (T)((double)rs[k][c] - alpha * (double)qs[k][c]);
        acc += (double)rs[k][c] * (double)rs[k][c];
      }
    }
    VRES_STAMP(3);
    vres_allreduce_begin(acc, a.ar, tag + 1u);
    VRES_STAMP(4);
    MFS_VISC_LDS_BARRIER();      // the record goes first: the faces' write-through stores would delay it everywhere
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (!(flags[k] & 64u)) continue;
      int ck = cell[k];
      asm volatile("" : "+v"(ck));
      const int lx = ck & 255, ly = (ck >> 8) & 255, z = ck >> 16;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (flags[k] & (1u << c)) vres_publish<T>(mir, a.off[c] + g.fidx(c, g.xa + lx, g.ya + ly, z), rs[k][c], tag + 1u);
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (flags[k] & (1u << c)) xs[k][c] = (T)((double)xs[k][c] + alpha * (double)ds[k][c]);
    }
    d_current = false;
    // the neighbours' faces -> registers.  kVResHB requests per thread in flight at a time, ONE wait for the batch (a wait
    // per element made this loop 7 dependent round trips long: half the iteration); what has not landed yet is polled
    bool hok = true;
    T rhv[kVResHK];
#pragma unroll
    for (int jb = 0; jb < kVResHK; jb += kVResHB) {
      vres_u64x2 w[kVResHB];
#pragma unroll
      for (int j = 0; j < kVResHB; ++j) {
        w[j][0] = 0; w[j][1] = 0;
        if (hm[jb + j] >= 0) {
          if (Gran<T>::N == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(w[j][0]) : "v"(mir + hm[jb + j]) : "memory");
          else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[j]) : "v"(mir + 2 * (int64_t)hm[jb + j]) : "memory");
        }
      }
      static_assert(kVResHB == 4, "the wait below names four request registers");
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])::"memory");
#pragma unroll
      for (int j = 0; j < kVResHB; ++j) {
        rhv[jb + j] = (T)0;
        if (hm[jb + j] < 0) continue;
        T rh;
        bool got;
        if (Gran<T>::N == 1) {
          got = (unsigned)(w[j][0] >> 32) == tag + 1u;
          rh = (T)__uint_as_float((unsigned)w[j][0]);
        } else {
          got = (unsigned)(w[j][0] >> 32) == tag + 1u && (unsigned)(w[j][1] >> 32) == tag + 1u;
          rh = (T)__longlong_as_double((long long)((w[j][1] << 32) | (w[j][0] & 0xffffffffull)));
        }
        if (!got && hok && !vres_fetch<T>(mir, hm[jb + j], tag + 1u, a.timeout_ticks, &rh)) hok = false;
        rhv[jb + j] = rh;
      }
    }
    if (!hok) s_fail = 1;
    VRES_STAMP(5);
    const double rr = vres_allreduce_end(a.ar, W, tag + 1u, a.timeout_ticks, &ok);
    VRES_STAMP(6);      // (its barrier publishes s_fail too)
    if (!ok) { if (tid == 0) slab_fail(scal, kErrArTimeout); break; }
    if (s_fail) { if (tid == 0) slab_fail(scal, kErrHaloTimeout); break; }
    const bool conv = rr < tol2;
    const int bad = cg_health(dq, rr);
    const double beta = rr / delta;
    if (blockIdx.x == 0 && tid == 0) {      // k_update_rdx's bookkeeping
      const int64_t itc = it0 + it;
      if (2 * itc + 2 < a.hist_cap) { a.hist[2 * itc + 1] = dq; a.hist[2 * itc + 2] = rr; }
      scal[S_ITERS] = (double)(itc + 1);
      scal[S_RING + (par ^ 1)] = rr;
      scal[S_DQ] = dq; scal[S_RR] = rr; scal[S_DELTA] = delta; scal[S_LASTRR] = rr; scal[S_ALPHA] = alpha;
      if (bad) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; }
      else if (conv) scal[S_DONE] = 1.0; else scal[S_BETA] = beta;
    }
    if (bad || conv) { d_current = true; break; }        // the reference leaves d alone when it stops (:605-606)
    // ---- d = r + beta d: own faces from registers, the halo ring from the neighbours' published r          (:607-610)
    delta = rr;
    const bool more = it + 1 < a.n_iter;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      int ck = cell[k];
      asm volatile("" : "+v"(ck));
      const int lx = ck & 255, ly = (ck >> 8) & 255, z = ck >> 16;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (!(flags[k] & (1u << c))) continue;
        ds[k][c] = (T)((double)rs[k][c] + beta * (double)ds[k][c]);
        if (more) img[c * g.isz + g.lofs(lx, ly, z)] = ds[k][c];
      }
    }
    d_current = true;
    if (!more) break;
#pragma unroll
    for (int j = 0; j < kVResHK; ++j)
      if (hm[j] >= 0) img[hl[j]] = (T)((double)rhv[j] + beta * (double)img[hl[j]]);
    VRES_STAMP(7);
    MFS_VISC_LDS_BARRIER();
    VRES_STAMP(8);
  }
#ifdef MFS_VRES_STAMP
  if (blockIdx.x == 0 && tid == 0 && st_n > 100)
    printf("viscosity resident stamps, ns per iteration over %d: rows %.0f | begin1 %.0f | end1 %.0f | r update %.0f | begin2 %.0f | publish+x+halo fetch %.0f | end2 %.0f | d+halo %.0f | barrier %.0f | loop top %.0f\n",
           st_n, 10.0 * st_acc[0] / st_n, 10.0 * st_acc[1] / st_n, 10.0 * st_acc[2] / st_n, 10.0 * st_acc[3] / st_n,
           10.0 * st_acc[4] / st_n, 10.0 * st_acc[5] / st_n, 10.0 * st_acc[6] / st_n, 10.0 * st_acc[7] / st_n,
           10.0 * st_acc[8] / st_n, 10.0 * st_acc[9] / st_n);
#endif
  // ---- state back to the arrays
  if (!ran || !d_current) return;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int lx = cell[k] & 255, ly = (cell[k] >> 8) & 255, z = cell[k] >> 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (!(flags[k] & (1u << c))) continue;
      const int64_t i = a.off[c] + g.fidx(c, g.xa + lx, g.ya + ly, z);
      X[i] = xs[k][c]; R[i] = rs[k][c]; D[i] = ds[k][c]; Q[i] = qs[k][c];
    }
  }
}

// ------------------------------------------------------------------ host side --------------
struct VResPlan {
  bool ok = false;
  int W = 0, Px = 0, Py = 0, bxm = 0, bym = 0, kc = 0;
  size_t lds = 0;
};

// the decomposition with the least halo among the factorisations of W that fit (KC <= kc_max, LDS <= kVResLdsMax)
static inline VResPlan vres_plan(int Nx, int Ny, int Nz, size_t elt, int W, int kc_max) {
  VResPlan best;
  if (Nx < 3 || Ny < 3 || Nz < 3 || Nz > 4000) return best;
  int64_t best_halo = -1;
  for (int Px = 1; Px <= W; ++Px) {
    if (W % Px) continue;
    const int Py = W / Px;
    const int bxm = (Nx - 1 + Px - 1) / Px, bym = (Ny - 1 + Py - 1) / Py;
    if (bxm < 1 || bym < 1 || bxm > 255 || bym > 255) continue;
    if ((int64_t)(Px - 1) * bxm >= Nx - 1 || (int64_t)(Py - 1) * bym >= Ny - 1) continue;     // no empty boxes
    const int64_t items = (int64_t)bxm * bym * Nz;
    const int kc = (int)((items + kVResBlock - 1) / kVResBlock);
    const int64_t ringc = 2 * (bym + 2) + 2 * bxm;
    const size_t lds = (size_t)3 * (bxm + 2) * (bym + 2) * (Nz + 1) * elt;
    if (kc > kc_max || lds > kVResLdsMax || 3 * ringc * Nz > (int64_t)kVResHK * kVResBlock) continue;
    const int64_t halo = (int64_t)bxm + bym;
    if (best_halo < 0 || halo < best_halo) {
      best_halo = halo;
      best.ok = true; best.W = W; best.Px = Px; best.Py = Py; best.bxm = bxm; best.bym = bym; best.kc = kc; best.lds = lds;
    }
  }
  return best;
}

static inline size_t vres_ws_bytes(int64_t n, size_t elt) {
  if (n > kVResMaxDofs) return 0;
  const size_t gran = elt == 4 ? 1 : 2;
  return align_up((size_t)kVResRing * kVResMaxW * kVResRecStride * 8, 4096) + align_up(2 * (size_t)n * gran * 8, 4096);
}

}  // namespace mfs
