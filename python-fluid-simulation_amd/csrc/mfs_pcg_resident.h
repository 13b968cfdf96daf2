// mfs_pcg_resident.h -- the whole CG loop of a SMALL grid in one launch (gfx950).
//
// The reference's own scene is 48x80x48 cells (3D_viscous_fluid_sim.ipynb): 184 k unknowns, 1.5 MB per vector.  At that
// size an iteration of PressureCGSolver3D.py:207-221 moves next to nothing; what it costs is synchronisation -- two
// dot products per iteration, each a point every cell has to wait for.  With kernel boundaries as those points the
// iteration cannot go below two launches' worth of ramp, drain and dependent round trips (~11 us measured with every
// avoidable round trip removed, tools/small_iter.py).  This kernel keeps the state where no launch can reach it:
//
//   * W (<= 64) workgroups of 512 threads stay resident for up to `n_iter` iterations; workgroup (px, py) owns the
//     x/y box [xa, xb) x [ya, yb) of interior columns, all of z.  A thread holds KV z-vectors of x, r, d and q in
//     REGISTERS -- and the seven coefficients of those cells as well (dense, read once per launch): 44-48 registers
//     per vector, KV <= 4 vectors per thread, the whole 512 KB register file of a CU in use.  Vector-interleaved:
//     item = thread + k * 512 of the box's flattened (column, z-vector) space, so LDS and global accesses of a wave
//     are contiguous.  Inside the loop nothing is loaded from memory but the neighbours' r faces.
//   * the box's d, with one halo layer in x and y, lives in an LDS image; the stencil reads its neighbours there.
//   * a dot product = workgroup sum -> ONE self-validating 16-byte record {tag|lo, tag|hi} per workgroup in a small
//     global table -> every workgroup's first wave polls the W records (a lane each) and adds them with the same
//     butterfly: all workgroups hold the bit-identical total and take the same decisions.  No counter, no flag, no
//     fence: an aligned 8-byte granule is never torn and carries its own episode tag (the scheme of mfs_p2p.h,
//     between workgroups instead of GPUs).  Cost ~1 us per dot product instead of a kernel boundary plus tail.
//   * halos: after the r update every workgroup publishes the r vectors of its box faces as tagged granules into a
//     global mirror; while the r.r total is in flight they arrive, and each workgroup forms its neighbours' d_new
//     cells itself (d_halo = r_halo + beta * d_halo_old -- k_update_d's arithmetic) -- no second exchange.
//   * bookkeeping (convergence test :218, history, beta :220) is cg_book of mfs_cg_core.h, run by workgroup 0.
//
// Every wait is a bounded spin; a timeout raises the engine's error word (kErrArTimeout / kErrHaloTimeout), all
// workgroups leave WITHOUT writing their state back, the host reports MFS_E_TIMEOUT.  The one timeout that can happen on
// healthy hardware -- a GPU shared with other work does not give the launch all its workgroups at once -- shows at the
// very first dot product of a launch, before anything has been written anywhere: it raises kErrNotResident instead
// (short bound, `first_timeout_ticks`), and mfs_pcg3d_poll turns that into "this engine uses the launch-per-phase loop
// from now on" -- arrays and scalar block are exactly as they were before the launch.  Workgroups never wait for anything a peer publishes AFTER
// waiting itself for them within the same episode, so the launch cannot deadlock as long as all W workgroups get a
// CU -- W <= 64 of 256, one workgroup per CU by LDS size; nothing else runs on the stream.
//
// Arithmetic: the stencil row is stencil_vec's expression, the vector updates are k_update_xr's / k_update_d's; only
// the grouping of the two dot products differs from the launch-per-phase loops (so results agree to rounding, not
// bit for bit -- tests/test_resident_gpu.py states the tolerances).  State enters and leaves through the same arrays
// and scalar block as mfs_pcg3d_iterate's other forms (d_j in the ping-pong buffer j & 1, d_{j+1} owed), so batches of
// either kind can follow each other.
#pragma once
#include "mfs_cg_core.h"
#include "mfs_pcg_apply.h"

namespace mfs {

constexpr int kResBlock = 512;
constexpr int kResMaxW = 64;
constexpr int kResRing = 4;
constexpr int64_t kResMaxCells = 1 << 20;          // grids beyond this never qualify (mirror sizing)
constexpr size_t kResLdsMax = 150 * 1024;
constexpr int kResRecStrideMax = 1024;             // u64 words between two workgroups' records (8 KiB), at most

// measurement build only (tools/build_variant.sh NAME -DMFS_RES_STAMP): workgroup 0 accumulates the wall-clock span of
// every phase of the loop and prints the averages when the launch ends
#ifdef MFS_RES_STAMP
#define RES_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const u64 t_ = wall_clock64(); st_acc[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define RES_STAMP(i) do { } while (0)
#endif

struct ResArgs {
  void *x, *r, *q, *dbuf[2];
  void* zb;                                         // JAC: z = r / diag written back beside r
  const void *diag, *cx, *cy, *cz, *cz2;
  const unsigned char* cls;
  int Nx, Ny, Nz, Px, Py, bxm, bym;
  double* scal;
  double* hist;
  int64_t hist_cap;
  int64_t j0;
  int n_iter;
  u64* ar;                                          // [kResRing][kResMaxW] records of 2 granules, `rec_stride` u64 apart
  int rec_stride;
  u64* mirror;                                      // [2][n * Gran<T>::N] granules: r of the box faces, by global cell index
  unsigned tag0;                                    // first episode tag of this launch (>= 1; 2 per iteration)
  u64 timeout_ticks, first_timeout_ticks;
  int test_drop_wg;                                 // fault injection (MFS_RES_TEST_DROP_WG): this workgroup never shows up; -1 none
};

__device__ __forceinline__ u64 dev_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dev_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// both granules of a record in ONE request (a torn 16-byte read is harmless: each half validates itself)
typedef unsigned long long res_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ res_u64x2 dev_load2(const u64* p) {
  res_u64x2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// A dot product in two halves, so that work which does not need the total can sit between them.
// begin: workgroup sum (fixed order) -> this workgroup's record in the table.
__device__ __forceinline__ void res_allreduce_begin(double v, u64* ar, int rec_stride, unsigned tag) {
  __shared__ double s_w[kResBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  v = wave_sum(v);
  if (lane == 0) s_w[wave] = v;
  MFS_LDS_BARRIER();            // only LDS is shared inside the workgroup: stores to the mirror / arrays stay in flight
  if (wave == 0 && lane == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kResBlock / kWave; ++w) t += s_w[w];
    u64* tab = ar + ((size_t)(tag % kResRing) * kResMaxW + blockIdx.x) * rec_stride;
    const u64 bits = (u64)__double_as_longlong(t);
    dev_store(tab + 0, ((u64)tag << 32) | (bits & 0xffffffffull));
    dev_store(tab + 1, ((u64)tag << 32) | (bits >> 32));
  }
}

// end: the first wave polls the W records (a lane each) and adds them in one fixed tree -- every workgroup holds the
// bit-identical total; known to every thread on return.  *ok false on timeout.  The first polls read no clock (a
// clock read is itself a memory-latency operation; records normally land within a poll or two).
__device__ __forceinline__ double res_allreduce_end(u64* ar, int rec_stride, int W, unsigned tag, u64 timeout_ticks, bool* ok) {
  __shared__ double s_tot;
  __shared__ int s_ok;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (wave == 0) {
    double c = 0.0;
    bool good = true;
    if (lane < W) {
      const u64* g = ar + ((size_t)(tag % kResRing) * kResMaxW + lane) * rec_stride;
      res_u64x2 w = dev_load2(g);
      bool got = (w[0] >> 32) == tag && (w[1] >> 32) == tag;
      for (int spin = 0; spin < 64 && !got; ++spin) {
        w = dev_load2(g);
        got = (w[0] >> 32) == tag && (w[1] >> 32) == tag;
      }
      if (!got) {
        const u64 t0 = wall_clock64();
        for (;;) {
          __builtin_amdgcn_s_sleep(1);
          w = dev_load2(g);
          if ((w[0] >> 32) == tag && (w[1] >> 32) == tag) break;
          if (wall_clock64() - t0 > timeout_ticks) { good = false; break; }
        }
      }
      c = __longlong_as_double((long long)((w[1] << 32) | (w[0] & 0xffffffffull)));
    }
    good = __all(good);
    c = wave_sum(c);                                   // the same tree in every workgroup
    if (lane == 0) { s_tot = c; s_ok = good ? 1 : 0; }
  }
  MFS_LDS_BARRIER();
  *ok = s_ok != 0;
  return s_tot;
}

// the same exchange for TWO values at once (Jacobi: r.r and r.z): two records, episodes tag1 / tag2, one barrier each way
__device__ __forceinline__ void res_allreduce_begin2(double v1, double v2, u64* ar, int rec_stride, unsigned tag1, unsigned tag2) {
  __shared__ double s_w2[2][kResBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  v1 = wave_sum(v1);
  v2 = wave_sum(v2);
  if (lane == 0) { s_w2[0][wave] = v1; s_w2[1][wave] = v2; }
  MFS_LDS_BARRIER();
  if (wave == 0 && lane < 2) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kResBlock / kWave; ++w) t += s_w2[lane][w];
    const unsigned tag = lane == 0 ? tag1 : tag2;
    u64* tab = ar + ((size_t)(tag % kResRing) * kResMaxW + blockIdx.x) * rec_stride;
    const u64 bits = (u64)__double_as_longlong(t);
    dev_store(tab + 0, ((u64)tag << 32) | (bits & 0xffffffffull));
    dev_store(tab + 1, ((u64)tag << 32) | (bits >> 32));
  }
}

__device__ __forceinline__ void res_allreduce_end2(u64* ar, int rec_stride, int W, unsigned tag1, unsigned tag2, u64 timeout_ticks,
                                                   bool* ok, double* t1, double* t2) {
  __shared__ double s_tot2[2];
  __shared__ int s_ok2;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (wave == 0) {
    double c[2] = {0.0, 0.0};
    bool good = true;
    if (lane < W) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const unsigned tag = m == 0 ? tag1 : tag2;
        const u64* g = ar + ((size_t)(tag % kResRing) * kResMaxW + lane) * rec_stride;
        res_u64x2 w = dev_load2(g);
        bool got = (w[0] >> 32) == tag && (w[1] >> 32) == tag;
        for (int spin = 0; spin < 64 && !got; ++spin) { w = dev_load2(g); got = (w[0] >> 32) == tag && (w[1] >> 32) == tag; }
        if (!got && good) {
          const u64 t0 = wall_clock64();
          for (;;) {
            __builtin_amdgcn_s_sleep(1);
            w = dev_load2(g);
            if ((w[0] >> 32) == tag && (w[1] >> 32) == tag) break;
            if (wall_clock64() - t0 > timeout_ticks) { good = false; break; }
          }
        }
        c[m] = __longlong_as_double((long long)((w[1] << 32) | (w[0] & 0xffffffffull)));
      }
    }
    good = __all(good);
    const double a1 = wave_sum(c[0]), a2 = wave_sum(c[1]);
    if (lane == 0) { s_tot2[0] = a1; s_tot2[1] = a2; s_ok2 = good ? 1 : 0; }
  }
  MFS_LDS_BARRIER();
  *ok = s_ok2 != 0;
  *t1 = s_tot2[0];
  *t2 = s_tot2[1];
}

// r of one z-vector as tagged granules (agent scope), and back
__device__ __forceinline__ void dev_store2(u64* p, u64 a, u64 b) {
  // 16-byte write-through store of TWO granules (a store torn between its halves is harmless: each validates itself).
  // The trailing s_nop is part of the instruction's contract -- see sys_store2 in mfs_p2p.h.
  res_u64x2 v = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <typename T, int VEC>
__device__ __forceinline__ void res_publish(u64* buf, int64_t elem, vec_t<T, VEC> v, unsigned tag) {
  const u64 t = (u64)tag << 32;
  u64* g = buf + elem * Gran<T>::N;
  u64 w[VEC * Gran<T>::N];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if (Gran<T>::N == 1) {
      w[j] = t | (u64)__float_as_uint((float)v[j]);
    } else {
      const u64 bits = (u64)__double_as_longlong((double)v[j]);
      w[2 * j] = t | (bits & 0xffffffffull);
      w[2 * j + 1] = t | (bits >> 32);
    }
  }
#pragma unroll
  for (int k = 0; k < VEC * Gran<T>::N; k += 2) dev_store2(g + k, w[k], w[k + 1]);      // elem is a multiple of VEC: 16-byte aligned
}

template <typename T, int VEC>
__device__ __forceinline__ bool res_fetch(const u64* buf, int64_t elem, unsigned tag, u64 timeout_ticks, vec_t<T, VEC>* out) {
  constexpr int NG = VEC * Gran<T>::N;
  const u64* g = buf + elem * Gran<T>::N;
  u64 w[NG];
  bool all = true;
#pragma unroll
  for (int k = 0; k < NG; ++k) { w[k] = dev_load(g + k); all = all && (unsigned)(w[k] >> 32) == tag; }
  if (!all) {
    const u64 t0 = wall_clock64();
    for (;;) {
      __builtin_amdgcn_s_sleep(1);
      all = true;
#pragma unroll
      for (int k = 0; k < NG; ++k) { w[k] = dev_load(g + k); all = all && (unsigned)(w[k] >> 32) == tag; }
      if (all) break;
      if (wall_clock64() - t0 > timeout_ticks) return false;
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if (Gran<T>::N == 1) (*out)[j] = (T)__uint_as_float((unsigned)w[j]);
    else (*out)[j] = (T)__longlong_as_double((long long)((w[2 * j + 1] << 32) | (w[2 * j] & 0xffffffffull)));
  }
  return true;
}

// geometry of one workgroup's box
struct ResBox {
  int xa, xb, ya, yb;            // interior columns owned: [xa, xb) x [ya, yb)
  int bx, by;                    // extents (0 if the box is empty)
  int pitch_y;                   // LDS image: element offset of (lx, ly, z) = ((lx + 1) * pitch_y + (ly + 1)) * Nz + z
};

// JAC: the opt-in Jacobi-preconditioned iteration (z = r / diag, delta = r.z; stopping rule r.r < tol^2 unchanged): z of the
// box faces travels instead of r, r.r and r.z share one exchange (two records), THREE episode tags per iteration, and z is
// written back beside r (the launch-per-phase Jacobi loop's operand, mfs_pcg.hip).
template <typename T, int VEC, int KV, bool ASYM, bool JAC = false>
__global__ void __launch_bounds__(kResBlock, 2)
k_pcg_resident(ResArgs a) {
  double* const scal = a.scal;
  // raised before this launch: uniform over the grid.  Agent-scope load: a workgroup scheduled after others of this launch
  // gave up must see THEIR flag too (a plain load may hit a stale line in this XCD's L2)
  if (__hip_atomic_load(scal + S_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) return;
  if ((int)blockIdx.x == a.test_drop_wg) return;
  extern __shared__ __align__(16) unsigned char res_smem[];
  T* const img = reinterpret_cast<T*>(res_smem);
  T* const rhalo = img + (size_t)(a.bxm + 2) * (a.bym + 2) * a.Nz;      // the neighbours' r on the halo, [nh] vectors
  __shared__ int s_fail;
  const int tid = threadIdx.x, W = gridDim.x;
  const int Nz = a.Nz, nzv = Nz / VEC;
  const int64_t sx = (int64_t)a.Ny * Nz;
  T* const x = (T*)a.x; T* const r = (T*)a.r; T* const q = (T*)a.q;
  const T *dg = (const T*)a.diag, *cx = (const T*)a.cx, *cy = (const T*)a.cy, *cz = (const T*)a.cz, *cz2 = (const T*)a.cz2;
  (void)cz2;

  ResBox b;
  {
    const int px = blockIdx.x / a.Py, py = blockIdx.x % a.Py;
    b.xa = 1 + px * a.bxm; b.xb = min(b.xa + a.bxm, a.Nx - 1);
    b.ya = 1 + py * a.bym; b.yb = min(b.ya + a.bym, a.Ny - 1);
    b.bx = max(b.xb - b.xa, 0); b.by = max(b.yb - b.ya, 0);
    if (b.bx == 0 || b.by == 0) { b.bx = 0; b.by = 0; }
    b.pitch_y = a.bym + 2;
  }
  const int lsx = b.pitch_y * Nz;                        // LDS stride of one x step
  const int items = b.bx * b.by * nzv;

  // ---- this thread's vectors
  int gofs[KV], lofs[KV];
  unsigned flags[KV];                                    // bit 0 active, 1 first, 2 last, 3 box face (publish)
  vec_t<T, VEC> xv[KV], rv[KV], dv[KV], qv[KV];
  vec_t<T, VEC> c_dg[KV], c_xm[KV], c_xp[KV], c_ym[KV], c_yp[KV], c_zm[KV], c_zm2[ASYM ? KV : 1];
  T c_zr[KV];                                            // cz just right of the vector (the +z weight of its last cell)
  const int64_t j0 = a.j0;
  const T* dsrc = (const T*)a.dbuf[(j0 == 0 ? 0 : (j0 - 1)) & 1];
  const double beta0 = j0 == 0 ? 0.0 : scal[S_BETA];
  if (tid == 0) s_fail = 0;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int item = tid + k * kResBlock;
    const bool act = item < items;
    const int it_ = act ? item : 0;
    const int col = it_ / nzv, zv = it_ - col * nzv;
    const int lx = b.by > 0 ? col / b.by : 0, ly = b.by > 0 ? col - lx * b.by : 0;
    gofs[k] = (int)((int64_t)(b.xa + lx) * sx + (int64_t)(b.ya + ly) * Nz + zv * VEC);
    lofs[k] = ((lx + 1) * b.pitch_y + (ly + 1)) * Nz + zv * VEC;
    const bool face = lx == 0 || lx == b.bx - 1 || ly == 0 || ly == b.by - 1;
    unsigned f = (act ? 1u : 0u) | (zv == 0 ? 2u : 0u) | (zv == nzv - 1 ? 4u : 0u) | (face ? 8u : 0u);
    xv[k] = vec_t<T, VEC>{}; rv[k] = vec_t<T, VEC>{}; dv[k] = vec_t<T, VEC>{}; qv[k] = vec_t<T, VEC>{};
    c_dg[k] = vec_t<T, VEC>{}; c_xm[k] = c_dg[k]; c_xp[k] = c_dg[k]; c_ym[k] = c_dg[k]; c_yp[k] = c_dg[k]; c_zm[k] = c_dg[k];
    if (ASYM) c_zm2[k] = c_dg[k];
    c_zr[k] = (T)0;
    if (act) {
      const int64_t g = gofs[k];
      c_dg[k] = vload<T, VEC>(dg + g);
      c_xm[k] = vload<T, VEC>(cx + g); c_xp[k] = vload<T, VEC>(cx + g + sx);
      c_ym[k] = vload<T, VEC>(cy + g); c_yp[k] = vload<T, VEC>(cy + g + Nz);
      c_zm[k] = vload<T, VEC>(cz + g);
      if (ASYM) c_zm2[k] = vload<T, VEC>(cz2 + g);
      if (zv != nzv - 1) c_zr[k] = cz[g + VEC];
      xv[k] = vload<T, VEC>(x + gofs[k]);
      rv[k] = vload<T, VEC>(r + gofs[k]);
      const vec_t<T, VEC> dp = vload<T, VEC>(dsrc + gofs[k]);
      if (j0 == 0) dv[k] = dp;
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          dv[k][j] = (T)((JAC ? (double)(T)jac_z((double)rv[k][j], (double)c_dg[k][j]) : (double)rv[k][j]) + beta0 * (double)dp[j]);
      }
      vstore<T, VEC>(img + lofs[k], dv[k]);
    }
    flags[k] = f;
  }
  // ---- the halo of the image: faces x = xa-1, x = xb (by columns each), y = ya-1, y = yb (bx columns each)
  const int nh = items > 0 ? 2 * (b.by + b.bx) * nzv : 0;
  auto halo_of = [&](int h, int* g_out, int* l_out, bool* has_owner) {
    const int colh = h / nzv, zv = h - colh * nzv;
    int gx, gy, lx, ly;
    if (colh < 2 * b.by) { const bool hi = colh >= b.by; ly = hi ? colh - b.by : colh; lx = hi ? b.bx : -1; }
    else { const int c2 = colh - 2 * b.by; const bool hi = c2 >= b.bx; lx = hi ? c2 - b.bx : c2; ly = hi ? b.by : -1; }
    gx = b.xa + lx; gy = b.ya + ly;
    *g_out = (int)((int64_t)gx * sx + (int64_t)gy * Nz + zv * VEC);
    *l_out = ((lx + 1) * b.pitch_y + (ly + 1)) * Nz + zv * VEC;
    *has_owner = gx >= 1 && gx < a.Nx - 1 && gy >= 1 && gy < a.Ny - 1;      // else a domain boundary column: never updated
  };
  for (int h = tid; h < nh; h += kResBlock) {
    int g, l; bool own;
    halo_of(h, &g, &l, &own);
    const vec_t<T, VEC> dp = vload<T, VEC>(dsrc + g);
    vec_t<T, VEC> o = dp;
    if (j0 != 0 && own) {
      vec_t<T, VEC> rh = vload<T, VEC>(r + g);
      if (JAC) {
        const vec_t<T, VEC> gh = vload<T, VEC>(dg + g);
#pragma unroll
        for (int j = 0; j < VEC; ++j) rh[j] = (T)jac_z((double)rh[j], (double)gh[j]);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (T)((double)rh[j] + beta0 * (double)dp[j]);
    }
    vstore<T, VEC>(img + l, o);
  }
  double delta = scal[S_RING + (int)(j0 & 1)];
  const double tol2 = scal[S_TOL2];
  const int64_t it0 = (int64_t)scal[S_ITERS];
  __syncthreads();

  int64_t jl = j0;                                       // the iteration whose d the registers hold at exit
  bool ran = false;
#ifdef MFS_RES_STAMP
  u64 st_acc[10] = {}, st_last = wall_clock64();
  int st_n = 0;
#endif
  for (int it = 0; it < a.n_iter; ++it) {
#ifdef MFS_RES_STAMP
    ++st_n;
#endif
    RES_STAMP(9);
    const int64_t jj = j0 + it;
    const int par = (int)(jj & 1);
    const unsigned tag = a.tag0 + (JAC ? 3u : 2u) * (unsigned)it;
    jl = jj; ran = true;
    // ---- q = A d, d.q
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if (!(flags[k] & 1u)) continue;
      const bool first = flags[k] & 2u, last = flags[k] & 4u;
      const double czr = (double)c_zr[k];
      const T* c0 = img + lofs[k];
      const vec_t<T, VEC> vxm = vload<T, VEC>(c0 - lsx), vxp = vload<T, VEC>(c0 + lsx);
      const vec_t<T, VEC> vym = vload<T, VEC>(c0 - Nz), vyp = vload<T, VEC>(c0 + Nz);
      const double zl = (double)c0[-1], zr = (double)c0[VEC];
      const vec_t<T, VEC> vc = dv[k];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const double zm = j == 0 ? zl : (double)vc[j == 0 ? 0 : j - 1];
        const double zp = j == VEC - 1 ? zr : (double)vc[j == VEC - 1 ? j : j + 1];
        const double czp = j == VEC - 1 ? czr : (double)c_zm[k][j == VEC - 1 ? j : j + 1];
        const double czm2 = ASYM ? (double)c_zm2[ASYM ? k : 0][j] : (double)c_zm[k][j];
        double val = 0;
        val -= (double)c_xp[k][j] * (double)vxp[j];
        val -= (double)c_xm[k][j] * (double)vxm[j];
        val -= (double)c_yp[k][j] * (double)vyp[j];
        val -= (double)c_ym[k][j] * (double)vym[j];
        val -= czp * zp;
        val -= czm2 * zm;
        val += (double)c_dg[k][j] * (double)vc[j];
        const bool bnd = (first && j == 0) || (last && j == VEC - 1);
        qv[k][j] = bnd ? (T)0 : (T)val;
        if (!bnd) acc += (double)vc[j] * (double)qv[k][j];
      }
    }
    bool ok;
    RES_STAMP(0);
    res_allreduce_begin(acc, a.ar, a.rec_stride, tag);
    RES_STAMP(1);
    const double dq = res_allreduce_end(a.ar, a.rec_stride, W, tag, it == 0 ? a.first_timeout_ticks : a.timeout_ticks, &ok);
    RES_STAMP(2);
    if (!ok) { if (tid == 0) slab_fail(scal, it == 0 ? kErrNotResident : kErrArTimeout); ran = false; break; }
    // ---- r -= alpha q ; r.r ; faces of r -> mirror ; then, while the r.r records travel, x += alpha d
    const double alpha = delta / dq;
    u64* const mir = a.mirror + (size_t)par * (size_t)a.Nx * sx * Gran<T>::N;
    acc = 0.0;
    double acz = 0.0;
    vec_t<T, VEC> zv_[JAC ? KV : 1];                       // JAC: z = r / diag, rounded to the state type like the stored z
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if (JAC) zv_[JAC ? k : 0] = vec_t<T, VEC>{};
      if (!(flags[k] & 1u)) continue;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        rv[k][j] = (T)((double)rv[k][j] - alpha * (double)qv[k][j]);
        acc += (double)rv[k][j] * (double)rv[k][j];
        if (JAC) {
          const double zz = jac_z((double)rv[k][j], (double)c_dg[k][j]);
          zv_[JAC ? k : 0][j] = (T)zz;
          acz += (double)rv[k][j] * zz;
        }
      }
    }
    RES_STAMP(3);
    if (JAC) res_allreduce_begin2(acc, acz, a.ar, a.rec_stride, tag + 1u, tag + 2u);
    else res_allreduce_begin(acc, a.ar, a.rec_stride, tag + 1u);
    RES_STAMP(4);
    // faces after the record HAS BEEN ISSUED (second barrier: the other waves do not overtake wave 0): their
    // write-through stores take a microsecond to drain, and a record queued behind them is a microsecond late everywhere
    MFS_LDS_BARRIER();
#pragma unroll
    for (int k = 0; k < KV; ++k)
      if ((flags[k] & 9u) == 9u) res_publish<T, VEC>(mir, gofs[k], JAC ? zv_[JAC ? k : 0] : rv[k], tag + 1u);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) xv[k][j] = (T)((double)xv[k][j] + alpha * (double)dv[k][j]);
    }
    // the neighbours' faces (published without waiting for anybody, so waiting for them here cannot deadlock) -> LDS
    bool hok = true;
    for (int h = tid; h < nh; h += kResBlock) {
      int g, l; bool own;
      halo_of(h, &g, &l, &own);
      if (!own) continue;
      vec_t<T, VEC> rh;
      if (!res_fetch<T, VEC>(mir, g, tag + 1u, a.timeout_ticks, &rh)) { hok = false; break; }
      vstore<T, VEC>(rhalo + (size_t)h * VEC, rh);
    }
    if (!hok) s_fail = 1;
    RES_STAMP(5);
    double rr, rz = 0.0;
    if (JAC) res_allreduce_end2(a.ar, a.rec_stride, W, tag + 1u, tag + 2u, a.timeout_ticks, &ok, &rr, &rz);
    else rr = res_allreduce_end(a.ar, a.rec_stride, W, tag + 1u, a.timeout_ticks, &ok);      // (its barrier publishes s_fail too)
    if (!ok) { if (tid == 0) slab_fail(scal, kErrArTimeout); ran = false; break; }
    const double dnew = JAC ? rz : rr;                      // delta of the next iteration
    if (s_fail) { if (tid == 0) slab_fail(scal, kErrHaloTimeout); ran = false; break; }
    if (blockIdx.x == 0 && tid == 0) {
      // cg_book (mfs_cg_core.h) with its operands already in registers: stores only, nothing on this thread's path waits
      const int64_t itc = it0 + it;
      if (2 * itc + 2 < a.hist_cap) { a.hist[2 * itc + 1] = dq; a.hist[2 * itc + 2] = rr; }
      scal[S_ITERS] = (double)(itc + 1);
      scal[S_RING + (par ^ 1)] = dnew;
      scal[S_RR] = rr; scal[S_DQ] = dq; scal[S_DELTA] = delta; scal[S_LASTRR] = rr;
      if (JAC) scal[S_RZ] = rz;
      scal[S_ALPHA] = delta / dq;
      if (const int bad = cg_health(dq, rr)) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; }
      else if (rr < tol2) scal[S_DONE] = 1.0;
      else scal[S_BETA] = dnew / delta;
    }
    RES_STAMP(6);
    if (cg_health(dq, rr) != 0 || rr < tol2 || it + 1 == a.n_iter) break;      // d_{j+1} is owed, as after every batch
    // ---- d = r + beta d: own vectors from registers, the halo from the neighbours' published r
    const double beta = dnew / delta;
    delta = dnew;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if (!(flags[k] & 1u)) continue;
#pragma unroll
      for (int j = 0; j < VEC; ++j) dv[k][j] = (T)((JAC ? (double)zv_[JAC ? k : 0][j] : (double)rv[k][j]) + beta * (double)dv[k][j]);
      vstore<T, VEC>(img + lofs[k], dv[k]);
    }
    for (int h = tid; h < nh; h += kResBlock) {
      int g, l; bool own;
      halo_of(h, &g, &l, &own);
      if (!own) continue;
      const vec_t<T, VEC> rh = vload<T, VEC>(rhalo + (size_t)h * VEC);      // written by this very thread above
      vec_t<T, VEC> o = vload<T, VEC>(img + l);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (T)((double)rh[j] + beta * (double)o[j]);
      vstore<T, VEC>(img + l, o);
    }
    RES_STAMP(7);
    MFS_LDS_BARRIER();
    RES_STAMP(8);
  }
#ifdef MFS_RES_STAMP
  if (blockIdx.x == 0 && tid == 0 && st_n > 100)
    printf("resident stamps, ns per iteration over %d: stencil %.0f | begin1 %.0f | end1 %.0f | r+publish %.0f | begin2 %.0f | x+halo fetch %.0f | end2 %.0f | d+halo %.0f | barrier %.0f | loop top %.0f\n",
           st_n, 10.0 * st_acc[0] / st_n, 10.0 * st_acc[1] / st_n, 10.0 * st_acc[2] / st_n, 10.0 * st_acc[3] / st_n,
           10.0 * st_acc[4] / st_n, 10.0 * st_acc[5] / st_n, 10.0 * st_acc[6] / st_n, 10.0 * st_acc[7] / st_n,
           10.0 * st_acc[8] / st_n, 10.0 * st_acc[9] / st_n);
#endif
  // ---- state back to the arrays: x, r, q and d_jl (buffer jl & 1)
  if (!ran) return;
  T* const dout = (T*)a.dbuf[jl & 1];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    if (!(flags[k] & 1u)) continue;
    vstore<T, VEC>(x + gofs[k], xv[k]);
    vstore<T, VEC>(r + gofs[k], rv[k]);
    vstore<T, VEC>(dout + gofs[k], dv[k]);
    if (JAC) {      // z of the final r: the operand the launch-per-phase Jacobi loop (and pcg_home_d) takes from the engine's z buffer
      vec_t<T, VEC> zo;
#pragma unroll
      for (int j = 0; j < VEC; ++j) zo[j] = (T)jac_z((double)rv[k][j], (double)c_dg[k][j]);
      vstore<T, VEC>((T*)a.zb + gofs[k], zo);
    }
    const bool first = flags[k] & 2u, last = flags[k] & 4u;
    if (!first && !last) vstore<T, VEC>(q + gofs[k], qv[k]);
    else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const bool bnd = (first && j == 0) || (last && j == VEC - 1);
        if (!bnd) q[gofs[k] + j] = qv[k][j];
      }
    }
  }
}

// ------------------------------------------------------------------ host side --------------
struct ResPlan {
  bool ok = false;
  int W = 0, Px = 0, Py = 0, bxm = 0, bym = 0, kv = 0;
  size_t lds = 0;
};

// the decomposition with the least halo among the factorisations of W that fit (KV <= 4, LDS image <= kResLdsMax)
static inline ResPlan res_plan(int Nx, int Ny, int Nz, int vec, size_t elt, int W) {
  ResPlan best;
  if (Nx < 3 || Ny < 3 || Nz % vec != 0 || Nz < 2 * vec) return best;
  const int nzv = Nz / vec;
  int64_t best_halo = -1;
  for (int Px = 1; Px <= W; ++Px) {
    if (W % Px) continue;
    const int Py = W / Px;
    const int bxm = (Nx - 2 + Px - 1) / Px, bym = (Ny - 2 + Py - 1) / Py;
    if (bxm < 1 || bym < 1) continue;
    const int64_t items = (int64_t)bxm * bym * nzv;
    const int kv = (int)((items + kResBlock - 1) / kResBlock);
    const size_t lds = ((size_t)(bxm + 2) * (bym + 2) + 2 * (size_t)(bxm + bym)) * Nz * elt;     // d image + the halo's r
    // registers: 4 vectors per thread fit (with spills that cost little) for 4-byte state; for 8-byte state the fourth
    // makes the loop slower than the launch-per-phase one (64^3: 11.8 vs 11.4 us per iteration)
    if (kv > (elt == 4 ? 4 : 3) || lds > kResLdsMax) continue;
    const int64_t halo = (int64_t)bxm + bym;
    if (best_halo < 0 || halo < best_halo) {
      best_halo = halo;
      best.ok = true; best.W = W; best.Px = Px; best.Py = Py; best.bxm = bxm; best.bym = bym; best.kv = kv; best.lds = lds;
    }
  }
  return best;
}

static inline size_t res_ws_bytes(int64_t n, size_t elt) {
  if (n > kResMaxCells) return 0;
  const size_t gran = elt == 4 ? 1 : 2;
  return align_up((size_t)kResRing * kResMaxW * kResRecStrideMax * 8, 4096) + align_up(2 * (size_t)n * gran * 8, 4096);
}

}  // namespace mfs
