// mfs_pcg_apply.h -- the per-iteration stencil kernel of the pressure CG (gfx950).
//
//   out = A v,  A = the reference's 7-point ghost-fluid operator
//   (solver/PressureCGSolver3D.py:52-130) in precomputed-coefficient form:
//   out[c] = diag[c] v[c] - cx[x+1] v[x+1] - cx[x] v[x-1] - cy[y+1] v[y+1]
//            - cy[y] v[y-1] - cz[z+1] v[z+1] - cz[z] v[z-1]
//   accumulated in the reference's order (+x -x +y -y +z -z, then diag), in fp64.
//   ASYM: the same launch serves the density solver's operator (solver/DensityCGSolver3D.py:
//   118-207), which differs in `diag` (built by its own setup) and in the -z tap, whose weight
//   is a fifth coefficient array cz2 (the reference reads wz[x,y,z+1] there, :184).
//
// Algorithmic HBM traffic: v, diag, cx, cy, cz read once, out written once
// = 6 scalars per cell.  Everything below is about making the
// hardware move exactly those bytes at HBM speed:
//
//  * one work item = one 16-byte z-vector (4 fp32 / 2 fp64 cells) of an interior
//    row; consecutive lanes = consecutive vectors, so every global access is a
//    full 1 KiB wave-wide request on the contiguous axis.
//  * variant MARCH (default): a workgroup owns a 256-vector tile of the (y,z)
//    plane and marches over a chunk of x planes.  v[x-1], v[x], v[x+1], cx[x],
//    cx[x+1] stay in registers across steps, so the x neighbours cost no memory
//    traffic at all; the in-plane neighbours (y+-1 rows, z+-1 cells) come from an
//    LDS-staged copy of the current plane tile (double buffered, one barrier per
//    plane, LDS-only wait so global prefetches stay in flight across it).
//    Operands of step x+1 are requested before step x is computed.
//  * XCD-aware order: blocks with equal blockIdx%8 share an XCD and its L2; each
//    label takes a contiguous range of (x-chunk, tile) work so the y-halo rows a
//    tile re-reads were just fetched by its neighbour tile on the same L2.
//  * the d.q partial sums ride along (wave shuffle tree -> LDS -> one double per
//    block); no atomics, bitwise reproducible.
#pragma once
#include "mfs_common.h"
#include "mfs_cg_core.h"

// workgroup barrier that waits on LDS traffic only: global prefetches stay in flight across it
#define MFS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#ifndef MFS_MARCH_MIN_WAVES
#define MFS_MARCH_MIN_WAVES 3      // waves per SIMD the march kernel is compiled for (register budget: 168 VGPRs, no spills)
#endif

namespace mfs {

constexpr int kApplyBlock = 256;
constexpr int kXcds = 8;

// One z-vector of the stencil.  zl / zr = v just left / right of the vector,
// czr = cz just right of it.  `first`/`last`: the vector holds the boundary cell
// z=0 / z=Nz-1, which is neither computed nor stored (PressureCGSolver3D.py:55-57).
template <typename T, int VEC>
__device__ __forceinline__ void stencil_vec(T* __restrict__ out_ptr, vec_t<T, VEC> vc, vec_t<T, VEC> vxp,
                                            vec_t<T, VEC> vxm, vec_t<T, VEC> vyp, vec_t<T, VEC> vym,
                                            vec_t<T, VEC> dg, vec_t<T, VEC> cxp, vec_t<T, VEC> cxm,
                                            vec_t<T, VEC> cyp, vec_t<T, VEC> cym, vec_t<T, VEC> czm, double zl,
                                            double zr, double czr, bool first, bool last, bool active, double& acc,
                                            vec_t<T, VEC> czm2) {   // czm2: weight of the -z tap (= czm unless ASYM)
  vec_t<T, VEC> o;
#ifdef MFS_APPLY_NATIVE_MATH   // experiment: arithmetic in the storage type
  typedef T C;
#else
  typedef double C;
#endif
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const C zm = j == 0 ? (C)zl : (C)vc[j == 0 ? 0 : j - 1];
    const C zp = j == VEC - 1 ? (C)zr : (C)vc[j == VEC - 1 ? j : j + 1];
    const C czp = j == VEC - 1 ? (C)czr : (C)czm[j == VEC - 1 ? j : j + 1];
    C val = 0;
    val -= (C)cxp[j] * (C)vxp[j];
    val -= (C)cxm[j] * (C)vxm[j];
    val -= (C)cyp[j] * (C)vyp[j];
    val -= (C)cym[j] * (C)vym[j];
    val -= czp * zp;
    val -= (C)czm2[j] * zm;
    val += (C)dg[j] * (C)vc[j];
    o[j] = (T)val;
    const bool bnd = (first && j == 0) || (last && j == VEC - 1);
    if (active && !bnd) acc += (double)vc[j] * (double)o[j];
  }
  if (!active) return;
  if (!first && !last) {
    vstore<T, VEC>(out_ptr, o);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool bnd = (first && j == 0) || (last && j == VEC - 1);
      if (!bnd) out_ptr[j] = o[j];
    }
  }
}

// planes [xb, xe) and, optionally, a second range [xb2, xe2) handled by the same launch
// (the multi-GPU driver applies its two edge planes {1, L-2} in one launch)
// items / runrem / count (round 3, optional): the (tile, plane) pairs that compute anything -- every other pair is a tile
// whose vectors are all ZERO rows with r = d = 0 (air): its q, d and x never change, and the march does not visit it.
// items[k] = tile * np + plane, ascending; runrem[k] = consecutive items from k on within the tile (one march); *count
// = how many.  Null: the dense sequence of all pairs.
struct ApplyArgs {
  int Nx, Ny, Nz, xb, xe, xchunk, xb2, xe2;
  const int* items = nullptr;
  const int* runrem = nullptr;
  const int* count = nullptr;
};

// ------------------------------------------------------------- variant 0 ----
// direct loads, no register/LDS reuse: every neighbour is its own global load and
// the caches do the rest.  Kept as the baseline the other variants are measured
// against (and as the generic fallback).
template <typename T, int VEC>
__global__ void __launch_bounds__(kApplyBlock)
k_pcg_apply_direct(const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ diag,
                   const T* __restrict__ cx, const T* __restrict__ cy, const T* __restrict__ cz, ApplyArgs a,
                   double* __restrict__ partial, const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  const int nzv = a.Nz / VEC;
  const int64_t ipp = (int64_t)(a.Ny - 2) * nzv;
  const int n1 = a.xe - a.xb;
  const int np = n1 + (a.xe2 - a.xb2);
  const int nch = min((int)gridDim.x, kXcds);
  const int xcd = blockIdx.x % nch, slot = blockIdx.x / nch;
  const int nblk = (gridDim.x - xcd + nch - 1) / nch;
  const int p0 = (int)((int64_t)np * xcd / nch), p1 = (int)((int64_t)np * (xcd + 1) / nch);
  const int64_t end = (int64_t)p1 * ipp, stride = (int64_t)nblk * kApplyBlock;
  const int64_t sx = (int64_t)a.Ny * a.Nz, sy = a.Nz;
  double acc = 0.0;
  for (int64_t it = (int64_t)p0 * ipp + (int64_t)slot * kApplyBlock + threadIdx.x; it < end; it += stride) {
    const int px = (int)(it / ipp);
    const int rem = (int)(it - (int64_t)px * ipp);
    const int yy = rem / nzv, zv = rem - (rem / nzv) * nzv;
    const int xx = px < n1 ? a.xb + px : a.xb2 + (px - n1);
    const int64_t base = (int64_t)xx * sx + (int64_t)(yy + 1) * sy + (int64_t)zv * VEC;
    const bool first = zv == 0, last = zv == nzv - 1;
    const auto vc = vload<T, VEC>(v + base);
    const double zl = first ? 0.0 : (double)v[base - 1];
    const double zr = last ? 0.0 : (double)v[base + VEC];
    const double czr = last ? 0.0 : (double)cz[base + VEC];
    stencil_vec<T, VEC>(out + base, vc, vload<T, VEC>(v + base + sx), vload<T, VEC>(v + base - sx),
                        vload<T, VEC>(v + base + sy), vload<T, VEC>(v + base - sy), vload<T, VEC>(diag + base),
                        vload<T, VEC>(cx + base + sx), vload<T, VEC>(cx + base), vload<T, VEC>(cy + base + sy),
                        vload<T, VEC>(cy + base), vload<T, VEC>(cz + base), zl, zr, czr, first, last, true, acc,
                        vload<T, VEC>(cz + base));
  }
  const double tot = block_sum<kApplyBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// ------------------------------------------------------ coefficient access ---
// The seven coefficients of the cells of one z-vector of one plane, as fetched
// ("raw") plus the vector's class.  Dense mode: raw is the truth.  Compressed
// mode (COMP): a byte per z-vector (k_pcg_classify) says whether every computed
// cell of the vector is a ZERO row (all seven coefficients 0: non-fluid), a REGULAR
// row (six face weights 1, diag 6: fluid cell away from solids and the free
// surface) or anything else (MIXED).  Only MIXED vectors read the four coefficient
// arrays; the others are reconstructed from the class, bit-identically.  In a
// fluid solve almost every vector is ZERO or REGULAR, so the kernel's HBM traffic
// drops from 6 to ~2.5 scalars per cell.  Lanes of non-MIXED vectors issue no
// coefficient request at all (exec-masked loads over class constants).
// (kClsZero / kClsRegular / kClsMixed: mfs_cg_core.h -- the Jacobi r / z update reads the class bytes too)

template <typename T, int VEC>
struct CoefVec {
  vec_t<T, VEC> dg, cxm, cxp, cym, cyp, czm, czm2;
  unsigned char cls;
};

// LOAD_CXM = false: the caller carries cx[x] over from the previous plane's cxp (the same array element;
// for a ZERO / REGULAR vector the class constant stands for it on every computed cell)
template <typename T, int VEC, bool COMP, int NT, bool LOAD_CXM = true, bool ASYM = false>
__device__ __forceinline__ CoefVec<T, VEC> coef_load(const T* __restrict__ diag, const T* __restrict__ cx,
                                                     const T* __restrict__ cy, const T* __restrict__ cz,
                                                     int64_t b, int64_t sx, int Nz, unsigned char cls,
                                                     const T* __restrict__ cz2 = nullptr) {
  CoefVec<T, VEC> c;
  c.cls = cls;
  if (COMP) {
    // class constants first (every lane), then an exec-masked overwrite for the MIXED lanes only: lanes
    // of ZERO / REGULAR vectors issue no memory request at all, and nothing after the branch reads the
    // loaded registers, so the loads stay in flight until the stencil consumes them.
    const T f = cls == kClsRegular ? (T)1 : (T)0, d = cls == kClsRegular ? (T)6 : (T)0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { c.dg[j] = d; c.cxm[j] = f; c.cxp[j] = f; c.cym[j] = f; c.cyp[j] = f; c.czm[j] = f; c.czm2[j] = f; }
    if (cls == kClsMixed) {
      c.dg = (NT & 1) ? vload_nt<T, VEC>(diag + b) : vload<T, VEC>(diag + b);
      if (LOAD_CXM) c.cxm = vload<T, VEC>(cx + b);
      c.cxp = (NT & 2) ? vload_nt<T, VEC>(cx + b + sx) : vload<T, VEC>(cx + b + sx);
      c.cym = (NT & 4) ? vload_nt<T, VEC>(cy + b) : vload<T, VEC>(cy + b);
      c.cyp = (NT & 4) ? vload_nt<T, VEC>(cy + b + Nz) : vload<T, VEC>(cy + b + Nz);
      c.czm = (NT & 1) ? vload_nt<T, VEC>(cz + b) : vload<T, VEC>(cz + b);
      if (ASYM) c.czm2 = (NT & 1) ? vload_nt<T, VEC>(cz2 + b) : vload<T, VEC>(cz2 + b);
    }
    if (!ASYM) c.czm2 = c.czm;
    return c;
  }
  c.dg = (NT & 1) ? vload_nt<T, VEC>(diag + b) : vload<T, VEC>(diag + b);
  c.cxm = c.dg;                                             // dense mode carries cx[x] over from the previous step
  c.cxp = (NT & 2) ? vload_nt<T, VEC>(cx + b + sx) : vload<T, VEC>(cx + b + sx);
  c.cym = (NT & 4) ? vload_nt<T, VEC>(cy + b) : vload<T, VEC>(cy + b);
  c.cyp = (NT & 4) ? vload_nt<T, VEC>(cy + b + Nz) : vload<T, VEC>(cy + b + Nz);
  c.czm = (NT & 1) ? vload_nt<T, VEC>(cz + b) : vload<T, VEC>(cz + b);
  if (ASYM) c.czm2 = (NT & 1) ? vload_nt<T, VEC>(cz2 + b) : vload<T, VEC>(cz2 + b);
  else c.czm2 = c.czm;
  return c;
}

// class of every z-vector from the dense coefficient arrays (once per solve)
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_pcg_classify(const T* __restrict__ diag, const T* __restrict__ cx, const T* __restrict__ cy,
               const T* __restrict__ cz, int Nx, int Ny, int Nz, unsigned char* __restrict__ cls,
               const T* __restrict__ cz2) {   // cz2: the asymmetric operator's -z weights, or null
  const int nzv = Nz / VEC;
  const int64_t nvec = (int64_t)Nx * Ny * nzv;
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (iv >= nvec) return;
  const int zv = (int)(iv % nzv), y = (int)((iv / nzv) % Ny), x = (int)(iv / ((int64_t)nzv * Ny));
  const int64_t sx = (int64_t)Ny * Nz, b = iv * VEC;
  bool all_zero = true, all_reg = true;
  if (x > 0 && x < Nx - 1 && y > 0 && y < Ny - 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int z = zv * VEC + j;
      if (z == 0 || z == Nz - 1) continue;                  // boundary cells are never computed: don't care
      const int64_t c = b + j;
      // symmetric operator: the -z weight is cz[c]; asymmetric: cz2[c] (cz[c] is then only cell c-1's +z weight)
      const T k[7] = {diag[c], cx[c], cx[c + sx], cy[c], cy[c + Nz], cz2 ? cz2[c] : cz[c], cz[c + 1]};
      bool z0 = true, r1 = k[0] == (T)6;
#pragma unroll
      for (int q = 0; q < 7; ++q) { z0 = z0 && k[q] == (T)0; if (q) r1 = r1 && k[q] == (T)1; }
      // Asymmetric operator: cz[c] of the vector's FIRST cell is no coefficient of this vector at all -- it is the +z
      // weight of the last cell of the vector to the left, which takes it from this lane's registers (czr in the
      // march).  The class constant must therefore stand for it too.  (Symmetric operator: it is k[5], covered.
      // Found by the 20^3 density golden: a REGULAR vector left of a ZERO one got czr = 0.)
      if (cz2 && j == 0) { z0 = z0 && cz[c] == (T)0; r1 = r1 && cz[c] == (T)1; }
      all_zero = all_zero && z0;
      all_reg = all_reg && r1;
    }
  }
  cls[iv] = all_zero ? kClsZero : (all_reg ? kClsRegular : kClsMixed);
}

// ------------------------------------------------------------- variant 1/2 --
// x-marching.  LDS=true stages the current plane tile (+ one row of halo each
// side) in LDS; LDS=false reads the y neighbours straight from global/L1.
//
// Tile: 256 consecutive z-vectors of the flattened interior (y,z) plane -- because
// a row is contiguous and interior rows are adjacent, that is ONE contiguous
// memory segment [m0, m0 + 256*VEC) of each plane, its y-1 / y+1 neighbours are
// the same segment shifted by -/+ Nz, and the LDS image is simply
//   [ Nz halo | 256*VEC tile | Nz halo ]   elements of T.
// Where the operand vector comes from.  Plain: v itself.  FUSE: the CG's direction update folded into
// the stencil -- v = d_new = r + beta * d_old computed on the fly for every vector the march touches
// (own vectors, halo rows, the x-1 / x+1 planes of the prologue), with exactly the arithmetic of
// k_update_d, and written back once per own vector to the OTHER d buffer (neighbouring workgroups
// still read d_old's halos, hence the ping-pong).  Saves the separate 3-scalar pass over r and d.
template <typename T, int VEC>
struct RawVec { vec_t<T, VEC> a, b; };        // as fetched: v (plain)  or  r and d_old (fused)

template <typename T, int VEC, bool FUSE>
struct VSrc {
  const T* __restrict__ v;
  const T* __restrict__ r;
  const T* __restrict__ d_old;
  double beta;
  // issue the loads only; nothing here depends on their data, so they stay in flight
  __device__ __forceinline__ RawVec<T, VEC> raw(int64_t off) const {
    RawVec<T, VEC> w;
    if (!FUSE) { w.a = vload<T, VEC>(v + off); w.b = w.a; }
    else { w.a = vload<T, VEC>(r + off); w.b = vload<T, VEC>(d_old + off); }
    return w;
  }
  // consume: the operand vector (k_update_d's arithmetic when fused)
  __device__ __forceinline__ vec_t<T, VEC> fin(const RawVec<T, VEC>& w) const {
    if (!FUSE) return w.a;
    vec_t<T, VEC> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = (T)((double)w.a[j] + beta * (double)w.b[j]);
    return o;
  }
  __device__ __forceinline__ vec_t<T, VEC> ld(int64_t off) const { return fin(raw(off)); }
};

// PD = prefetch depth in planes for the operand vector and its halo rows (the long-latency streams):
// their loads for plane x+1+PD are issued at step x and consumed PD steps later, which keeps PD
// planes of v per wave in flight -- the march is latency-paced, so bytes in flight are what set its speed.
// XDEF (with FUSE): the PREVIOUS iteration's solution update x += alpha d_old rides in this launch too -- each
// vector is updated by the step that owns it, d_old re-read from cache -- so that the x/r update kernel only
// touches r and q.  The march is latency-paced, extra streams are nearly free for it; same values as k_update_xr's.
// BOOK (with FUSE, small grids): the launch also CLOSES the previous iteration -- every workgroup folds the r.r partials
// of the preceding x/r update itself (block_total_of: same order, same value in every workgroup), takes the convergence
// decision and beta from that, and workgroup 0 writes the bookkeeping (cg_book).  The update kernel then needs no
// in-launch reduction tail (store -> ticket -> ticket -> acquire -> partials -> scalars: four dependent round trips that are
// most of a 184 k-cell iteration).  The workgroups agree by construction: the decision is a function of values that
// were complete before the launch.
struct BookArgs {
  double* scal;
  double* hist;
  int64_t hist_cap;
  const double* part_rr;
  int npart, par;
};

template <typename T, int VEC, bool LDS, int NT, bool COMP, bool FUSE, int PD, bool ASYM = false, bool XDEF = false,
          bool BOOK = false, bool LMASK = false>
__global__ void __launch_bounds__(kApplyBlock, MFS_MARCH_MIN_WAVES)
k_pcg_apply_march(const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ diag,
                  const T* __restrict__ cx, const T* __restrict__ cy, const T* __restrict__ cz,
                  const unsigned char* __restrict__ cls, ApplyArgs a, double* __restrict__ partial,
                  const double* __restrict__ done_flag, const T* __restrict__ fr, const T* __restrict__ fd_old,
                  T* __restrict__ fd_new, const double* __restrict__ beta_ptr, const T* __restrict__ cz2,
                  T* __restrict__ xdef = nullptr, const double* __restrict__ alpha_ptr = nullptr, BookArgs bk = BookArgs{}) {
  static_assert(!XDEF || FUSE, "the deferred x update rides on the fused direction update");
  static_assert(!LMASK || (COMP && FUSE && PD == 1), "lane-level masking: compressed, fused, prefetch depth 1");
  static_assert(!BOOK || FUSE, "closing the previous iteration rides on the fused direction update");
  static_assert(kApplyBlock == kBlock, "block_total_of is written for kBlock threads");
  double alpha_x = (XDEF && !BOOK) ? *alpha_ptr : 0.0;
  static_assert(!FUSE || LDS, "the fused direction update is implemented on the LDS march");
  // BOOK: the flag, the scalars and this lane's share of the r.r partials are REQUESTED here and consumed after the first
  // march's own loads have been issued (book_pending below): one memory round trip instead of three in sequence.
  double bk_dn = 0.0, bk_delta = 0.0, bk_dq = 0.0, bk_tol2 = 0.0, bk_acc = 0.0;
  bool book_pending = BOOK;
  double beta_v = 0.0;
  if (BOOK) {
    bk_dn = bk.scal[S_DONE]; bk_delta = bk.scal[S_RING + bk.par]; bk_dq = bk.scal[S_DQ]; bk_tol2 = bk.scal[S_TOL2];
    for (int i = threadIdx.x; i < bk.npart; i += kBlock) bk_acc += bk.part_rr[i];      // block_total_of's order
  } else {
    if (done_flag && *done_flag != 0.0) return;
    if (FUSE) beta_v = *beta_ptr;
  }
  VSrc<T, VEC, FUSE> src{v, fr, fd_old, beta_v};
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* const smem = reinterpret_cast<T*>(smem_raw);
  const int Nz = a.Nz;
  const int nzv = Nz / VEC;
  const int ipp = (a.Ny - 2) * nzv;                         // interior z-vectors per plane
  const int tiles = (ipp + kApplyBlock - 1) / kApplyBlock;
  const int n1 = a.xe - a.xb;
  const int np = n1 + (a.xe2 - a.xb2);
  // Work = the sequence of (tile, plane) pairs, tile-major.  It is cut into gridDim
  // equal contiguous segments (+-1 pair): every workgroup marches the same number of
  // planes, so all CUs finish together whatever the grid shape; a segment that runs
  // off the end of its tile's x range simply restarts the pipeline on the next tile.
  // `xchunk` > 0 additionally caps the length of one march.
  const bool sparse = a.items != nullptr;
  const int64_t total = sparse ? (int64_t)*a.count : (int64_t)tiles * np;

  const int G = gridDim.x;
  const int nch = min(G, kXcds);
  const int xcd = blockIdx.x % nch, slot = blockIdx.x / nch;
  // logical segment id: blocks of one XCD (equal blockIdx % 8) get adjacent segments
  const int per = G / nch, extra = G - per * nch;           // first `extra` labels own one more block
  const int seg = xcd * per + min(xcd, extra) + slot;
  const int64_t s0 = total * seg / G, s1 = total * (seg + 1) / G;
  const int64_t sx = (int64_t)a.Ny * Nz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile_elems = kApplyBlock * VEC;
  const int buf_elems = tile_elems + 2 * Nz;                // one LDS plane image
  double acc = 0.0;

  // BOOK: closing the previous iteration -- at the workgroup's first march (its loads in flight), or after the loop for a
  // workgroup that has no march at all (sparse work lists can be shorter than the grid)
  auto close_previous = [&]() -> bool {
    book_pending = false;
    __shared__ double s_rr;
    const double t = block_sum<kBlock>(bk_acc);
    if (threadIdx.x == 0) s_rr = t;
    __syncthreads();
    const double rr = s_rr;                        // == block_total_of(bk.part_rr, bk.npart)
    // the flag may have been raised by an EARLIER launch, or by workgroup 0 of this one for the very decision every
    // workgroup takes here: same outcome
    const bool stop = bk_dn != 0.0 || cg_health(bk_dq, rr) != 0 || rr < bk_tol2;
    if (bk_dn == 0.0 && blockIdx.x == 0 && threadIdx.x == 0) cg_book(bk.scal, bk.hist, bk.hist_cap, bk.par, bk_dq, rr);
    src.beta = rr / bk_delta;                      // cg_book's expressions: the values it leaves in S_BETA
    if (XDEF) alpha_x = bk_delta / bk_dq;          // and S_ALPHA (the step of the iteration being closed)
    return stop;
  };

  for (int64_t i = s0; i < s1;) {
    const int64_t it_ = sparse ? (int64_t)a.items[i] : i;
    const int tile = (int)(it_ / np);
    const int pl = (int)(it_ - (int64_t)tile * np);        // plane slot inside the (up to two) ranges
    const int x0 = pl < n1 ? a.xb + pl : a.xb2 + (pl - n1);
    int len = (int)min((int64_t)((pl < n1 ? a.xe : a.xe2) - x0), s1 - i);
    if (sparse) len = min(len, a.runrem[i]);
    if (a.xchunk > 0) len = min(len, a.xchunk);
    const int x1 = x0 + len;
    i += len;
    const int item_raw = tile * kApplyBlock + tid;
    const bool active = item_raw < ipp;
    const int item = active ? item_raw : ipp - 1;           // clamp: inactive lanes load valid addresses
    const int yy = item / nzv, zv = item - yy * nzv;
    const bool first = zv == 0, last = zv == nzv - 1;
    const int64_t m = (int64_t)(yy + 1) * Nz + (int64_t)zv * VEC;   // offset of this vector inside a plane
    // halo ownership: the tile's lower halo is the Nz elements before its first
    // vector, the upper halo the Nz elements after its last one.
    const int64_t m0 = (int64_t)Nz + (int64_t)tile * tile_elems;    // plane offset of the tile's first vector
    const int tile_len = min(tile_elems, ipp * VEC - tile * tile_elems);  // elements really in this tile

    int64_t base = (int64_t)x0 * sx + m;
    const RawVec<T, VEC> wm = src.raw(base - sx), wc = src.raw(base), wp = src.raw(base + sx);
    // classes of this vector in planes x0 and x0+1 (compressed mode), then plane x0's coefficients
    // MASK (template flag LMASK; sparse launches of the fused PD-1 forms when the solve's dead fraction makes it pay -- it costs
    // ~1 % where every lane of a listed pair is live, the pool scene of the bench, and saves 10 % of the launch where half
    // of them are dead, the 256^3 time step): a DEAD vector (kClsDead: ZERO row, r = d = 0 when the solve's lists
    // were built) moves nothing in the steady state -- its r, d_old, x loads and its q, d_new, x stores are exec-masked
    // (zeros stand for the operands: exactly what the loads would return; q and the partner d buffer hold +0 there since
    // the solve began).  A liquid body covers PART of a tile's rows: in the 256^3 time step half the lanes of a listed
    // pair are air.  The class of plane x+2 must be known when its operand load is issued: classes run three planes ahead.
    // The prologue's own three planes and the halo rows are fetched unmasked.
    constexpr bool MASK = LMASK;       // (a template flag, chosen per launch on the host: as a run-time flag it cost the pool scene 2 %)
    const bool mask_on = MASK && sparse;
    unsigned char cls_n = kClsMixed, cls_c = kClsMixed, cls_n2 = kClsMixed;
    if (COMP) {
      cls_c = cls[base / VEC];
      cls_n = cls[(base + (x0 + 1 < x1 ? sx : 0)) / VEC];
      if (MASK) cls_n2 = cls[((int64_t)min(x0 + 2, x1) * sx + m) / VEC];
    }
    // this lane's first halo vector of plane x0 (requested with the rest of the batch, consumed when the image is staged)
    RawVec<T, VEC> wh = {};
    const bool h0 = LDS && tid < 2 * nzv;
    if (h0) wh = src.raw((int64_t)x0 * sx + (tid < nzv ? m0 - Nz + tid * VEC : m0 + tile_len + (tid - nzv) * VEC));
    if (BOOK && book_pending) {                      // uniform over the workgroup: its first march
      if (close_previous()) return;
    }
    vec_t<T, VEC> vm = src.fin(wm), vc = src.fin(wc), vp = src.fin(wp);
    CoefVec<T, VEC> cc = coef_load<T, VEC, COMP, NT, true, ASYM>(diag, cx, cy, cz, base, sx, Nz, cls_c, cz2);
    if (!COMP) cc.cxm = vload<T, VEC>(cx + base);

    if (LDS) {
      // stage plane x0 (tile + halos) into buffer 0
      T* b0 = smem;
      if (active) vstore<T, VEC>(b0 + Nz + tid * VEC, vc);   // inactive lanes sit past tile_len = in the halo
      if (h0) vstore<T, VEC>(b0 + (tid < nzv ? tid * VEC : Nz + tile_len + (tid - nzv) * VEC), src.fin(wh));
      for (int s = tid + kApplyBlock; s < 2 * nzv; s += kApplyBlock) {
        const bool lo = s < nzv;
        const int j = lo ? s : s - nzv;
        vstore<T, VEC>(b0 + (lo ? j * VEC : Nz + tile_len + j * VEC),
                       src.ld((int64_t)x0 * sx + (lo ? m0 - Nz + j * VEC : m0 + tile_len + j * VEC)));
      }
      MFS_LDS_BARRIER();
    }
    // The two halo rows of a plane (Nz elements below the tile, Nz above) are 2 * nzv vectors: thread t
    // owns halo vector t (low row first, then the high row), so the halo prefetch costs ONE vector of
    // registers per plane in flight; rows with more than 256 halo vectors take the slow path below.
    const bool hact = LDS && tid < 2 * nzv;
    const bool hlow = tid < nzv;
    const int64_t hg = hlow ? m0 - Nz + (int64_t)tid * VEC : m0 + tile_len + (int64_t)(tid - nzv) * VEC;   // in-plane offset
    const int hl = hlow ? tid * VEC : Nz + tile_len + (tid - nzv) * VEC;                                     // LDS image offset
    // in flight from here on: the operand vector of planes x0+2 .. x0+PD and the halo rows of planes
    // x0+1 .. x0+PD (raw; consumed PD steps after issue)
    RawVec<T, VEC> Q[PD > 1 ? PD - 1 : 1], H[PD];
#pragma unroll
    for (int k = 0; k < PD - 1; ++k) Q[k] = src.raw((int64_t)min(x0 + 2 + k, x1) * sx + m);
#pragma unroll
    for (int k = 0; k < PD; ++k) {
      H[k] = RawVec<T, VEC>{};
      if (hact) H[k] = src.raw((int64_t)min(x0 + 1 + k, x1) * sx + hg);
    }

    for (int x = x0; x < x1; ++x) {
      const bool more = x + 1 < x1;
      const int cur = (x - x0) & 1;
      // ---- prefetch everything step x+1 needs (plane x+2 of v and cx; plane x+1 of the rest)
      const int64_t nb = base + sx;                         // this vector in plane x+1
      // The last step re-requests its own plane (addresses stay valid, values unused):
      // unconditional loads keep the prefetch free of control flow, so the compiler
      // waits for them where they are consumed (the rotation below), not here.
      const int64_t nn = more ? nb : base;                  // plane x+1 (or x again on the last step)
      const int64_t n2 = more ? nb + sx : nb;               // plane x+2 (or x+1 again)
      const bool dead_c = mask_on && cc.cls == kClsDead;    // this step's own vector
      RawVec<T, VEC> qn = {};                               // operand vector, plane x+1+PD (MASK: not for a dead vector)
      if (!(mask_on && cls_n2 == kClsDead)) qn = src.raw((int64_t)min(x + 1 + PD, x1) * sx + m);
      vec_t<T, VEC> xo = {}, dxo = {};                      // XDEF: this step's own x vector and d_old (cache hit)
      if (XDEF && !dead_c) { xo = vload_nt<T, VEC>(xdef + base); dxo = vload<T, VEC>(fd_old + base); }
      // plane x+1's coefficients (class known since the previous step); class of plane x+2
      CoefVec<T, VEC> cn = coef_load<T, VEC, COMP, NT, false, ASYM>(diag, cx, cy, cz, nn, sx, Nz, more ? cls_n : cc.cls, cz2);
      cn.cxm = cc.cxp;                                      // cx[x+1] was this step's upper-face weight
      unsigned char cls_nn = kClsMixed;                     // plain: class of plane x+2; MASK: of plane x+3 (x+2's is cls_n2)
      if (COMP) cls_nn = MASK ? cls[((int64_t)min(x + 3, x1) * sx + m) / VEC] : cls[n2 / VEC];
      RawVec<T, VEC> hn = {};                               // this thread's halo vector of plane x+1+PD
      if (hact) hn = src.raw((int64_t)min(x + 1 + PD, x1) * sx + hg);
      // ---- in-plane neighbours of plane x
      vec_t<T, VEC> vym, vyp;
      double zl, zr;
      if (LDS) {
        const T* bc = smem + cur * buf_elems;
        vym = vload<T, VEC>(bc + tid * VEC);                 // (tile offset + Nz) - Nz
        vyp = vload<T, VEC>(bc + 2 * Nz + tid * VEC);
        zl = (double)bc[Nz + tid * VEC - 1];
        zr = (double)bc[Nz + tid * VEC + VEC];
      } else {
        vym = vload<T, VEC>(v + base - Nz);
        vyp = vload<T, VEC>(v + base + Nz);
        const T l = __shfl_up(vc[VEC - 1], 1, 64), r = __shfl_down(vc[0], 1, 64);
        zl = (double)((lane == 0 && !first) ? v[base - 1] : l);
        zr = (double)((lane == 63 && !last) ? v[base + VEC] : r);
      }
      const T czs = __shfl_down(cc.czm[0], 1, 64);
      const double czr = (double)((lane == 63 && !last) ? cz[base + VEC] : czs);
      stencil_vec<T, VEC>(out + base, vc, vp, vm, vyp, vym, cc.dg, cc.cxp, cc.cxm, cc.cyp, cc.cym, cc.czm, zl, zr, czr,
                          first, last, active && !dead_c, acc, cc.czm2);
      if (FUSE && active && !dead_c) vstore<T, VEC>(fd_new + base, vc);   // d_new of this vector (its z-boundary cells are 0 + beta*0)
      if (XDEF && active && !dead_c) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) xo[j] = (T)((double)xo[j] + alpha_x * (double)dxo[j]);
        vstore_nt<T, VEC>(xdef + base, xo);
      }
      // ---- rotate; publish plane x+1 to the other LDS buffer
      if (more) {
        if (LDS) {
          T* bn = smem + (cur ^ 1) * buf_elems;
          if (active) vstore<T, VEC>(bn + Nz + tid * VEC, vp);
          if (hact) vstore<T, VEC>(bn + hl, src.fin(H[0]));
          for (int s = tid + kApplyBlock; s < 2 * nzv; s += kApplyBlock) {   // more halo vectors than threads
            const bool lo = s < nzv;
            const int j = lo ? s : s - nzv;
            vstore<T, VEC>(bn + (lo ? j * VEC : Nz + tile_len + j * VEC),
                           src.ld((int64_t)(x + 1) * sx + (lo ? m0 - Nz + j * VEC : m0 + tile_len + j * VEC)));
          }
          MFS_LDS_BARRIER();
        }
        vm = vc; vc = vp; vp = src.fin(PD == 1 ? qn : Q[0]); cc = cn;
        if (MASK) { cls_n = cls_n2; cls_n2 = cls_nn; } else cls_n = cls_nn;
#pragma unroll
        for (int k = 0; k + 1 < PD - 1; ++k) Q[k] = Q[k + 1];
        if (PD > 1) Q[PD - 2] = qn;
#pragma unroll
        for (int k = 0; k + 1 < PD; ++k) H[k] = H[k + 1];
        H[PD - 1] = hn;
        base = nb;
      }
    }
    if (LDS) MFS_LDS_BARRIER();  // next work unit reuses buffer 0
  }
  if (BOOK && book_pending) {                        // no march in this workgroup: the iteration is closed all the same
    if (close_previous()) return;
  }
  const double tot = block_sum<kApplyBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

}  // namespace mfs
