// mfs_pcg_slab.h -- kernels of the slab-decomposed (multi-GPU) pressure CG loop that talk
// to the neighbours' windows directly (mfs_p2p.h).  One iteration j on every rank, all on
// ONE stream, no host synchronisation and no library collective inside the loop:
//
//   k_slab_edge_d       d_j = r + beta d_{j-1} on the two edge planes {1, L-2}; each plane
//                       is stored locally AND, as tagged granules, into the neighbour's window
//   <stencil launch>    planes [2, L-2): the single-GPU LDS march with the direction update
//                       folded in (mfs_pcg_apply.h) -- runs while the edge planes travel
//   k_slab_edge_apply   q on the two edge planes, ghost plane read from the own window
//                       (granules re-read until they carry this iteration's tag); its LAST block
//                       adds up the d.q partials, sends the sum into every rank's window and
//                       adds the world's slots in rank order -> d.q
//   k_update_xr         alpha, x += alpha d, r -= alpha q (mfs_cg_core.h); its LAST block does the
//                       same for r.r, then the convergence test / history / beta bookkeeping
//
// Four launches per iteration (one GPU: two).
//
// The arithmetic of every cell is the single-GPU kernels' (same stencil_vec, same update
// expressions), so a slab solve differs from the single-domain one only in the order in
// which the dot products' partial sums are added.
#pragma once
#include "mfs_cg_core.h"
#include "mfs_p2p.h"
#include "mfs_pcg_apply.h"

namespace mfs {

// the (one or two) edge planes of a slab of L local planes and who receives them
struct SlabEdge {
  int np;
  int plane[2];
  int to_left[2], to_right[2];   // plane k is the left / right neighbour's ghost
};

static inline SlabEdge slab_edges(int L, int rank, int world) {
  SlabEdge e{};
  const int lo = 1, hi = L - 2;
  if (hi < lo) return e;
  e.np = hi > lo ? 2 : 1;
  e.plane[0] = lo; e.plane[1] = hi;
  const bool left = rank > 0, right = rank < world - 1;
  e.to_left[0] = left; e.to_right[0] = (hi == lo) && right;
  e.to_left[1] = 0;    e.to_right[1] = (hi > lo) && right;
  return e;
}

// d_new = r + beta d_old on the edge planes (exactly k_update_d's expression); FIRST: iteration 0,
// the direction vector is d_old itself and nothing is written locally.
// xdef != null (never with FIRST): the deferred solution update x += alpha d_old of the previous iteration on
// the edge planes too (the interior launch does it for its own planes).
template <typename T, int VEC, bool FIRST>
__global__ void __launch_bounds__(kBlock)
k_slab_edge_d(const T* __restrict__ r, const T* __restrict__ d_old, T* __restrict__ d_new, int64_t plane_elems,
              SlabEdge e, const double* __restrict__ scal, P2pDev pd, int par, unsigned tag, T* __restrict__ xdef) {
  if (scal[S_DONE] != 0.0) return;
  const double beta = FIRST ? 0.0 : scal[S_BETA];
  const double alpha_x = (!FIRST && xdef) ? scal[S_ALPHA] : 0.0;
  const int64_t nv = plane_elems / VEC, stride = (int64_t)gridDim.x * blockDim.x;
  for (int k = 0; k < e.np; ++k) {
    const int64_t p0 = (int64_t)e.plane[k] * plane_elems;
    u64* const left = e.to_left[k] ? pd.send[0][par] : nullptr;
    u64* const right = e.to_right[k] ? pd.send[1][par] : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
      vec_t<T, VEC> dv = vload<T, VEC>(d_old + p0 + i * VEC);
      if (!FIRST) {
        if (xdef) {
          vec_t<T, VEC> xv = vload<T, VEC>(xdef + p0 + i * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) xv[j] = (T)((double)xv[j] + alpha_x * (double)dv[j]);
          vstore<T, VEC>(xdef + p0 + i * VEC, xv);
        }
        const vec_t<T, VEC> rv = vload<T, VEC>(r + p0 + i * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dv[j] = (T)((double)rv[j] + beta * (double)dv[j]);
        vstore<T, VEC>(d_new + p0 + i * VEC, dv);
      }
      if (left) gran_store_vec<T, VEC>(left, i * VEC, dv, tag);
      if (right) gran_store_vec<T, VEC>(right, i * VEC, dv, tag);
    }
  }
}

// q = A d on the edge planes.  Direct loads (two planes of work: the march's register reuse has
// nothing to amortise); the x-1 / x+1 operand of a plane next to a neighbour comes from the window,
// re-read until its granules carry this iteration's tag.
template <typename T, int VEC>
__global__ void __launch_bounds__(kApplyBlock)
k_slab_edge_apply(const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ diag, const T* __restrict__ cx,
                  const T* __restrict__ cy, const T* __restrict__ cz, const T* __restrict__ czm2, int L, int Ny, int Nz,
                  SlabEdge e,   // czm2: weights of the -z tap (= cz except for the density operator, DensityCGSolver3D.py:184)
                  double* __restrict__ partial_all, int n_before, double* __restrict__ scal, P2pDev pd, int par,
                  unsigned tag, unsigned* ticket, int ar_ring, unsigned ar_tag) {
  if (scal[S_DONE] != 0.0) return;
  const bool has_left = pd.rank > 0, has_right = pd.rank < pd.world - 1;
  const u64* const ghost_lo = pd.recv[0][par];
  const u64* const ghost_hi = pd.recv[1][par];
  const int nzv = Nz / VEC;
  const int64_t ipp = (int64_t)(Ny - 2) * nzv;
  const int64_t sx = (int64_t)Ny * Nz, sy = Nz;
  const int64_t items = (int64_t)e.np * ipp, stride = (int64_t)gridDim.x * kApplyBlock;
  double acc = 0.0;
  bool lost = false;
  for (int64_t it = (int64_t)blockIdx.x * kApplyBlock + threadIdx.x; it < items; it += stride) {
    const int px = (int)(it / ipp);
    const int rem = (int)(it - (int64_t)px * ipp);
    const int yy = rem / nzv, zv = rem - yy * nzv;
    const int xx = e.plane[px];
    const int64_t in_plane = (int64_t)(yy + 1) * sy + (int64_t)zv * VEC;
    const int64_t base = (int64_t)xx * sx + in_plane;
    const bool first = zv == 0, last = zv == nzv - 1;
    const auto vc = vload<T, VEC>(v + base);
    vec_t<T, VEC> vxm, vxp;
    if (xx == 1 && has_left) lost = lost || !gran_load_vec<T, VEC>(ghost_lo, in_plane, tag, pd.timeout_ticks, &vxm);
    else vxm = vload<T, VEC>(v + base - sx);
    if (xx == L - 2 && has_right) lost = lost || !gran_load_vec<T, VEC>(ghost_hi, in_plane, tag, pd.timeout_ticks, &vxp);
    else vxp = vload<T, VEC>(v + base + sx);
    if (lost) break;
    const double zl = first ? 0.0 : (double)v[base - 1];
    const double zr = last ? 0.0 : (double)v[base + VEC];
    const double czr = last ? 0.0 : (double)cz[base + VEC];
    stencil_vec<T, VEC>(out + base, vc, vxp, vxm, vload<T, VEC>(v + base + sy), vload<T, VEC>(v + base - sy),
                        vload<T, VEC>(diag + base), vload<T, VEC>(cx + base + sx), vload<T, VEC>(cx + base),
                        vload<T, VEC>(cy + base + sy), vload<T, VEC>(cy + base), vload<T, VEC>(cz + base), zl, zr, czr,
                        first, last, true, acc, vload<T, VEC>(czm2 + base));
  }
  if (lost) slab_fail(scal, 2);
  const double tot = block_sum<kApplyBlock>(acc);
  // the last block to get here adds up ALL of this iteration's d.q partials (the interior launch's
  // [0, n_before), then this launch's), sends the sum to every rank and takes the world's total
  double dq;
  if (!last_block_total(partial_all, n_before + blockIdx.x, tot, n_before + gridDim.x, ticket, gridDim.x, &dq)) return;
  if (threadIdx.x >= kWave) return;
  bool ok;
  dq = slab_allreduce_wave(pd, ar_ring, ar_tag, dq, &ok);
  if (threadIdx.x != 0) return;
  if (!ok) { slab_fail(scal, 1); return; }
  scal[S_DQ] = dq;
}

// ONE block: this rank's partial sums -> every rank's window -> the world's total in rank order ->
// scal[S_RR].  Used by `begin` (delta0); inside the loop the reductions ride in the last block of
// k_slab_edge_apply (d.q) and of k_update_xr (r.r + bookkeeping).
static __global__ void __launch_bounds__(kBlock)
k_slab_allreduce_rr(const double* __restrict__ partial, int count, double* __restrict__ scal, P2pDev pd, int ring,
                    unsigned tag, int which = S_RR) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
  const double loc = block_sum<kBlock>(acc);          // thread 0
  __shared__ double s_loc;
  if (threadIdx.x == 0) s_loc = loc;
  __syncthreads();
  if (threadIdx.x >= kWave) return;
  bool ok;
  const double tot = slab_allreduce_wave(pd, ring, tag, s_loc, &ok);
  if (threadIdx.x != 0) return;
  if (!ok) { slab_fail(scal, 1); return; }
  scal[which] = tot;
}

}  // namespace mfs
