// mfs_pcg_apply.h -- the per-iteration stencil kernel of the pressure CG (gfx950).
//
//   out = A v,  A = the reference's 7-point ghost-fluid operator
//   (solver/PressureCGSolver3D.py:52-130) in precomputed-coefficient form:
//   out[c] = diag[c] v[c] - cx[x+1] v[x+1] - cx[x] v[x-1] - cy[y+1] v[y+1]
//            - cy[y] v[y-1] - cz[z+1] v[z+1] - cz[z] v[z-1]
//   accumulated in the reference's order (+x -x +y -y +z -z, then diag), in fp64.
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
                                            double zr, double czr, bool first, bool last, bool active, double& acc) {
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
    val -= (C)czm[j] * zm;
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
struct ApplyArgs {
  int Nx, Ny, Nz, xb, xe, xchunk, xb2, xe2;
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
                        vload<T, VEC>(cy + base), vload<T, VEC>(cz + base), zl, zr, czr, first, last, true, acc);
  }
  const double tot = block_sum<kApplyBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
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
template <typename T, int VEC, bool LDS, int NT>
__global__ void __launch_bounds__(kApplyBlock)
k_pcg_apply_march(const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ diag,
                  const T* __restrict__ cx, const T* __restrict__ cy, const T* __restrict__ cz, ApplyArgs a,
                  double* __restrict__ partial, const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
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
  const int64_t total = (int64_t)tiles * np;
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

  for (int64_t i = s0; i < s1;) {
    const int tile = (int)(i / np);
    const int pl = (int)(i - (int64_t)tile * np);          // plane slot inside the (up to two) ranges
    const int x0 = pl < n1 ? a.xb + pl : a.xb2 + (pl - n1);
    int len = (int)min((int64_t)((pl < n1 ? a.xe : a.xe2) - x0), s1 - i);
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
    vec_t<T, VEC> vm = vload<T, VEC>(v + base - sx), vc = vload<T, VEC>(v + base), vp = vload<T, VEC>(v + base + sx);
    vec_t<T, VEC> cxm = vload<T, VEC>(cx + base);
    vec_t<T, VEC> cxp = (NT & 2) ? vload_nt<T, VEC>(cx + base + sx) : vload<T, VEC>(cx + base + sx);
    vec_t<T, VEC> dg = (NT & 1) ? vload_nt<T, VEC>(diag + base) : vload<T, VEC>(diag + base);
    vec_t<T, VEC> cym = (NT & 4) ? vload_nt<T, VEC>(cy + base) : vload<T, VEC>(cy + base);
    vec_t<T, VEC> cyp = (NT & 4) ? vload_nt<T, VEC>(cy + base + Nz) : vload<T, VEC>(cy + base + Nz);
    vec_t<T, VEC> czm = (NT & 1) ? vload_nt<T, VEC>(cz + base) : vload<T, VEC>(cz + base);

    if (LDS) {
      // stage plane x0 (tile + halos) into buffer 0
      T* b0 = smem;
      if (active) vstore<T, VEC>(b0 + Nz + tid * VEC, vc);   // inactive lanes sit past tile_len = in the halo
      for (int h = tid * VEC; h < Nz; h += kApplyBlock * VEC) {
        vstore<T, VEC>(b0 + h, vload<T, VEC>(v + (int64_t)x0 * sx + m0 - Nz + h));
        vstore<T, VEC>(b0 + Nz + tile_len + h, vload<T, VEC>(v + (int64_t)x0 * sx + m0 + tile_len + h));
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // halo vectors of plane x0+1 (published to LDS at the end of step x0)
    const int hofs = tid * VEC;                             // halo slot owned by this thread (if < Nz)
    vec_t<T, VEC> hlo = {}, hhi = {};
    if (LDS && hofs < Nz) {
      const int64_t hp = (int64_t)min(x0 + 1, x1) * sx + m0;
      hlo = vload<T, VEC>(v + hp - Nz + hofs);
      hhi = vload<T, VEC>(v + hp + tile_len + hofs);
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
      const vec_t<T, VEC> vpp = vload<T, VEC>(v + n2);
      const vec_t<T, VEC> cxpp = (NT & 2) ? vload_nt<T, VEC>(cx + n2) : vload<T, VEC>(cx + n2);
      const vec_t<T, VEC> dg_n = (NT & 1) ? vload_nt<T, VEC>(diag + nn) : vload<T, VEC>(diag + nn);
      const vec_t<T, VEC> cym_n = (NT & 4) ? vload_nt<T, VEC>(cy + nn) : vload<T, VEC>(cy + nn);
      const vec_t<T, VEC> cyp_n = (NT & 4) ? vload_nt<T, VEC>(cy + nn + Nz) : vload<T, VEC>(cy + nn + Nz);
      const vec_t<T, VEC> czm_n = (NT & 1) ? vload_nt<T, VEC>(cz + nn) : vload<T, VEC>(cz + nn);
      vec_t<T, VEC> hlo_n = {}, hhi_n = {};                           // halos of plane x+2, published one step later
      if (LDS && hofs < Nz) {
        const int64_t hp = (int64_t)(more ? x + 2 : x + 1) * sx + m0;
        hlo_n = vload<T, VEC>(v + hp - Nz + hofs);
        hhi_n = vload<T, VEC>(v + hp + tile_len + hofs);
      }
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
      const T czs = __shfl_down(czm[0], 1, 64);
      const double czr = (double)((lane == 63 && !last) ? cz[base + VEC] : czs);
      stencil_vec<T, VEC>(out + base, vc, vp, vm, vyp, vym, dg, cxp, cxm, cyp, cym, czm, zl, zr, czr, first, last,
                          active, acc);
      // ---- rotate; publish plane x+1 to the other LDS buffer
      if (more) {
        if (LDS) {
          T* bn = smem + (cur ^ 1) * buf_elems;
          if (active) vstore<T, VEC>(bn + Nz + tid * VEC, vp);
          if (hofs < Nz) {
            vstore<T, VEC>(bn + hofs, hlo);
            vstore<T, VEC>(bn + Nz + tile_len + hofs, hhi);
          }
          for (int h = hofs + kApplyBlock * VEC; h < Nz; h += kApplyBlock * VEC) {   // rows longer than one tile
            vstore<T, VEC>(bn + h, vload<T, VEC>(v + (int64_t)(x + 1) * sx + m0 - Nz + h));
            vstore<T, VEC>(bn + Nz + tile_len + h, vload<T, VEC>(v + (int64_t)(x + 1) * sx + m0 + tile_len + h));
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        vm = vc; vc = vp; vp = vpp; cxm = cxp; cxp = cxpp; dg = dg_n; cym = cym_n; cyp = cyp_n; czm = czm_n;
        hlo = hlo_n; hhi = hhi_n;
        base = nb;
      }
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // next work unit reuses buffer 0
  }
  const double tot = block_sum<kApplyBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

}  // namespace mfs
