// mfs_vcg_march.h -- the per-iteration kernel of the viscosity CG on the pressure kernel's recipe (gfx950).
//
//   q = A d,  A = the reference's variational-viscosity operator, three coupled face components
//   (solver/ViscosityCGSolver3D.py:248-456), rows evaluated by vcg_row_s (mfs_visc.hip) -- the same code,
//   tap order and arithmetic as every other form of the operator in this library, so results are bit-identical.
//
// Algorithmic HBM traffic per cell: d_x, d_y, d_z read, q_x, q_y, q_z written, 7 volume classes read = 13 scalars,
// + 3 mask bytes.  What this kernel is about is moving exactly those bytes, once, in 16-byte pieces:
//
//  * one work item = one 16-byte z-vector (4 fp32 / 2 fp64 cells) of an interior row y in [1, Ny-2]; all three
//    operator rows (u, v, w faces) of its cells are computed by the same thread (they share most operands).
//  * a workgroup owns a tile of 256 consecutive z-vectors of the flattened interior (y, z) plane and MARCHES ALONG X.
//    The x-1 / x / x+1 planes of the three velocity components live in registers across steps, as do the volume
//    samples that a step shares with the next one (cell centres of x-1, the xy- and xz-edge classes of x+1): x
//    neighbours cost no memory traffic.
//  * in-plane neighbours (y+-1 rows, z+-1 cells) of the three velocity components come from an LDS image of the plane
//    tile plus one row of halo each side, per component, double buffered, ONE barrier per plane that waits on lgkmcnt
//    only, so the global prefetch of the next plane stays in flight across it.  The image mirrors memory: for the u
//    and v components (rows of Nz) every access is an aligned 16-byte one; the w component has rows of Nz+1 elements,
//    which no vector alignment survives -- its global accesses are unaligned 16-byte ones (the hardware splits them)
//    and its LDS accesses scalar.  The step's one look into plane x+1 off its own row (u[x+1,y-1,z], u[x+1,y,z-1]:
//    the cross taps of the v and w rows) is served from the image published in the SAME step, read behind the barrier.
//  * the seven volume classes and the three mask arrays are solver-owned and stored with one uniform, 16-byte aligned
//    pitch (struct Compact): one per-thread offset serves them all, own vectors and the three neighbour rows
//    (C[y-1], EXY[y+1], EYZ[y+1]) are aligned 16-byte loads that hit in L1/L2 behind the owner's.
//  * balanced, XCD-aware work split exactly as in mfs_pcg_apply.h: the (tile, plane) sequence, tile-major, is cut into
//    gridDim equal contiguous segments; blocks with equal blockIdx % 8 (one XCD, one L2) get adjacent segments.
//  * the d.q partial sums ride along; no atomics, bitwise reproducible.
#pragma once

#ifndef MFS_VMARCH_MIN_WAVES
#define MFS_VMARCH_MIN_WAVES 2     // waves per SIMD the kernel is compiled for (256 VGPRs)
#endif

namespace mfs {

constexpr int kVmBlock = 256;

// 16-byte vector whose address is only element-aligned (rows of Nz+1 elements)
template <typename T, int N> struct UVecT { typedef T type __attribute__((ext_vector_type(N), aligned(sizeof(T)))); };
template <typename T> struct UVecT<T, 1> { typedef T type __attribute__((ext_vector_type(1))); };

template <typename T, int VEC>
__device__ __forceinline__ vec_t<T, VEC> vload_u(const T* p) {
  const typename UVecT<T, VEC>::type u = *reinterpret_cast<const typename UVecT<T, VEC>::type*>(p);
  vec_t<T, VEC> o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) o[j] = u[j];
  return o;
}
template <typename T, int VEC>
__device__ __forceinline__ void vstore_u(T* p, vec_t<T, VEC> v) {
  typename UVecT<T, VEC>::type u;
#pragma unroll
  for (int j = 0; j < VEC; ++j) u[j] = v[j];
  *reinterpret_cast<typename UVecT<T, VEC>::type*>(p) = u;
}

// The operands of one step, in registers.  vcg_row_s asks for samples by (array, offset); every request below is
// resolved at compile time (J and the tap table entries are constants after unrolling) to one register element.
template <typename T, int VEC>
struct VmRegs {
  typedef vec_t<T, VEC> V;
  // velocity: own rows of planes x-1, x, x+1; rows y-1 / y+1 of plane x; the z-1 / z+VEC cells of the own row;
  // and the few samples off those: u[x+1,y-1,z], u[x+1,y,z-1], v[x-1,y+1,z], v[x,y+1,z-1], w[x-1,y,z+1], w[x,y-1,z+1]
  V um, uc, up, uym, uyp, upym;
  T uzl, uzr, upzl;
  V vm, vc, vp, vym, vyp, vmyp;
  T vzl, vzr, vypzl;
  V wm, wc, wp, wyp;
  T wym[VEC + 1];               // w[x, y-1, z0 .. z0+VEC]
  T wzl, wzr, wmzr;             // w[x,y,z0-1], w[x,y,z0+VEC], w[x-1,y,z0+VEC]
  // volume classes (1 xy-edge, 2 xz-edge, 3 x-face, 4 yz-edge, 5 y-face, 6 z-face, 7 cell centre)
  V fx, fy, fz, cc, cm, cym, exyc, exyp, exyyp, exzc, exzp, eyzc, eyzyp;
  T czl, exzzr, eyzzr;          // C[x,y,z0-1], EXZ[x,y,z0+VEC], EYZ[x,y,z0+VEC]
};

template <typename T, int VEC, int J>
struct VmSampler {
  const VmRegs<T, VEC>& r;
  // element J+1 / J-1 of the own-row vector `a`, with the neighbours' cells at the ends
  static __device__ __forceinline__ T zp(const vec_t<T, VEC>& a, T right) { return J < VEC - 1 ? a[J < VEC - 1 ? J + 1 : J] : right; }
  static __device__ __forceinline__ T zm(const vec_t<T, VEC>& a, T left) { return J > 0 ? a[J > 0 ? J - 1 : J] : left; }

  __device__ __forceinline__ double vel(int comp, int dx, int dy, int dz) const {
    const int key = comp * 27 + (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1);
    switch (key) {
      // ---- u
      case 0 * 27 + 1 * 9 + 1 * 3 + 1: return (double)r.uc[J];
      case 0 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.up[J];
      case 0 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.um[J];
      case 0 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.uyp[J];
      case 0 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.uym[J];
      case 0 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.uc, r.uzr);
      case 0 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.uc, r.uzl);
      case 0 * 27 + 2 * 9 + 0 * 3 + 1: return (double)r.upym[J];
      case 0 * 27 + 2 * 9 + 1 * 3 + 0: return (double)zm(r.up, r.upzl);
      // ---- v
      case 1 * 27 + 1 * 9 + 1 * 3 + 1: return (double)r.vc[J];
      case 1 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.vp[J];
      case 1 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.vm[J];
      case 1 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.vyp[J];
      case 1 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.vym[J];
      case 1 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.vc, r.vzr);
      case 1 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.vc, r.vzl);
      case 1 * 27 + 0 * 9 + 2 * 3 + 1: return (double)r.vmyp[J];
      case 1 * 27 + 1 * 9 + 2 * 3 + 0: return (double)zm(r.vyp, r.vypzl);
      // ---- w
      case 2 * 27 + 1 * 9 + 1 * 3 + 1: return (double)r.wc[J];
      case 2 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.wp[J];
      case 2 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.wm[J];
      case 2 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.wyp[J];
      case 2 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.wym[J];
      case 2 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.wc, r.wzr);
      case 2 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.wc, r.wzl);
      case 2 * 27 + 0 * 9 + 1 * 3 + 2: return (double)zp(r.wm, r.wmzr);
      case 2 * 27 + 1 * 9 + 0 * 3 + 2: return (double)r.wym[J + 1];
    }
    __builtin_trap();            // a tap the register file does not hold: the tap table and this kernel disagree
  }

  __device__ __forceinline__ double vol(int p, int ox, int oy, int oz) const {
    const int key = p * 27 + (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1);
    switch (key) {
      case 3 * 27 + 13: return (double)r.fx[J];
      case 5 * 27 + 13: return (double)r.fy[J];
      case 6 * 27 + 13: return (double)r.fz[J];
      case 7 * 27 + 13: return (double)r.cc[J];
      case 7 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.cm[J];
      case 7 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.cym[J];
      case 7 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.cc, r.czl);
      case 1 * 27 + 13: return (double)r.exyc[J];
      case 1 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.exyp[J];
      case 1 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.exyyp[J];
      case 2 * 27 + 13: return (double)r.exzc[J];
      case 2 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.exzp[J];
      case 2 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.exzc, r.exzzr);
      case 4 * 27 + 13: return (double)r.eyzc[J];
      case 4 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.eyzyp[J];
      case 4 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.eyzc, r.eyzzr);
    }
    __builtin_trap();
  }
  __device__ __forceinline__ bool tap_ok(int, int, int, int) const { return true; }
};

// the three rows of cell J of the vector
template <typename T, int VEC, int J>
__device__ __forceinline__ void vm_cell(const VmRegs<T, VEC>& rg, double k1, double k2, unsigned mu, unsigned mv, unsigned mw,
                                        vec_t<T, VEC>& ou, vec_t<T, VEC>& ov, vec_t<T, VEC>& ow, double (&own)[3][VEC]) {
  const VmSampler<T, VEC, J> smp{rg};
  const bool oku = ((mu >> (8 * J)) & 0xffu) != 0, okv = ((mv >> (8 * J)) & 0xffu) != 0, okw = ((mw >> (8 * J)) & 0xffu) != 0;
  ou[J] = (T)vcg_row_s<0, false>(smp, k1, k2, oku, own[0][J]);
  ov[J] = (T)vcg_row_s<1, false>(smp, k1, k2, okv, own[1][J]);
  ow[J] = (T)vcg_row_s<2, false>(smp, k1, k2, okw, own[2][J]);
}

template <typename T, int VEC, int J>
struct VmCells {
  static __device__ __forceinline__ void run(const VmRegs<T, VEC>& rg, double k1, double k2, unsigned mu, unsigned mv,
                                             unsigned mw, vec_t<T, VEC>& ou, vec_t<T, VEC>& ov, vec_t<T, VEC>& ow,
                                             double (&own)[3][VEC]) {
    vm_cell<T, VEC, J>(rg, k1, k2, mu, mv, mw, ou, ov, ow, own);
    VmCells<T, VEC, J + 1>::run(rg, k1, k2, mu, mv, mw, ou, ov, ow, own);
  }
};
template <typename T, int VEC>
struct VmCells<T, VEC, VEC> {
  static __device__ __forceinline__ void run(const VmRegs<T, VEC>&, double, double, unsigned, unsigned, unsigned,
                                             vec_t<T, VEC>&, vec_t<T, VEC>&, vec_t<T, VEC>&, double (&)[3][VEC]) {}
};

// VEC mask bytes at a VEC-aligned byte offset, as an unsigned (byte J = cell J)
template <int VEC>
__device__ __forceinline__ unsigned vm_mask(const unsigned char* p) {
  if (VEC == 4) return *reinterpret_cast<const unsigned*>(p);
  if (VEC == 2) return *reinterpret_cast<const unsigned short*>(p);
  return *p;
}

// store the computed cells of one component's vector (the z = 0 / z = Nz-1 cells of a row are array-boundary faces:
// never written) and accumulate own . out over them
template <typename T, int VEC, bool ALIGNED>
__device__ __forceinline__ void vm_store(T* p, vec_t<T, VEC> o, const double (&own)[VEC], bool first, bool last, double& acc) {
  if (!first && !last) {
    if (ALIGNED) vstore<T, VEC>(p, o); else vstore_u<T, VEC>(p, o);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc += own[j] * (double)o[j];
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool bnd = (first && j == 0) || (last && j == VEC - 1);
      if (!bnd) { p[j] = o[j]; acc += own[j] * (double)o[j]; }
    }
  }
}

// element counts of one LDS plane buffer (u image, v image, w image)
template <int VEC>
__host__ __device__ inline int vm_su(int Nz) { return 2 * Nz + kVmBlock * VEC; }
template <int VEC>
__host__ __device__ inline int vm_sw(int Nz) { return 2 * (Nz + 1) + kVmBlock * VEC + (kVmBlock * VEC) / Nz + 2 + VEC; }
template <int VEC>
__host__ __device__ inline int vm_buf_elems(int Nz) { return (2 * vm_su<VEC>(Nz) + vm_sw<VEC>(Nz) + 3) / 4 * 4; }

// slabs: the three boundary slabs (u at x = Nx-1, v at y = Ny-1, w at z = Nz-1) ride as extra blocks, as in k_vcg_apply_all
template <typename T, int VEC>
__global__ void __launch_bounds__(kVmBlock, MFS_VMARCH_MIN_WAVES)
k_vcg_apply_march(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ ox, T* __restrict__ oy, T* __restrict__ oz,
                  int gmain, Box3 b0, Box3 b1, Box3 b2, int g0, int g1, double* __restrict__ partial,
                  const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  double acc = 0.0;
  if ((int)blockIdx.x >= gmain) {
    const int b = (int)blockIdx.x - gmain, g2 = (int)gridDim.x - gmain - g0 - g1;
    if (b < g0) acc = vcg_slab_rows<T, 0, false>(c, k1, k2, v, ox, b0, b, g0);
    else if (b < g0 + g1) acc = vcg_slab_rows<T, 1, false>(c, k1, k2, v, oy, b1, b - g0, g1);
    else acc = vcg_slab_rows<T, 2, false>(c, k1, k2, v, oz, b2, b - g0 - g1, g2);
    const double tot = block_sum<kVmBlock>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    return;
  }
  typedef vec_t<T, VEC> V;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* const smem = reinterpret_cast<T*>(smem_raw);
  const int Nx = c.N[0], Ny = c.N[1], Nz = c.N[2], W1 = Nz + 1;
  const int nzv = Nz / VEC;
  const int ipp = (Ny - 2) * nzv;                          // interior z-vectors per plane
  const int tiles = (ipp + kVmBlock - 1) / kVmBlock;
  const int np = Nx - 2;                                    // planes 1 .. Nx-2
  const int64_t total = (int64_t)tiles * np;
  const int G = gmain;
  const int nch = min(G, 8);
  const int xcd = blockIdx.x % nch, slot = blockIdx.x / nch;
  const int per = G / nch, extra = G - per * nch;
  const int seg = xcd * per + min(xcd, extra) + slot;      // blocks of one XCD get adjacent segments
  const int64_t s0 = total * seg / G, s1 = total * (seg + 1) / G;
  const int64_t su = (int64_t)Ny * Nz, sv = (int64_t)(Ny + 1) * Nz, sw = (int64_t)Ny * W1, sc = c.plane();
  const T* const U = v.p[0];
  const T* const Vv = v.p[1];
  const T* const W = v.p[2];
  const T* const C1 = (const T*)c.vol[1];
  const T* const C2 = (const T*)c.vol[2];
  const T* const C3 = (const T*)c.vol[3];
  const T* const C4 = (const T*)c.vol[4];
  const T* const C5 = (const T*)c.vol[5];
  const T* const C6 = (const T*)c.vol[6];
  const T* const C7 = (const T*)c.vol[7];
  const unsigned char* const M3 = c.msk[3];
  const unsigned char* const M5 = c.msk[5];
  const unsigned char* const M6 = c.msk[6];
  const int tid = threadIdx.x;
  const int SU = vm_su<VEC>(Nz), BUF = vm_buf_elems<VEC>(Nz);
  const int tile_elems = kVmBlock * VEC;

  for (int64_t i = s0; i < s1;) {
    const int tile = (int)(i / np);
    const int pl = (int)(i - (int64_t)tile * np);
    const int x0 = 1 + pl;
    const int len = (int)min((int64_t)(Nx - 1 - x0), s1 - i);
    const int x1 = x0 + len;
    i += len;
    const int item_raw = tile * kVmBlock + tid;
    const bool active = item_raw < ipp;
    const int item = active ? item_raw : ipp - 1;           // clamp: inactive lanes load valid addresses
    const int yy = item / nzv, zv = item - yy * nzv;
    const bool first = zv == 0, last = zv == nzv - 1;
    const int y = yy + 1, z0 = zv * VEC;
    const int o_uv = y * Nz + z0, o_w = y * W1 + z0, o_c = y * c.pz + z0;   // in-plane offsets (u / v, w, class arrays)
    const int tile_items = min(kVmBlock, ipp - tile * kVmBlock);
    const int tile_len = tile_items * VEC;
    const int yy0 = (tile * kVmBlock) / nzv, zv0 = tile * kVmBlock - yy0 * nzv;
    const int yyl = (tile * kVmBlock + tile_items - 1) / nzv;
    const int m0 = Nz + tile * tile_elems;                   // in-plane offset of the tile's first u / v vector
    const int m0w = (yy0 + 1) * W1 + zv0 * VEC;              // ... of its first w vector
    const int tile_len_w = tile_len + (yyl - yy0);           // the w tile spans one more element per row crossed
    // LDS offsets of this thread's vectors inside a plane buffer (images mirror memory: [halo row | tile | halo row])
    const int lu = Nz + tid * VEC;
    const int lw = 2 * SU + W1 + tid * VEC + (yy - yy0);
    // halo ownership.  u, v: the Nz elements below the tile and the Nz above = 2 * nzv aligned vectors, thread t owns
    // vector t.  w: two rows of Nz+1 elements = 2 * (nzv + 1) unaligned vectors, the last of a row pulled back so that
    // it ends with the row (it overlaps its predecessor; both write the same values).
    const bool hact = tid < 2 * nzv;
    const bool hlow = tid < nzv;
    const int hg = hlow ? m0 - Nz + tid * VEC : m0 + tile_len + (tid - nzv) * VEC;
    const int hl = hlow ? tid * VEC : Nz + tile_len + (tid - nzv) * VEC;
    const int nhw = nzv + 1;
    const bool hwact = tid < 2 * nhw;
    const bool hwlow = tid < nhw;
    const int hwk = min((hwlow ? tid : tid - nhw) * VEC, W1 - VEC);
    const int hgw = hwlow ? m0w - W1 + hwk : m0w + tile_len_w + hwk;
    const int hlw = 2 * SU + (hwlow ? hwk : W1 + tile_len_w + hwk);

    VmRegs<T, VEC> rg;
    // ---- prologue: planes x0-1, x0, x0+1 of the own rows; the carried samples; plane x0's images
    {
      const T* const u0 = U + (int64_t)x0 * su;
      const T* const v0 = Vv + (int64_t)x0 * sv;
      const T* const w0 = W + (int64_t)x0 * sw;
      rg.um = vload<T, VEC>(u0 - su + o_uv); rg.uc = vload<T, VEC>(u0 + o_uv); rg.up = vload<T, VEC>(u0 + su + o_uv);
      rg.vm = vload<T, VEC>(v0 - sv + o_uv); rg.vc = vload<T, VEC>(v0 + o_uv); rg.vp = vload<T, VEC>(v0 + sv + o_uv);
      rg.wm = vload_u<T, VEC>(w0 - sw + o_w); rg.wc = vload_u<T, VEC>(w0 + o_w); rg.wp = vload_u<T, VEC>(w0 + sw + o_w);
      rg.wmzr = w0[-sw + o_w + VEC];
      rg.vmyp = vload<T, VEC>(v0 - sv + o_uv + Nz);
      rg.cm = vload<T, VEC>(C7 + (int64_t)(x0 - 1) * sc + o_c);
      rg.exyp = vload<T, VEC>(C1 + (int64_t)x0 * sc + o_c);     // rotated into exyc / exzc at the top of the first step
      rg.exzp = vload<T, VEC>(C2 + (int64_t)x0 * sc + o_c);
      T* const b = smem;
      if (active) {
        vstore<T, VEC>(b + lu, rg.uc);
        vstore<T, VEC>(b + SU + lu, rg.vc);
#pragma unroll
        for (int j = 0; j < VEC; ++j) b[lw + j] = rg.wc[j];
      }
      if (hact) {
        vstore<T, VEC>(b + hl, vload<T, VEC>(u0 + hg));
        vstore<T, VEC>(b + SU + hl, vload<T, VEC>(v0 + hg));
      }
      if (hwact) {
        const V h = vload_u<T, VEC>(w0 + hgw);
#pragma unroll
        for (int j = 0; j < VEC; ++j) b[hlw + j] = h[j];
      }
    }
    // halo vectors of plane x0+1 (published during step x0)
    V Hu = {}, Hv = {}, Hw = {};
    {
      const int xh = min(x0 + 1, Nx - 1);
      if (hact) { Hu = vload<T, VEC>(U + (int64_t)xh * su + hg); Hv = vload<T, VEC>(Vv + (int64_t)xh * sv + hg); }
      if (hwact) Hw = vload_u<T, VEC>(W + (int64_t)xh * sw + hgw);
    }
    MFS_VISC_LDS_BARRIER();

    for (int x = x0; x < x1; ++x) {
      const int cur = (x - x0) & 1;
      T* const bc = smem + cur * BUF;
      T* const bn = smem + (cur ^ 1) * BUF;
      // ---- (1) prefetch: own rows and halo vectors of plane x+2 (clamped on the last planes: values unused)
      const int x2 = min(x + 2, Nx - 1);
      const V un = vload<T, VEC>(U + (int64_t)x2 * su + o_uv);
      const V vn = vload<T, VEC>(Vv + (int64_t)x2 * sv + o_uv);
      const V wn = vload_u<T, VEC>(W + (int64_t)x2 * sw + o_w);
      V hun = {}, hvn = {}, hwn = {};
      if (hact) { hun = vload<T, VEC>(U + (int64_t)x2 * su + hg); hvn = vload<T, VEC>(Vv + (int64_t)x2 * sv + hg); }
      if (hwact) hwn = vload_u<T, VEC>(W + (int64_t)x2 * sw + hgw);
      // ---- (2) this plane's volume samples and masks
      const int64_t pc = (int64_t)x * sc + o_c;
      rg.exyc = rg.exyp; rg.exzc = rg.exzp;
      rg.fx = vload<T, VEC>(C3 + pc); rg.fy = vload<T, VEC>(C5 + pc); rg.fz = vload<T, VEC>(C6 + pc);
      rg.cc = vload<T, VEC>(C7 + pc); rg.cym = vload<T, VEC>(C7 + pc - c.pz); rg.czl = C7[pc - 1];
      rg.exyp = vload<T, VEC>(C1 + pc + sc); rg.exyyp = vload<T, VEC>(C1 + pc + c.pz);
      rg.exzp = vload<T, VEC>(C2 + pc + sc); rg.exzzr = C2[pc + VEC];
      rg.eyzc = vload<T, VEC>(C4 + pc); rg.eyzyp = vload<T, VEC>(C4 + pc + c.pz); rg.eyzzr = C4[pc + VEC];
      const unsigned mu = vm_mask<VEC>(M3 + pc), mv = vm_mask<VEC>(M5 + pc), mw = vm_mask<VEC>(M6 + pc);
      // ---- (3) in-plane neighbours of plane x from its LDS images
      rg.uyp = vload<T, VEC>(bc + lu + Nz); rg.uym = vload<T, VEC>(bc + lu - Nz);
      rg.uzl = bc[lu - 1]; rg.uzr = bc[lu + VEC];
      rg.vyp = vload<T, VEC>(bc + SU + lu + Nz); rg.vym = vload<T, VEC>(bc + SU + lu - Nz);
      rg.vzl = bc[SU + lu - 1]; rg.vzr = bc[SU + lu + VEC]; rg.vypzl = bc[SU + lu + Nz - 1];
#pragma unroll
      for (int j = 0; j <= VEC; ++j) rg.wym[j] = bc[lw - W1 + j];
#pragma unroll
      for (int j = 0; j < VEC; ++j) rg.wyp[j] = bc[lw + W1 + j];
      rg.wzl = bc[lw - 1]; rg.wzr = bc[lw + VEC];
      // ---- (4) publish plane x+1's images (always: the last step still needs u of plane x1); one barrier per plane
      if (active) {
        vstore<T, VEC>(bn + lu, rg.up);
        vstore<T, VEC>(bn + SU + lu, rg.vp);
#pragma unroll
        for (int j = 0; j < VEC; ++j) bn[lw + j] = rg.wp[j];
      }
      if (hact) { vstore<T, VEC>(bn + hl, Hu); vstore<T, VEC>(bn + SU + hl, Hv); }
      if (hwact) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) bn[hlw + j] = Hw[j];
      }
      MFS_VISC_LDS_BARRIER();
      // ---- (5) the two looks into plane x+1 off the own row
      rg.upym = vload<T, VEC>(bn + lu - Nz);
      rg.upzl = bn[lu - 1];
      // ---- (6) the three rows of every cell of the vector
      V qu, qv, qw;
      double own[3][VEC];
      VmCells<T, VEC, 0>::run(rg, k1, k2, mu, mv, mw, qu, qv, qw, own);
      // ---- (7) stores, d.q
      if (active) {
        vm_store<T, VEC, true>(ox + (int64_t)x * su + o_uv, qu, own[0], first, last, acc);
        vm_store<T, VEC, true>(oy + (int64_t)x * sv + o_uv, qv, own[1], first, last, acc);
        vm_store<T, VEC, false>(oz + (int64_t)x * sw + o_w, qw, own[2], first, last, acc);
      }
      // ---- (8) rotate
      rg.um = rg.uc; rg.uc = rg.up; rg.up = un;
      rg.vm = rg.vc; rg.vc = rg.vp; rg.vp = vn;
      rg.wm = rg.wc; rg.wc = rg.wp; rg.wp = wn;
      rg.wmzr = rg.wzr;
      rg.vmyp = rg.vyp;
      rg.cm = rg.cc;
      Hu = hun; Hv = hvn; Hw = hwn;
    }
    MFS_VISC_LDS_BARRIER();      // the next march stages into buffer 0
  }
  const double tot = block_sum<kVmBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

}  // namespace mfs
