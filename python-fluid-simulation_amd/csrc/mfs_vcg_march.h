// mfs_vcg_march.h -- the per-iteration kernel of the viscosity CG on the pressure kernel's recipe (gfx950).
//
//   q = A d,  A = the reference's variational-viscosity operator, three coupled face components
//   (solver/ViscosityCGSolver3D.py:248-456), rows evaluated by vcg_row_s (mfs_visc.hip) -- the same code,
//   tap order and arithmetic as every other form of the operator in this library, so results are bit-identical.
//
// Algorithmic HBM traffic per cell: d_x, d_y, d_z read, q_x, q_y, q_z written, 7 volume classes read = 13 scalars,
// + 1 packed mask byte.  The kernel moves those bytes once, in 16-byte pieces:
//
//  * one work item = one 16-byte z-vector (4 fp32 / 2 fp64 cells) of an interior row y in [1, Ny-2]; the three
//    operator rows (u, v, w faces) of its cells are computed by the same thread, component after component, the
//    accumulation chains of the vector's cells interleaved (vcg_row_n) so that they hide each other's latency.
//  * a workgroup owns a tile of 256 consecutive z-vectors of the flattened interior (y, z) plane and MARCHES ALONG X.
//  * the velocity operands live in LDS: a ring of four plane slots, each holding the three components' images of the
//    plane tile plus one row of halo each side ([Nz | tile | Nz], mirroring the memory of the u / v components, so every
//    LDS access is an aligned 16-byte one; the w component -- rows of Nz+1 in memory, which no vector alignment survives --
//    is loaded / stored with unaligned 16-byte global accesses and kept in the same row-of-Nz image).  Planes x-1, x and
//    x+1 are read (own rows, y+-1 rows), plane x+2 -- requested at the top of the step, a whole step in flight -- is
//    written at its end; ONE barrier per plane, waiting on lgkmcnt only.  The ring is the register file of the march:
//    nothing velocity is carried in VGPRs across steps, each phase reads what it needs and lets it go.
//  * z-1 / z+VEC neighbours come from the neighbouring lane by DPP (wave_shr / wave_shl); only where a wave's 64 vectors
//    do not start on a row boundary do lanes 0 / 63 read theirs (one lane, exec-masked).
//  * the seven volume classes and the packed mask bytes are solver-owned and stored with one uniform, 16-byte aligned
//    pitch and at one constant stride (struct Compact): one per-thread offset and one base pointer serve them all.  The
//    samples only one phase reads are replaced, right after that phase, by the NEXT step's (a whole step in flight); the
//    x-1 / x+1 samples a step shares with its neighbours are carried in registers.
//  * balanced, XCD-aware work split exactly as in mfs_pcg_apply.h: the (tile, plane) sequence, tile-major, is cut into
//    gridDim equal contiguous segments; blocks with equal blockIdx % 8 (one XCD, one L2) get adjacent segments.
//  * the d.q partial sums ride along; no atomics, bitwise reproducible.
// What bounds it (256^3 fp32, MI355X; tools/vm_stamps.py, tools/pmc_sq.sh, probes below): not HBM (1.2 GB moved in 208 us,
// the same time with the arithmetic removed and with one workgroup per CU instead of two) but the length of a wave's own
// instruction stream between two barriers -- ~1000 instructions per plane, of which ~20 vector-memory and ~75 LDS
// instructions whose issue the wave pays itself.
#pragma once

#ifndef MFS_VMARCH_MIN_WAVES
#define MFS_VMARCH_MIN_WAVES 2     // waves per SIMD the kernel is compiled for (256 VGPRs)
#endif

// Wave priority around the row arithmetic (MFS_VM_PRIO: A/B knob, tools/build_variant.sh).  2 (default, fp32 state): the
// LDS / vector-memory phases of a step run at raised priority, the rows at base priority -- the two waves of a SIMD (one per
// resident workgroup) then tend to sit in complementary phases instead of queueing for the same unit.  Same-box A/B
// (profiles/r02_visc_prio_ab.txt), apply us per launch: 256^3 fp32 198.3 -> 192.3, 128^3 30.3 -> 29.3; fp64 state: 256^3 -1 %,
// 128^3 +0.5 % (left off there).  1: the reverse (rows first): 256^3 fp32 194.4, 128^3 31.0.  0: none.
#ifndef MFS_VM_PRIO
#define MFS_VM_PRIO 2
#endif
#if MFS_VM_PRIO == 1
#define MFS_VM_ROWS_BEGIN() do { if (sizeof(T) == 4) __builtin_amdgcn_s_setprio(2); } while (0)
#define MFS_VM_ROWS_END() do { if (sizeof(T) == 4) __builtin_amdgcn_s_setprio(0); } while (0)
#elif MFS_VM_PRIO == 2
#define MFS_VM_ROWS_BEGIN() do { if (sizeof(T) == 4) __builtin_amdgcn_s_setprio(0); } while (0)
#define MFS_VM_ROWS_END() do { if (sizeof(T) == 4) __builtin_amdgcn_s_setprio(2); } while (0)
#else
#define MFS_VM_ROWS_BEGIN() do {} while (0)
#define MFS_VM_ROWS_END() do {} while (0)
#endif

#ifndef MFS_VM_FUSE_TOP
#define MFS_VM_FUSE_TOP 0       // FUSE: 1 = all three components of plane x+2 requested at the top of the step, 0 = one per phase
#endif

#ifndef MFS_VM_CELL_GROUP
#define MFS_VM_CELL_GROUP 0     // cells of a vector whose accumulation chains are interleaved (0: all of them)
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

// lane l receives lane l-1's / l+1's value (lane 0 / 63 keep their own): the z-1 / z+VEC neighbour of a vector is an
// element of the neighbouring lane's vector -- one DPP move instead of a 4-byte load that drags in a whole line
template <typename T>
__device__ __forceinline__ T vm_from_left(T v) {
  if (sizeof(T) == 4) {
    const int i = __builtin_bit_cast(int, (float)v);
    return (T)__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
  }
  const long long d = __builtin_bit_cast(long long, (double)v);
  const int lo = (int)d, hi = (int)(d >> 32);
  const unsigned rlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  const unsigned rhi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return (T)__builtin_bit_cast(double, ((long long)rhi << 32) | rlo);
}
template <typename T>
__device__ __forceinline__ T vm_from_right(T v) {
  if (sizeof(T) == 4) {
    const int i = __builtin_bit_cast(int, (float)v);
    return (T)__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
  }
  const long long d = __builtin_bit_cast(long long, (double)v);
  const int lo = (int)d, hi = (int)(d >> 32);
  const unsigned rlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  const unsigned rhi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return (T)__builtin_bit_cast(double, ((long long)rhi << 32) | rlo);
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

template <typename T, int VEC>
struct VmSampler {
  const VmRegs<T, VEC>& r;
  int j0;        // the sampler's cell j is cell j0 + j of the vector
  // element J+1 / J-1 of the own-row vector `a`, with the neighbours' cells at the ends
  static __device__ __forceinline__ T zp(const vec_t<T, VEC>& a, T right, int J) { return J < VEC - 1 ? a[J < VEC - 1 ? J + 1 : J] : right; }
  static __device__ __forceinline__ T zm(const vec_t<T, VEC>& a, T left, int J) { return J > 0 ? a[J > 0 ? J - 1 : J] : left; }

  __device__ __forceinline__ double vel(int jj, int comp, int dx, int dy, int dz) const {
    const int J = j0 + jj;
    const int key = comp * 27 + (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1);
    switch (key) {
      // ---- u
      case 0 * 27 + 1 * 9 + 1 * 3 + 1: return (double)r.uc[J];
      case 0 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.up[J];
      case 0 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.um[J];
      case 0 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.uyp[J];
      case 0 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.uym[J];
      case 0 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.uc, r.uzr, J);
      case 0 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.uc, r.uzl, J);
      case 0 * 27 + 2 * 9 + 0 * 3 + 1: return (double)r.upym[J];
      case 0 * 27 + 2 * 9 + 1 * 3 + 0: return (double)zm(r.up, r.upzl, J);
      // ---- v
      case 1 * 27 + 1 * 9 + 1 * 3 + 1: return (double)r.vc[J];
      case 1 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.vp[J];
      case 1 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.vm[J];
      case 1 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.vyp[J];
      case 1 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.vym[J];
      case 1 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.vc, r.vzr, J);
      case 1 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.vc, r.vzl, J);
      case 1 * 27 + 0 * 9 + 2 * 3 + 1: return (double)r.vmyp[J];
      case 1 * 27 + 1 * 9 + 2 * 3 + 0: return (double)zm(r.vyp, r.vypzl, J);
      // ---- w
      case 2 * 27 + 1 * 9 + 1 * 3 + 1: return (double)r.wc[J];
      case 2 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.wp[J];
      case 2 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.wm[J];
      case 2 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.wyp[J];
      case 2 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.wym[J];
      case 2 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.wc, r.wzr, J);
      case 2 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.wc, r.wzl, J);
      case 2 * 27 + 0 * 9 + 1 * 3 + 2: return (double)zp(r.wm, r.wmzr, J);
      case 2 * 27 + 1 * 9 + 0 * 3 + 2: return (double)r.wym[J + 1];
    }
    __builtin_trap();            // a tap the register file does not hold: the tap table and this kernel disagree
  }

  __device__ __forceinline__ double vol(int jj, int p, int ox, int oy, int oz) const {
    const int J = j0 + jj;
    const int key = p * 27 + (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1);
    switch (key) {
      case 3 * 27 + 13: return (double)r.fx[J];
      case 5 * 27 + 13: return (double)r.fy[J];
      case 6 * 27 + 13: return (double)r.fz[J];
      case 7 * 27 + 13: return (double)r.cc[J];
      case 7 * 27 + 0 * 9 + 1 * 3 + 1: return (double)r.cm[J];
      case 7 * 27 + 1 * 9 + 0 * 3 + 1: return (double)r.cym[J];
      case 7 * 27 + 1 * 9 + 1 * 3 + 0: return (double)zm(r.cc, r.czl, J);
      case 1 * 27 + 13: return (double)r.exyc[J];
      case 1 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.exyp[J];
      case 1 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.exyyp[J];
      case 2 * 27 + 13: return (double)r.exzc[J];
      case 2 * 27 + 2 * 9 + 1 * 3 + 1: return (double)r.exzp[J];
      case 2 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.exzc, r.exzzr, J);
      case 4 * 27 + 13: return (double)r.eyzc[J];
      case 4 * 27 + 1 * 9 + 2 * 3 + 1: return (double)r.eyzyp[J];
      case 4 * 27 + 1 * 9 + 1 * 3 + 2: return (double)zp(r.eyzc, r.eyzzr, J);
    }
    __builtin_trap();
  }
  __device__ __forceinline__ bool tap_ok(int, int, int, int, int) const { return true; }
};

// row AXIS of all cells of the vector, MFS_VM_CELL_GROUP cells at a time (their accumulation chains interleaved:
// vcg_row_n); own . out joins `acc` unless the cell is one of the row's two array-boundary cells (z = 0 / z = Nz-1:
// never stored) or the lane is a clamped duplicate (`count` false)
template <typename T, int VEC, int AXIS>
__device__ __forceinline__ void vm_row(const VmRegs<T, VEC>& rg, double k1, double k2, unsigned m, vec_t<T, VEC>& q, bool first,
                                       bool last, bool count, double& acc) {
  constexpr int G = MFS_VM_CELL_GROUP > 0 ? MFS_VM_CELL_GROUP : VEC;
  constexpr int NC = (G < VEC && VEC % G == 0) ? G : VEC;
#pragma unroll
  for (int j0 = 0; j0 < VEC; j0 += NC) {
    const VmSampler<T, VEC> smp{rg, j0};
    bool ok[NC];
    double out[NC], own[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) ok[j] = ((m >> (8 * (j0 + j) + AXIS)) & 1u) != 0;     // byte J of m: cell J; bit AXIS: this row
#ifdef MFS_VM_PROBE_NOMATH     // timing probe (WRONG results): every operand touched once, no operator arithmetic
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 14; ++t) { const VTap tp = kTaps[AXIS][t]; a += (float)smp.vel(j, tp.comp, tp.dx, tp.dy, tp.dz); }
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int ax = kD0[AXIS][0] + kVolOff[k][0], ay = kD0[AXIS][1] + kVolOff[k][1], az = kD0[AXIS][2] + kVolOff[k][2];
        a += (float)smp.vol(j, ((ax & 1) << 2) | ((ay & 1) << 1) | (az & 1), fdiv2(ax), fdiv2(ay), fdiv2(az));
      }
      own[j] = smp.vel(j, AXIS, 0, 0, 0);
      out[j] = ok[j] ? (double)a : 0.0;
    }
#else
    vcg_row_n<AXIS, false, NC>(smp, k1, k2, ok, out, own);
#endif
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int J = j0 + j;
      q[J] = (T)out[j];
      const bool bnd = (J == 0 && first) || (J == VEC - 1 && last);
      if (count && !bnd) acc += own[j] * (double)q[J];
    }
  }
}

// VEC mask bytes at a VEC-aligned byte offset, as an unsigned (byte J = cell J)
template <int VEC>
__device__ __forceinline__ unsigned vm_mask(const unsigned char* p) {
  if (VEC == 4) return *reinterpret_cast<const unsigned*>(p);
  if (VEC == 2) return *reinterpret_cast<const unsigned short*>(p);
  return *p;
}

// store the computed cells of one component's vector (the z = 0 / z = Nz-1 cells of a row are array-boundary faces:
// never written)
template <typename T, int VEC, bool ALIGNED, int NT>
__device__ __forceinline__ void vm_store(T* p, vec_t<T, VEC> o, bool first, bool last) {
  if (!first && !last) {
    if (ALIGNED) { if (NT & 4) vstore_nt<T, VEC>(p, o); else vstore<T, VEC>(p, o); }
    else vstore_u<T, VEC>(p, o);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool bnd = (first && j == 0) || (last && j == VEC - 1);
      if (!bnd) p[j] = o[j];
    }
  }
}

// LDS: a ring of kVmRing plane slots (planes x-1, x, x+1 being read, x+2 being written), each the three components'
// images [Nz halo | tile | Nz halo]
constexpr int kVmRing = 4;
template <int VEC, int BLOCK = kVmBlock>
__host__ __device__ inline int vm_su(int Nz) { return 2 * Nz + BLOCK * VEC; }
template <typename T, int VEC, int BLOCK = kVmBlock, int RING = kVmRing>
__host__ __device__ inline size_t vm_lds_bytes(int Nz) { return (size_t)RING * 3 * vm_su<VEC, BLOCK>(Nz) * sizeof(T); }

#define MFS_VM_PIN() __builtin_amdgcn_sched_barrier(0)
// nontemporal hints (template NT; the host turns them on when the launch's working set exceeds the Infinity Cache, as the
// pressure engine does).  Bit 0: loads of the classes whose lines nobody reads twice in a launch (x-, y-, z-face classes,
// xz-edge class: own vectors only, their z-neighbours come by DPP); bit 1: the own rows of C / EXY / EYZ as well (A/B only:
// their y-neighbour rows are then re-fetched, 216 -> 241 us); bit 2: the q stores (read next by the r update, long after
// they have left the caches at this size).  Same-box A/B at 256^3 fp32, us per launch: none 216.1, loads 216.2, stores
// 209.1, loads + stores 204.1 (CG iteration 504 -> 487).
#define MFS_VM_LD1(p) ((NT & 1) ? vload_nt<T, VEC>(p) : vload<T, VEC>(p))
#define MFS_VM_LD2(p) ((NT & 2) ? vload_nt<T, VEC>(p) : vload<T, VEC>(p))
// In-kernel stamps (diagnostic build only, -DMFS_VM_STAMPS; tools/vm_stamps.py): where a step's cycles go.  One stamp =
// s_memtime + lgkmcnt(0) in one statement, fenced by scheduling barriers; the segment sums of the first 64 workgroups'
// waves go to the upper half of the partial-sum array, which nothing else reads.
#ifdef MFS_VM_STAMPS
#define MFS_VM_STAMP(k)                                                                  \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long t_;                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    stamp_sum[k] += t_ - stamp_last;                                                      \
    stamp_last = t_;                                                                      \
  } while (0)
#else
#define MFS_VM_STAMP(k) do {} while (0)
#endif
#ifdef MFS_VM_PROBE_NOSCAL     // traffic probe (WRONG results): no z-neighbour scalar loads, no mask loads
#define MFS_VM_SC(expr) ((T)1)
#define MFS_VM_MK(expr) 0x01010101u
#else
#define MFS_VM_SC(expr) (expr)
#define MFS_VM_MK(expr) (expr)
#endif
#ifdef MFS_VM_PROBE_NONBR      // traffic probe (WRONG results): class-array neighbour rows read as own rows
#define MFS_VM_NBR(off) 0
#else
#define MFS_VM_NBR(off) (off)
#endif

// FUSE (the fused loop of mfs_vcg3d_iterate, iteration j >= 1): the launch also performs the two vector updates the
// reference does between two applies (ViscosityCGSolver3D.py:595-597, :609-610) for every face it owns --
//     x += alpha_{j-1} d_{j-1} ;  d_j = r_j + beta_{j-1} d_{j-1} ;  q = A d_j
// `v` is then d_{j-1} (read only), d_j goes to the OTHER buffer of the pair {bound d, engine buffer} (fz.dn: neighbouring
// workgroups and the slab blocks still read d_{j-1}), and every operand of the operator -- own vectors, halo rows, the
// x-1 / x+1 planes of a march, the slab rows' taps -- is formed on the fly as (T)(r + beta d_{j-1}): the arithmetic of
// k_update_d, rounded to the storage type like the stored d_j, so q is bit-identical to the three-kernel loop.  A face
// is WRITTEN (d_j, x) by exactly one owner: planes [x0, x1) of a march's tile, the cells q is stored for.
// 13 + 12 scalars per cell instead of 13 + 15 in two launches, and one dependent launch less per iteration.
template <typename T>
struct VmFuse {
  const T* r[3];          // residual components (read)
  T* dn[3];               // d_j (written once per owned face)
  T* x[3];                // solution (read-modify-write by the owner)
  const double* scal;     // S_ALPHA / S_BETA of the iteration closed by the previous launch
};

// slabs: the three boundary slabs (u at x = Nx-1, v at y = Ny-1, w at z = Nz-1) ride as extra blocks, as in k_vcg_apply_all
// BLOCK: threads per workgroup = z-vectors per tile.  256 (two workgroups per CU) everywhere but fp64 state with rows so long
// that the ring of a 256-vector tile -- two rows and two halo rows at Nz = 256 -- exceeds half the CU's LDS: there ONE
// workgroup of 512 threads (a tile of four rows + two halo rows: the same eight waves per CU, 1.5x instead of 2x halo
// traffic) replaces one of 256.
// RING: plane slots of the LDS ring.  4: planes x-1, x, x+1 are read while x+2 is written -- ONE barrier per plane.  3 (rows so
// long that four slots of a 512-vector tile exceed the CU's LDS: fp64 380 < Nz <= 512, fp32 760 < Nz <= 1024): plane x+2 takes the
// slot of plane x-1, after a second barrier.
// COMP (compressed class access, round 3): bits 4-5 of the mask byte of a vector's FIRST cell (k_vcg_classify, once per
// set-up) say whether every class sample this vector's step LOADS -- the seven classes at the vector itself, C at x-1 and
// y-1, EXY at x+1 and y+1, EXZ at x+1, EYZ at y+1 -- is +0.0 (kVmClsZero: outside the liquid), 1.0 (kVmClsOne: inside it)
// or anything else (0: MIXED).  Only MIXED vectors read the class arrays; the others take the constant, so their registers
// hold exactly the values a load would have returned and everything downstream -- the rows, the DPP z-neighbours taken from
// the neighbouring lanes' registers, the samples carried across steps -- is bit-identical to dense access.  In a liquid
// solve most of the domain is air or bulk liquid: 7 of the kernel's 13 scalars per cell are then not moved at all.  The
// class travels with the mask byte, requested two steps ahead so that it is known when the next step's loads are issued.
constexpr unsigned kVmClsZero = 1, kVmClsOne = 2;
template <typename T, int VEC, int WAVES, int NT, bool FUSE = false, int BLOCK = kVmBlock, int RING = kVmRing, bool COMP = false>
__global__ void __launch_bounds__(BLOCK, WAVES)
k_vcg_apply_march(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ ox, T* __restrict__ oy, T* __restrict__ oz,
                  int gmain, Box3 b0, Box3 b1, Box3 b2, int g0, int g1, double* __restrict__ partial,
                  const double* __restrict__ done_flag, VmFuse<T> fz) {
  // the flag is REQUESTED here and tested where the first march's own loads have been issued: on a small grid the
  // launch is a chain of memory round trips, and flag -> planes was two of them (loads past a raised flag are harmless)
  const double dn = done_flag ? *done_flag : 0.0;
  const T bulk = COMP ? *(const T*)c.bulk : (T)0;           // the constant of class kVmClsOne (k_vcg_classify compared against it)
  const double f_alpha = FUSE ? fz.scal[S_ALPHA] : 0.0, f_beta = FUSE ? fz.scal[S_BETA] : 0.0;
  double acc = 0.0;
  if ((int)blockIdx.x >= gmain) {
    if (dn != 0.0) return;
    const int b = (int)blockIdx.x - gmain, g2 = (int)gridDim.x - gmain - g0 - g1;
    if constexpr (FUSE) {
      const Vec3T<T> rr{{fz.r[0], fz.r[1], fz.r[2]}};
      if (b < g0) acc = vcg_slab_rows_fused<T, 0>(c, k1, k2, v, rr, f_alpha, f_beta, ox, fz.dn[0], fz.x[0], b0, b, g0);
      else if (b < g0 + g1) acc = vcg_slab_rows_fused<T, 1>(c, k1, k2, v, rr, f_alpha, f_beta, oy, fz.dn[1], fz.x[1], b1, b - g0, g1);
      else acc = vcg_slab_rows_fused<T, 2>(c, k1, k2, v, rr, f_alpha, f_beta, oz, fz.dn[2], fz.x[2], b2, b - g0 - g1, g2);
    } else {
      if (b < g0) acc = vcg_slab_rows<T, 0, false>(c, k1, k2, v, ox, b0, b, g0);
      else if (b < g0 + g1) acc = vcg_slab_rows<T, 1, false>(c, k1, k2, v, oy, b1, b - g0, g1);
      else acc = vcg_slab_rows<T, 2, false>(c, k1, k2, v, oz, b2, b - g0 - g1, g2);
    }
    const double tot = block_sum<BLOCK>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    return;
  }
  typedef vec_t<T, VEC> V;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* const smem = reinterpret_cast<T*>(smem_raw);
  const int Nx = c.N[0], Ny = c.N[1], Nz = c.N[2], W1 = Nz + 1;
  const int nzv = Nz / VEC;
  const int ipp = (Ny - 2) * nzv;                          // interior z-vectors per plane
  const int tiles = (ipp + BLOCK - 1) / BLOCK;
  const int np = Nx - 2;                                    // planes 1 .. Nx-2
  // COMP, work list (the launches of a single-domain CG loop): only the listed (tile, plane) pairs are visited -- every
  // other pair is all air, its q = +0 was stored by the solve's initial q = A x and nothing else writes it
  const bool listed = COMP && c.items != nullptr;
  const int64_t total = listed ? (int64_t)*c.count : (int64_t)tiles * np;
  const int G = gmain;
  const int nch = min(G, 8);
  const int xcd = blockIdx.x % nch, slot = blockIdx.x / nch;
  const int per = G / nch, extra = G - per * nch;
  const int seg = xcd * per + min(xcd, extra) + slot;      // blocks of one XCD get adjacent segments
  // plain: gridDim equal segments of the (tile, plane) sequence.  COMP: equal-COST segments (k_vcg_balance at set-up: a
  // plane where the whole tile is air costs a fraction of one that computes) -- else the launch lasts as long as the
  // marches that lie entirely in the liquid, however empty the rest of the domain is
  int64_t s0 = total * seg / G, s1 = total * (seg + 1) / G;
  if (COMP && !listed && c.seg_g == G) { s0 = c.seg[seg]; s1 = c.seg[seg + 1]; }
  const int64_t su = (int64_t)Ny * Nz, sv = (int64_t)(Ny + 1) * Nz, sw = (int64_t)Ny * W1, sc = c.plane();
  const T* const U = v.p[0];
  const T* const Vv = v.p[1];
  const T* const W = v.p[2];
  // the seven class arrays sit at one constant stride in the engine's workspace (and so do the three mask arrays):
  // one base pointer each instead of ten
  const T* const C1 = (const T*)c.vol[1];
  const int64_t cs = (const T*)c.vol[2] - (const T*)c.vol[1];
  const unsigned char* const MP = c.msk;
  const int tid = threadIdx.x;
  const int SU = vm_su<VEC, BLOCK>(Nz), BUF = 3 * SU;
  const int tile_elems = BLOCK * VEC;
#ifdef MFS_VM_STAMPS
  unsigned long long stamp_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

  for (int64_t i = s0; i < s1;) {
    int tile, x0, len;
    if (listed) {                     // entry = tile * Nx + x; a run of consecutive entries is one march
      const int it = c.items[i];
      tile = it / Nx;
      x0 = it - tile * Nx;
      len = (int)min((int64_t)c.runrem[i], s1 - i);
    } else {
      tile = (int)(i / np);
      x0 = 1 + (int)(i - (int64_t)tile * np);
      len = (int)min((int64_t)(Nx - 1 - x0), s1 - i);
    }
    const int x1 = x0 + len;
    i += len;
    const int item_raw = tile * BLOCK + tid;
    const bool active = item_raw < ipp;
    const int item = active ? item_raw : ipp - 1;           // clamp: inactive lanes load valid addresses
    const int yy = item / nzv, zv = item - yy * nzv;
    const bool first = zv == 0, last = zv == nzv - 1;
    const int y = yy + 1, z0 = zv * VEC;
    const int o_uv = y * Nz + z0, o_w = y * W1 + z0, o_c = y * c.pz + z0;   // in-plane offsets (u / v, w, class arrays)
    const int tile_items = min(BLOCK, ipp - tile * BLOCK);
    const int tile_len = tile_items * VEC;
    const int m0 = Nz + tile * tile_elems;                   // in-plane (u-layout) offset of the tile's first vector
    // LDS offset of this thread's vectors inside an image.  All three images use the u layout -- rows of Nz elements,
    // [Nz halo | tile | Nz halo], mirroring the memory of the u / v components, every access 16-byte aligned.  The w
    // component's (Nz+1)-th element of a row is not kept: only the row's never-computed last cell would tap it.
    const int lu = Nz + tid * VEC;
    // halo ownership: the Nz elements below the tile and the Nz above = 2 * nzv vectors, thread t owns vector t
    const bool hact = tid < 2 * nzv;
    const bool hlow = tid < nzv;
    const int hg = hlow ? m0 - Nz + tid * VEC : m0 + tile_len + (tid - nzv) * VEC;     // in-plane, u / v
    const int hl = hlow ? tid * VEC : Nz + tile_len + (tid - nzv) * VEC;               // in the image
    const int hgw = hg + hg / Nz;                                                     // in-plane, w (rows of Nz+1)

    // one plane's own vectors and halo vectors: global -> registers -> its ring slot
    struct Plane { V u, v, w, hu, hv, hw; };
    auto fetch = [&](int xp) {
      Plane p;
      p.u = vload<T, VEC>(U + (int64_t)xp * su + o_uv);
      p.v = vload<T, VEC>(Vv + (int64_t)xp * sv + o_uv);
      p.w = vload_u<T, VEC>(W + (int64_t)xp * sw + o_w);
      p.hu = V{}; p.hv = V{}; p.hw = V{};
#ifdef MFS_VM_PROBE_NOHALO     // traffic probe (WRONG results): what do the halo rows cost?
      if (false) {
#else
      if (hact) {
#endif
        p.hu = vload<T, VEC>(U + (int64_t)xp * su + hg);
        p.hv = vload<T, VEC>(Vv + (int64_t)xp * sv + hg);
        p.hw = vload_u<T, VEC>(W + (int64_t)xp * sw + hgw);
      }
      return p;
    };
    auto publish = [&](T* b, const Plane& p) {
      if (active) { vstore<T, VEC>(b + lu, p.u); vstore<T, VEC>(b + SU + lu, p.v); vstore<T, VEC>(b + 2 * SU + lu, p.w); }
      if (hact) { vstore<T, VEC>(b + hl, p.hu); vstore<T, VEC>(b + SU + hl, p.hv); vstore<T, VEC>(b + 2 * SU + hl, p.hw); }
    };

    // FUSE: one component of one plane -- d_{j-1} and r of the own vector and of the halo vector, x of the own vector --
    // requested (fload), and later turned into d_j for the plane's image, and, on a plane this march owns, into the
    // stored d_j and the updated x (fform).  Component by component, so that 5 vectors are in flight at a time.
    struct CompLd { V d, hd, r, hr, x; };
    auto fload = [&](int comp, int xp, bool owned) {
      CompLd l;
      l.hd = V{}; l.hr = V{}; l.x = V{};
      if constexpr (FUSE) {
        const int64_t ps = comp == 0 ? su : (comp == 1 ? sv : sw);
        const int64_t oo = (int64_t)xp * ps + (comp == 2 ? o_w : o_uv), oh = (int64_t)xp * ps + (comp == 2 ? hgw : hg);
        if (comp < 2) {
          l.d = vload<T, VEC>(v.p[comp] + oo); l.r = vload<T, VEC>(fz.r[comp] + oo);
          if (hact) { l.hd = vload<T, VEC>(v.p[comp] + oh); l.hr = vload<T, VEC>(fz.r[comp] + oh); }
          if (owned) l.x = (NT & 4) ? vload_nt<T, VEC>(fz.x[comp] + oo) : vload<T, VEC>(fz.x[comp] + oo);
        } else {
          l.d = vload_u<T, VEC>(v.p[comp] + oo); l.r = vload_u<T, VEC>(fz.r[comp] + oo);
          if (hact) { l.hd = vload_u<T, VEC>(v.p[comp] + oh); l.hr = vload_u<T, VEC>(fz.r[comp] + oh); }
          if (owned) l.x = vload_u<T, VEC>(fz.x[comp] + oo);
        }
      } else {
        l.d = V{}; l.r = V{};
      }
      return l;
    };
    auto fform = [&](int comp, int xp, bool owned, const CompLd& l, T* slot) {
      if constexpr (FUSE) {
        V dj, hj, xn;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          dj[j] = (T)__builtin_fma(f_beta, (double)l.d[j], (double)l.r[j]);
          hj[j] = (T)__builtin_fma(f_beta, (double)l.hd[j], (double)l.hr[j]);
          xn[j] = (T)__builtin_fma(f_alpha, (double)l.d[j], (double)l.x[j]);
        }
        if (active) vstore<T, VEC>(slot + comp * SU + lu, dj);
        if (hact) vstore<T, VEC>(slot + comp * SU + hl, hj);
        if (owned && active) {
          const int64_t ps = comp == 0 ? su : (comp == 1 ? sv : sw);
          const int64_t oo = (int64_t)xp * ps + (comp == 2 ? o_w : o_uv);
          if (comp < 2) { vm_store<T, VEC, true, 0>(fz.dn[comp] + oo, dj, first, last); vm_store<T, VEC, true, NT>(fz.x[comp] + oo, xn, first, last); }
          else { vm_store<T, VEC, false, 0>(fz.dn[comp] + oo, dj, first, last); vm_store<T, VEC, false, 0>(fz.x[comp] + oo, xn, first, last); }
        }
      }
    };

    // z neighbours by DPP.  Lane l-1 / l+1 of the wave holds the row's previous / next vector whenever this lane's
    // vector is not the first / last of its row (then the neighbour is never used); only lane 0 / 63 can have its
    // neighbour in another wave, and only when a wave's 64 vectors do not start on a row boundary, i.e. 64 % nzv != 0:
    // those single lanes then read the value themselves (exec-masked: one lane).
    const bool fix = (64 % nzv) != 0;
    const bool fixl = fix && (tid & 63) == 0 && !first, fixr = fix && (tid & 63) == 63 && !last;
    auto ZL = [&](const V& a, const T* at) { T r = vm_from_left<T>(a[VEC - 1]); if (fixl) r = at[-1]; return r; };
    auto ZR = [&](const V& a, const T* at) { T r = vm_from_right<T>(a[0]); if (fixr) r = at[VEC]; return r; };

    // COMP, tile level: byte (tile, x) of c.tw says whether the TILE computes anything at plane x -- all its vectors air
    // otherwise (k_vcg_tile_flags); tw_at(x) packs the flags of planes x .. x+3 into bits 0-3.  Where a whole tile is air for a stretch of planes
    // the march neither fetches nor stages those planes, skips the barrier and only stores its zeros: deep air costs
    // 12 bytes per cell instead of 53.  Everything derived from it is uniform over the workgroup.
    // the flags of 64 consecutive planes at a time: ONE wave-wide byte load + a ballot -> a 64-bit mask in scalar
    // registers (a per-step load of the flag would be a memory round trip per step: in air that IS the step)
    const unsigned char* const twp = COMP ? c.tw + (int64_t)tile * Nx : nullptr;
    int wb = x0;                       // plane of bit 0 of the window
    unsigned long long win = 0;
    auto win_load = [&](int base) {
      const int xx = base + (tid & 63);
      const bool busy = xx < x1 && twp[xx] != 0;          // steps at or beyond x1 do not exist for this march
      win = __builtin_amdgcn_ballot_w64(busy);
      wb = base;
    };
    auto tw_at = [&](int xx) -> unsigned {                // bits 0-3: the tile computes something at plane xx .. xx+3
      if (xx + 3 - wb >= 64) win_load(xx);
      return (unsigned)((win >> (xx - wb)) & 0xfull);
    };
    if (COMP) win_load(x0);
    unsigned tw_cur = COMP ? tw_at(x0) : 0xfu;
    bool dirty = false;              // an image-reading step has run since the last barrier
    VmRegs<T, VEC> rg;
    unsigned msk = 0, mskn = 0;      // mask bytes (+ COMP: class bits) of this step's vector; COMP: of the next step's too
    // ---- prologue: the images of planes x0-1, x0, x0+1 into ring slots 0, 1, 2; the volume samples of the first step
    {
      if constexpr (FUSE) {
        // all nine component loads of the three planes in one batch, then formed into the ring (and, for the planes this
        // march owns, into d_j and x)
        const bool ob = true, oc = x0 + 1 < x1;
        const CompLd a0 = fload(0, x0 - 1, false), a1 = fload(1, x0 - 1, false), a2 = fload(2, x0 - 1, false);
        const CompLd b0_ = fload(0, x0, ob), b1_ = fload(1, x0, ob), b2_ = fload(2, x0, ob);
        const CompLd c0 = fload(0, x0 + 1, oc), c1 = fload(1, x0 + 1, oc), c2 = fload(2, x0 + 1, oc);
        if (dn != 0.0) return;                                 // uniform over the grid
        fform(0, x0 - 1, false, a0, smem); fform(1, x0 - 1, false, a1, smem); fform(2, x0 - 1, false, a2, smem);
        fform(0, x0, ob, b0_, smem + BUF); fform(1, x0, ob, b1_, smem + BUF); fform(2, x0, ob, b2_, smem + BUF);
        fform(0, x0 + 1, oc, c0, smem + 2 * BUF); fform(1, x0 + 1, oc, c1, smem + 2 * BUF); fform(2, x0 + 1, oc, c2, smem + 2 * BUF);
      } else {
        // COMP: a plane is staged only if a step of this march that computes anything (tile not all air) taps it
        Plane pa = Plane{}, pb = Plane{}, pc = Plane{};
        if (tw_cur & 1u) pa = fetch(x0 - 1);
        if (tw_cur & 3u) pb = fetch(x0);
        if (tw_cur & 7u) pc = fetch(x0 + 1);
        if (dn != 0.0) return;                                   // uniform over the grid
        if (tw_cur & 1u) publish(smem, pa);
        if (tw_cur & 3u) publish(smem + BUF, pb);
        if (tw_cur & 7u) publish(smem + 2 * BUF, pc);
      }
      const T* const q = C1 + (int64_t)x0 * sc + o_c;          // class 1 of this vector; class p at q + (p-1)*cs
      rg.czl = (T)0; rg.exzzr = (T)0; rg.eyzzr = (T)0;
      if (fixl) rg.czl = q[6 * cs - 1];
      if (fixr) { rg.exzzr = q[cs + VEC]; rg.eyzzr = q[3 * cs + VEC]; }
      if constexpr (COMP) {
        // the step's class first (one more dependent round trip per march, behind the plane fetches already in flight)
        msk = vm_mask<VEC>(MP + (int64_t)x0 * sc + o_c);
        mskn = vm_mask<VEC>(MP + (int64_t)min(x0 + 1, Nx - 2) * sc + o_c);
        const unsigned cl = (msk >> 4) & 3u;
        const T cv = cl == kVmClsOne ? bulk : (T)0;
        V k;
#pragma unroll
        for (int j = 0; j < VEC; ++j) k[j] = cv;
        rg.cm = k; rg.exyc = k; rg.exzc = k; rg.fx = k; rg.fy = k; rg.fz = k; rg.cc = k; rg.cym = k;
        rg.exyp = k; rg.exyyp = k; rg.exzp = k; rg.eyzc = k; rg.eyzyp = k;
        if (cl == 0) {
          rg.cm = vload<T, VEC>(q + 6 * cs - sc);
          rg.exyc = vload<T, VEC>(q); rg.exzc = vload<T, VEC>(q + cs);
          rg.fx = vload<T, VEC>(q + 2 * cs); rg.fy = vload<T, VEC>(q + 4 * cs); rg.fz = vload<T, VEC>(q + 5 * cs);
          rg.cc = vload<T, VEC>(q + 6 * cs); rg.cym = vload<T, VEC>(q + 6 * cs - c.pz);
          rg.exyp = vload<T, VEC>(q + sc); rg.exyyp = vload<T, VEC>(q + c.pz);
          rg.exzp = vload<T, VEC>(q + cs + sc);
          rg.eyzc = vload<T, VEC>(q + 3 * cs); rg.eyzyp = vload<T, VEC>(q + 3 * cs + c.pz);
        }
      } else {
        rg.cm = vload<T, VEC>(q + 6 * cs - sc);
        rg.exyc = vload<T, VEC>(q); rg.exzc = vload<T, VEC>(q + cs);
        rg.fx = vload<T, VEC>(q + 2 * cs); rg.fy = vload<T, VEC>(q + 4 * cs); rg.fz = vload<T, VEC>(q + 5 * cs);
        rg.cc = vload<T, VEC>(q + 6 * cs); rg.cym = vload<T, VEC>(q + 6 * cs - MFS_VM_NBR(c.pz));
        rg.exyp = vload<T, VEC>(q + sc); rg.exyyp = vload<T, VEC>(q + MFS_VM_NBR(c.pz));
        rg.exzp = vload<T, VEC>(q + cs + sc);
        rg.eyzc = vload<T, VEC>(q + 3 * cs); rg.eyzyp = vload<T, VEC>(q + 3 * cs + MFS_VM_NBR(c.pz));
      }
    }
    if constexpr (!COMP) msk = MFS_VM_MK(vm_mask<VEC>(MP + (int64_t)x0 * sc + o_c));
    MFS_VM_STAMP(0);                                             // prologue

    for (int x = x0; x < x1; ++x) {
      const int k = x - x0;
      const T* const bm = smem + (k % RING) * BUF;             // plane x-1
      const T* const bc = smem + ((k + 1) % RING) * BUF;       // plane x
      const T* const bn = smem + ((k + 2) % RING) * BUF;       // plane x+1
      T* const bw = smem + ((k + 3) % RING) * BUF;             // plane x+2 (written at the end of this step)
      // ---- (1) ONE barrier per plane: plane x+1's images (published at the end of the previous step) are complete,
      //      and nobody reads plane x-2's slot any more (it is written at the end of this step)
      // plain: every step.  COMP: only when this step reads the images (it must see plane x+1, staged at the end of the
      // previous step), or stages a plane into a slot that somebody may still be reading (an image-reading step ran since
      // the last barrier)
      const bool reads = !COMP || (tw_cur & 1u) != 0;
      const bool stage = (x + 2 <= x1) && (!COMP || (tw_cur & 0xeu) != 0);
      if (reads || (stage && dirty)) { MFS_VISC_LDS_BARRIER(); dirty = false; }
      if (COMP && reads) dirty = true;
      unsigned tw_nxt = 0xfu;
      if (COMP) tw_nxt = x + 1 < x1 ? tw_at(x + 1) : 0u;
      MFS_VM_STAMP(1);                                           // barrier
      // ---- (2) in flight for the whole step: own rows and halo vectors of plane x+2 (clamped at the end: unused)
      // (the last two steps of a march need no further plane, its last step no further class samples: wave-uniform skips)
      const bool need_plane = stage, more = x + 1 < x1;
      Plane pn = Plane{};
      const bool own_n = x + 2 < x1;                              // FUSE: plane x+2 is one of this march's own
      CompLd fl = CompLd{}, fl1 = CompLd{}, fl2 = CompLd{};
      if (need_plane) {
        if constexpr (FUSE) {
          fl = fload(0, x + 2, own_n);
          if (MFS_VM_FUSE_TOP) { fl1 = fload(1, x + 2, own_n); fl2 = fload(2, x + 2, own_n); }
        } else {
          pn = fetch(x + 2);
        }
      }
      const T* const qn = C1 + (int64_t)min(x + 1, Nx - 2) * sc + o_c;      // next step's class samples
      const unsigned char* const mqn = MP + (int64_t)min(x + 1, Nx - 2) * sc + o_c;
      MFS_VM_PIN();
      MFS_VM_STAMP(2);                                           // issue of plane x+2's loads
      // ---- (3) u rows.  Every velocity operand comes from the images (the ring is the register file of the march:
      //      nothing velocity is carried across steps); the registers of the class samples only this phase reads take
      //      the next step's afterwards (a whole step in flight)
      // COMP: a wave whose 64 vectors are ALL air at this step (every class sample 0: wave-uniform, one ballot) skips the
      // image reads and the rows of the step -- with all seven volumes 0 every coefficient of the three rows is 0 and
      // q = 0 whatever the (finite) velocities are.  The skipped arithmetic would have produced +-0 (the sign following
      // the operands' signs) or NaN for a non-finite operand; the skip stores +0: equal in value (==), not in the sign
      // bit of a zero.  Planes are still fetched and published (neighbouring waves tap them), q is still stored.
      const bool skip = COMP && (!reads || __builtin_amdgcn_ballot_w64(((msk >> 4) & 3u) != kVmClsZero) == 0);
      // work-list launches: q of an all-air VECTOR is +0 since the solve's initial q = A x, like that of an all-air pair --
      // its three stores are not issued (lane level: a liquid body covers part of a tile's rows)
      const bool qst = active && !(listed && ((msk >> 4) & 3u) == kVmClsZero);
      V qu = V{};
      if (!skip) {
      rg.um = vload<T, VEC>(bm + lu); rg.uc = vload<T, VEC>(bc + lu); rg.up = vload<T, VEC>(bn + lu);
      rg.uyp = vload<T, VEC>(bc + lu + Nz); rg.uym = vload<T, VEC>(bc + lu - Nz);
      rg.uzl = ZL(rg.uc, bc + lu); rg.uzr = ZR(rg.uc, bc + lu);
      rg.vc = vload<T, VEC>(bc + SU + lu); rg.vm = vload<T, VEC>(bm + SU + lu);
      rg.vyp = vload<T, VEC>(bc + SU + lu + Nz); rg.vmyp = vload<T, VEC>(bm + SU + lu + Nz);
      rg.wc = vload<T, VEC>(bc + 2 * SU + lu); rg.wzr = ZR(rg.wc, bc + 2 * SU + lu);
      rg.wm = vload<T, VEC>(bm + 2 * SU + lu); rg.wmzr = ZR(rg.wm, bm + 2 * SU + lu);
      { const T t = vm_from_right<T>(rg.exzc[0]); rg.exzzr = fixr ? rg.exzzr : t; }      // EXZ[x, y, z0+VEC]
      MFS_VM_STAMP(3);                                           // u: image reads (issued and landed)
      MFS_VM_ROWS_BEGIN();
      vm_row<T, VEC, 0>(rg, k1, k2, msk, qu, first, last, active, acc);
      MFS_VM_ROWS_END();
      }
      MFS_VM_STAMP(4);                                           // u: rows (includes the wait for this step's class samples)
      if (COMP ? qst : active) vm_store<T, VEC, true, NT>(ox + (int64_t)x * su + o_uv, qu, first, last);
      MFS_VM_PIN();
      // COMP: the next step's class (its mask word has been in flight for a whole step) and the constant that stands
      // for every sample of a ZERO / ONE vector; only MIXED vectors (ldn) issue class loads -- exec-masked, lane by lane
      const unsigned cln = COMP ? ((mskn >> 4) & 3u) : 0u;
      const bool ldn = more && cln == 0;
      V kv;
#pragma unroll
      for (int j = 0; j < VEC; ++j) kv[j] = cln == kVmClsOne ? bulk : (T)0;
      V fxn = kv, ccn = kv, exyypn = kv;
      T exzzrn = (T)0;
      if (ldn) { fxn = MFS_VM_LD1(qn + 2 * cs); ccn = MFS_VM_LD2(qn + 6 * cs); exyypn = vload<T, VEC>(qn + MFS_VM_NBR(c.pz)); }
      if (more && fixr) exzzrn = qn[cs + VEC];
      if constexpr (FUSE) if (need_plane) { fform(0, x + 2, own_n, fl, bw); if (!MFS_VM_FUSE_TOP) fl1 = fload(1, x + 2, own_n); }
      MFS_VM_PIN();
      MFS_VM_STAMP(5);                                           // u: store + issue of the next step's samples
      // ---- (4) v rows
      V qv = V{};
      if (!skip) {
      rg.vc = vload<T, VEC>(bc + SU + lu); rg.vp = vload<T, VEC>(bn + SU + lu); rg.vm = vload<T, VEC>(bm + SU + lu);
      rg.vyp = vload<T, VEC>(bc + SU + lu + Nz); rg.vym = vload<T, VEC>(bc + SU + lu - Nz);
      rg.vzl = ZL(rg.vc, bc + SU + lu); rg.vzr = ZR(rg.vc, bc + SU + lu);
      rg.up = vload<T, VEC>(bn + lu); rg.upym = vload<T, VEC>(bn + lu - Nz);
      rg.uc = vload<T, VEC>(bc + lu); rg.uym = vload<T, VEC>(bc + lu - Nz);
      rg.wc = vload<T, VEC>(bc + 2 * SU + lu); rg.wzr = ZR(rg.wc, bc + 2 * SU + lu);
      {
        const V t = vload<T, VEC>(bc + 2 * SU + lu - Nz);
#pragma unroll
        for (int j = 0; j < VEC; ++j) rg.wym[j] = t[j];
        rg.wym[VEC] = ZR(t, bc + 2 * SU + lu - Nz);
      }
      { const T t = vm_from_right<T>(rg.eyzc[0]); rg.eyzzr = fixr ? rg.eyzzr : t; }      // EYZ[x, y, z0+VEC]
      MFS_VM_STAMP(6);                                           // v: image reads
      MFS_VM_ROWS_BEGIN();
      vm_row<T, VEC, 1>(rg, k1, k2, msk, qv, first, last, active, acc);
      MFS_VM_ROWS_END();
      }
      MFS_VM_STAMP(7);                                           // v: rows
      if (COMP ? qst : active) vm_store<T, VEC, true, NT>(oy + (int64_t)x * sv + o_uv, qv, first, last);
      MFS_VM_PIN();
      V fyn = kv, cymn = kv, exypn = kv;
      T eyzzrn = (T)0;
      unsigned mskl = 0;               // plain: the next step's mask bytes; COMP: those of the step after it
      if (ldn) { fyn = MFS_VM_LD1(qn + 4 * cs); cymn = vload<T, VEC>(qn + 6 * cs - MFS_VM_NBR(c.pz)); exypn = MFS_VM_LD2(qn + sc); }
      if (more) {
        if (fixr) eyzzrn = qn[3 * cs + VEC];
        mskl = MFS_VM_MK(vm_mask<VEC>(COMP ? MP + (int64_t)min(x + 2, Nx - 2) * sc + o_c : mqn));
      }
      if constexpr (FUSE) if (need_plane) { fform(1, x + 2, own_n, fl1, bw); if (!MFS_VM_FUSE_TOP) fl2 = fload(2, x + 2, own_n); }
      MFS_VM_PIN();
      MFS_VM_STAMP(8);                                           // v: store + issue
      // ---- (5) w rows
      V qw = V{};
      if (!skip) {
      rg.wc = vload<T, VEC>(bc + 2 * SU + lu); rg.wp = vload<T, VEC>(bn + 2 * SU + lu); rg.wm = vload<T, VEC>(bm + 2 * SU + lu);
      rg.wyp = vload<T, VEC>(bc + 2 * SU + lu + Nz);
      {
        const V t = vload<T, VEC>(bc + 2 * SU + lu - Nz);
#pragma unroll
        for (int j = 0; j < VEC; ++j) rg.wym[j] = t[j];
      }
      rg.wzl = ZL(rg.wc, bc + 2 * SU + lu); rg.wzr = ZR(rg.wc, bc + 2 * SU + lu);
      rg.up = vload<T, VEC>(bn + lu); rg.upzl = ZL(rg.up, bn + lu);
      rg.uc = vload<T, VEC>(bc + lu); rg.uzl = ZL(rg.uc, bc + lu);
      rg.vc = vload<T, VEC>(bc + SU + lu); rg.vzl = ZL(rg.vc, bc + SU + lu);
      rg.vyp = vload<T, VEC>(bc + SU + lu + Nz); rg.vypzl = ZL(rg.vyp, bc + SU + lu + Nz);
      { const T t = vm_from_left<T>(rg.cc[VEC - 1]); rg.czl = fixl ? rg.czl : t; }        // C[x, y, z0-1]
      MFS_VM_STAMP(9);                                           // w: image reads
      MFS_VM_ROWS_BEGIN();
      vm_row<T, VEC, 2>(rg, k1, k2, msk, qw, first, last, active, acc);
      MFS_VM_ROWS_END();
      }
      MFS_VM_STAMP(10);                                          // w: rows
      if (COMP ? qst : active) vm_store<T, VEC, false, NT>(oz + (int64_t)x * sw + o_w, qw, first, last);
      MFS_VM_PIN();
      V fzn = kv, exzpn = kv, eyzcn = kv, eyzypn = kv;
      T czln = (T)0;
      if (ldn) {
        fzn = MFS_VM_LD1(qn + 5 * cs); exzpn = MFS_VM_LD1(qn + cs + sc); eyzcn = MFS_VM_LD2(qn + 3 * cs);
        eyzypn = vload<T, VEC>(qn + 3 * cs + MFS_VM_NBR(c.pz));
      }
      if (more && fixl) czln = qn[6 * cs - 1];
      MFS_VM_PIN();
      // ---- (6) plane x+2 into its slot (its loads have had the whole step); next step's class samples take over
      static_assert(!FUSE || RING == 4, "the fused form writes plane x+2's slot while the step reads the others");
      if (RING == 3 && need_plane && reads) { MFS_VISC_LDS_BARRIER(); dirty = false; }      // plane x+2 reuses plane x-1's slot: everybody has read it
      if (need_plane) { if constexpr (FUSE) fform(2, x + 2, own_n, fl2, bw); else publish(bw, pn); }
      MFS_VM_STAMP(11);                                          // w: store + issue, publish of plane x+2 (waits for its loads)
      rg.cm = rg.cc; rg.cc = ccn;
      rg.exyc = rg.exyp; rg.exyp = exypn; rg.exzc = rg.exzp; rg.exzp = exzpn;
      rg.fx = fxn; rg.exyyp = exyypn; rg.exzzr = exzzrn;
      rg.fy = fyn; rg.cym = cymn; rg.eyzzr = eyzzrn;
      rg.fz = fzn; rg.eyzc = eyzcn; rg.eyzyp = eyzypn; rg.czl = czln;
      if constexpr (COMP) { msk = mskn; mskn = mskl; } else msk = mskl;
      tw_cur = tw_nxt;
    }
    if (!COMP || dirty) MFS_VISC_LDS_BARRIER();      // the next march stages into the ring while a slow wave may still read this one's planes
  }
#ifdef MFS_VM_STAMPS
  if (blockIdx.x < 64 && (tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) partial[4096 + (blockIdx.x * 4 + tid / 64) * 12 + k] = (double)stamp_sum[k];
  }
#endif
  const double tot = block_sum<BLOCK>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

}  // namespace mfs
