// mfs_visc.hip -- ViscosityCGSolver3D on gfx950: the variational-viscosity operator
// (3 coupled face components, 15 taps per row), its RHS, the 3-sweep extrapolation,
// the write-back, and the CG engine around them.
//
// Reference: solver/ViscosityCGSolver3D.py.  Rows (u,v,w) are described by ONE tap
// table (kTaps below; SURVEY.md Appendix A, re-read against :248-456) from which
// both the operator (mask `sphi >= 0`, sign as listed) and the RHS (mask `sphi < 0`,
// opposite sign, :41-246) are instantiated -- the same table drives the oracle.
//
// Two forms of the operator:
//  * k_visc_row_direct: straight from the caller's doubled-grid `sphi` / `vol`
//    arrays (stride-2 reads).  Used for the module-level drop-ins `matvecmul` /
//    `initialize_solver` and once per solve for the RHS.
//  * k_vcg_apply: the per-iteration kernel.  Once per solve k_vcg_setup
//    de-interleaves `vol` into its 7 used parity classes (cell centres, 3 face
//    classes, 3 edge classes) and `sphi >= 0` at the 3 face
//    classes into byte masks, all unit-stride.  A tap's mask sample is always the
//    validity of the tapped face itself, so the masks are per-DOF bytes.
//    Algorithmic traffic: 3 (v) + 7 (vol classes) + 3 (out) scalars + 3 mask bytes
//    per cell, vs 8x-strided reads of two (2N+1)^3 arrays in the reference.
#include "mfs_cg_core.h"
#include "mfs_p2p.h"

// workgroup barrier that waits on LDS traffic only: the global prefetch of the next plane stays in flight
#define MFS_VISC_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace mfs {

struct VTap { int fac, vol, sgn, comp, dx, dy, dz, mx, my, mz; };
// vol sample index: 0=c 1=R 2=L 3=T 4=B 5=F 6=K   (offsets from the face's doubled index D)
__device__ constexpr int kVolOff[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
// D = 2*(x,y,z) + kD0[row]
__device__ constexpr int kD0[3][3] = {{0, 1, 1}, {1, 0, 1}, {1, 1, 0}};
// diag = c + k*(fR*R + fL*L + fT*T + fB*B + fF*F + fK*K)    (:268, :338, :408)
__device__ constexpr int kDiagFac[3][6] = {{2, 2, 1, 1, 1, 1}, {1, 1, 2, 2, 1, 1}, {1, 1, 1, 1, 2, 2}};
__device__ constexpr VTap kTaps[3][14] = {
    {  // u-row :271-314
        {2, 1, -1, 0, 1, 0, 0, 2, 0, 0},   {2, 2, -1, 0, -1, 0, 0, -2, 0, 0}, {1, 3, -1, 0, 0, 1, 0, 0, 2, 0},
        {1, 4, -1, 0, 0, -1, 0, 0, -2, 0}, {1, 5, -1, 0, 0, 0, 1, 0, 0, 2},   {1, 6, -1, 0, 0, 0, -1, 0, 0, -2},
        {1, 3, -1, 1, 0, 1, 0, 1, 1, 0},   {1, 3, +1, 1, -1, 1, 0, -1, 1, 0}, {1, 4, +1, 1, 0, 0, 0, 1, -1, 0},
        {1, 4, -1, 1, -1, 0, 0, -1, -1, 0}, {1, 5, -1, 2, 0, 0, 1, 1, 0, 1},  {1, 5, +1, 2, -1, 0, 1, -1, 0, 1},
        {1, 6, +1, 2, 0, 0, 0, 1, 0, -1},  {1, 6, -1, 2, -1, 0, 0, -1, 0, -1},
    },
    {  // v-row :341-384
        {1, 1, -1, 1, 1, 0, 0, 2, 0, 0},   {1, 2, -1, 1, -1, 0, 0, -2, 0, 0}, {2, 3, -1, 1, 0, 1, 0, 0, 2, 0},
        {2, 4, -1, 1, 0, -1, 0, 0, -2, 0}, {1, 5, -1, 1, 0, 0, 1, 0, 0, 2},   {1, 6, -1, 1, 0, 0, -1, 0, 0, -2},
        {1, 1, -1, 0, 1, 0, 0, 1, 1, 0},   {1, 1, +1, 0, 1, -1, 0, 1, -1, 0}, {1, 2, +1, 0, 0, 0, 0, -1, 1, 0},
        {1, 2, -1, 0, 0, -1, 0, -1, -1, 0}, {1, 5, -1, 2, 0, 0, 1, 0, 1, 1},  {1, 5, +1, 2, 0, -1, 1, 0, -1, 1},
        {1, 6, +1, 2, 0, 0, 0, 0, 1, -1},  {1, 6, -1, 2, 0, -1, 0, 0, -1, -1},
    },
    {  // w-row :411-454
        {1, 1, -1, 2, 1, 0, 0, 2, 0, 0},   {1, 2, -1, 2, -1, 0, 0, -2, 0, 0}, {1, 3, -1, 2, 0, 1, 0, 0, 2, 0},
        {1, 4, -1, 2, 0, -1, 0, 0, -2, 0}, {2, 5, -1, 2, 0, 0, 1, 0, 0, 2},   {2, 6, -1, 2, 0, 0, -1, 0, 0, -2},
        {1, 1, -1, 0, 1, 0, 0, 1, 0, 1},   {1, 1, +1, 0, 1, 0, -1, 1, 0, -1}, {1, 2, +1, 0, 0, 0, 0, -1, 0, 1},
        {1, 2, -1, 0, 0, 0, -1, -1, 0, -1}, {1, 3, -1, 1, 0, 1, 0, 0, 1, 1},  {1, 3, +1, 1, 0, 1, -1, 0, 1, -1},
        {1, 4, +1, 1, 0, 0, 0, 0, -1, 1},  {1, 4, -1, 1, 0, 0, -1, 0, -1, -1},
    },
};

struct G3 {
  int N[3];
  __host__ __device__ int sh(int comp, int ax) const { return N[ax] + (comp == ax ? 1 : 0); }
  __host__ __device__ int64_t nface(int comp) const { return (int64_t)sh(comp, 0) * sh(comp, 1) * sh(comp, 2); }
  __device__ int64_t fidx(int comp, int x, int y, int z) const {
    return ((int64_t)x * sh(comp, 1) + y) * sh(comp, 2) + z;
  }
  __device__ int64_t dg(int i, int j, int k) const { return ((int64_t)i * (2 * N[1] + 1) + j) * (2 * N[2] + 1) + k; }
};

struct V3 { const void* p[3]; };
struct W3 { void* p[3]; };

// ----------------------------------------------------- direct (doubled grid) --
template <int AXIS, bool RHS>
__global__ void __launch_bounds__(256)
k_visc_row_direct(G3 g, double scale, double mu, V3 v, int vdt, void* out, int odt, const void* sphi, int sdt,
                  const void* vol, int voldt) {
  const int s1 = g.sh(AXIS, 1), s2 = g.sh(AXIS, 2), s0 = g.sh(AXIS, 0);
  const int64_t n = (int64_t)s0 * s1 * s2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % s2), y = (int)((i / s2) % s1), x = (int)(i / ((int64_t)s2 * s1));
  if (x == 0 || x >= s0 - 1 || y == 0 || y >= s1 - 1 || z == 0 || z >= s2 - 1) return;   // array-boundary faces untouched
  const int Dx = 2 * x + kD0[AXIS][0], Dy = 2 * y + kD0[AXIS][1], Dz = 2 * z + kD0[AXIS][2];
  if (ldx(sphi, sdt, g.dg(Dx, Dy, Dz)) < 0) { stx(out, odt, i, 0.0); return; }            // solid face
  double vs[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) vs[k] = ldx(vol, voldt, g.dg(Dx + kVolOff[k][0], Dy + kVolOff[k][1], Dz + kVolOff[k][2]));
  const double own = ldx(v.p[AXIS], vdt, i);
  double val;
  if (RHS) {
    val = own * vs[0];
  } else {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double t = kDiagFac[AXIS][k] == 2 ? 2 * vs[k + 1] : vs[k + 1];
      s = k == 0 ? t : s + t;
    }
    val = (vs[0] + scale * mu * s) * own;
  }
#pragma unroll
  for (int t = 0; t < 14; ++t) {
    const VTap tp = kTaps[AXIS][t];
    const double k = tp.fac == 2 ? 2 * scale * mu : scale * mu;
    const double term = k * vs[tp.vol] * ldx(v.p[tp.comp], vdt, g.fidx(tp.comp, x + tp.dx, y + tp.dy, z + tp.dz));
    const double m = ldx(sphi, sdt, g.dg(Dx + tp.mx, Dy + tp.my, Dz + tp.mz));
    if (RHS) { if (m < 0) val -= tp.sgn * term; }
    else     { if (m >= 0) val += tp.sgn * term; }
  }
  stx(out, odt, i, val);
}

// ------------------------------------------------------------ extrapolation --
// valid = sphi(face) >= 0 for every face of the component (:479-481)
template <int AXIS>
__global__ void __launch_bounds__(256) k_visc_valid(G3 g, const void* sphi, int sdt, unsigned char* valid) {
  const int s1 = g.sh(AXIS, 1), s2 = g.sh(AXIS, 2);
  const int64_t n = g.nface(AXIS);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % s2), y = (int)((i / s2) % s1), x = (int)(i / ((int64_t)s2 * s1));
  valid[i] = ldx(sphi, sdt, g.dg(2 * x + kD0[AXIS][0], 2 * y + kD0[AXIS][1], 2 * z + kD0[AXIS][2])) >= 0 ? 1 : 0;
}

// one Jacobi sweep (:8-39): every face is written (copy-through or new value), so
// the in/out buffers ping-pong exactly like the reference's new_v / new_valid copies.
__global__ void __launch_bounds__(256)
k_visc_extrap_sweep(int s0, int s1, int s2, const void* vin, void* vout, int vdt, const unsigned char* valid_in,
                    unsigned char* valid_out) {
  const int64_t n = (int64_t)s0 * s1 * s2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % s2), y = (int)((i / s2) % s1), x = (int)(i / ((int64_t)s2 * s1));
  double nv = ldx(vin, vdt, i);
  unsigned char va = valid_in[i];
  const bool interior = !(x == 0 || x >= s0 - 1 || y == 0 || y >= s1 - 1 || z == 0 || z >= s2 - 1);
  if (interior && !va) {
    double val = 0.0;
    int count = 0;
    const int64_t sx = (int64_t)s1 * s2, sy = s2;
    const int64_t nb[6] = {i + sx, i - sx, i + sy, i - sy, i + 1, i - 1};
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (valid_in[nb[k]]) { val += ldx(vin, vdt, nb[k]); ++count; }
    if (count > 0) { nv = val / count; va = 1; }
  }
  stx(vout, vdt, i, nv);
  valid_out[i] = va;
}

// apply_viscosity_kernel (:458-470): x,y,z in [1, N-1]
__global__ void __launch_bounds__(256)
k_visc_writeback(G3 g, W3 v, int vdt, V3 o, int odt, const void* sphi, int sdt) {
  const int64_t n = (int64_t)g.N[0] * g.N[1] * g.N[2];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % g.N[2]), y = (int)((i / g.N[2]) % g.N[1]), x = (int)(i / ((int64_t)g.N[2] * g.N[1]));
  if (x == 0 || y == 0 || z == 0) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (ldx(sphi, sdt, g.dg(2 * x + kD0[c][0], 2 * y + kD0[c][1], 2 * z + kD0[c][2])) >= 0) {
      const int64_t f = g.fidx(c, x, y, z);
      stx(v.p[c], vdt, f, ldx(o.p[c], odt, f));
    }
  }
}

// ------------------------------------------ notebook grid kernels (8(f) rank 1) --
// validity = grid mass > 0   (notebook `extrapolate`, 3D_viscous_fluid_sim.ipynb code cell 7)
__global__ void __launch_bounds__(256) k_grid_valid_mass(int64_t n, const void* m, int mdt, unsigned char* valid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) valid[i] = ldx(m, mdt, i) > 0 ? 1 : 0;
}

// boundary_condition_{x,y,z} (notebook code cell 5).  dv = 0 on array-boundary faces and on faces at
// least dx away from the solid; else minus the AXIS component of the inward-normal part of the
// solid-relative velocity, faded by (1 - sphi/dx).  The velocity*mass products are formed in the
// arrays' own dtype (fp32 in the notebook) and accumulated in fp64, as numba types them.
// (The reference stores dv = 0 before its bounds check, i.e. out of bounds for the rounded-up
// part of its launch grid; only in-range faces are written here.)
template <int AXIS>
__global__ void __launch_bounds__(256)
k_grid_boundary_condition(G3 g, V3 gv, int vdt, V3 gm, int mdt, const void* sphi, int sdt, const void* sv, int svdt,
                          double dx, void* dv, int dvdt) {
  const int s0 = g.sh(AXIS, 0), s1 = g.sh(AXIS, 1), s2 = g.sh(AXIS, 2);
  const int64_t n = (int64_t)s0 * s1 * s2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % s2), y = (int)((i / s2) % s1), x = (int)(i / ((int64_t)s2 * s1));
  if (x == 0 || x >= s0 - 1 || y == 0 || y >= s1 - 1 || z == 0 || z >= s2 - 1) { stx(dv, dvdt, i, 0.0); return; }
  const int Dx = 2 * x + kD0[AXIS][0], Dy = 2 * y + kD0[AXIS][1], Dz = 2 * z + kD0[AXIS][2];
  const double ndist = ldx(sphi, sdt, g.dg(Dx, Dy, Dz)) / dx;
  if (ndist >= 1) { stx(dv, dvdt, i, 0.0); return; }
  double vel[3] = {0.0, 0.0, 0.0};
  vel[AXIS] = ldx(gv.p[AXIS], vdt, i);
  const bool f32prod = vdt == MFS_F32 && mdt == MFS_F32;
  // the two other components: 4 samples each, offsets as written in the notebook's loops
  constexpr int CA = AXIS == 0 ? 1 : 0, CB = AXIS == 2 ? 1 : 2;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const int comp = w == 0 ? CA : CB;
    double msum = 0.0, vsum = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int ox, oy, oz;
        if (AXIS == 0) { ox = -p; oy = w == 0 ? q : 0; oz = w == 0 ? 0 : q; }          // x: (x-ix, y+iy, z) / (x-ix, y, z+iy)
        else if (AXIS == 1) { ox = w == 0 ? q : 0; oy = -p; oz = w == 0 ? 0 : q; }     // y: (x+iz, y-iy, z) / (x, y-iy, z+iz)
        else { ox = w == 0 ? q : 0; oy = w == 0 ? 0 : q; oz = -p; }                    // z: (x+ix, y, z-iz) / (x, y+ix, z-iz)
        const int64_t f = g.fidx(comp, x + ox, y + oy, z + oz);
        const double mm = ldx(gm.p[comp], mdt, f), vv = ldx(gv.p[comp], vdt, f);
        msum += mm;
        vsum += f32prod ? (double)((float)vv * (float)mm) : vv * mm;
      }
    vel[comp] = vsum / msum;
  }
  const int64_t sc = 3 * g.dg(Dx, Dy, Dz);
  const double rx = vel[0] - ldx(sv, svdt, sc + 0), ry = vel[1] - ldx(sv, svdt, sc + 1), rz = vel[2] - ldx(sv, svdt, sc + 2);
  const double snx = ldx(sphi, sdt, g.dg(Dx + 1, Dy, Dz)) - ldx(sphi, sdt, g.dg(Dx - 1, Dy, Dz));
  const double sny = ldx(sphi, sdt, g.dg(Dx, Dy + 1, Dz)) - ldx(sphi, sdt, g.dg(Dx, Dy - 1, Dz));
  const double snz = ldx(sphi, sdt, g.dg(Dx, Dy, Dz + 1)) - ldx(sphi, sdt, g.dg(Dx, Dy, Dz - 1));
  const double sn_inv = 1.0 / (snx * snx + sny * sny + snz * snz);
  const double s = snx * rx + sny * ry + snz * rz;
  const double proj = (s < 0 ? s : 0.0) * (AXIS == 0 ? snx : (AXIS == 1 ? sny : snz)) * sn_inv;   // min(0, s): NaN -> 0
  stx(dv, dvdt, i, -proj * (1.0 - ndist));
}

// ------------------------------------------------------------ compact form ---
// parity class code p = (i&1)<<2 | (j&1)<<1 | (k&1) of a doubled-grid node;
// logical class array dims: odd axis -> N, even axis -> N+1; compact index = node >> 1.
// Storage: EVERY class array (and mask array) is laid out with the same pitches -- (N[0]+1) planes of py = N[1]+1 rows
// of pz = N[2]+4 elements -- so that one per-thread offset addresses the same cell in all of them, rows start 16-byte
// aligned whenever N[2] % 4 == 0 (the (N+1)-long rows of the even-z classes would otherwise break vector alignment on
// every row, SURVEY.md 7 "hard parts"), and the z-1 / z+VEC neighbours of any vector are addressable.  Pads are 0.
struct Compact {
  const void* vol[8];           // state dtype; [0] (e,e,e) unused
  const unsigned char* msk;     // one byte per compact index: bit c = the face of component c there is not solid
                                // (`sphi >= 0` at the three face classes 3 (e,o,o), 5 (o,e,o), 6 (o,o,e))
  const unsigned char* tw;      // compressed class access, tile level: byte (tile, x) != 0: the march tile computes something
                                // at plane x (k_vcg_tile_flags); built for tiles of `tw_block` vectors
  int tw_block;
  const int* seg;               // ... and the cost-balanced cut of the (tile, plane) sequence into seg_g segments (k_vcg_balance)
  int seg_g;
  const int* items;             // work list of the loop's march launches (null: every pair): indices tile * Nx + x of the busy
  const int* runrem;            // (tile, plane) pairs, ascending; runrem[k] = consecutive entries from k on (one march); *count
  const int* count;
  const void* bulk;             // ONE value of the state dtype: the largest volume sample of the set-up (what a sub-cell
                                // deep inside the liquid carries: 1 up to the rounding of lvol / (cell_vol / 8)) -- the
                                // constant of the compressed class access' second uniform class
  int N[3];
  int py, pz;                   // row / element pitch (see above)
  __host__ __device__ int dim(int p, int ax) const { return N[ax] + (((p >> (2 - ax)) & 1) ? 0 : 1); }
  __host__ __device__ int64_t idx(int x, int y, int z) const { return ((int64_t)x * py + y) * pz + z; }
  __host__ __device__ int64_t plane() const { return (int64_t)py * pz; }
  __host__ __device__ int64_t stored() const { return (int64_t)(N[0] + 1) * py * pz; }   // elements of one array
};
__host__ __device__ constexpr int face_class(int comp) { return comp == 0 ? 3 : (comp == 1 ? 5 : 6); }
__host__ __device__ constexpr int fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }

template <typename T>
__global__ void __launch_bounds__(256)
k_vcg_setup(int Nx, int Ny, int Nz, int py, int pz, const void* sphi, int sdt, const void* vol, int voldt, T* o1, T* o2,
            T* o3, T* o4, T* o5, T* o6, T* o7, unsigned char* mp) {
  const int d1 = 2 * Ny + 1, d2 = 2 * Nz + 1;
  const int64_t n = (int64_t)(2 * Nx + 1) * d1 * d2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = (int)(i % d2), j = (int)((i / d2) % d1), ii = (int)(i / ((int64_t)d2 * d1));
  const int p = ((ii & 1) << 2) | ((j & 1) << 1) | (k & 1);
  if (p == 0) return;
  const int64_t ci = ((int64_t)(ii >> 1) * py + (j >> 1)) * pz + (k >> 1);
  T* dst = p == 1 ? o1 : p == 2 ? o2 : p == 3 ? o3 : p == 4 ? o4 : p == 5 ? o5 : p == 6 ? o6 : o7;
  dst[ci] = (T)ldx(vol, voldt, i);
  if ((p == 3 || p == 5 || p == 6) && ldx(sphi, sdt, i) >= 0) {
    // three nodes share the byte of a compact index: OR the component's bit into its (zeroed) 32-bit word
    const unsigned bit = 1u << (p == 3 ? 0 : (p == 5 ? 1 : 2));
    atomicOr(reinterpret_cast<unsigned*>(mp + (ci & ~(int64_t)3)), bit << (8 * (int)(ci & 3)));
  }
}

// the largest positive volume sample over the seven class arrays -> *out (bit pattern order = value order for positive
// floats: one unsigned atomic max per wave)
template <typename T>
__global__ void __launch_bounds__(256)
k_vcg_bulk_value(Compact c, T* out) {
  typedef typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type U;
  const int64_t n = c.stored();
  U best = 0;
  for (int p = 1; p < 8; ++p) {
    const T* a = (const T*)c.vol[p];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const T v = a[i];
      if (v > (T)0) { const U b = __builtin_bit_cast(U, v); best = b > best ? b : best; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    U other;
    if (sizeof(T) == 4) other = (U)__shfl_xor((unsigned)best, o, 64);
    else other = (U)(((unsigned long long)__shfl_xor((unsigned)((unsigned long long)best >> 32), o, 64) << 32) | __shfl_xor((unsigned)best, o, 64));
    best = other > best ? other : best;
  }
  if ((threadIdx.x & 63) == 0 && best != 0) atomicMax(reinterpret_cast<U*>(out), best);
}

// Compressed class access of the x-marching kernel (mfs_vcg_march.h, COMP): the class of every interior z-vector -- are
// ALL class samples its step loads (and the three its rows take from the neighbouring vectors) +0.0 (bit 4), all its own exactly the
// set-up's bulk value (bit 5), or anything else (neither)? -- into bits 4-5
// of the mask byte of the vector's first cell (every reader of the mask bytes extracts single bits 0-2).  Bit patterns
// are compared (a -0.0 is "anything else"), so the constant that stands for a load IS what the load returns.  Runs behind
// k_vcg_setup in the same stream, once per set-up.
template <typename T> __device__ __forceinline__ bool vcg_is_pos_zero(T v);
template <> __device__ __forceinline__ bool vcg_is_pos_zero<float>(float v) { return __float_as_uint(v) == 0u; }
template <> __device__ __forceinline__ bool vcg_is_pos_zero<double>(double v) { return __double_as_longlong(v) == 0ll; }

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_vcg_classify(Compact c, unsigned char* mp) {
  const int Nx = c.N[0], Ny = c.N[1], nzv = c.N[2] / VEC;
  const int64_t total = (int64_t)(Nx - 2) * (Ny - 2) * nzv;
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (iv >= total) return;
  const int zv = (int)(iv % nzv), y = 1 + (int)((iv / nzv) % (Ny - 2)), x = 1 + (int)(iv / ((int64_t)nzv * (Ny - 2)));
  const int64_t base = c.idx(x, y, zv * VEC), sc = c.plane();
  bool zero = true, one = true;
  const T bulk = *(const T*)c.bulk;
  auto chk = [&](int p, int64_t off) {
    const T* a = (const T*)c.vol[p] + base + off;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const T v = a[j]; zero = zero && vcg_is_pos_zero<T>(v); one = one && v == bulk; }
  };
#pragma unroll
  for (int p = 1; p < 8; ++p) chk(p, 0);
  chk(7, -sc); chk(7, -c.pz);          // C at x-1, y-1
  chk(1, sc); chk(1, c.pz);            // EXY at x+1, y+1
  chk(2, sc);                          // EXZ at x+1
  chk(4, c.pz);                        // EYZ at y+1
  // ZERO also promises that every ROW of the vector's cells is empty (the march skips all-ZERO waves and tiles, and its work-list
  // launches do not store q for a ZERO vector): the three class operands the rows take from the NEIGHBOURING vectors' registers
  // -- C at z0-1, EXZ and EYZ at z0+VEC -- must be +0 as well.  (A liquid surface lying exactly on a vector boundary has
  // all of this vector's own samples 0 and EXZ[z0+VEC] > 0: the last cell's u row is then not empty.)
  if (zero) {
    if (zv > 0) zero = vcg_is_pos_zero<T>(((const T*)c.vol[7])[base - 1]);
    if (zero && zv < nzv - 1) zero = vcg_is_pos_zero<T>(((const T*)c.vol[2])[base + VEC]) && vcg_is_pos_zero<T>(((const T*)c.vol[4])[base + VEC]);
  }
  mp[base] = (unsigned char)((mp[base] & 0x0f) | (zero ? 0x10 : (one ? 0x20 : 0)));
}

// tile flags of the marching kernel (see Compact::tw): one WAVE per (tile, x) decides whether the tile computes anything
// at plane x (any vector not all air) -> flag byte (0 at the two boundary planes: no step there)
template <int VEC>
__global__ void __launch_bounds__(256)
k_vcg_tile_flags(Compact c, int block, unsigned char* flags) {
  const int Nx = c.N[0], Ny = c.N[1], nzv = c.N[2] / VEC;
  const int ipp = (Ny - 2) * nzv, tiles = (ipp + block - 1) / block;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  if (w >= (int64_t)tiles * Nx) return;
  const int tile = (int)(w / Nx), x = (int)(w - (int64_t)tile * Nx);
  bool busy = false;
  if (x >= 1 && x <= Nx - 2) {
    const int v1 = min(ipp, (tile + 1) * block);
    for (int it = tile * block + lane; it < v1; it += kWave) {
      const int yy = it / nzv, zv = it - yy * nzv;
      busy = busy || ((c.msk[c.idx(x, yy + 1, zv * VEC)] >> 4) & 3u) != 1u;
    }
  }
  const bool any = __builtin_amdgcn_ballot_w64(busy) != 0;
  if (lane == 0) flags[w] = any ? 1 : 0;
}

// Cost-balanced segments of the march's (tile, plane) sequence (tile-major, planes 1 .. Nx-2): item cost kVmCostBusy
// where the tile computes something at the plane, 1 where it is all air.  seg[k] = the item whose cost interval holds
// total * k / G; seg[G] = n.  ONE block: chunk sums, a scan over the 1024 chunks, then every thread walks its chunk.
constexpr int kVmCostBusy = 6;
__global__ void __launch_bounds__(1024)
k_vcg_balance(const unsigned char* flags, int tiles, int Nx, int G, int* seg) {
  const int np = Nx - 2;
  const int64_t n = (int64_t)tiles * np;
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024, i0 = min(n, (int64_t)t * chunk), i1 = min(n, i0 + chunk);
  auto cost = [&](int64_t i) -> long long { const int64_t tl = i / np; return flags[tl * Nx + 1 + (i - tl * np)] ? kVmCostBusy : 1; };
  long long sum = 0;
  for (int64_t i = i0; i < i1; ++i) sum += cost(i);
  __shared__ long long s_pre[1024];
  s_pre[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = t >= o ? s_pre[t - o] : 0;
    __syncthreads();
    s_pre[t] += v;
    __syncthreads();
  }
  const long long total = s_pre[1023];
  long long p = s_pre[t] - sum;                            // cost before this chunk
  for (int64_t i = i0; i < i1; ++i) {
    const long long c = cost(i);
    long long k = (p * G + total - 1) / total;             // smallest k with total * k / G >= p  (or one below: checked)
    for (; k < G && total * k / G < p + c; ++k)
      if (total * k / G >= p) seg[k] = (int)i;
    p += c;
  }
  if (t == 0) seg[G] = (int)n;
}

// census of the classes k_vcg_classify stored: out[0] ZERO, out[1] ONE, out[2] MIXED vectors (diagnostics / bench line)
template <int VEC>
__global__ void __launch_bounds__(256)
k_vcg_class_census(Compact c, unsigned long long* out) {
  const int Nx = c.N[0], Ny = c.N[1], nzv = c.N[2] / VEC;
  const int64_t total = (int64_t)(Nx - 2) * (Ny - 2) * nzv;
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (iv >= total) return;
  const int zv = (int)(iv % nzv), y = 1 + (int)((iv / nzv) % (Ny - 2)), x = 1 + (int)(iv / ((int64_t)nzv * (Ny - 2)));
  const unsigned k = (c.msk[c.idx(x, y, zv * VEC)] >> 4) & 3u;
  atomicAdd(out + (k == 1 ? 0 : (k == 2 ? 1 : 2)), 1ull);
}

template <typename T>
struct Vec3T { const T* p[3]; };

struct Box3 { int lo[3], hi[3]; };   // face-index box [lo, hi) of one row

// One operator row at face (x,y,z) on the compact arrays, written against a SAMPLER that hands out the
// row's operands: S::vol(p, ox, oy, oz) = volume class p at compact index (x+ox, y+oy, z+oz),
// S::vel(comp, dx, dy, dz) = velocity component at face (x+dx, y+dy, z+dz), S::tap_ok(...) = validity of
// that face.  The global sampler reads the arrays directly (every load unconditional, so the ~40 loads of
// a face are in flight together); the LDS sampler of the tiled kernel reads the staged plane tiles.
// MASK=false drops the tap masks: in the CG the operand is `d`, which is exactly 0 on solid and
// array-boundary faces, so they cannot matter; the initial q = A x (x = extrapolated velocity, non-zero
// on solid faces) and the stand-alone apply use MASK=true.  The row's OWN mask (solid face -> out = 0)
// always applies.
// NC cells at once: `smp` hands out the operands of cell j = 0 .. NC-1 (vol(j, p, ox, oy, oz), vel(j, comp, dx, dy, dz),
// tap_ok(j, ...)).  The statements are ordered tap-major, cell-minor, so the NC accumulation chains -- each a strictly
// dependent sequence of 15 fp64 operations -- sit interleaved in the instruction stream and hide each other's latency;
// per cell the operations and their order are exactly those of the one-cell form below (same rounding, bit for bit).
template <int AXIS, bool MASK, int NC, typename S>
__device__ __forceinline__ void vcg_row_n(const S& smp, double k1, double k2, const bool (&own_ok)[NC], double (&out)[NC],
                                          double (&own_out)[NC]) {
  // The arithmetic is spelled out -- contraction off, fused multiply-adds written where they are wanted -- so that
  // every kernel built on this function (one cell per lane, LDS tiles, x-marching vectors, boundary slabs) rounds
  // identically whatever else the compiler finds around the call: where `(diag * own) - (k vol) * nb` may fuse
  // either product into the subtraction, two kernels used to differ in the last bit.
#pragma clang fp contract(off)
  double vs[NC][7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int ax = kD0[AXIS][0] + kVolOff[k][0], ay = kD0[AXIS][1] + kVolOff[k][1], az = kD0[AXIS][2] + kVolOff[k][2];
    const int p = ((ax & 1) << 2) | ((ay & 1) << 1) | (az & 1);
#pragma unroll
    for (int j = 0; j < NC; ++j) vs[j][k] = smp.vol(j, p, fdiv2(ax), fdiv2(ay), fdiv2(az));
  }
  double val[NC], s[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) own_out[j] = smp.vel(j, AXIS, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < 6; ++k) {      // 2 * vol is exact: the fused form of `s + 2 vol` rounds like the reference's
    const bool two = kDiagFac[AXIS][k] == 2;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (k == 0) s[j] = two ? 2 * vs[j][1] : vs[j][1];
      else s[j] = two ? __builtin_fma(2.0, vs[j][k + 1], s[j]) : s[j] + vs[j][k + 1];
    }
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) val[j] = __builtin_fma(k1, s[j], vs[j][0]) * own_out[j];          // diag * v   (:268-269)
#pragma unroll
  for (int t = 0; t < 14; ++t) {
    const VTap tp = kTaps[AXIS][t];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      double nbv = smp.vel(j, tp.comp, tp.dx, tp.dy, tp.dz);
      // the tap's mask sample is the validity of the tapped face itself (see header)
      if (MASK) nbv = smp.tap_ok(j, tp.comp, tp.dx, tp.dy, tp.dz) ? nbv : 0.0;
      const double kv = (tp.fac == 2 ? k2 : k1) * vs[j][tp.vol];   // `scale * mu * vol` first, as the reference evaluates it
      val[j] = __builtin_fma(tp.sgn > 0 ? kv : -kv, nbv, val[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) out[j] = own_ok[j] ? val[j] : 0.0;
}

// one cell: S::vol(p, ox, oy, oz), S::vel(comp, dx, dy, dz), S::tap_ok(comp, dx, dy, dz)
template <typename S>
struct OneCellSampler {
  const S& s;
  __device__ __forceinline__ double vol(int, int p, int ox, int oy, int oz) const { return s.vol(p, ox, oy, oz); }
  __device__ __forceinline__ double vel(int, int comp, int dx, int dy, int dz) const { return s.vel(comp, dx, dy, dz); }
  __device__ __forceinline__ bool tap_ok(int, int comp, int dx, int dy, int dz) const { return s.tap_ok(comp, dx, dy, dz); }
};

template <int AXIS, bool MASK, typename S>
__device__ __forceinline__ double vcg_row_s(const S& smp, double k1, double k2, bool own_ok, double& own_out) {
  const OneCellSampler<S> one{smp};
  const bool ok[1] = {own_ok};
  double out[1], own[1];
  vcg_row_n<AXIS, MASK, 1>(one, k1, k2, ok, out, own);
  own_out = own[0];
  return out[0];
}

template <typename T>
struct GlobalSampler {
  const Compact& c;
  const Vec3T<T>& v;
  int x, y, z;
  __device__ __forceinline__ double vol(int p, int ox, int oy, int oz) const {
    return (double)((const T*)c.vol[p])[c.idx(x + ox, y + oy, z + oz)];
  }
  __device__ __forceinline__ int64_t fidx(int comp, int dx, int dy, int dz) const {
    const int cs1 = c.N[1] + (comp == 1), cs2 = c.N[2] + (comp == 2);
    return ((int64_t)(x + dx) * cs1 + (y + dy)) * cs2 + (z + dz);
  }
  __device__ __forceinline__ double vel(int comp, int dx, int dy, int dz) const {
    return (double)v.p[comp][fidx(comp, dx, dy, dz)];
  }
  __device__ __forceinline__ bool tap_ok(int comp, int dx, int dy, int dz) const {
    return ((c.msk[c.idx(x + dx, y + dy, z + dz)] >> comp) & 1) != 0;
  }
};

template <typename T, int AXIS, bool MASK>
__device__ __forceinline__ double vcg_row(const Compact& c, double k1, double k2, const Vec3T<T>& v, int x, int y,
                                          int z, double& own_out) {
  const GlobalSampler<T> smp{c, v, x, y, z};
  const bool own_ok = ((c.msk[c.idx(x, y, z)] >> AXIS) & 1) != 0;
  return vcg_row_s<AXIS, MASK>(smp, k1, k2, own_ok, own_out);
}

// The diagonal of the operator on the flat [u | v | w] layout (opt-in Jacobi loop): vcg_row_n's own expression
// diag = vol_c + k (fR R + fL L + fT T + fB B + fF F + fK K) (:268, :338, :408) on interior non-solid faces, 0 elsewhere.
template <typename T, int AXIS>
__global__ void __launch_bounds__(256) k_vcg_diag(Compact c, double k1, T* __restrict__ out) {
#pragma clang fp contract(off)
  const int s0 = c.N[0] + (AXIS == 0), s1 = c.N[1] + (AXIS == 1), s2 = c.N[2] + (AXIS == 2);
  const int64_t n = (int64_t)s0 * s1 * s2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % s2), y = (int)((i / s2) % s1), x = (int)(i / ((int64_t)s2 * s1));
  double dg = 0.0;
  const bool interior = x > 0 && x < s0 - 1 && y > 0 && y < s1 - 1 && z > 0 && z < s2 - 1;
  if (interior && ((c.msk[c.idx(x, y, z)] >> AXIS) & 1) != 0) {
    double vs[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int ax = kD0[AXIS][0] + kVolOff[k][0], ay = kD0[AXIS][1] + kVolOff[k][1], az = kD0[AXIS][2] + kVolOff[k][2];
      const int p = ((ax & 1) << 2) | ((ay & 1) << 1) | (az & 1);
      vs[k] = (double)((const T*)c.vol[p])[c.idx(x + fdiv2(ax), y + fdiv2(ay), z + fdiv2(az))];
    }
    double sm = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bool two = kDiagFac[AXIS][k] == 2;
      if (k == 0) sm = two ? 2 * vs[1] : vs[1];
      else sm = two ? __builtin_fma(2.0, vs[k + 1], sm) : sm + vs[k + 1];
    }
    dg = __builtin_fma(k1, sm, vs[0]);
  }
  out[i] = (T)dg;
}

// rows of ONE component over a box of faces (used for the three boundary slabs the
// fused kernel does not cover, and as the simple reference form)
template <typename T, int AXIS, bool MASK>
__global__ void __launch_bounds__(256)
k_vcg_apply_row(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ out, Box3 bx, double* __restrict__ partial,
                const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  const int s1 = c.N[1] + (AXIS == 1), s2 = c.N[2] + (AXIS == 2);
  const int i0 = bx.hi[0] - bx.lo[0], i1 = bx.hi[1] - bx.lo[1], i2 = bx.hi[2] - bx.lo[2];
  const int64_t nint = (int64_t)i0 * i1 * i2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < nint; it += stride) {
    const int z = bx.lo[2] + (int)(it % i2), y = bx.lo[1] + (int)((it / i2) % i1), x = bx.lo[0] + (int)(it / ((int64_t)i2 * i1));
    double own;
    const double val = vcg_row<T, AXIS, MASK>(c, k1, k2, v, x, y, z, own);
    const T o = (T)val;
    out[((int64_t)x * s1 + y) * s2 + z] = o;
    acc += own * (double)o;
  }
  const double tot = block_sum<256>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// the three boundary slabs (u at x = Nx-1, v at y = Ny-1, w at z = Nz-1) in ONE launch: blocks [0, g0) take
// the first box, [g0, g0+g1) the second, the rest the third (one launch instead of three per iteration)
template <typename T, int AXIS, bool MASK>
__device__ __forceinline__ double vcg_slab_rows(const Compact& c, double k1, double k2, const Vec3T<T>& v,
                                                T* __restrict__ out, const Box3& bx, int blk, int nblk) {
  const int s1 = c.N[1] + (AXIS == 1), s2 = c.N[2] + (AXIS == 2);
  const int i0 = bx.hi[0] - bx.lo[0], i1 = bx.hi[1] - bx.lo[1], i2 = bx.hi[2] - bx.lo[2];
  const int64_t nint = (int64_t)i0 * i1 * i2;
  const int64_t stride = (int64_t)nblk * blockDim.x;
  double acc = 0.0;
  for (int64_t it = (int64_t)blk * blockDim.x + threadIdx.x; it < nint; it += stride) {
    const int z = bx.lo[2] + (int)(it % i2), y = bx.lo[1] + (int)((it / i2) % i1), x = bx.lo[0] + (int)(it / ((int64_t)i2 * i1));
    double own;
    const double val = vcg_row<T, AXIS, MASK>(c, k1, k2, v, x, y, z, own);
    const T o = (T)val;
    out[((int64_t)x * s1 + y) * s2 + z] = o;
    acc += own * (double)o;
  }
  return acc;
}

// The slab rows of the FUSED loop (mfs_vcg_march.h, FUSE): every operand is d_j = (T)(r + beta d_{j-1}) formed on the
// fly from the two arrays (the stored d_j belongs to other blocks of the same launch), the row's own face gets its d_j
// stored and its x updated -- k_update_d's arithmetic, face by face.
template <typename T>
struct FusedSampler {
  const Compact& c;
  const Vec3T<T>& dp;      // d_{j-1}
  const Vec3T<T>& r;
  double beta;
  int x, y, z;
  __device__ __forceinline__ double vol(int p, int ox, int oy, int oz) const {
    return (double)((const T*)c.vol[p])[c.idx(x + ox, y + oy, z + oz)];
  }
  __device__ __forceinline__ int64_t fidx(int comp, int dx, int dy, int dz) const {
    const int cs1 = c.N[1] + (comp == 1), cs2 = c.N[2] + (comp == 2);
    return ((int64_t)(x + dx) * cs1 + (y + dy)) * cs2 + (z + dz);
  }
  __device__ __forceinline__ double vel(int comp, int dx, int dy, int dz) const {
    const int64_t i = fidx(comp, dx, dy, dz);
    return (double)(T)__builtin_fma(beta, (double)dp.p[comp][i], (double)r.p[comp][i]);
  }
  __device__ __forceinline__ bool tap_ok(int, int, int, int) const { return true; }
};

template <typename T, int AXIS>
__device__ __forceinline__ double vcg_slab_rows_fused(const Compact& c, double k1, double k2, const Vec3T<T>& dp,
                                                      const Vec3T<T>& r, double alpha, double beta, T* __restrict__ out,
                                                      T* __restrict__ dn, T* __restrict__ xs, const Box3& bx, int blk, int nblk) {
  const int s1 = c.N[1] + (AXIS == 1), s2 = c.N[2] + (AXIS == 2);
  const int i0 = bx.hi[0] - bx.lo[0], i1 = bx.hi[1] - bx.lo[1], i2 = bx.hi[2] - bx.lo[2];
  const int64_t nint = (int64_t)i0 * i1 * i2;
  const int64_t stride = (int64_t)nblk * blockDim.x;
  double acc = 0.0;
  for (int64_t it = (int64_t)blk * blockDim.x + threadIdx.x; it < nint; it += stride) {
    const int z = bx.lo[2] + (int)(it % i2), y = bx.lo[1] + (int)((it / i2) % i1), x = bx.lo[0] + (int)(it / ((int64_t)i2 * i1));
    const FusedSampler<T> smp{c, dp, r, beta, x, y, z};
    const bool own_ok = ((c.msk[c.idx(x, y, z)] >> AXIS) & 1) != 0;
    double own;
    const double val = vcg_row_s<AXIS, false>(smp, k1, k2, own_ok, own);
    const T o = (T)val;
    const int64_t i = ((int64_t)x * s1 + y) * s2 + z;
    out[i] = o;
    dn[i] = (T)own;
    xs[i] = (T)__builtin_fma(alpha, (double)dp.p[AXIS][i], (double)xs[i]);
    acc += own * (double)o;
  }
  return acc;
}

template <typename T, bool MASK>
__global__ void __launch_bounds__(256)
k_vcg_apply_slabs(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ ox, T* __restrict__ oy, T* __restrict__ oz,
                  Box3 b0, Box3 b1, Box3 b2, int g0, int g1, double* __restrict__ partial,
                  const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  const int b = blockIdx.x, g2 = gridDim.x - g0 - g1;
  double acc;
  if (b < g0) acc = vcg_slab_rows<T, 0, MASK>(c, k1, k2, v, ox, b0, b, g0);
  else if (b < g0 + g1) acc = vcg_slab_rows<T, 1, MASK>(c, k1, k2, v, oy, b1, b - g0, g1);
  else acc = vcg_slab_rows<T, 2, MASK>(c, k1, k2, v, oz, b2, b - g0 - g1, g2);
  const double tot = block_sum<256>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and
// its L2), so neighbouring tiles -- which read each other's halo rows -- would sit on different L2s and every
// halo line would be fetched from HBM twice.  Give each XCD a CONTIGUOUS range of logical tile ids instead.
__device__ __forceinline__ int xcd_logical_block(int on) {
  if (!on) return blockIdx.x;
  const int G = gridDim.x, nch = min(G, 8);
  const int xcd = blockIdx.x % nch, slot = blockIdx.x / nch;
  const int per = G / nch, extra = G - per * nch;           // the first `extra` XCDs own one more block
  return xcd * per + min(xcd, extra) + slot;
}

// The per-iteration kernel: all three rows of cell (x,y,z) by one thread, over the
// box x in [1,Nx-2], y in [1,Ny-2], z in [1,Nz-2] where every row is an interior
// face.  The three rows share most of their operands (27 distinct velocity and 16
// distinct volume samples instead of 45 + 21); written from the one tap table, the
// duplicates are merged by the compiler (all inputs are read-only __restrict__).
// Lanes run along z (coalesced rows), 4 rows of y per workgroup, marching over a
// chunk of x planes so the x-1 / x+1 operands are L1/L2 hits of the previous step.
template <typename T, bool MASK>
__global__ void __launch_bounds__(256)
k_vcg_apply_fused(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ ox, T* __restrict__ oy,
                  T* __restrict__ oz, int xchunk, int nbz, int nby, int xcd_order, double* __restrict__ partial,
                  const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  const int Nx = c.N[0], Ny = c.N[1], Nz = c.N[2];
  const int lb = xcd_logical_block(xcd_order);
  const int bz = lb % nbz, by = (lb / nbz) % nby, bxk = lb / (nbz * nby);
  const int zr = 1 + bz * 64 + (threadIdx.x & 63), yr = 1 + by * 4 + (threadIdx.x >> 6);
  const bool active = zr <= Nz - 2 && yr <= Ny - 2;
  const int z = min(zr, Nz - 2), y = min(yr, Ny - 2);      // inactive lanes recompute a valid cell, store nothing
  const int x0 = 1 + bxk * xchunk, x1 = min(x0 + xchunk, Nx - 1);
  double acc = 0.0;
  for (int x = x0; x < x1; ++x) {
    double o0, o1, o2;
    const double r0 = vcg_row<T, 0, MASK>(c, k1, k2, v, x, y, z, o0);
    const double r1 = vcg_row<T, 1, MASK>(c, k1, k2, v, x, y, z, o1);
    const double r2 = vcg_row<T, 2, MASK>(c, k1, k2, v, x, y, z, o2);
    if (active) {
      const T t0 = (T)r0, t1 = (T)r1, t2 = (T)r2;
      ox[((int64_t)x * Ny + y) * Nz + z] = t0;
      oy[((int64_t)x * (Ny + 1) + y) * Nz + z] = t1;
      oz[((int64_t)x * Ny + y) * (Nz + 1) + z] = t2;
      acc += o0 * (double)t0 + o1 * (double)t1 + o2 * (double)t2;
    }
  }
  const double tot = block_sum<256>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// The fused box AND the three boundary slabs in one launch (one kernel boundary less per CG iteration -- what counts on
// launch-bound grids): blocks [0, gmain) are k_vcg_apply_fused's, the rest k_vcg_apply_slabs'; same block -> partial-sum
// slot mapping as the two launches, so the dot product is bit-identical.
template <typename T, bool MASK>
__global__ void __launch_bounds__(256)
k_vcg_apply_all(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ ox, T* __restrict__ oy, T* __restrict__ oz,
                int xchunk, int nbz, int nby, int xcd_order, int gmain, Box3 b0, Box3 b1, Box3 b2, int g0, int g1,
                double* __restrict__ partial, const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  double acc = 0.0;
  if ((int)blockIdx.x < gmain) {
    const int Nx = c.N[0], Ny = c.N[1], Nz = c.N[2];
    int lb = blockIdx.x;
    if (xcd_order) {                                         // xcd_logical_block over the first gmain blocks
      const int nch = min(gmain, 8), xcd = lb % nch, slot = lb / nch, per = gmain / nch, extra = gmain - per * nch;
      lb = xcd * per + min(xcd, extra) + slot;
    }
    const int bz = lb % nbz, by = (lb / nbz) % nby, bxk = lb / (nbz * nby);
    const int zr = 1 + bz * 64 + (threadIdx.x & 63), yr = 1 + by * 4 + (threadIdx.x >> 6);
    const bool active = zr <= Nz - 2 && yr <= Ny - 2;
    const int z = min(zr, Nz - 2), y = min(yr, Ny - 2);
    const int x0 = 1 + bxk * xchunk, x1 = min(x0 + xchunk, Nx - 1);
    for (int x = x0; x < x1; ++x) {
      double o0, o1, o2;
      const double r0 = vcg_row<T, 0, MASK>(c, k1, k2, v, x, y, z, o0);
      const double r1 = vcg_row<T, 1, MASK>(c, k1, k2, v, x, y, z, o1);
      const double r2 = vcg_row<T, 2, MASK>(c, k1, k2, v, x, y, z, o2);
      if (active) {
        const T t0 = (T)r0, t1 = (T)r1, t2 = (T)r2;
        ox[((int64_t)x * Ny + y) * Nz + z] = t0;
        oy[((int64_t)x * (Ny + 1) + y) * Nz + z] = t1;
        oz[((int64_t)x * Ny + y) * (Nz + 1) + z] = t2;
        acc += o0 * (double)t0 + o1 * (double)t1 + o2 * (double)t2;
      }
    }
  } else {
    const int b = (int)blockIdx.x - gmain, g2 = (int)gridDim.x - gmain - g0 - g1;
    if (b < g0) acc = vcg_slab_rows<T, 0, MASK>(c, k1, k2, v, ox, b0, b, g0);
    else if (b < g0 + g1) acc = vcg_slab_rows<T, 1, MASK>(c, k1, k2, v, oy, b1, b - g0, g1);
    else acc = vcg_slab_rows<T, 2, MASK>(c, k1, k2, v, oz, b2, b - g0 - g1, g2);
  }
  const double tot = block_sum<256>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// The per-iteration kernel, LDS-staged: a workgroup owns a TY x TZ tile of cells and marches over a
// chunk of x planes.  The ten operand arrays of the operator (3 velocity components, 7 volume classes)
// are staged plane by plane into LDS as (TY+2) x (TZ+2) tiles -- one ring of three planes (x-1, x, x+1)
// per array -- with coalesced row-segment loads issued one plane ahead into registers; every one of the
// 43 samples a cell's three rows need is then an LDS read at a constant offset.  Global traffic per cell
// drops from 43 L2-served loads (the vector L1 cannot hold the working set of 12 waves) to the 10 streams
// plus halo.  Arithmetic, operand order and results are those of k_vcg_apply_fused (same vcg_row_s).
template <typename T, int TY, int TZ>
struct VTile {
  static constexpr int PY = TY + 2, PZ = TZ + 2, PLANE = PY * PZ, NARR = 10, BLOCK = TY * TZ;
  static constexpr int NSTG = (PLANE + BLOCK - 1) / BLOCK;          // staging elements per thread per array
  static constexpr size_t lds_bytes() { return (size_t)NARR * 3 * PLANE * sizeof(T); }
};

template <typename T, int TY, int TZ>
struct LdsSampler {
  typedef VTile<T, TY, TZ> V;
  const T* base;          // smem + this thread's centre (ty + 1, tz + 1)
  int slot[3];            // element offsets of the ring slots holding planes x-1, x, x+1
  // array ids: 0..2 velocity components, 2 + p volume class p (1..7)
  __device__ __forceinline__ double at(int arr, int ox, int oy, int oz) const {
    return (double)base[arr * 3 * V::PLANE + slot[ox + 1] + oy * V::PZ + oz];
  }
  __device__ __forceinline__ double vol(int p, int ox, int oy, int oz) const { return at(2 + p, ox, oy, oz); }
  __device__ __forceinline__ double vel(int comp, int dx, int dy, int dz) const { return at(comp, dx, dy, dz); }
  __device__ __forceinline__ bool tap_ok(int, int, int, int) const { return true; }
};

template <typename T, int TY, int TZ>
__global__ void __launch_bounds__(TY * TZ)
k_vcg_apply_tiled(Compact c, double k1, double k2, Vec3T<T> v, T* __restrict__ ox, T* __restrict__ oy,
                  T* __restrict__ oz, int xchunk, int nbz, int nby, int xcd_order, double* __restrict__ partial,
                  const double* __restrict__ done_flag) {
  typedef VTile<T, TY, TZ> V;
  if (done_flag && *done_flag != 0.0) return;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* const smem = reinterpret_cast<T*>(smem_raw);
  const int Nx = c.N[0], Ny = c.N[1], Nz = c.N[2];
  const int tid = threadIdx.x;
  const int lb = xcd_logical_block(xcd_order);
  const int bz = lb % nbz, by = (lb / nbz) % nby, bxk = lb / (nbz * nby);
  const int y0 = 1 + by * TY, z0 = 1 + bz * TZ;               // first cell of the tile
  const int ty = tid / TZ, tz = tid - ty * TZ;
  const int yr = y0 + ty, zr = z0 + tz;
  const bool active = zr <= Nz - 2 && yr <= Ny - 2;
  const int y = min(yr, Ny - 2), z = min(zr, Nz - 2);          // mask / store addresses of inactive lanes stay valid
  const int x0 = 1 + bxk * xchunk, x1 = min(x0 + xchunk, Nx - 1);
  // the ten arrays: base pointer, rows per plane, row length
  const T* ap[V::NARR];
  int d1[V::NARR], d2[V::NARR];
#pragma unroll
  for (int a = 0; a < 3; ++a) { ap[a] = v.p[a]; d1[a] = Ny + (a == 1); d2[a] = Nz + (a == 2); }
  int q1[V::NARR], q2[V::NARR];                               // storage pitches (rows per plane, elements per row)
#pragma unroll
  for (int a = 0; a < 3; ++a) { q1[a] = d1[a]; q2[a] = d2[a]; }
#pragma unroll
  for (int p = 1; p < 8; ++p) {
    ap[2 + p] = (const T*)c.vol[p]; d1[2 + p] = c.dim(p, 1); d2[2 + p] = c.dim(p, 2); q1[2 + p] = c.py; q2[2 + p] = c.pz;
  }
  // staging slots of this thread: tile element e = tid + k * BLOCK  <->  (y0 - 1 + e / PZ, z0 - 1 + e % PZ)
  int se[V::NSTG], sgy[V::NSTG], sgz[V::NSTG];
#pragma unroll
  for (int k = 0; k < V::NSTG; ++k) {
    se[k] = tid + k * V::BLOCK;
    const int ly = se[k] / V::PZ;
    sgy[k] = y0 - 1 + ly;
    sgz[k] = z0 - 1 + (se[k] - ly * V::PZ);
  }
  T stg[V::NARR][V::NSTG];
  auto stage_load = [&](int X) {
#pragma unroll
    for (int a = 0; a < V::NARR; ++a) {
#pragma unroll
      for (int k = 0; k < V::NSTG; ++k) {
        const bool ok = se[k] < V::PLANE && sgy[k] < d1[a] && sgz[k] < d2[a];
        // clamped address, unconditional load: the prefetch stays free of control flow
        const int gy = min(sgy[k], d1[a] - 1), gz = min(sgz[k], d2[a] - 1);
        const T val = ap[a][((int64_t)X * q1[a] + gy) * q2[a] + gz];
        stg[a][k] = ok ? val : (T)0;
      }
    }
  };
  auto stage_store = [&](int slot_off) {
#pragma unroll
    for (int a = 0; a < V::NARR; ++a) {
#pragma unroll
      for (int k = 0; k < V::NSTG; ++k)
        if (se[k] < V::PLANE) smem[a * 3 * V::PLANE + slot_off + se[k]] = stg[a][k];
    }
  };
  LdsSampler<T, TY, TZ> smp;
  smp.base = smem + (ty + 1) * V::PZ + tz + 1;
  smp.slot[0] = 0; smp.slot[1] = V::PLANE; smp.slot[2] = 2 * V::PLANE;
  // prologue: planes x0-1, x0, x0+1
  stage_load(x0 - 1); stage_store(smp.slot[0]);
  stage_load(x0);     stage_store(smp.slot[1]);
  stage_load(x0 + 1); stage_store(smp.slot[2]);
  MFS_VISC_LDS_BARRIER();
  double acc = 0.0;
  for (int x = x0; x < x1; ++x) {
    const bool more = x + 1 < x1;
    if (more) stage_load(x + 2);                               // in flight while this plane is computed
    const int64_t f0 = ((int64_t)x * Ny + y) * Nz + z, f1 = ((int64_t)x * (Ny + 1) + y) * Nz + z,
                  f2 = ((int64_t)x * Ny + y) * (Nz + 1) + z;
    const unsigned fm = c.msk[c.idx(x, y, z)];
    const bool ok0 = (fm & 1) != 0, ok1 = (fm & 2) != 0, ok2 = (fm & 4) != 0;
    double o0, o1, o2;
    const double r0 = vcg_row_s<0, false>(smp, k1, k2, ok0, o0);
    const double r1 = vcg_row_s<1, false>(smp, k1, k2, ok1, o1);
    const double r2 = vcg_row_s<2, false>(smp, k1, k2, ok2, o2);
    if (active) {
      const T t0 = (T)r0, t1 = (T)r1, t2 = (T)r2;
      ox[f0] = t0;
      oy[f1] = t1;
      oz[f2] = t2;
      acc += o0 * (double)t0 + o1 * (double)t1 + o2 * (double)t2;
    }
    if (more) {
      MFS_VISC_LDS_BARRIER();                                  // everyone is done with plane x-1's slot
      stage_store(smp.slot[0]);                                // ... which now receives plane x+2
      MFS_VISC_LDS_BARRIER();
      const int s0 = smp.slot[0];
      smp.slot[0] = smp.slot[1]; smp.slot[1] = smp.slot[2]; smp.slot[2] = s0;
    }
  }
  const double tot = block_sum<V::BLOCK>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

}  // namespace mfs

#include "mfs_vcg_march.h"
#include "mfs_vcg_resident.h"

using namespace mfs;

struct mfs_vcg3d {
  G3 g;
  int dt;
  int64_t nf[3], off[3], n;
  CgCore c;
  char* ws;
  size_t ws_bytes;
  Compact cp;
  double k1, k2;
  bool is_setup;
  int grid_row;
  int mask_cg;   // 1: keep the tap masks in CG applies too (debug / A-B)
  int tiled;     // 1: CG applies use the LDS-staged kernel (default); 0: the direct-load fused kernel
  int xchunk_tiled;
  int xcd_order;   // 1 / 0: XCD-contiguous tile order on / off; -1 auto
  int split_x;     // 1: the x update rides in the direction-update kernel (default)
  int skip_top_x;  // slab decomposition: the u row at x = Nx-1 belongs to the right neighbour (a ghost here)
  int merge_slabs; // 1 / 0: the fused box and the boundary slabs in one launch (default) / two
  int march;       // 1 (default): CG applies run the x-marching vector kernel (mfs_vcg_march.h) where the grid allows it
  int march_bpc;   // its workgroups per CU (2: what its register budget makes resident)
  int march_vec;   // experiment: 2 = 8-byte vectors for fp32 state
  mfs_p2p* p2p;    // window transport of the slab loop (mfs_vcg3d_attach_p2p); null: the caller moves halos / scalars
  int64_t last_iters;   // iterations of this engine's previous converged solve (sizes the first batch of the next one)
  int jacobi;      // 1: opt-in Jacobi-preconditioned loop (mfs_vcg3d_set_jacobi; NOT the reference's iteration)
  void* diag;      // its diagonal (n elements, built by setup when the flag is on)
  double* part_rz; // its r.z partials
  bool diag_ready;
  // small grids: the whole loop as one resident launch per batch (mfs_vcg_resident.h): 1 / 0 / -1 auto (MFS_VISC_RESIDENT)
  int resident;
  VResPlan res;
  u64 *res_ar, *res_mirror;
  unsigned res_epoch;                                  // episode tags handed out so far
  u64 res_timeout_ticks, res_first_timeout_ticks;      // read once at creation
  int res_drop_wg;                                     // MFS_VRES_TEST_DROP_WG: fault injection, tests only
  int res_creg;                                        // MFS_VRES_CREG (default 1): volume samples in registers where they fit
  int sparse_vec;       // 1 (default; MFS_VISC_SPARSE): the vector phases of a single-domain solve sweep live chunks only
  int64_t sparse_min;   // ... from this many unknowns on (smaller grids run the resident / merged loops)
  int* live_ws;         // flags | list | count
  bool classes_ready;   // the compressed class access' classes / tile flags / balanced cut match the current set-up
  int *list_items, *list_runrem, *list_count;      // work list of the loop's march launches (Compact::items), built with the classes
  bool list_ready;
  int compress;    // 1 (default; MFS_VISC_COMPRESS): the march reads the class arrays only for MIXED vectors (k_vcg_classify)
  int march_nt;    // MFS_VISC_MARCH_NT: nontemporal class loads / q stores (-1 auto by size), read at creation
  int fuse;        // 1: mfs_vcg3d_iterate / solve fold the direction and x updates into the march (2 launches per iteration); default 0
  void* d2;        // ping-pong partner of the bound d for that loop (n elements, zero outside the faces the loop writes)
  bool fused_run;  // the fused loop has run since begin: x lags by one update, d_j may sit in d2 (vcg_home settles both)
};

// elements one class / mask array occupies (uniform pitches, see struct Compact)
static int64_t class_count(const int64_t gres[3], int /*p*/) {
  return (gres[0] + 1) * (gres[1] + 1) * (gres[2] + 4);
}

// byte distance between consecutive class arrays in the workspace.  The pad staggers the seven streams a workgroup reads at
// the same in-plane offset (MFS_VISC_CLASS_PAD: A/B knob, a multiple of 256; read at every call, so keep it fixed in a process)
static size_t class_stride_bytes(const int64_t gres[3], int dt) {
  const size_t pad = (size_t)std::max(256, env_int("MFS_VISC_CLASS_PAD", 4096)) / 256 * 256;
  return align_up((size_t)class_count(gres, 0) * dtype_size(dt), 4096) + pad;
}

constexpr int kVmMaxSegs = 4096;       // workgroups of a march launch, at most (cus x blocks per CU)

// bytes of the marching kernel's tile words: (tiles of >= 256 vectors per plane) x Nx
static size_t tile_words_bytes(const int64_t gres[3], int dt) {
  const int64_t vec = dt == MFS_F32 ? 4 : 2;
  const int64_t tiles = (gres[1] * (gres[2] / vec + 1) + 255) / 256 + 1;
  return align_up((size_t)(tiles * (gres[0] + 1)), 4096);
}

static int check_gres(const int64_t gres[3]) {
  MFS_REQUIRE(gres != nullptr, "gres is null");
  for (int a = 0; a < 3; ++a) MFS_REQUIRE(gres[a] >= 1 && gres[a] <= 2048, "grid resolution out of range [1,2048]");
  return MFS_OK;
}

static G3 make_g(const int64_t gres[3]) {
  G3 g;
  for (int a = 0; a < 3; ++a) g.N[a] = (int)gres[a];
  return g;
}

template <bool RHS>
static int launch_rows_direct(const int64_t gres[3], double scale, double mu, const void* vx, const void* vy,
                              const void* vz, int v_dt, void* ox, void* oy, void* oz, int o_dt, const void* sphi,
                              int sphi_dt, const void* vol, int vol_dt, hipStream_t st) {
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(vx && vy && vz && ox && oy && oz && sphi && vol, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(o_dt) && dtype_ok(sphi_dt) && dtype_ok(vol_dt), "dtype");
  G3 g = make_g(gres);
  V3 v{{vx, vy, vz}};
  hipLaunchKernelGGL((k_visc_row_direct<0, RHS>), dim3(cdiv(g.nface(0), 256)), dim3(256), 0, st, g, scale, mu, v, v_dt,
                     ox, o_dt, sphi, sphi_dt, vol, vol_dt);
  hipLaunchKernelGGL((k_visc_row_direct<1, RHS>), dim3(cdiv(g.nface(1), 256)), dim3(256), 0, st, g, scale, mu, v, v_dt,
                     oy, o_dt, sphi, sphi_dt, vol, vol_dt);
  hipLaunchKernelGGL((k_visc_row_direct<2, RHS>), dim3(cdiv(g.nface(2), 256)), dim3(256), 0, st, g, scale, mu, v, v_dt,
                     oz, o_dt, sphi, sphi_dt, vol, vol_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

// the three slabs of interior faces outside the fused box: u at x = Nx-1, v at y = Ny-1, w at z = Nz-1
static Box3 vslab_box(const mfs_vcg3d* h, int ax) {
  Box3 b;
  for (int a = 0; a < 3; ++a) { b.lo[a] = 1; b.hi[a] = h->g.sh(ax, a) - 1; }
  b.lo[ax] = h->g.N[ax] - 1;
  b.hi[ax] = h->g.N[ax];
  return b;
}
static int vslab_grid(const Box3& b) {
  const int64_t n = (int64_t)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
  return (int)std::max<int64_t>(1, std::min<int64_t>(256, (n + 255) / 256));
}

// can the x-marching vector kernel serve this engine / these operands?  (rows of whole 16-byte vectors, 16-byte aligned
// vectors, one halo vector per thread, the ring of plane buffers within the CU's LDS: two workgroups per CU up to 80 KB
// each, one beyond that -- measured at 256^3 fp32: one workgroup per CU runs within 7 % of two, the kernel is paced
// by its instruction stream, not by occupancy)
constexpr size_t kVmMaxLds = 152 * 1024;
// geometry of the march for this engine: threads per workgroup (= z-vectors per tile) * 10 + ring slots; 0: rows too long.
// 256 x 4 wherever two such workgroups share a CU; 512 x 4 where only one 256-vector workgroup would fit anyway (fp64 from
// Nz ~ 150, fp32 from Nz ~ 300); 512 x 3 (a second barrier per plane) where even that ring exceeds the LDS (fp64 380 < Nz <= 512)
template <typename T, int VEC>
static int vcg_march_geom(const mfs_vcg3d* h) {
  const int Nz = h->g.N[2], nzv2 = 2 * (Nz / VEC);
  const int knob = env_int("MFS_VISC_MARCH_BLOCK", 0);          // A/B: 256 / 512 / 5123 force, 0 auto
  const bool f2564 = nzv2 <= 256 && vm_lds_bytes<T, VEC, 256, 4>(Nz) <= kVmMaxLds;
  const bool f5124 = nzv2 <= 512 && vm_lds_bytes<T, VEC, 512, 4>(Nz) <= kVmMaxLds;
  const bool f5123 = nzv2 <= 512 && vm_lds_bytes<T, VEC, 512, 3>(Nz) <= kVmMaxLds;
  if (knob == 256 && f2564) return 2564;
  if (knob == 512 && f5124) return 5124;
  if (knob == 5123 && f5123) return 5123;
  if (f2564 && vm_lds_bytes<T, VEC, 256, 4>(Nz) <= 80 * 1024) return 2564;
  if (f5124) return 5124;
  if (f2564) return 2564;
  if (f5123) return 5123;
  return 0;
}

template <typename T, int VEC = VecOf<T>::N>
static bool vcg_march_ok(const mfs_vcg3d* h, const void* v, const void* out) {
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2];
  if (!h->march || Nx < 3 || Ny < 3 || Nz % VEC != 0 || Nz < 2 * VEC) return false;
  if (vcg_march_geom<T, VEC>(h) == 0) return false;
  if (((uintptr_t)v % 16) != 0 || ((uintptr_t)out % 16) != 0) return false;
  if ((int64_t)(Ny + 1) * (Nz + 4) > (int64_t)0x7fffffff / 2) return false;     // 32-bit in-plane offsets
  return true;
}

template <typename T, int VEC, int WAVES, int NT, bool FUSE = false, int BLOCK = kVmBlock, int RING = kVmRing, bool COMP = false>
static int vcg_march_launch_blk(mfs_vcg3d* h, const Vec3T<T>& vv, T* ob, double* partial, const double* done, hipStream_t st,
                                int* nparts, const VmFuse<T>* fz) {
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2];
  const int ipp = (Ny - 2) * (Nz / VEC), tiles = (ipp + BLOCK - 1) / BLOCK;
  const int64_t total = (int64_t)tiles * (Nx - 2);
  const size_t lds = vm_lds_bytes<T, VEC, BLOCK, RING>(Nz);
  const int bpc = lds > 80 * 1024 ? 1 : h->march_bpc;
  const int gmain = (int)std::max<int64_t>(1, std::min<int64_t>(total, (int64_t)h->c.cus * bpc));
  const Box3 b0 = vslab_box(h, 0), b1 = vslab_box(h, 1), b2 = vslab_box(h, 2);
  const int g0 = h->skip_top_x ? 0 : vslab_grid(b0), g1 = vslab_grid(b1), g2 = vslab_grid(b2);
  static bool attr_set = false;        // per instantiation: more than the default 64 KB of dynamic LDS
  if (!attr_set) {
    MFS_HIP_TRY(hipFuncSetAttribute((const void*)k_vcg_apply_march<T, VEC, WAVES, NT, FUSE, BLOCK, RING, COMP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kVmMaxLds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_vcg_apply_march<T, VEC, WAVES, NT, FUSE, BLOCK, RING, COMP>), dim3(gmain + g0 + g1 + g2), dim3(BLOCK), lds, st, h->cp, h->k1,
                     h->k2, vv, ob + h->off[0], ob + h->off[1], ob + h->off[2], gmain, b0, b1, b2, g0, g1, partial, done,
                     fz ? *fz : VmFuse<T>{});
  MFS_LAUNCH_CHECK();
  *nparts = gmain + g0 + g1 + g2;
  return MFS_OK;
}

template <typename T, int VEC, int WAVES, int NT, bool FUSE = false, bool COMP = false>
static int vcg_march_launch_nt(mfs_vcg3d* h, const Vec3T<T>& vv, T* ob, double* partial, const double* done, hipStream_t st,
                               int* nparts, const VmFuse<T>* fz = nullptr) {
  if constexpr (VEC == VecOf<T>::N && WAVES == MFS_VMARCH_MIN_WAVES) {
    const int geom = vcg_march_geom<T, VEC>(h);
    if (geom == 5124) return vcg_march_launch_blk<T, VEC, WAVES, NT, FUSE, 512, 4, COMP>(h, vv, ob, partial, done, st, nparts, fz);
    if constexpr (!FUSE) {
      if (geom == 5123) return vcg_march_launch_blk<T, VEC, WAVES, NT, FUSE, 512, 3, COMP>(h, vv, ob, partial, done, st, nparts, fz);
    }
  }
  return vcg_march_launch_blk<T, VEC, WAVES, NT, FUSE, 256, 4, COMP>(h, vv, ob, partial, done, st, nparts, fz);
}

// ---- live chunks of the flat CG vectors (mfs_cg_core.h LiveMap): a face is DEAD when its operator row is empty -- an
// array-boundary face, a solid face, or all seven volume samples of its row 0 -- and r = d = 0 there at the start of the
// loop (b - A x0 = 0): then q, r and d stay exactly 0 and x never changes.  A chunk of 32 unknowns is live when any of
// its faces is not dead.  Built once per solve (single-domain loops), behind the initial residual.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_vcg_live_flags(Compact c, const T* __restrict__ r, const T* __restrict__ d, int64_t n, int64_t o1, int64_t o2, int* __restrict__ flags) {
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i0 = iv * VEC;
  bool live = false;
  if (i0 < n) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int64_t i = i0 + j;
      if (i >= n) break;
      if (r[i] != (T)0 || d[i] != (T)0) { live = true; continue; }
      const int comp = i >= o2 ? 2 : (i >= o1 ? 1 : 0);
      const int64_t f = i - (comp == 2 ? o2 : (comp == 1 ? o1 : 0));
      const int s1 = c.N[1] + (comp == 1), s2 = c.N[2] + (comp == 2), s0 = c.N[0] + (comp == 0);
      const int z = (int)(f % s2), y = (int)((f / s2) % s1), x = (int)(f / ((int64_t)s2 * s1));
      if (x < 1 || x > s0 - 2 || y < 1 || y > s1 - 2 || z < 1 || z > s2 - 2) continue;          // array boundary: never computed
      if (!((c.msk[c.idx(x, y, z)] >> comp) & 1)) continue;                                       // solid face: q = 0
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int ax = kD0[comp][0] + kVolOff[k][0], ay = kD0[comp][1] + kVolOff[k][1], az = kD0[comp][2] + kVolOff[k][2];
        const int p = ((ax & 1) << 2) | ((ay & 1) << 1) | (az & 1);
        if (((const T*)c.vol[p])[c.idx(x + fdiv2(ax), y + fdiv2(ay), z + fdiv2(az))] != (T)0) live = true;
      }
    }
  }
  if (live) flags[i0 / kLiveChunk] = 1;      // (a vector never straddles a chunk: both are multiples of VEC unknowns)
}

static int vcg_build_live(mfs_vcg3d* h, hipStream_t st) {
  h->c.live = LiveMap{nullptr, nullptr, 0};
  if (!h->sparse_vec || !core_vec_ok(h->c) || h->n < h->sparse_min) return MFS_OK;
  const int vec = h->dt == MFS_F32 ? 4 : 2;
  const int nchunks = (int)((h->n + kLiveChunk - 1) / kLiveChunk);
  int* flags = h->live_ws;
  int* list = flags + nchunks;
  int* count = list + nchunks;
  MFS_HIP_TRY(hipMemsetAsync(flags, 0, (size_t)nchunks * sizeof(int), st));
  const int64_t nvec = (h->n + vec - 1) / vec;
  if (h->dt == MFS_F32)
    hipLaunchKernelGGL((k_vcg_live_flags<float, 4>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, h->cp, (const float*)h->c.r, (const float*)h->c.d, h->n, h->off[1], h->off[2], flags);
  else
    hipLaunchKernelGGL((k_vcg_live_flags<double, 2>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, h->cp, (const double*)h->c.r, (const double*)h->c.d, h->n, h->off[1], h->off[2], flags);
  if (int e = core_compact_flags<int>(flags, nchunks, list, count, count + 64, st)) return e;
  MFS_LAUNCH_CHECK();
  int shift = 0;
  while ((1 << shift) < kLiveChunk / vec) ++shift;
  h->c.live = LiveMap{list, count, shift};
  return MFS_OK;
}

// classes, tile flags and the cost-balanced cut of the marching kernel's compressed class access, once per set-up, at the
// first launch that wants them (grids that run the resident small-grid loop never do: four launches less per solve)
static int vcg_build_classes(mfs_vcg3d* h, hipStream_t stream) {
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2];
  unsigned char* mp = (unsigned char*)h->cp.msk;
  h->classes_ready = true;
  h->list_ready = false;
  h->cp.tw_block = 0;
  h->cp.seg_g = 0;
  if (Nx >= 3 && Ny >= 3) {    // classes of the interior z-vectors for the march's compressed class access
    const int vec = h->dt == MFS_F32 ? 4 : 2;
    if (Nz % vec == 0 && Nz >= 2 * vec) {
      MFS_HIP_TRY(hipMemsetAsync((void*)h->cp.bulk, 0, 8, stream));
      const int gb = (int)std::min<int64_t>(1024, cdiv(h->cp.stored(), 256));
      if (h->dt == MFS_F32) hipLaunchKernelGGL((k_vcg_bulk_value<float>), dim3(gb), dim3(256), 0, stream, h->cp, (float*)h->cp.bulk);
      else hipLaunchKernelGGL((k_vcg_bulk_value<double>), dim3(gb), dim3(256), 0, stream, h->cp, (double*)h->cp.bulk);
      const int64_t nvec = (int64_t)(Nx - 2) * (Ny - 2) * (Nz / vec);
      if (h->dt == MFS_F32) hipLaunchKernelGGL((k_vcg_classify<float, 4>), dim3(cdiv(nvec, 256)), dim3(256), 0, stream, h->cp, mp);
      else hipLaunchKernelGGL((k_vcg_classify<double, 2>), dim3(cdiv(nvec, 256)), dim3(256), 0, stream, h->cp, mp);
      MFS_LAUNCH_CHECK();
      // ... and the tile words, for the tile size the march will run with on this grid
      const int geom = h->dt == MFS_F32 ? vcg_march_geom<float, 4>(h) : vcg_march_geom<double, 2>(h);
      h->cp.tw_block = geom / 10;
      if (h->cp.tw_block > 0) {
        const int ipp = (Ny - 2) * (Nz / vec), tiles = (ipp + h->cp.tw_block - 1) / h->cp.tw_block;
        const int64_t nt = (int64_t)tiles * Nx;
        const int64_t gr[3] = {Nx, Ny, Nz};
        MFS_REQUIRE((size_t)nt <= tile_words_bytes(gr, h->dt), "tile flags do not fit their workspace");
        unsigned char* const flags = (unsigned char*)h->cp.tw;
        if (vec == 4) hipLaunchKernelGGL((k_vcg_tile_flags<4>), dim3(cdiv(nt * kWave, 256)), dim3(256), 0, stream, h->cp, h->cp.tw_block, flags);
        else hipLaunchKernelGGL((k_vcg_tile_flags<2>), dim3(cdiv(nt * kWave, 256)), dim3(256), 0, stream, h->cp, h->cp.tw_block, flags);
        MFS_LAUNCH_CHECK();
        // the cost-balanced cut for the grid the march will be launched with on this engine
        const size_t ldsb = h->dt == MFS_F32 ? (size_t)(geom % 10) * 3 * (2 * Nz + h->cp.tw_block * 4) * 4
                                             : (size_t)(geom % 10) * 3 * (2 * Nz + h->cp.tw_block * 2) * 8;
        const int bpc = ldsb > 80 * 1024 ? 1 : h->march_bpc;
        const int64_t total = (int64_t)tiles * (Nx - 2);
        const int G = (int)std::max<int64_t>(1, std::min<int64_t>(total, (int64_t)h->c.cus * bpc));
        h->cp.seg_g = 0;
        if (G <= kVmMaxSegs && total < 0x7fffffff) {
          hipLaunchKernelGGL(k_vcg_balance, dim3(1), dim3(1024), 0, stream, flags, tiles, Nx, G, (int*)h->cp.seg);
          MFS_LAUNCH_CHECK();
          h->cp.seg_g = G;
        }
        // the work list of the CG loop's launches: the busy pairs only (an all-air pair's q is +0 and stays what the
        // initial q = A x stored there)
        if (nt < 0x7fffffff && core_compact_scratch_ints(nt) + 16 <= 4096 / sizeof(int)) {      // (the scratch shares the count's page)
          if (int e = core_compact_flags<unsigned char>(flags, (int)nt, h->list_items, h->list_count, h->list_count + 16, stream)) return e;
          hipLaunchKernelGGL(k_list_runs, dim3(cdiv(nt, 256)), dim3(256), 0, stream, h->list_items, h->list_count, Nx, h->list_runrem);
          MFS_LAUNCH_CHECK();
          h->list_ready = true;
        }
      }
    }
  }
  return MFS_OK;
}

template <typename T, int VEC, int WAVES>
static int vcg_march_launch(mfs_vcg3d* h, const Vec3T<T>& vv, T* ob, double* partial, const double* done, hipStream_t st,
                            int* nparts) {
  // nontemporal class loads / q stores once operand, result and classes (13 arrays) exceed the Infinity Cache
  // (MFS_VISC_MARCH_NT = 0 / 1 overrides, read at engine creation)
  const int knob = h->march_nt;
  const bool nt = knob < 0 ? (13.0 * (double)h->g.N[0] * h->g.N[1] * h->g.N[2] * sizeof(T) > 200e6) : (knob != 0);
  if constexpr (VEC == VecOf<T>::N && WAVES == MFS_VMARCH_MIN_WAVES) {
    // compressed class access (classes, tile flags and the balanced cut built once per set-up, for this tile size)
    if (h->compress && !h->classes_ready) { if (int e = vcg_build_classes(h, st)) return e; }
    if (h->compress && h->cp.tw_block > 0 && h->cp.tw_block == vcg_march_geom<T, VEC>(h) / 10)
      return nt ? vcg_march_launch_nt<T, VEC, WAVES, 5, false, true>(h, vv, ob, partial, done, st, nparts)
                : vcg_march_launch_nt<T, VEC, WAVES, 0, false, true>(h, vv, ob, partial, done, st, nparts);
  }
  return nt ? vcg_march_launch_nt<T, VEC, WAVES, 5>(h, vv, ob, partial, done, st, nparts)
            : vcg_march_launch_nt<T, VEC, WAVES, 0>(h, vv, ob, partial, done, st, nparts);
}

// the stencil launch of fused iteration j >= 1: q = A d_j with d_j = r + beta d_{j-1} formed on the fly and stored to
// `d_cur`, x += alpha d_{j-1} for every owned face (mfs_vcg_march.h, FUSE)
template <typename T>
static int vcg_march_fused(mfs_vcg3d* h, const void* d_prev, void* d_cur, hipStream_t st, int* nparts) {
  constexpr int VEC = VecOf<T>::N;
  const T* dp = (const T*)d_prev;
  T* dc = (T*)d_cur;
  T* xb = (T*)h->c.x;
  const T* rb = (const T*)h->c.r;
  Vec3T<T> vv{{dp + h->off[0], dp + h->off[1], dp + h->off[2]}};
  VmFuse<T> fz;
  for (int a = 0; a < 3; ++a) { fz.r[a] = rb + h->off[a]; fz.dn[a] = dc + h->off[a]; fz.x[a] = xb + h->off[a]; }
  fz.scal = h->c.scal;
  const int knob = h->march_nt;
  const bool nt = knob < 0 ? (13.0 * (double)h->g.N[0] * h->g.N[1] * h->g.N[2] * sizeof(T) > 200e6) : (knob != 0);
  return nt ? vcg_march_launch_nt<T, VEC, MFS_VMARCH_MIN_WAVES, 5, true>(h, vv, (T*)h->c.q, h->c.part_dq, h->c.scal + S_DONE, st, nparts, &fz)
            : vcg_march_launch_nt<T, VEC, MFS_VMARCH_MIN_WAVES, 0, true>(h, vv, (T*)h->c.q, h->c.part_dq, h->c.scal + S_DONE, st, nparts, &fz);
}

template <typename T, bool MASK>
static int vcg_apply_TM(mfs_vcg3d* h, const void* v, void* out, double* partial, const double* done, hipStream_t st,
                        int* nparts) {
  const T* vb = (const T*)v;
  T* ob = (T*)out;
  Vec3T<T> vv{{vb + h->off[0], vb + h->off[1], vb + h->off[2]}};
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2];
  int used = 0;
  if constexpr (sizeof(T) == 4) if (!MASK && !h->tiled && h->march_vec == 2 && vcg_march_ok<T, 2>(h, v, out))
    return vcg_march_launch<T, 2, 3>(h, vv, ob, partial, done, st, nparts);      // experiment: 8-byte vectors, 3 waves / SIMD
  if (!MASK && !h->tiled && vcg_march_ok<T>(h, v, out))
    return vcg_march_launch<T, VecOf<T>::N, MFS_VMARCH_MIN_WAVES>(h, vv, ob, partial, done, st, nparts);
  // (1) the box where all three rows are interior faces: one fused launch
  if (Nx >= 3 && Ny >= 3 && Nz >= 3) {
    const int nbz = (Nz - 2 + 63) / 64, nby = (Ny - 2 + 3) / 4;
    int xchunk = 8;
    while ((int64_t)nbz * nby * ((Nx - 2 + xchunk - 1) / xchunk) > 4096) xchunk *= 2;
    const int grid = nbz * nby * ((Nx - 2 + xchunk - 1) / xchunk);
    // XCD-contiguous tile order pays while the operands are Infinity-Cache resident (128^3: -9 %); beyond that the
    // plain order is faster (256^3: +7 % with it) -- auto unless MFS_VISC_XCD is set
    const double ws = 13.0 * (double)Nx * Ny * Nz * sizeof(T);
    const int xcd = h->xcd_order < 0 ? (ws < 200e6 ? 1 : 0) : h->xcd_order;
    if (!MASK && h->tiled) {
      // LDS-staged tiles: 4 x 64 cells (fp32) / 4 x 32 (fp64) so that three planes of ten arrays fit 48 KB
      constexpr int TZ = sizeof(T) == 4 ? 64 : 32;
      const int tbz = (Nz - 2 + TZ - 1) / TZ;
      int xc = h->xchunk_tiled;
      while ((int64_t)tbz * nby * ((Nx - 2 + xc - 1) / xc) > 4096) xc *= 2;
      const int tgrid = tbz * nby * ((Nx - 2 + xc - 1) / xc);
      typedef VTile<T, 4, TZ> Tile;
      const size_t lds = Tile::lds_bytes();
      hipLaunchKernelGGL((k_vcg_apply_tiled<T, 4, TZ>), dim3(tgrid), dim3(4 * TZ), lds, st, h->cp,
                         h->k1, h->k2, vv, ob + h->off[0], ob + h->off[1], ob + h->off[2], xc, tbz, nby, xcd, partial + used,
                         done);
      used += tgrid;
    } else if (h->merge_slabs != 0) {
      // in-process A/B (tools/visc_ab.py), us per iteration, two launches -> one: 64^3 fp64 35.0 -> 28.0, 128^3 fp32
      // 70.7 -> 67.4, 192^3 244.8 -> 250.0, 256^3 681 -> 672 (run-to-run and box-to-box spread at 256^3: 618 .. 685)
      // one launch for the box and the three boundary slabs (below): same partial slots as the two launches
      const Box3 b0 = vslab_box(h, 0), b1 = vslab_box(h, 1), b2 = vslab_box(h, 2);
      const int g0 = h->skip_top_x ? 0 : vslab_grid(b0), g1 = vslab_grid(b1), g2 = vslab_grid(b2);
      hipLaunchKernelGGL((k_vcg_apply_all<T, MASK>), dim3(grid + g0 + g1 + g2), dim3(256), 0, st, h->cp, h->k1, h->k2, vv,
                         ob + h->off[0], ob + h->off[1], ob + h->off[2], xchunk, nbz, nby, xcd, grid, b0, b1, b2, g0, g1,
                         partial + used, done);
      used += grid + g0 + g1 + g2;
      MFS_LAUNCH_CHECK();
      *nparts = used;
      return MFS_OK;
    } else {
      hipLaunchKernelGGL((k_vcg_apply_fused<T, MASK>), dim3(grid), dim3(256), 0, st, h->cp, h->k1, h->k2, vv,
                         ob + h->off[0], ob + h->off[1], ob + h->off[2], xchunk, nbz, nby, xcd, partial + used, done);
      used += grid;
    }
  }
  // (2) the three slabs of interior faces outside that box: u at x = Nx-1, v at y = Ny-1, w at z = Nz-1
  auto slab = [&](int ax) { return vslab_box(h, ax); };
  auto sgrid = [&](const Box3& b) { return vslab_grid(b); };
  if (Nx >= 3 && Ny >= 3 && Nz >= 3) {
    const Box3 b0 = slab(0), b1 = slab(1), b2 = slab(2);
    const int g0 = h->skip_top_x ? 0 : sgrid(b0), g1 = sgrid(b1), g2 = sgrid(b2);
    hipLaunchKernelGGL((k_vcg_apply_slabs<T, MASK>), dim3(g0 + g1 + g2), dim3(256), 0, st, h->cp, h->k1, h->k2, vv,
                       ob + h->off[0], ob + h->off[1], ob + h->off[2], b0, b1, b2, g0, g1, partial + used, done);
    used += g0 + g1 + g2;
  } else {   // degenerate grids: whichever slabs exist, one launch each
    if (Nx >= 2 && Ny >= 3 && Nz >= 3 && !h->skip_top_x) {
      const Box3 b = slab(0); const int g = sgrid(b);
      hipLaunchKernelGGL((k_vcg_apply_row<T, 0, MASK>), dim3(g), dim3(256), 0, st, h->cp, h->k1, h->k2, vv, ob + h->off[0], b, partial + used, done);
      used += g;
    }
    if (Nx >= 3 && Ny >= 2 && Nz >= 3) {
      const Box3 b = slab(1); const int g = sgrid(b);
      hipLaunchKernelGGL((k_vcg_apply_row<T, 1, MASK>), dim3(g), dim3(256), 0, st, h->cp, h->k1, h->k2, vv, ob + h->off[1], b, partial + used, done);
      used += g;
    }
    if (Nx >= 3 && Ny >= 3 && Nz >= 2) {
      const Box3 b = slab(2); const int g = sgrid(b);
      hipLaunchKernelGGL((k_vcg_apply_row<T, 2, MASK>), dim3(g), dim3(256), 0, st, h->cp, h->k1, h->k2, vv, ob + h->off[2], b, partial + used, done);
      used += g;
    }
  }
  MFS_LAUNCH_CHECK();
  *nparts = used;
  return MFS_OK;
}

// masked = false only for operands that are 0 on solid / array-boundary faces (the CG's d)
static int vcg_apply(mfs_vcg3d* h, const void* v, void* out, double* partial, bool use_done, bool masked,
                     hipStream_t st, int* nparts) {
  if (h->g.N[0] < 2 || h->g.N[1] < 2 || h->g.N[2] < 2) { *nparts = 0; return MFS_OK; }
  const double* done = use_done ? h->c.scal + S_DONE : nullptr;
  if (h->dt == MFS_F32)
    return masked ? vcg_apply_TM<float, true>(h, v, out, partial, done, st, nparts)
                  : vcg_apply_TM<float, false>(h, v, out, partial, done, st, nparts);
  return masked ? vcg_apply_TM<double, true>(h, v, out, partial, done, st, nparts)
                : vcg_apply_TM<double, false>(h, v, out, partial, done, st, nparts);
}

// ---- window transport of the slab loop (mfs_p2p.h): the edge planes of the direction vector and the two dot products
// travel as self-validating granules written by these kernels straight into the neighbours' windows -- no collective
// call, no host code inside the loop.  A receive buffer holds the three components' planes back to back.
struct VSlabPlanes {
  int64_t off[3];        // element offset of component c in the flat vectors
  int64_t pe[3];         // elements of one plane of component c
  int L;                 // local cell planes (the u array has L + 1, only planes 0 .. L-1 take part)
};

// One launch moves both directions: blocks [0, nsend) store this rank's two edge planes (all three components) into the
// neighbours' receive buffers, the remaining blocks unpack the own window into the ghost planes, re-reading any granule
// that does not carry this iteration's tag yet.  The senders never wait, and they are the lower-numbered blocks
// (dispatched first), so the waiting blocks cannot starve them.
template <typename T>
__global__ void __launch_bounds__(256)
k_vslab_exchange(T* __restrict__ d, VSlabPlanes g, double* __restrict__ scal, P2pDev pd, int par, unsigned tag, int nsend) {
  if (scal[S_DONE] != 0.0) return;
  const int64_t tot = g.pe[0] + g.pe[1] + g.pe[2];
  const bool sender = (int)blockIdx.x < nsend;
  const int nblk = sender ? nsend : (int)gridDim.x - nsend, blk = sender ? (int)blockIdx.x : (int)blockIdx.x - nsend;
  const int64_t stride = (int64_t)nblk * blockDim.x;
  bool lost = false;
  for (int side = 0; side < 2 && !lost; ++side) {
    u64* const dst = pd.send[side][par];       // [0]: left neighbour's high-ghost buffer, [1]: right neighbour's low-ghost
    const u64* const src = pd.recv[side][par];
    if (sender ? dst == nullptr : (side == 0 ? pd.rank == 0 : pd.rank == pd.world - 1)) continue;
    const int plane = sender ? (side == 0 ? 1 : g.L - 2) : (side == 0 ? 0 : g.L - 1);
    for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < tot; i += stride) {
      const int c = i < g.pe[0] ? 0 : (i < g.pe[0] + g.pe[1] ? 1 : 2);
      const int64_t e = i - (c == 0 ? 0 : (c == 1 ? g.pe[0] : g.pe[0] + g.pe[1]));
      T* const cell = d + g.off[c] + (int64_t)plane * g.pe[c] + e;
      vec_t<T, 1> v;
      if (sender) {
        v[0] = *cell;
        gran_store_vec<T, 1>(dst, i, v, tag);
      } else {
        if (!gran_load_vec<T, 1>(src, i, tag, pd.timeout_ticks, &v)) { lost = true; break; }
        *cell = v[0];
      }
    }
  }
  if (lost) slab_fail(scal, 2);
}

// ONE block: this rank's partial sums (k_reduce's order) -> every rank's window -> the world's total in rank order -> scal[slot]
static __global__ void __launch_bounds__(kBlock)
k_vslab_allreduce(const double* __restrict__ partial, int count, double* __restrict__ scal, int slot, int check_done, P2pDev pd,
                  int ring, unsigned tag) {
  if (check_done && scal[S_DONE] != 0.0) return;
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
  scal[slot] = tot;
}

static VSlabPlanes vslab_planes(const mfs_vcg3d* h) {
  VSlabPlanes g;
  for (int c = 0; c < 3; ++c) { g.off[c] = h->off[c]; g.pe[c] = (int64_t)h->g.sh(c, 1) * h->g.sh(c, 2); }
  g.L = h->g.N[0];
  return g;
}
static unsigned vslab_tag(const mfs_p2p* p, int64_t episode) {
  return 0x80000000u | ((p->epoch & 0x7ffu) << 20) | (unsigned)(episode & 0xfffff);
}
static int vslab_allreduce(mfs_vcg3d* h, int slot, int check_done, int64_t episode, hipStream_t st) {
  const double* part = slot == S_DQ ? h->c.part_dq : (slot == S_RZ ? h->part_rz : h->c.part_rr);
  const int count = slot == S_DQ ? h->c.n_part_dq : h->c.n_part_rr;      // (r.z partials: one per block of the same kernel as r.r)
  hipLaunchKernelGGL(k_vslab_allreduce, dim3(1), dim3(kBlock), 0, st, part, count, h->c.scal, slot, check_done, h->p2p->dev,
                     (int)(episode & (kArRing - 1)), vslab_tag(h->p2p, episode));
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

// Work list of the CG loops' march launches (Compact::items): attached for the duration of one loop call when the solve
// has live-chunk vector phases (vcg_build_live: single-domain and window slab loops) -- q of an all-air (tile, plane) pair is
// +0 since the solve's initial q = A x and nothing else writes it.  Stand-alone / phase-API applies never see the list.
struct VcgListScope {
  mfs_vcg3d* h;
  int err = MFS_OK;
  VcgListScope(mfs_vcg3d* h_, hipStream_t st) : h(h_) {
    if (h->c.live.list && h->compress && !h->mask_cg) {
      if (!h->classes_ready) err = vcg_build_classes(h, st);
      if (!err && h->list_ready) { h->cp.items = h->list_items; h->cp.runrem = h->list_runrem; h->cp.count = h->list_count; }
    }
  }
  ~VcgListScope() { h->cp.items = nullptr; h->cp.runrem = nullptr; h->cp.count = nullptr; }
};

template <typename T>
static int vslab_iteration(mfs_vcg3d* h, hipStream_t st) {
  mfs_p2p* p = h->p2p;
  const int64_t j = h->c.iter_enq;
  const int par = (int)(j & 1);
  const VSlabPlanes g = vslab_planes(h);
  const unsigned halo_tag = vslab_tag(p, j + 1);
  const int64_t tot = g.pe[0] + g.pe[1] + g.pe[2];
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(256, (tot + 255) / 256));
  int e, np = 0;
  if (p->world > 1 && g.L >= 3) {   // d's edge planes into the neighbours' windows, the ghosts out of the own one
    hipLaunchKernelGGL((k_vslab_exchange<T>), dim3(2 * grid), dim3(256), 0, st, (T*)h->c.d, g, h->c.scal, p->dev, par, halo_tag,
                       grid);
    MFS_LAUNCH_CHECK();
  }
  {
    VcgListScope list(h, st);
    if (list.err) return list.err;
    if ((e = vcg_apply(h, h->c.d, h->c.q, h->c.part_dq, true, h->mask_cg != 0, st, &np))) return e;
  }
  h->c.n_part_dq = np;
  if (h->jacobi) {
    // opt-in Jacobi iteration: three all-reduce episodes (d.q, r.r, r.z; begin used 0 and 1), the generic Jacobi kernels
    // reading the all-reduced scalars (npart = 0); ghost planes: r = 0 there, so z = 0 and the local r.z counts owned faces
    constexpr int VEC = VecOf<T>::N;
    const bool vec = core_vec_ok(h->c);
    const int gv = core_vec_grid(h->c, vec);
    if ((e = vslab_allreduce(h, S_DQ, 1, 3 * j + 2, st))) return e;
    if (vec)
      hipLaunchKernelGGL((k_jac_update_xr<T, VEC>), dim3(gv), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)h->c.d, (T*)h->c.r,
                         (const T*)h->c.q, (const T*)h->diag, h->n, h->c.scal, h->c.part_rr, h->part_rz, par, h->c.part_dq, 0);
    else
      hipLaunchKernelGGL((k_jac_update_xr<T, 1>), dim3(gv), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)h->c.d, (T*)h->c.r,
                         (const T*)h->c.q, (const T*)h->diag, h->n, h->c.scal, h->c.part_rr, h->part_rz, par, h->c.part_dq, 0);
    MFS_LAUNCH_CHECK();
    h->c.n_part_rr = gv;
    if ((e = vslab_allreduce(h, S_RR, 1, 3 * j + 3, st))) return e;
    if ((e = vslab_allreduce(h, S_RZ, 1, 3 * j + 4, st))) return e;
    if (vec)
      hipLaunchKernelGGL((k_jac_update_d<T, VEC>), dim3(gv), dim3(kBlock), 0, st, (T*)h->c.d, (const T*)h->c.r, (const T*)h->diag,
                         h->n, h->c.scal, h->c.hist, kHistCap, par, h->c.part_rr, h->part_rz, 0);
    else
      hipLaunchKernelGGL((k_jac_update_d<T, 1>), dim3(gv), dim3(kBlock), 0, st, (T*)h->c.d, (const T*)h->c.r, (const T*)h->diag,
                         h->n, h->c.scal, h->c.hist, kHistCap, par, h->c.part_rr, h->part_rz, 0);
    MFS_LAUNCH_CHECK();
    ++h->c.iter_enq;
    return MFS_OK;
  }
  if ((e = vslab_allreduce(h, S_DQ, 1, 2 * j + 1, st))) return e;     // local partials -> d.q over all ranks
  if ((e = core_update_xr(h->c, false, st))) return e;
  if ((e = vslab_allreduce(h, S_RR, 1, 2 * j + 2, st))) return e;     // r.r over all ranks
  return core_update_d(h->c, false, st);
}

extern "C" {

size_t mfs_visc_extrapolate3d_workspace_bytes(const int64_t gres[3], int v_dt) {
  if (!gres || !dtype_ok(v_dt)) return 0;
  G3 g = make_g(gres);
  size_t tot = 0;
  for (int c = 0; c < 3; ++c) tot += align_up((size_t)g.nface(c) * dtype_size(v_dt), 256) + 2 * align_up((size_t)g.nface(c), 256);
  return tot;
}

// shared by the viscosity solver's extrapolate (validity = sphi >= 0 at the face) and the notebook's
// (validity = grid mass > 0): num_iter Jacobi sweeps, ping-pong buffers in `workspace`
static int extrapolate_impl(const int64_t gres[3], int num_iter, void* vx, void* vy, void* vz, int v_dt,
                            const void* sphi, int sphi_dt, const void* const mass[3], int m_dt, void* workspace,
                            size_t workspace_bytes, hipStream_t st) {
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(vx && vy && vz && workspace, "null array");
  MFS_REQUIRE(dtype_ok(v_dt), "dtype");
  MFS_REQUIRE(num_iter >= 0, "num_iter");
  MFS_REQUIRE(workspace_bytes >= mfs_visc_extrapolate3d_workspace_bytes(gres, v_dt), "workspace too small");
  MFS_REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
  G3 g = make_g(gres);
  void* v[3] = {vx, vy, vz};
  char* p = (char*)workspace;
  for (int c = 0; c < 3; ++c) {
    const int64_t n = g.nface(c);
    void* tmp = p; p += align_up((size_t)n * dtype_size(v_dt), 256);
    unsigned char* va = (unsigned char*)p; p += align_up((size_t)n, 256);
    unsigned char* vb = (unsigned char*)p; p += align_up((size_t)n, 256);
    const int grid = cdiv(n, 256);
    if (mass) {
      hipLaunchKernelGGL(k_grid_valid_mass, dim3(grid), dim3(256), 0, st, n, mass[c], m_dt, va);
    } else {
      if (c == 0) hipLaunchKernelGGL((k_visc_valid<0>), dim3(grid), dim3(256), 0, st, g, sphi, sphi_dt, va);
      if (c == 1) hipLaunchKernelGGL((k_visc_valid<1>), dim3(grid), dim3(256), 0, st, g, sphi, sphi_dt, va);
      if (c == 2) hipLaunchKernelGGL((k_visc_valid<2>), dim3(grid), dim3(256), 0, st, g, sphi, sphi_dt, va);
    }
    void *cur = v[c], *oth = tmp;
    unsigned char *mcur = va, *moth = vb;
    for (int it = 0; it < num_iter; ++it) {
      hipLaunchKernelGGL(k_visc_extrap_sweep, dim3(grid), dim3(256), 0, st, g.sh(c, 0), g.sh(c, 1), g.sh(c, 2), cur, oth,
                         v_dt, mcur, moth);
      std::swap(cur, oth);
      std::swap(mcur, moth);
    }
    MFS_LAUNCH_CHECK();
    if (cur != v[c]) MFS_HIP_TRY(hipMemcpyAsync(v[c], cur, (size_t)n * dtype_size(v_dt), hipMemcpyDeviceToDevice, st));
  }
  return MFS_OK;
}

// the two pieces of `extrapolate` for callers that must act between sweeps (slab decomposition: ghost planes of
// the values AND of the validity travel after every sweep)
int mfs_visc_valid3d(const int64_t gres[3], int comp, const void* sphi, int sphi_dt, unsigned char* valid,
                     mfs_stream stream) {
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(sphi && valid, "null array");
  MFS_REQUIRE(dtype_ok(sphi_dt), "dtype");
  MFS_REQUIRE(comp >= 0 && comp < 3, "comp must be 0, 1 or 2");
  G3 g = make_g(gres);
  const int grid = cdiv(g.nface(comp), 256);
  hipStream_t st = (hipStream_t)stream;
  if (comp == 0) hipLaunchKernelGGL((k_visc_valid<0>), dim3(grid), dim3(256), 0, st, g, sphi, sphi_dt, valid);
  if (comp == 1) hipLaunchKernelGGL((k_visc_valid<1>), dim3(grid), dim3(256), 0, st, g, sphi, sphi_dt, valid);
  if (comp == 2) hipLaunchKernelGGL((k_visc_valid<2>), dim3(grid), dim3(256), 0, st, g, sphi, sphi_dt, valid);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_visc_extrapolate_sweep3d(const int64_t gres[3], int comp, const void* v_in, void* v_out, int v_dt,
                                 const unsigned char* valid_in, unsigned char* valid_out, mfs_stream stream) {
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(v_in && v_out && valid_in && valid_out, "null array");
  MFS_REQUIRE(v_in != v_out && valid_in != valid_out, "a sweep reads the old arrays and writes new ones");
  MFS_REQUIRE(dtype_ok(v_dt), "dtype");
  MFS_REQUIRE(comp >= 0 && comp < 3, "comp must be 0, 1 or 2");
  G3 g = make_g(gres);
  hipLaunchKernelGGL(k_visc_extrap_sweep, dim3(cdiv(g.nface(comp), 256)), dim3(256), 0, (hipStream_t)stream,
                     g.sh(comp, 0), g.sh(comp, 1), g.sh(comp, 2), v_in, v_out, v_dt, valid_in, valid_out);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_visc_extrapolate3d(const int64_t gres[3], int num_iter, void* vx, void* vy, void* vz, int v_dt,
                           const void* sphi, int sphi_dt, void* workspace, size_t workspace_bytes,
                           mfs_stream stream) {
  MFS_REQUIRE(sphi, "null array");
  MFS_REQUIRE(dtype_ok(sphi_dt), "dtype");
  return extrapolate_impl(gres, num_iter, vx, vy, vz, v_dt, sphi, sphi_dt, nullptr, 0, workspace, workspace_bytes,
                          (hipStream_t)stream);
}

int mfs_grid_extrapolate3d(const int64_t gres[3], int num_iter, void* vx, void* vy, void* vz, int v_dt,
                           const void* mx, const void* my, const void* mz, int m_dt, void* workspace,
                           size_t workspace_bytes, mfs_stream stream) {
  MFS_REQUIRE(mx && my && mz, "null array");
  MFS_REQUIRE(dtype_ok(m_dt), "dtype");
  const void* const mass[3] = {mx, my, mz};
  return extrapolate_impl(gres, num_iter, vx, vy, vz, v_dt, nullptr, 0, mass, m_dt, workspace, workspace_bytes,
                          (hipStream_t)stream);
}

int mfs_grid_boundary_condition3d(const int64_t gres[3], const void* gvx, const void* gvy, const void* gvz, int v_dt,
                                  const void* gmx, const void* gmy, const void* gmz, int m_dt, const void* sphi,
                                  int sphi_dt, const void* sv, int sv_dt, double dx, void* dvx, void* dvy, void* dvz,
                                  int dv_dt, mfs_stream stream) {
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(gvx && gvy && gvz && gmx && gmy && gmz && sphi && sv && dvx && dvy && dvz, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(m_dt) && dtype_ok(sphi_dt) && dtype_ok(sv_dt) && dtype_ok(dv_dt), "dtype");
  G3 g = make_g(gres);
  V3 gv{{gvx, gvy, gvz}}, gm{{gmx, gmy, gmz}};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((k_grid_boundary_condition<0>), dim3(cdiv(g.nface(0), 256)), dim3(256), 0, st, g, gv, v_dt, gm, m_dt,
                     sphi, sphi_dt, sv, sv_dt, dx, dvx, dv_dt);
  hipLaunchKernelGGL((k_grid_boundary_condition<1>), dim3(cdiv(g.nface(1), 256)), dim3(256), 0, st, g, gv, v_dt, gm, m_dt,
                     sphi, sphi_dt, sv, sv_dt, dx, dvy, dv_dt);
  hipLaunchKernelGGL((k_grid_boundary_condition<2>), dim3(cdiv(g.nface(2), 256)), dim3(256), 0, st, g, gv, v_dt, gm, m_dt,
                     sphi, sphi_dt, sv, sv_dt, dx, dvz, dv_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_visc_rhs3d(const int64_t gres[3], double scale, double mu, const void* vx, const void* vy, const void* vz,
                   int v_dt, const void* sphi, int sphi_dt, const void* vol, int vol_dt, void* b_x, void* b_y,
                   void* b_z, int b_dt, mfs_stream stream) {
  return launch_rows_direct<true>(gres, scale, mu, vx, vy, vz, v_dt, b_x, b_y, b_z, b_dt, sphi, sphi_dt, vol, vol_dt,
                                  (hipStream_t)stream);
}

int mfs_visc_apply3d(const int64_t gres[3], double scale, double mu, const void* vx, const void* vy, const void* vz,
                     int v_dt, void* out_x, void* out_y, void* out_z, int out_dt, const void* sphi, int sphi_dt,
                     const void* vol, int vol_dt, mfs_stream stream) {
  MFS_REQUIRE(vx != out_x && vy != out_y && vz != out_z, "apply cannot run in place");
  return launch_rows_direct<false>(gres, scale, mu, vx, vy, vz, v_dt, out_x, out_y, out_z, out_dt, sphi, sphi_dt, vol,
                                   vol_dt, (hipStream_t)stream);
}

int mfs_visc_writeback3d(const int64_t gres[3], void* vx, void* vy, void* vz, int v_dt, const void* out_x,
                         const void* out_y, const void* out_z, int out_dt, const void* sphi, int sphi_dt,
                         mfs_stream stream) {
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(vx && vy && vz && out_x && out_y && out_z && sphi, "null array");
  MFS_REQUIRE(dtype_ok(v_dt) && dtype_ok(out_dt) && dtype_ok(sphi_dt), "dtype");
  G3 g = make_g(gres);
  W3 v{{vx, vy, vz}};
  V3 o{{out_x, out_y, out_z}};
  const int64_t n = gres[0] * gres[1] * gres[2];
  hipLaunchKernelGGL(k_visc_writeback, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, v, v_dt, o, out_dt,
                     sphi, sphi_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

}  // extern "C"

// the operator's diagonal on the flat layout (once per solve, Jacobi loop only)
static int vcg_build_diag(mfs_vcg3d* h, hipStream_t st) {
  if (h->diag_ready) return MFS_OK;
#define MFS_VDIAG(TT, AX) \
  hipLaunchKernelGGL((k_vcg_diag<TT, AX>), dim3(cdiv(h->nf[AX], 256)), dim3(256), 0, st, h->cp, h->k1, (TT*)h->diag + h->off[AX])
  if (h->dt == MFS_F32) { MFS_VDIAG(float, 0); MFS_VDIAG(float, 1); MFS_VDIAG(float, 2); }
  else                  { MFS_VDIAG(double, 0); MFS_VDIAG(double, 1); MFS_VDIAG(double, 2); }
#undef MFS_VDIAG
  MFS_LAUNCH_CHECK();
  h->diag_ready = true;
  return MFS_OK;
}

extern "C" {

// ------------------------------------------------------------------ engine ---
size_t mfs_vcg3d_workspace_bytes(const int64_t gres[3], int dt) {
  if (!gres || !dtype_ok(dt)) return 0;
  size_t tot = core_ws_bytes() + 4096;
  for (int p = 1; p < 8; ++p) tot += class_stride_bytes(gres, dt);
  tot += align_up((size_t)class_count(gres, 0), 4096);       // the packed mask bytes
  tot += align_up((size_t)mfs_vcg3d_dofs(gres) * dtype_size(dt), 4096);   // partner of d (fused loop)
  tot += align_up((size_t)mfs_vcg3d_dofs(gres) * dtype_size(dt), 4096);   // diagonal (Jacobi loop)
  tot += align_up((size_t)kMaxPartials * 8, 4096);                        // r.z partials (Jacobi loop)
  tot += vres_ws_bytes(mfs_vcg3d_dofs(gres), dtype_size(dt));             // resident loop: records + face mirror (0 if too big)
  tot += 4096;                                                            // the bulk volume value (compressed class access)
  tot += core_live_ws_bytes(mfs_vcg3d_dofs(gres));                             // live chunks of the flat vectors
  tot += tile_words_bytes(gres, dt);                                      // its tile words
  tot += align_up((size_t)(kVmMaxSegs + 1) * sizeof(int), 4096);          // ... and the cost-balanced segment starts
  tot += 2 * tile_words_bytes(gres, dt) * sizeof(int) + 4096;             // ... and the work list of the loop's launches
  return tot;
}

int64_t mfs_vcg3d_dofs(const int64_t gres[3]) {
  if (!gres) return 0;
  G3 g = make_g(gres);
  return g.nface(0) + g.nface(1) + g.nface(2);
}

int mfs_vcg3d_create(mfs_vcg3d** out, const int64_t gres[3], int dt, void* workspace, size_t workspace_bytes,
                     mfs_stream stream) {
  MFS_REQUIRE(out && workspace, "null argument");
  if (int e = check_gres(gres)) return e;
  MFS_REQUIRE(dtype_ok(dt), "dtype");
  MFS_REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
  MFS_REQUIRE(workspace_bytes >= mfs_vcg3d_workspace_bytes(gres, dt), "workspace too small");
  mfs_vcg3d* h = new mfs_vcg3d();
  h->g = make_g(gres);
  h->dt = dt;
  h->n = 0;
  for (int c = 0; c < 3; ++c) { h->nf[c] = h->g.nface(c); h->off[c] = h->n; h->n += h->nf[c]; }
  if (int e = core_init(h->c, dt, h->n)) { delete h; return e; }
  h->ws = (char*)workspace; h->ws_bytes = workspace_bytes;
  char* p = core_carve(h->c, h->ws);
  p = (char*)align_up((uintptr_t)p, 4096);
  for (int a = 0; a < 3; ++a) h->cp.N[a] = h->g.N[a];
  h->cp.py = h->g.N[1] + 1;
  h->cp.pz = h->g.N[2] + 4;
  h->cp.vol[0] = nullptr;
  for (int q = 1; q < 8; ++q) { h->cp.vol[q] = p; p += class_stride_bytes(gres, dt); }
  h->cp.msk = (unsigned char*)p; p += align_up((size_t)class_count(gres, 0), 4096);
  h->d2 = p; p += align_up((size_t)h->n * dtype_size(dt), 4096);
  h->diag = p; p += align_up((size_t)h->n * dtype_size(dt), 4096);
  h->part_rz = (double*)p; p += align_up((size_t)kMaxPartials * 8, 4096);
  h->cp.bulk = p; p += 4096;
  h->cp.tw = (const unsigned char*)p; p += tile_words_bytes(gres, dt);
  h->cp.tw_block = 0;
  h->cp.seg = (const int*)p; p += align_up((size_t)(kVmMaxSegs + 1) * sizeof(int), 4096);
  h->cp.seg_g = 0;
  h->cp.items = nullptr; h->cp.runrem = nullptr; h->cp.count = nullptr;
  h->list_items = (int*)p; p += tile_words_bytes(gres, dt) * sizeof(int);
  h->list_runrem = (int*)p; p += tile_words_bytes(gres, dt) * sizeof(int);
  h->list_count = (int*)p; p += 4096;
  h->list_ready = false;
  h->resident = env_int("MFS_VISC_RESIDENT", -1);
  h->res = VResPlan{};
  h->res_ar = nullptr; h->res_mirror = nullptr; h->res_epoch = 0;
  if (vres_ws_bytes(h->n, dtype_size(dt)) > 0) {
    h->res_ar = (u64*)p;
    h->res_mirror = (u64*)(p + align_up((size_t)kVResRing * kVResMaxW * kVResRecStride * 8, 4096));
    p += vres_ws_bytes(h->n, dtype_size(dt));
    // as many workgroups as fit half the chip, fewer if the boxes then get too thin to be worth their halo
    // as many workgroups as the chip has CUs at most (MFS_VRES_W caps it); fewer if the boxes then get too thin
    const int wmax = std::max(1, std::min(std::min(kVResMaxW, h->c.cus), env_int("MFS_VRES_W", 256)));
    int w0 = 1;
    while (2 * w0 <= wmax) w0 *= 2;
    for (int w = w0; w >= 16 && !h->res.ok; w /= 2)
      h->res = vres_plan(h->g.N[0], h->g.N[1], h->g.N[2], dtype_size(dt), w, dt == MFS_F32 ? 6 : 4);
  }
  h->res_timeout_ticks = (u64)std::max(1, env_int("MFS_VRES_TIMEOUT_MS", 2000)) * 100000ull;      // wall clock: 100 MHz
  h->res_first_timeout_ticks = (u64)std::max(1, env_int("MFS_VRES_FIRST_TIMEOUT_MS", 250)) * 100000ull;
  h->res_drop_wg = env_int("MFS_VRES_TEST_DROP_WG", -1);
  h->res_creg = env_int("MFS_VRES_CREG", 1);
  h->jacobi = env_int("MFS_VISC_JACOBI", 0);
  h->last_iters = 0;
  h->diag_ready = false;
  h->compress = env_int("MFS_VISC_COMPRESS", 1);
  h->classes_ready = false;
  h->sparse_vec = env_int("MFS_VISC_SPARSE", 1);
  h->sparse_min = (int64_t)env_int("MFS_VISC_SPARSE_MIN", 1 << 21);
  h->live_ws = (int*)p; p += core_live_ws_bytes(h->n);
  h->march_nt = env_int("MFS_VISC_MARCH_NT", -1);
  h->fuse = env_int("MFS_VISC_FUSE", 0);   // measured slower than the three-launch loop: opt-in
  h->fused_run = false;
  h->grid_row = std::min(kMaxPartials / 3, h->c.cus * env_int("MFS_VISC_BLOCKS_PER_CU", 8));
  h->is_setup = false;
  h->mask_cg = env_int("MFS_VISC_MASK_CG", 0);
  h->tiled = env_int("MFS_VISC_TILED", 0);   // measured slower than the direct-load kernel; kept selectable
  h->xcd_order = env_int("MFS_VISC_XCD", -1);
  h->split_x = env_int("MFS_VISC_SPLIT_X", 1);
  h->skip_top_x = 0;
  h->p2p = nullptr;
  h->merge_slabs = env_int("MFS_VISC_MERGE", 1);
  h->march = env_int("MFS_VISC_MARCH", 1);
  h->march_bpc = std::max(1, env_int("MFS_VISC_MARCH_BPC", 2));
  h->march_vec = env_int("MFS_VISC_MARCH_VEC", 0);
  h->xchunk_tiled = std::max(1, env_int("MFS_VISC_XCHUNK", 32));
  h->k1 = h->k2 = 0.0;
  if (hipMemsetAsync(workspace, 0, mfs_vcg3d_workspace_bytes(gres, dt), (hipStream_t)stream) != hipSuccess) {
    set_error("hipMemsetAsync(workspace) failed");
    core_free(h->c);
    delete h;
    return MFS_E_HIP;
  }
  *out = h;
  return MFS_OK;
}

int mfs_vcg3d_destroy(mfs_vcg3d* h) {
  if (!h) return MFS_OK;
  core_free(h->c);
  delete h;
  return MFS_OK;
}

int mfs_vcg3d_setup(mfs_vcg3d* h, double scale, double mu, const void* sphi, int sphi_dt, const void* vol, int vol_dt,
                    mfs_stream stream) {
  MFS_REQUIRE(h && sphi && vol, "null argument");
  MFS_REQUIRE(dtype_ok(sphi_dt) && dtype_ok(vol_dt), "dtype");
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2];
  const int64_t n = (int64_t)(2 * Nx + 1) * (2 * Ny + 1) * (2 * Nz + 1);
  unsigned char* mp = (unsigned char*)h->cp.msk;
  MFS_HIP_TRY(hipMemsetAsync(mp, 0, align_up((size_t)h->cp.stored(), 4), (hipStream_t)stream));   // the setup kernel ORs bits in
  void* const* v = (void* const*)h->cp.vol;
  if (h->dt == MFS_F32)
    hipLaunchKernelGGL((k_vcg_setup<float>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, Nx, Ny, Nz, h->cp.py,
                       h->cp.pz, sphi, sphi_dt, vol, vol_dt, (float*)v[1], (float*)v[2], (float*)v[3], (float*)v[4], (float*)v[5],
                       (float*)v[6], (float*)v[7], mp);
  else
    hipLaunchKernelGGL((k_vcg_setup<double>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, Nx, Ny, Nz, h->cp.py,
                       h->cp.pz, sphi, sphi_dt, vol, vol_dt, (double*)v[1], (double*)v[2], (double*)v[3], (double*)v[4],
                       (double*)v[5], (double*)v[6], (double*)v[7], mp);
  MFS_LAUNCH_CHECK();
  h->classes_ready = false;    // the march's compressed class access builds its classes at its first launch (vcg_build_classes)
  h->list_ready = false;
  h->c.live = LiveMap{nullptr, nullptr, 0};      // a solve's sparse lists belong to the operator it began with
  h->k1 = scale * mu;          // `scale * mu * ...`      (left to right, as the reference evaluates it)
  h->k2 = 2 * scale * mu;      // `2 * scale * mu * ...`
  h->is_setup = true;
  h->diag_ready = false;
  return h->jacobi ? vcg_build_diag(h, (hipStream_t)stream) : MFS_OK;
}

int mfs_vcg3d_apply(mfs_vcg3d* h, const void* v, void* out, mfs_stream stream) {
  MFS_REQUIRE(h && v && out && v != out, "null / aliased argument");
  MFS_REQUIRE(h->is_setup, "mfs_vcg3d_setup has not been called");
  int np = 0;
  if (int e = vcg_apply(h, v, out, h->c.part_dq, false, true, (hipStream_t)stream, &np)) return e;
  h->c.n_part_dq = np;
  return MFS_OK;
}

int mfs_vcg3d_apply_kernel(mfs_vcg3d* h) {
  if (!h) return 0;
  if (h->tiled) return 1;
  static const char aligned[16] __attribute__((aligned(16))) = {0};
  const void* d = h->c.d ? h->c.d : (const void*)aligned;
  const void* q = h->c.q ? h->c.q : (const void*)aligned;
  const bool ok = h->dt == MFS_F32 ? vcg_march_ok<float>(h, d, q) : vcg_march_ok<double>(h, d, q);
  return ok ? 2 : 0;
}

int mfs_vcg3d_bind(mfs_vcg3d* h, void* b, void* x, void* d, void* r, void* q) {
  MFS_REQUIRE(h, "null handle");
  return core_bind(h->c, b, x, d, r, q);
}

int mfs_vcg3d_begin(mfs_vcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  h->fused_run = false;
  h->c.live = LiveMap{nullptr, nullptr, 0};
  if (int e = core_begin_pre(h->c, tol, false, st)) return e;     // x keeps the extrapolated velocity (:569-573)
  int np = 0;
  if (int e = vcg_apply(h, h->c.x, h->c.q, h->c.part_dq, false, true, st, &np)) return e;   // :575
  if (h->jacobi) {       // opt-in: r = b - q, d = z = r / diag, delta0 = r.z (stopping rule stays r.r < tol^2)
    if (int e = vcg_build_diag(h, st)) return e;
    const int g2 = std::max(1, (int)std::min<int64_t>(h->c.grid_vec, (h->n + kBlock - 1) / kBlock));
    if (h->dt == MFS_F32)
      hipLaunchKernelGGL((k_jac_init<float>), dim3(g2), dim3(kBlock), 0, st, (const float*)h->c.b, (const float*)h->c.q,
                         (const float*)h->diag, (float*)h->c.d, (float*)h->c.r, h->n, h->c.part_rr, h->part_rz);
    else
      hipLaunchKernelGGL((k_jac_init<double>), dim3(g2), dim3(kBlock), 0, st, (const double*)h->c.b, (const double*)h->c.q,
                         (const double*)h->diag, (double*)h->c.d, (double*)h->c.r, h->n, h->c.part_rr, h->part_rz);
    h->c.n_part_rr = g2;
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st, h->c.part_rr, g2, h->c.scal, (int)S_RR, 0);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st, h->part_rz, g2, h->c.scal, (int)S_RZ, 0);
    hipLaunchKernelGGL(k_jac_begin_finish, dim3(1), dim3(64), 0, st, h->c.scal, h->c.hist);
    MFS_LAUNCH_CHECK();
    // the solve's sparse lists for the opt-in Jacobi loop too (z = r / diag is 0 wherever r is: a dead face stays dead); the
    // stored z (the z form keeps it in the fused loop's partner buffer) must then be 0 outside the live chunks
    if (h->p2p || h->skip_top_x) return MFS_OK;
    if (int e = vcg_build_live(h, st)) return e;
    if (h->c.live.list) MFS_HIP_TRY(hipMemsetAsync(h->d2, 0, (size_t)h->n * h->c.elt, st));
    return MFS_OK;
  }
  if (int e = core_begin_post(h->c, st)) return e;                // :577-585
  if (int e = core_begin_finish(h->c, st)) return e;
  return (h->p2p || h->skip_top_x || h->fuse) ? MFS_OK : vcg_build_live(h, st);      // single-domain, launch-per-phase loops only
}

// ---- slab decomposition along x (mfs/dist.py:SlabVCG): the phases of one iteration, so that the caller can put the
// halo exchange of d and the two scalar all-reduces (on the engine's scalar block) between them.
// skip_top_x: this rank's last u plane is a ghost of the right neighbour's first owned one -- not computed here,
// so q (and with b = 0 there, r) stay exactly 0 on it and the local dot products count owned faces only.
int mfs_vcg3d_set_slab(mfs_vcg3d* h, int skip_top_x) {
  MFS_REQUIRE(h, "null handle");
  h->skip_top_x = skip_top_x ? 1 : 0;
  return MFS_OK;
}

void* mfs_vcg3d_scalars(mfs_vcg3d* h) { return h ? h->c.scal : nullptr; }

int mfs_vcg3d_begin_local(mfs_vcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  h->fused_run = false;
  h->c.live = LiveMap{nullptr, nullptr, 0};      // callers that drive the phases themselves sweep every chunk
  if (int e = core_begin_pre(h->c, tol, false, st)) return e;
  int np = 0;
  if (int e = vcg_apply(h, h->c.x, h->c.q, h->c.part_dq, false, true, st, &np)) return e;
  return core_begin_post(h->c, st);          // local r.r in the scalar block: all-reduce it, then begin_finish
}

int mfs_vcg3d_begin_finish(mfs_vcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_begin_finish(h->c, (hipStream_t)stream);
}

int mfs_vcg3d_phase_apply(mfs_vcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.d && h->is_setup, "engine not bound / set up");
  int np = 0;
  if (int e = vcg_apply(h, h->c.d, h->c.q, h->c.part_dq, true, h->mask_cg != 0, (hipStream_t)stream, &np)) return e;
  h->c.n_part_dq = np;
  return MFS_OK;
}

int mfs_vcg3d_phase_reduce(mfs_vcg3d* h, int which, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(which == 0 || which == 1, "which must be 0 (d.q) or 1 (r.r)");
  return core_reduce(h->c, which, 1, (hipStream_t)stream);
}

int mfs_vcg3d_phase_update_xr(mfs_vcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_update_xr(h->c, false, (hipStream_t)stream);
}

int mfs_vcg3d_phase_update_d(mfs_vcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  return core_update_d(h->c, false, (hipStream_t)stream);
}

// window transport for the slab loop: `p` must have been created with plane_bytes = the three edge planes of this
// engine's vectors, (Ny*Nz + (Ny+1)*Nz + Ny*(Nz+1)) * sizeof(element); null detaches
int mfs_vcg3d_attach_p2p(mfs_vcg3d* h, mfs_p2p* p) {
  MFS_REQUIRE(h, "null handle");
  if (p) {
    MFS_REQUIRE(p->connected, "mfs_p2p_connect has not been called");
    const VSlabPlanes g = vslab_planes(h);
    MFS_REQUIRE((size_t)(g.pe[0] + g.pe[1] + g.pe[2]) * h->c.elt == p->plane_bytes,
                "window plane size != the three edge planes of this engine");
  }
  h->p2p = p;
  return MFS_OK;
}

int mfs_vcg3d_slab_begin(mfs_vcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup && h->p2p, "engine not bound / set up / no window attached");
  ++h->p2p->epoch;
  hipStream_t st = (hipStream_t)stream;
  h->fused_run = false;
  h->c.live = LiveMap{nullptr, nullptr, 0};
  if (int e = core_begin_pre(h->c, tol, false, st)) return e;
  int np = 0;
  if (int e = vcg_apply(h, h->c.x, h->c.q, h->c.part_dq, false, true, st, &np)) return e;
  if (h->jacobi) {       // r = b - q, d = z = r / diag; r.r and r.z over all ranks (episodes 0 and 1); delta0 = r.z
    if (int e = vcg_build_diag(h, st)) return e;
    const int g2 = std::max(1, (int)std::min<int64_t>(h->c.grid_vec, (h->n + kBlock - 1) / kBlock));
    if (h->dt == MFS_F32)
      hipLaunchKernelGGL((k_jac_init<float>), dim3(g2), dim3(kBlock), 0, st, (const float*)h->c.b, (const float*)h->c.q,
                         (const float*)h->diag, (float*)h->c.d, (float*)h->c.r, h->n, h->c.part_rr, h->part_rz);
    else
      hipLaunchKernelGGL((k_jac_init<double>), dim3(g2), dim3(kBlock), 0, st, (const double*)h->c.b, (const double*)h->c.q,
                         (const double*)h->diag, (double*)h->c.d, (double*)h->c.r, h->n, h->c.part_rr, h->part_rz);
    MFS_LAUNCH_CHECK();
    h->c.n_part_rr = g2;
    if (int e = vslab_allreduce(h, S_RR, 0, 0, st)) return e;
    if (int e = vslab_allreduce(h, S_RZ, 0, 1, st)) return e;
    hipLaunchKernelGGL(k_jac_begin_finish, dim3(1), dim3(64), 0, st, h->c.scal, h->c.hist);
    MFS_LAUNCH_CHECK();
    return MFS_OK;
  }
  if (int e = core_begin_post(h->c, st, false)) return e;        // d = r = b - q, partials of r.r
  if (int e = vslab_allreduce(h, S_RR, 0, 0, st)) return e;
  if (int e = core_begin_finish(h->c, st)) return e;
  // (a face's liveness is a property of its own rank: an empty row gives q = 0 whatever the ghost planes hold; ghost faces are
  // array-boundary faces of the local arrays -- never swept unless they share a chunk with a live face, as in the dense loop)
  return h->fuse ? MFS_OK : vcg_build_live(h, st);
}

int mfs_vcg3d_slab_iterate(mfs_vcg3d* h, int64_t n, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup && h->p2p, "engine not bound / set up / no window attached");
  for (int64_t i = 0; i < n; ++i) {
    const int e = h->dt == MFS_F32 ? vslab_iteration<float>(h, (hipStream_t)stream) : vslab_iteration<double>(h, (hipStream_t)stream);
    if (e) return e;
  }
  return MFS_OK;
}

// can the fused loop serve the engine as bound?  (the marching kernel's preconditions, 16-byte aligned CG vectors)
static bool vcg_fuse_ok(const mfs_vcg3d* h) {
  if ((h->dt == MFS_F32 ? vcg_march_geom<float, 4>(h) : vcg_march_geom<double, 2>(h)) % 10 == 3) return false;   // (no fused form on the 3-slot ring)
  if (!h->fuse || h->jacobi || !h->split_x || h->mask_cg || h->tiled || h->march_vec == 2 || !h->c.d || !core_vec_ok(h->c)) return false;
  return h->dt == MFS_F32 ? vcg_march_ok<float>(h, h->c.d, h->c.q) : vcg_march_ok<double>(h, h->c.d, h->c.q);
}

extern "C++" {
// one iteration of the opt-in Jacobi loop: stencil launch, x / r update with r.r and r.z (z = r / diag formed on the fly),
// direction update d = z + beta d with the bookkeeping -- the generic kernels of mfs_cg_core.h on the flat vectors
template <typename T, int VEC>
static int vcg_jac_iteration(mfs_vcg3d* h, hipStream_t st) {
  int e, np = 0;
  if ((e = vcg_apply(h, h->c.d, h->c.q, h->c.part_dq, true, false, st, &np))) return e;
  h->c.n_part_dq = np;
  const int grid = core_vec_grid(h->c, VEC > 1);
  const int par = (int)(h->c.iter_enq & 1);
  hipLaunchKernelGGL((k_jac_update_xr<T, VEC>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)h->c.d, (T*)h->c.r,
                     (const T*)h->c.q, (const T*)h->diag, h->n, h->c.scal, h->c.part_rr, h->part_rz, par, h->c.part_dq,
                     h->c.n_part_dq);
  hipLaunchKernelGGL((k_jac_update_d<T, VEC>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.d, (const T*)h->c.r,
                     (const T*)h->diag, h->n, h->c.scal, h->c.hist, kHistCap, par, h->c.part_rr, h->part_rz, grid);
  MFS_LAUNCH_CHECK();
  h->c.n_part_rr = grid;
  ++h->c.iter_enq;
  return MFS_OK;
}

// the same iteration with z STORED (aligned vectors): the r update reads diag once, writes r and z = r / diag (into the
// fused loop's partner buffer, idle here) and its last block closes the iteration; the second phase x += alpha d,
// d = z + beta d streams five arrays.  10 instead of 11 scalars per DOF, two reduction-free streaming kernels:
// 256^3 fp32 660 -> see profiles/r02_jacobi_time_step.txt
template <typename T, int VEC>
static int vcg_jac_iteration_z(mfs_vcg3d* h, hipStream_t st) {
  int e, np = 0;
  if ((e = vcg_apply(h, h->c.d, h->c.q, h->c.part_dq, true, false, st, &np))) return e;
  h->c.n_part_dq = np;
  const int grid = core_vec_grid(h->c, true);
  const int par = (int)(h->c.iter_enq & 1);
  hipLaunchKernelGGL((k_jac_update_rz<T, VEC, false>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.x, (const T*)h->c.d, (T*)h->c.r,
                     (const T*)h->c.q, (const T*)h->diag, (T*)h->d2, h->n, h->c.scal, h->c.part_rr, h->part_rz, par,
                     h->c.part_dq, h->c.n_part_dq, (const unsigned char*)nullptr, h->c.hist, kHistCap, h->c.tickets, JacSlab{}, h->c.live);
  const bool ntx = 5.0 * (double)h->n * h->c.elt > 200e6;
  if (ntx) hipLaunchKernelGGL((k_jac_dx<T, VEC, true>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.x, (T*)h->c.d, (const T*)h->d2, h->n,
                              h->c.scal, (double)h->c.iter_enq, h->c.live);
  else hipLaunchKernelGGL((k_jac_dx<T, VEC, false>), dim3(grid), dim3(kBlock), 0, st, (T*)h->c.x, (T*)h->c.d, (const T*)h->d2, h->n,
                          h->c.scal, (double)h->c.iter_enq, h->c.live);
  MFS_LAUNCH_CHECK();
  h->c.n_part_rr = grid;
  ++h->c.iter_enq;
  return MFS_OK;
}
}  // extern "C++"

// the resident loop (mfs_vcg_resident.h): a grid whose state fits W workgroups' registers and LDS, the reference's plain
// loop (no Jacobi, no fused / masked / slab form), a device with that many CUs
static bool vcg_resident_ok(const mfs_vcg3d* h) {
  return h->resident != 0 && h->res.ok && h->res_ar && !h->jacobi && !h->fuse && !h->mask_cg && !h->p2p && !h->skip_top_x &&
         h->c.x && h->c.cus >= h->res.W;
}

extern "C++" {
template <typename T>
static int vcg_launch_resident(mfs_vcg3d* h, const VResArgs& a, hipStream_t st) {
  const VResPlan& p = h->res;
#define MFS_VRES_ONE(KCC, CRG)                                                                                   \
  do {                                                                                                           \
    MFS_HIP_TRY(hipFuncSetAttribute((const void*)k_vcg_resident<T, KCC, CRG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)kVResLdsMax));                                                          \
    hipLaunchKernelGGL((k_vcg_resident<T, KCC, CRG>), dim3(p.W), dim3(kVResBlock), p.lds, st, a);                \
  } while (0)
  // volume samples in registers where KC cells' worth fit without spills (16 per cell, kept converted to fp64):
  // 8-byte state up to 2 cells, 4-byte up to 3
  const bool creg = h->res_creg != 0 && p.kc <= (sizeof(T) == 8 ? 2 : 3);
  if (p.kc <= 1) { if (creg) MFS_VRES_ONE(1, true); else MFS_VRES_ONE(1, false); }
  else if (p.kc == 2) { if (creg) MFS_VRES_ONE(2, true); else MFS_VRES_ONE(2, false); }
  else if (p.kc == 3) { if (creg) MFS_VRES_ONE(3, true); else MFS_VRES_ONE(3, false); }
  else if (p.kc == 4) MFS_VRES_ONE(4, false);
  else MFS_VRES_ONE(6, false);
#undef MFS_VRES_ONE
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}
}  // extern "C++"

static int vcg_iterate_resident(mfs_vcg3d* h, int64_t n, hipStream_t st) {
  while (n > 0) {
    const int nb = (int)std::min<int64_t>(n, 1 << 20);
    if (h->res_epoch > 0xf0000000u - 2u * (unsigned)nb) {      // tags about to wrap: start over on clean tables
      MFS_HIP_TRY(hipMemsetAsync(h->res_ar, 0, vres_ws_bytes(h->n, h->c.elt), st));
      h->res_epoch = 0;
    }
    VResArgs a{};
    a.x = h->c.x; a.r = h->c.r; a.q = h->c.q; a.d = h->c.d;
    a.c = h->cp; a.k1 = h->k1; a.k2 = h->k2;
    for (int c = 0; c < 3; ++c) a.off[c] = h->off[c];
    a.Px = h->res.Px; a.Py = h->res.Py; a.bxm = h->res.bxm; a.bym = h->res.bym;
    a.scal = h->c.scal; a.hist = h->c.hist; a.hist_cap = kHistCap;
    a.n_iter = nb;
    a.ar = h->res_ar; a.mirror = h->res_mirror;
    a.tag0 = h->res_epoch + 1u;
    a.timeout_ticks = h->res_timeout_ticks; a.first_timeout_ticks = h->res_first_timeout_ticks;
    a.test_drop_wg = h->res_drop_wg;
    int e = h->dt == MFS_F32 ? vcg_launch_resident<float>(h, a, st) : vcg_launch_resident<double>(h, a, st);
    if (e) return e;
    h->res_epoch += 2u * (unsigned)nb;
    h->c.iter_enq += nb;
    n -= nb;
  }
  return MFS_OK;
}

int mfs_vcg3d_iterate(mfs_vcg3d* h, int64_t n, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  if (h->jacobi) {
    MFS_REQUIRE(h->diag_ready, "Jacobi loop: mfs_vcg3d_begin has not built the diagonal");
    const bool vec = core_vec_ok(h->c);
    // z stored (two streaming kernels, a reduction tail) pays beyond the caches; launch-bound sizes keep the form whose
    // consumers fold the partial sums themselves (MFS_VISC_JACOBI_Z = 0 / 1 overrides)
    const int zk = env_int("MFS_VISC_JACOBI_Z", -1);
    const bool zform = zk < 0 ? (5.0 * (double)h->n * h->c.elt > 100e6) : (zk != 0);
    VcgListScope list(h, st);      // (single-domain solves with lists: the march visits the busy pairs only)
    if (list.err) return list.err;
    for (int64_t i = 0; i < n; ++i) {
      int e;
      if (vec && zform) e = h->dt == MFS_F32 ? vcg_jac_iteration_z<float, 4>(h, st) : vcg_jac_iteration_z<double, 2>(h, st);
      else if (vec) e = h->dt == MFS_F32 ? vcg_jac_iteration<float, 4>(h, st) : vcg_jac_iteration<double, 2>(h, st);
      else e = h->dt == MFS_F32 ? vcg_jac_iteration<float, 1>(h, st) : vcg_jac_iteration<double, 1>(h, st);
      if (e) return e;
    }
    return MFS_OK;
  }
  if (vcg_fuse_ok(h)) {
    // Fused loop, 2 launches per iteration:  A | R   A* | R   A* | R ...   A = plain march (iteration 0: d_0 = r_0 is there),
    // A* = march that first forms d_j = r + beta d_{j-1} (into the other buffer of {bound d, d2}) and lets x += alpha d_{j-1}
    // ride along, R = r -= alpha q whose last block closes the iteration (test :606, history, alpha, beta).  x lags one
    // update behind and d_j may sit in d2 until vcg_home (mfs_vcg3d_finish / the end of mfs_vcg3d_solve) settles both.
    for (int64_t i = 0; i < n; ++i) {
      int e, np = 0;
      const int64_t j = h->c.iter_enq;
      void* d_cur = (j & 1) ? h->d2 : h->c.d;
      void* d_prev = (j & 1) ? h->c.d : h->d2;
      if (j == 0) {
        if ((e = vcg_apply(h, d_cur, h->c.q, h->c.part_dq, true, false, st, &np))) return e;                    // :589
      } else {
        e = h->dt == MFS_F32 ? vcg_march_fused<float>(h, d_prev, d_cur, st, &np) : vcg_march_fused<double>(h, d_prev, d_cur, st, &np);
        if (e) return e;                                                                                          // :595-597 (x), :609-610, :589
      }
      h->c.n_part_dq = np;
      h->fused_run = true;
      XrTail tl{1, h->c.hist, kHistCap, nullptr, 0, 0};
      if ((e = core_update_xr(h->c, true, st, 1, d_cur, 0, -1, &tl, nullptr))) return e;                          // :592-601 (r), :604-608
      ++h->c.iter_enq;
    }
    return MFS_OK;
  }
  if (vcg_resident_ok(h)) return vcg_iterate_resident(h, n, st);
  // small problems: both vector phases, the r.r reduction and the bookkeeping in ONE launch whose workgroups exchange
  // their partial sums while resident (k_update_rdx, mfs_cg_core.h): 2 launches per iteration, no reduction tail
  const bool rdx = core_rdx_ok(h->c) && !h->p2p;
  // single-domain solves with live-chunk vector phases (vcg_build_live): the march visits only the busy (tile, plane)
  // pairs -- q of an all-air pair is +0 since the initial q = A x and nothing else writes it
  VcgListScope list(h, st);
  if (list.err) return list.err;
  for (int64_t i = 0; i < n; ++i) {
    int e, np = 0;
    if ((e = vcg_apply(h, h->c.d, h->c.q, h->c.part_dq, true, h->mask_cg != 0, st, &np))) return e;   // :589
    h->c.n_part_dq = np;
    if (rdx) {
      if ((e = core_update_rdx(h->c, st))) return e;                                  // :592-610
      continue;
    }
    if (h->split_x) {   // r -= alpha q here, x += alpha d rides in the direction update (8 instead of 9 scalars per DOF)
      if ((e = core_update_xr(h->c, true, st, 1))) return e;                          // :592-601 (r)
      if ((e = core_update_d(h->c, true, st, true))) return e;                        // :595-597 (x), :604-610
    } else {
      if ((e = core_update_xr(h->c, true, st))) return e;                             // :592-601
      if ((e = core_update_d(h->c, true, st))) return e;                              // :604-610
    }
  }
  return MFS_OK;
}

// what the fused loop owes when it stops after `iters` completed iterations: x += alpha d of the last one, the direction
// update d_iters = r + beta d_{iters-1} unless it converged (the reference updates d at the end of every non-converged
// iteration, :609-610), and d brought home to the bound array if it sits in the partner buffer
static int vcg_home(mfs_vcg3d* h, int64_t iters, bool converged, hipStream_t st) {
  if (!h->fused_run) return MFS_OK;
  h->fused_run = false;
  if (iters < 1) return MFS_OK;
  void* cur = ((iters - 1) & 1) ? h->d2 : h->c.d;             // holds d_{iters-1}
  const int grid = core_vec_grid(h->c, true);
  if (h->dt == MFS_F32) hipLaunchKernelGGL((k_x_axpy<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)h->c.x, (const float*)cur, h->n, h->c.scal);
  else hipLaunchKernelGGL((k_x_axpy<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)h->c.x, (const double*)cur, h->n, h->c.scal);
  MFS_LAUNCH_CHECK();
  if (!converged) {
    if (h->dt == MFS_F32) hipLaunchKernelGGL((k_d_axpy<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)cur, (const float*)h->c.r, h->n, h->c.scal);
    else hipLaunchKernelGGL((k_d_axpy<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)cur, (const double*)h->c.r, h->n, h->c.scal);
    MFS_LAUNCH_CHECK();
  }
  if (cur != h->c.d) MFS_HIP_TRY(hipMemcpyAsync(h->c.d, cur, (size_t)h->n * h->c.elt, hipMemcpyDeviceToDevice, st));
  return MFS_OK;
}

int mfs_vcg3d_poll(mfs_vcg3d* h, mfs_stream stream, int64_t* iters, int* done, double* delta, double* alpha,
                   double* beta) {
  MFS_REQUIRE(h, "null handle");
  hipStream_t st = (hipStream_t)stream;
  // The merged vector-phase launch did not get its workgroups together (a shared GPU): it -- and every launch queued behind
  // it -- wrote nothing to the CG vectors, so the state is the one the scalar block describes.  Clear the flag, switch
  // this engine to the launch-per-phase loop for good, and report the iterations that did complete.
  MFS_HIP_TRY(hipMemcpyAsync(h->c.pinned, h->c.scal, MFS_PCG_NSCALARS * sizeof(double), hipMemcpyDeviceToHost, st));
  MFS_HIP_TRY(hipStreamSynchronize(st));
  bool fresh = true;
  if ((int)h->c.pinned[S_ERR] == kErrNotResident) {
    MFS_HIP_TRY(hipMemsetAsync(h->c.scal + S_ERR, 0, sizeof(double), st));
    MFS_HIP_TRY(hipMemsetAsync(h->c.scal + S_DONE, 0, sizeof(double), st));
    if (vcg_resident_ok(h)) h->resident = 0;      // the resident loop did not get its workgroups: launch-per-phase from now on
    else h->c.rdx = 0;
    h->c.iter_enq = (int64_t)h->c.pinned[S_ITERS];
    fresh = false;
  }
  return core_poll(h->c, st, iters, done, delta, alpha, beta, fresh);
}

int mfs_vcg3d_solve(mfs_vcg3d* h, double tol, int64_t max_iter, int64_t check_every, mfs_stream stream,
                    int64_t* iters_host) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(max_iter >= 0 && check_every >= 1, "max_iter / check_every");
  if (int e = mfs_vcg3d_begin(h, tol, stream)) return e;
  int64_t enq = 0, iters = 0;
  int done = 0;
  // SHORT solves (up to 4 x check_every iterations): the first batch is sized by the previous solve of this engine
  // (consecutive time steps need about the same number of iterations): last count + 1/8 + 2 in one go, so that the solve costs
  // ONE look at the scalar block instead of one per `check_every` iterations (launches queued behind a converged iteration
  // return at their top; a problem that starts converged is caught by them as well).  Later batches, and long solves --
  // where an overshooting prediction costs more no-op launches than the looks it saves --: `check_every` as before.
  bool first = true;
  while (!done && enq < max_iter) {
    // (the resident loop stops by itself inside a batch -- one launch --: at least 1024 iterations per look there)
    const bool res = vcg_resident_ok(h);
    int64_t n = std::min(res ? std::max<int64_t>(check_every, 1024) : check_every, max_iter - enq);
    if (!res && first && h->last_iters > 0 && h->last_iters <= 4 * check_every)
      n = std::min<int64_t>(max_iter - enq, std::min<int64_t>(h->last_iters + h->last_iters / 8 + 2, h->last_iters + 256));
    first = false;
    if (int e = mfs_vcg3d_iterate(h, n, stream)) return e;
    if (int e = mfs_vcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
    enq = h->c.iter_enq;        // = enq + n, unless the poll has just taken a batch back (merged launch not resident)
  }
  if (max_iter == 0)       // (nothing iterated: the scalar block as begin left it)
    if (int e = mfs_vcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  if (done) h->last_iters = iters;
  if (int e = vcg_home(h, iters, done != 0, (hipStream_t)stream)) return e;
  if (iters_host) *iters_host = iters;
  return done ? MFS_OK : MFS_NOT_CONVERGED;
}

// for callers that drive begin / iterate themselves: settles what the fused loop owes (the last x update, the direction
// vector parked in the engine's partner buffer).  Host-synchronous (it needs the iteration count); the loop must be
// started again with mfs_vcg3d_begin afterwards.
int mfs_vcg3d_finish(mfs_vcg3d* h, mfs_stream stream) {
  MFS_REQUIRE(h, "null handle");
  int64_t iters = 0;
  int done = 0;
  if (int e = mfs_vcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  return vcg_home(h, iters, done != 0, (hipStream_t)stream);
}

// 1 / 0: fold the direction and x updates into the marching kernel (default 0; MFS_VISC_FUSE)
int mfs_vcg3d_set_compress(mfs_vcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->compress = on ? 1 : 0;
  return MFS_OK;
}

int mfs_vcg3d_class_census(mfs_vcg3d* h, int64_t counts_host[3], mfs_stream stream) {
  MFS_REQUIRE(h && counts_host, "null argument");
  MFS_REQUIRE(h->is_setup, "mfs_vcg3d_setup has not been called");
  if (!h->classes_ready) { if (int e = vcg_build_classes(h, (hipStream_t)stream)) return e; }
  counts_host[0] = counts_host[1] = counts_host[2] = 0;
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2], vec = h->dt == MFS_F32 ? 4 : 2;
  if (Nx < 3 || Ny < 3 || Nz % vec != 0 || Nz < 2 * vec) return MFS_OK;       // no marching kernel, no classes
  unsigned long long* dv = (unsigned long long*)h->part_rz;                   // scratch: 3 words of the Jacobi partials
  hipStream_t st = (hipStream_t)stream;
  MFS_HIP_TRY(hipStreamSynchronize(st));                                      // (diagnostic call: nothing may be using the scratch)
  MFS_HIP_TRY(hipMemsetAsync(dv, 0, 3 * sizeof(unsigned long long), st));
  const int64_t nvec = (int64_t)(Nx - 2) * (Ny - 2) * (Nz / vec);
  if (vec == 4) hipLaunchKernelGGL((k_vcg_class_census<4>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, h->cp, dv);
  else hipLaunchKernelGGL((k_vcg_class_census<2>), dim3(cdiv(nvec, 256)), dim3(256), 0, st, h->cp, dv);
  MFS_LAUNCH_CHECK();
  unsigned long long host[3];
  MFS_HIP_TRY(hipMemcpyAsync(host, dv, sizeof(host), hipMemcpyDeviceToHost, st));
  MFS_HIP_TRY(hipStreamSynchronize(st));
  MFS_HIP_TRY(hipMemsetAsync(dv, 0, 3 * sizeof(unsigned long long), st));
  for (int k = 0; k < 3; ++k) counts_host[k] = (int64_t)host[k];
  return MFS_OK;
}

int mfs_vcg3d_set_sparse(mfs_vcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->sparse_vec = on ? 1 : 0;
  return MFS_OK;
}

// out = {live chunks, chunks, listed (tile, plane) pairs, pairs} of the solve begun last (zeros where a list is off); host-synchronous
int mfs_vcg3d_sparse_info(mfs_vcg3d* h, mfs_stream stream, int64_t out[4]) {
  MFS_REQUIRE(h && out, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  hipStream_t st = (hipStream_t)stream;
  if (!h->c.live.count) return MFS_OK;
  int v = 0;
  MFS_HIP_TRY(hipMemcpyAsync(&v, h->c.live.count, sizeof(int), hipMemcpyDeviceToHost, st));
  MFS_HIP_TRY(hipStreamSynchronize(st));
  out[0] = v;
  out[1] = (h->n + kLiveChunk - 1) / kLiveChunk;
  if (h->compress && !h->mask_cg && h->classes_ready && h->list_ready) {
    MFS_HIP_TRY(hipMemcpyAsync(&v, h->list_count, sizeof(int), hipMemcpyDeviceToHost, st));
    MFS_HIP_TRY(hipStreamSynchronize(st));
    const int vec = h->dt == MFS_F32 ? 4 : 2;
    const int ipp = (h->g.N[1] - 2) * (h->g.N[2] / vec);
    out[2] = v;
    out[3] = (int64_t)((ipp + h->cp.tw_block - 1) / h->cp.tw_block) * (h->g.N[0] - 2);
  }
  return MFS_OK;
}

int mfs_vcg3d_set_fuse(mfs_vcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(!h->fused_run, "mfs_vcg3d_set_fuse inside a fused loop: call mfs_vcg3d_finish first");
  h->fuse = on ? 1 : 0;
  return MFS_OK;
}

// what mfs_vcg3d_iterate will do for the engine as bound: bit 0 fused direction + x update (2 launches per iteration)
int mfs_vcg3d_loop_info(mfs_vcg3d* h) {
  if (!h) return 0;
  if (!h->c.x || !h->is_setup) return h->jacobi ? 4 : 0;
  if (vcg_resident_ok(h)) return 8;
  return (vcg_fuse_ok(h) ? 1 : 0) | ((!h->jacobi && !vcg_fuse_ok(h) && core_rdx_ok(h->c) && !h->p2p) ? 2 : 0) | (h->jacobi ? 4 : 0);
}

// 1 / 0: allow the resident small-grid loop (default: on where the grid qualifies; env MFS_VISC_RESIDENT)
int mfs_vcg3d_set_resident(mfs_vcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->resident = on ? 1 : 0;
  return MFS_OK;
}

// OPT-IN Jacobi preconditioning of mfs_vcg3d_begin / iterate / solve (default off; env MFS_VISC_JACOBI=1): see mfs.h
int mfs_vcg3d_set_jacobi(mfs_vcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  if (h->jacobi != (on ? 1 : 0)) h->last_iters = 0;      // another iteration: the previous solve predicts nothing
  h->jacobi = on ? 1 : 0;
  return MFS_OK;
}

// 1 / 0: the merged vector phases of small problems (k_update_rdx; default 1, env MFS_RDX)
int mfs_vcg3d_set_merged(mfs_vcg3d* h, int on) {
  MFS_REQUIRE(h, "null handle");
  h->c.rdx = on ? 1 : 0;
  return MFS_OK;
}

int64_t mfs_vcg3d_history(mfs_vcg3d* h, double* out_host, int64_t cap, mfs_stream stream) {
  if (!h) { set_error("mfs_vcg3d_history: null handle"); return MFS_E_INVALID; }
  return core_history(h->c, out_host, cap, (hipStream_t)stream);
}

}  // extern "C"
