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
// class array dims: odd axis -> N, even axis -> N+1; compact index = node >> 1.
struct Compact {
  const void* vol[8];           // state dtype; [0] (e,e,e) unused
  const unsigned char* msk[8];  // only the three face classes 3 (e,o,o), 5 (o,e,o), 6 (o,o,e)
  int N[3];
  __host__ __device__ int dim(int p, int ax) const { return N[ax] + (((p >> (2 - ax)) & 1) ? 0 : 1); }
  __host__ __device__ int64_t count(int p) const { return (int64_t)dim(p, 0) * dim(p, 1) * dim(p, 2); }
};
__host__ __device__ constexpr int face_class(int comp) { return comp == 0 ? 3 : (comp == 1 ? 5 : 6); }
__host__ __device__ constexpr int fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }

template <typename T>
__global__ void __launch_bounds__(256)
k_vcg_setup(int Nx, int Ny, int Nz, const void* sphi, int sdt, const void* vol, int voldt, T* o1, T* o2, T* o3, T* o4,
            T* o5, T* o6, T* o7, unsigned char* m3, unsigned char* m5, unsigned char* m6) {
  const int d1 = 2 * Ny + 1, d2 = 2 * Nz + 1;
  const int64_t n = (int64_t)(2 * Nx + 1) * d1 * d2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = (int)(i % d2), j = (int)((i / d2) % d1), ii = (int)(i / ((int64_t)d2 * d1));
  const int p = ((ii & 1) << 2) | ((j & 1) << 1) | (k & 1);
  if (p == 0) return;
  const int e1 = Ny + ((j & 1) ? 0 : 1), e2 = Nz + ((k & 1) ? 0 : 1);
  const int64_t ci = ((int64_t)(ii >> 1) * e1 + (j >> 1)) * e2 + (k >> 1);
  T* dst = p == 1 ? o1 : p == 2 ? o2 : p == 3 ? o3 : p == 4 ? o4 : p == 5 ? o5 : p == 6 ? o6 : o7;
  dst[ci] = (T)ldx(vol, voldt, i);
  if (p == 3 || p == 5 || p == 6) {
    unsigned char* m = p == 3 ? m3 : (p == 5 ? m5 : m6);
    m[ci] = ldx(sphi, sdt, i) >= 0 ? 1 : 0;
  }
}

template <typename T>
struct Vec3T { const T* p[3]; };

struct Box3 { int lo[3], hi[3]; };   // face-index box [lo, hi) of one row

// One operator row at face (x,y,z) on the compact arrays.  All loads are
// unconditional (no load depends on a mask value), so the ~40 loads of a face are
// in flight together.  MASK=false drops the tap masks: in the CG the operand is
// `d`, which is exactly 0 on solid and array-boundary faces, so they cannot matter;
// the initial q = A x (x = extrapolated velocity, non-zero on solid faces) and the
// stand-alone apply use MASK=true.  The row's OWN mask (solid face -> out = 0) always applies.
template <typename T, int AXIS, bool MASK>
__device__ __forceinline__ double vcg_row(const Compact& c, double k1, double k2, const Vec3T<T>& v, int x, int y,
                                          int z, double& own_out) {
  const int s1 = c.N[1] + (AXIS == 1), s2 = c.N[2] + (AXIS == 2);
  const int64_t f = ((int64_t)x * s1 + y) * s2 + z;
  double vs[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int ax = kD0[AXIS][0] + kVolOff[k][0], ay = kD0[AXIS][1] + kVolOff[k][1], az = kD0[AXIS][2] + kVolOff[k][2];
    const int p = ((ax & 1) << 2) | ((ay & 1) << 1) | (az & 1);
    const int e1 = c.N[1] + ((ay & 1) ? 0 : 1), e2 = c.N[2] + ((az & 1) ? 0 : 1);
    vs[k] = (double)((const T*)c.vol[p])[((int64_t)(x + fdiv2(ax)) * e1 + (y + fdiv2(ay))) * e2 + (z + fdiv2(az))];
  }
  const double own = (double)v.p[AXIS][f];
  const bool own_ok = c.msk[face_class(AXIS)][f] != 0;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double t = kDiagFac[AXIS][k] == 2 ? 2 * vs[k + 1] : vs[k + 1];
    s = k == 0 ? t : s + t;
  }
  double val = (vs[0] + k1 * s) * own;
#pragma unroll
  for (int t = 0; t < 14; ++t) {
    const VTap tp = kTaps[AXIS][t];
    const int cs1 = c.N[1] + (tp.comp == 1), cs2 = c.N[2] + (tp.comp == 2);
    const int64_t nb = ((int64_t)(x + tp.dx) * cs1 + (y + tp.dy)) * cs2 + (z + tp.dz);
    double nbv = (double)v.p[tp.comp][nb];
    // the tap's mask sample is the validity of the tapped face itself (see header)
    if (MASK) nbv = c.msk[face_class(tp.comp)][nb] ? nbv : 0.0;
    val += tp.sgn * ((tp.fac == 2 ? k2 : k1) * vs[tp.vol] * nbv);
  }
  own_out = own;
  return own_ok ? val : 0.0;
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
                  T* __restrict__ oz, int xchunk, int nbz, int nby, double* __restrict__ partial,
                  const double* __restrict__ done_flag) {
  if (done_flag && *done_flag != 0.0) return;
  const int Nx = c.N[0], Ny = c.N[1], Nz = c.N[2];
  const int bz = blockIdx.x % nbz, by = (blockIdx.x / nbz) % nby, bxk = blockIdx.x / (nbz * nby);
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

}  // namespace mfs

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
};

static int64_t class_count(const int64_t gres[3], int p) {
  int64_t n = 1;
  for (int ax = 0; ax < 3; ++ax) n *= gres[ax] + (((p >> (2 - ax)) & 1) ? 0 : 1);
  return n;
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

template <typename T, bool MASK>
static int vcg_apply_TM(mfs_vcg3d* h, const void* v, void* out, double* partial, const double* done, hipStream_t st,
                        int* nparts) {
  const T* vb = (const T*)v;
  T* ob = (T*)out;
  Vec3T<T> vv{{vb + h->off[0], vb + h->off[1], vb + h->off[2]}};
  const int Nx = h->g.N[0], Ny = h->g.N[1], Nz = h->g.N[2];
  int used = 0;
  // (1) the box where all three rows are interior faces: one fused launch
  if (Nx >= 3 && Ny >= 3 && Nz >= 3) {
    const int nbz = (Nz - 2 + 63) / 64, nby = (Ny - 2 + 3) / 4;
    int xchunk = 8;
    while ((int64_t)nbz * nby * ((Nx - 2 + xchunk - 1) / xchunk) > 4096) xchunk *= 2;
    const int grid = nbz * nby * ((Nx - 2 + xchunk - 1) / xchunk);
    hipLaunchKernelGGL((k_vcg_apply_fused<T, MASK>), dim3(grid), dim3(256), 0, st, h->cp, h->k1, h->k2, vv,
                       ob + h->off[0], ob + h->off[1], ob + h->off[2], xchunk, nbz, nby, partial + used, done);
    used += grid;
  }
  // (2) the three slabs of interior faces outside that box: u at x = Nx-1, v at y = Ny-1, w at z = Nz-1
  auto slab = [&](int ax) {
    Box3 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = 1; b.hi[a] = h->g.sh(ax, a) - 1; }
    b.lo[ax] = h->g.N[ax] - 1;
    b.hi[ax] = h->g.N[ax];
    return b;
  };
  auto sgrid = [&](const Box3& b) {
    const int64_t n = (int64_t)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
    return (int)std::max<int64_t>(1, std::min<int64_t>(256, (n + 255) / 256));
  };
  if (Nx >= 2 && Ny >= 3 && Nz >= 3) {
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

// ------------------------------------------------------------------ engine ---
size_t mfs_vcg3d_workspace_bytes(const int64_t gres[3], int dt) {
  if (!gres || !dtype_ok(dt)) return 0;
  size_t tot = core_ws_bytes() + 4096;
  for (int p = 1; p < 8; ++p) tot += align_up((size_t)class_count(gres, p) * dtype_size(dt), 4096) + 4096;
  for (int p : {3, 5, 6}) tot += align_up((size_t)class_count(gres, p), 4096);
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
  h->cp.vol[0] = nullptr;
  for (int q = 0; q < 8; ++q) h->cp.msk[q] = nullptr;
  for (int q = 1; q < 8; ++q) { h->cp.vol[q] = p; p += align_up((size_t)class_count(gres, q) * h->c.elt, 4096) + 4096; }
  for (int q : {3, 5, 6}) { h->cp.msk[q] = (unsigned char*)p; p += align_up((size_t)class_count(gres, q), 4096); }
  h->grid_row = std::min(kMaxPartials / 3, h->c.cus * env_int("MFS_VISC_BLOCKS_PER_CU", 8));
  h->is_setup = false;
  h->mask_cg = env_int("MFS_VISC_MASK_CG", 0);
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
  unsigned char *m3 = (unsigned char*)h->cp.msk[3], *m5 = (unsigned char*)h->cp.msk[5], *m6 = (unsigned char*)h->cp.msk[6];
  void* const* v = (void* const*)h->cp.vol;
  if (h->dt == MFS_F32)
    hipLaunchKernelGGL((k_vcg_setup<float>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, Nx, Ny, Nz, sphi,
                       sphi_dt, vol, vol_dt, (float*)v[1], (float*)v[2], (float*)v[3], (float*)v[4], (float*)v[5],
                       (float*)v[6], (float*)v[7], m3, m5, m6);
  else
    hipLaunchKernelGGL((k_vcg_setup<double>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, Nx, Ny, Nz, sphi,
                       sphi_dt, vol, vol_dt, (double*)v[1], (double*)v[2], (double*)v[3], (double*)v[4],
                       (double*)v[5], (double*)v[6], (double*)v[7], m3, m5, m6);
  MFS_LAUNCH_CHECK();
  h->k1 = scale * mu;          // `scale * mu * ...`      (left to right, as the reference evaluates it)
  h->k2 = 2 * scale * mu;      // `2 * scale * mu * ...`
  h->is_setup = true;
  return MFS_OK;
}

int mfs_vcg3d_apply(mfs_vcg3d* h, const void* v, void* out, mfs_stream stream) {
  MFS_REQUIRE(h && v && out && v != out, "null / aliased argument");
  MFS_REQUIRE(h->is_setup, "mfs_vcg3d_setup has not been called");
  int np = 0;
  if (int e = vcg_apply(h, v, out, h->c.part_dq, false, true, (hipStream_t)stream, &np)) return e;
  h->c.n_part_dq = np;
  return MFS_OK;
}

int mfs_vcg3d_bind(mfs_vcg3d* h, void* b, void* x, void* d, void* r, void* q) {
  MFS_REQUIRE(h, "null handle");
  return core_bind(h->c, b, x, d, r, q);
}

int mfs_vcg3d_begin(mfs_vcg3d* h, double tol, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  if (int e = core_begin_pre(h->c, tol, false, st)) return e;     // x keeps the extrapolated velocity (:569-573)
  int np = 0;
  if (int e = vcg_apply(h, h->c.x, h->c.q, h->c.part_dq, false, true, st, &np)) return e;   // :575
  if (int e = core_begin_post(h->c, st)) return e;                // :577-585
  return core_begin_finish(h->c, st);
}

int mfs_vcg3d_iterate(mfs_vcg3d* h, int64_t n, mfs_stream stream) {
  MFS_REQUIRE(h && h->c.x && h->is_setup, "engine not bound / set up");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i = 0; i < n; ++i) {
    int e, np = 0;
    if ((e = vcg_apply(h, h->c.d, h->c.q, h->c.part_dq, true, h->mask_cg != 0, st, &np))) return e;   // :589
    h->c.n_part_dq = np;
    if ((e = core_update_xr(h->c, true, st))) return e;                               // :592-601
    if ((e = core_update_d(h->c, true, st))) return e;                                // :604-610
  }
  return MFS_OK;
}

int mfs_vcg3d_poll(mfs_vcg3d* h, mfs_stream stream, int64_t* iters, int* done, double* delta, double* alpha,
                   double* beta) {
  MFS_REQUIRE(h, "null handle");
  return core_poll(h->c, (hipStream_t)stream, iters, done, delta, alpha, beta);
}

int mfs_vcg3d_solve(mfs_vcg3d* h, double tol, int64_t max_iter, int64_t check_every, mfs_stream stream,
                    int64_t* iters_host) {
  MFS_REQUIRE(h, "null handle");
  MFS_REQUIRE(max_iter >= 0 && check_every >= 1, "max_iter / check_every");
  if (int e = mfs_vcg3d_begin(h, tol, stream)) return e;
  int64_t enq = 0, iters = 0;
  int done = 0;
  if (int e = mfs_vcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  while (!done && enq < max_iter) {
    const int64_t n = std::min(check_every, max_iter - enq);
    if (int e = mfs_vcg3d_iterate(h, n, stream)) return e;
    enq += n;
    if (int e = mfs_vcg3d_poll(h, stream, &iters, &done, nullptr, nullptr, nullptr)) return e;
  }
  if (iters_host) *iters_host = iters;
  return done ? MFS_OK : MFS_NOT_CONVERGED;
}

int64_t mfs_vcg3d_history(mfs_vcg3d* h, double* out_host, int64_t cap, mfs_stream stream) {
  if (!h) { set_error("mfs_vcg3d_history: null handle"); return MFS_E_INVALID; }
  return core_history(h->c, out_host, cap, (hipStream_t)stream);
}

}  // extern "C"
