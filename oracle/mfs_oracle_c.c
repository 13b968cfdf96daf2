/* mfs_oracle_c.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (never on a product path).
 *
 * Plain-C (OpenMP) restatement of the reference's pressure hot loop, used (a) as a
 * second, independent checker next to oracle/mfs_oracle.py and (b) as the
 * `cpu_baseline` leg of bench.py ("port", all host cores).  fp64 like the
 * reference.  Pinned against tests/golden/p3d_*.npz (tests/test_oracle_c.py), i.e.
 * against the reference's own source executed under CPython (see
 * tests/golden/make_goldens.py); not pinned against cupy/numba-CUDA execution.
 *
 * Follows, statement for statement:
 *   mfs_oracle_pressure_apply3d  <- solver/PressureCGSolver3D.py:52-130 (matvecmul_kernel)
 *   mfs_oracle_pressure_cg3d     <- solver/PressureCGSolver3D.py:198-223 (the CG loop)
 *   mfs_oracle_visc_apply3d      <- solver/ViscosityCGSolver3D.py:248-456 (matvecmul_{x,y,z}_kernel), from the tap
 *                                   tables of SURVEY.md Appendix A (re-read against the reference lines cited there)
 *   mfs_oracle_visc_cg3d         <- solver/ViscosityCGSolver3D.py:575-612 (the CG loop over the three components)
 * The viscosity half is pinned against tests/golden/v3d_*.npz (tests/test_oracle_c.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int mfs_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void mfs_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---- rounding variants (tests/test_history_envelope.py: how far does the residual history move when ONLY rounding
 * changes?).  Variant 0 is the oracle proper.  `dot_variant` permutes the summation order of the dot products,
 * `fma_mask` fuses multiply-adds: bit 0 in the dot products, bit 1 in the x / r / d updates, bit 2 in the operator's
 * accumulation.  Every variant is the reference's algorithm statement for statement; they differ in rounding only. */
static int g_dot_variant = 0, g_fma_mask = 0;
void mfs_oracle_set_variant(int dot_variant, int fma_mask) { g_dot_variant = dot_variant; g_fma_mask = fma_mask; }
#define MADD(bit, a, b, c) ((g_fma_mask & (bit)) ? fma((a), (b), (c)) : (a) * (b) + (c))

static inline double theta(double phi, double nphi) { /* :75  min(1, max(0.01, phi/(phi-nphi))) */
  double f = phi / (phi - nphi);
  if (f < 0.01) f = 0.01;
  if (f > 1.0) f = 1.0;
  return f;
}

void mfs_oracle_pressure_apply3d(const int64_t g[3], const double* v, double* out, const double* wx,
                                 const double* wy, const double* wz, const double* lphi) {
  const int64_t Nx = g[0], Ny = g[1], Nz = g[2];
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t x = 1; x < Nx - 1; ++x)       /* boundary cells ignored (:55-57) */
    for (int64_t y = 1; y < Ny - 1; ++y)
      for (int64_t z = 1; z < Nz - 1; ++z) {
        const int64_t c = (x * Ny + y) * Nz + z;
        const double phi = lphi[c];
        if (phi >= 0) { out[c] = 0; continue; }   /* :60-63 */
        double val = 0.0, diag = 0.0, nphi, w;
#define TAP(NB, W)                                   \
  nphi = lphi[NB]; w = (W);                          \
  if (nphi < 0) { val = MADD(4, -w, v[NB], val); diag += w; }     \
  else { diag += w / theta(phi, nphi); }
        TAP(c + Ny * Nz, wx[((x + 1) * Ny + y) * Nz + z])          /* +x :68-76 */
        TAP(c - Ny * Nz, wx[(x * Ny + y) * Nz + z])                /* -x */
        TAP(c + Nz, wy[(x * (Ny + 1) + y + 1) * Nz + z])           /* +y */
        TAP(c - Nz, wy[(x * (Ny + 1) + y) * Nz + z])               /* -y */
        TAP(c + 1, wz[(x * Ny + y) * (Nz + 1) + z + 1])            /* +z */
        TAP(c - 1, wz[(x * Ny + y) * (Nz + 1) + z])                /* -z */
#undef TAP
        val = MADD(4, diag, v[c], val);                            /* :128 */
        out[c] = val;
      }
}

/* Dot product with a FIXED summation order, whatever the thread count: partial sums over blocks of consecutive
 * products, then the partials added in block order (round 2's `omp reduction` combined the threads' sums in arrival
 * order -- the history, and with it a test, changed with OMP_NUM_THREADS).  Variants permute the order (see above). */
static double block_sum(const double* a, const double* b, int64_t lo, int64_t hi, int variant) {
  const int f = g_fma_mask & 1;
  if (variant == 3) {                                   /* backwards inside the block */
    double s = 0.0;
    for (int64_t i = hi - 1; i >= lo; --i) s = f ? fma(a[i], b[i], s) : s + a[i] * b[i];
    return s;
  }
  if (variant == 5) {                                   /* four interleaved chains (a vector unit's order) */
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = lo; i < hi; ++i) s4[(i - lo) & 3] = f ? fma(a[i], b[i], s4[(i - lo) & 3]) : s4[(i - lo) & 3] + a[i] * b[i];
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  if (variant == 7) {                                   /* extended-precision accumulator */
    long double s = 0.0L;
    for (int64_t i = lo; i < hi; ++i) s += (long double)a[i] * (long double)b[i];
    return (double)s;
  }
  double s = 0.0;
  for (int64_t i = lo; i < hi; ++i) s = f ? fma(a[i], b[i], s) : s + a[i] * b[i];
  return s;
}

static double dot(const double* a, const double* b, int64_t n) {
  static const int64_t kBlock[8] = {4096, 1024, 16384, 4096, 4096, 4096, 333, 4096};
  const int variant = g_dot_variant & 7;
  const int64_t B = kBlock[variant];
  const int64_t nb = (n + B - 1) / B;
  double stack_part[1024];
  double* part = nb <= 1024 ? stack_part : (double*)malloc((size_t)nb * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < nb; ++k) part[k] = block_sum(a, b, k * B, (k + 1) * B < n ? (k + 1) * B : n, variant);
  double s = 0.0;
  if (variant == 4) { for (int64_t k = nb - 1; k >= 0; --k) s += part[k]; }           /* partials in reverse */
  else if (g_dot_variant & 8) {                                                       /* pairwise tree over the partials */
    for (int64_t w = 1; w < nb; w *= 2)
      for (int64_t k = 0; k + w < nb; k += 2 * w) part[k] += part[k + w];
    s = nb ? part[0] : 0.0;
  } else { for (int64_t k = 0; k < nb; ++k) s += part[k]; }
  if (part != stack_part) free(part);
  return s;
}

/* returns the number of iterations performed; *converged = 1 if delta < tol^2 was reached.
 * history (if non-null) receives [delta0, dq1, delta1, ...] up to hist_cap values. */
int64_t mfs_oracle_pressure_cg3d(const int64_t g[3], const double* b, double* x, double* d, double* r, double* q,
                                 const double* wx, const double* wy, const double* wz, const double* lphi,
                                 double tol, int64_t max_iter, double* history, int64_t hist_cap, double* delta_out,
                                 int* converged) {
  const int64_t n = g[0] * g[1] * g[2];
  int64_t hn = 0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) x[i] *= 0.0;                     /* :198 */
  mfs_oracle_pressure_apply3d(g, x, q, wx, wy, wz, lphi);          /* :201 */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { d[i] = b[i] - q[i]; r[i] = d[i]; }   /* :202-203 */
  double delta = dot(r, r, n);                                     /* :204 */
  if (history && hn < hist_cap) history[hn++] = delta;
  int64_t it = 0;
  int conv = delta < tol * tol;
  if (!conv) {
    for (it = 1; it <= max_iter; ++it) {                           /* :207 */
      mfs_oracle_pressure_apply3d(g, d, q, wx, wy, wz, lphi);      /* :208 */
      const double dq = dot(d, q, n);
      const double alpha = delta / dq;                             /* :211 */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) { x[i] = MADD(2, alpha, d[i], x[i]); r[i] = MADD(2, -alpha, q[i], r[i]); }  /* :212-213 */
      const double old_delta = delta;
      delta = dot(r, r, n);                                        /* :216 */
      if (history && hn + 1 < hist_cap) { history[hn++] = dq; history[hn++] = delta; }
      if (delta < tol * tol) { conv = 1; break; }                  /* :218 */
      const double beta = delta / old_delta;                       /* :220 */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) d[i] = MADD(2, beta, d[i], r[i]);   /* :221 */
    }
    if (!conv) it = max_iter;
  }
  if (delta_out) *delta_out = delta;
  if (converged) *converged = conv;
  return it;
}

/* ------------------------------------------------------------------ viscosity ---
 * Doubled-grid conventions: sphi, vol are (2Nx+1, 2Ny+1, 2Nz+1); face (x,y,z) of component c sits at
 * D = 2*(x,y,z) + D0[c].  A tap = {factor (1|2), vol sample (0 c, 1 R, 2 L, 3 T, 4 B, 5 F, 6 K), sign, component,
 * (dx,dy,dz) into that component's array, mask offset relative to D}.                                             */
typedef struct { int fac, vol, sgn, comp, dx, dy, dz, mx, my, mz; } vtap;
static const int kD0[3][3] = {{0, 1, 1}, {1, 0, 1}, {1, 1, 0}};
static const int kVolOff[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
static const int kDiag[3][6] = {{2, 2, 1, 1, 1, 1}, {1, 1, 2, 2, 1, 1}, {1, 1, 1, 1, 2, 2}};   /* :268, :338, :408 */
static const vtap kTaps[3][14] = {
    {   /* u row :271-314 */
        {2, 1, -1, 0, 1, 0, 0, 2, 0, 0},    {2, 2, -1, 0, -1, 0, 0, -2, 0, 0},  {1, 3, -1, 0, 0, 1, 0, 0, 2, 0},
        {1, 4, -1, 0, 0, -1, 0, 0, -2, 0},  {1, 5, -1, 0, 0, 0, 1, 0, 0, 2},    {1, 6, -1, 0, 0, 0, -1, 0, 0, -2},
        {1, 3, -1, 1, 0, 1, 0, 1, 1, 0},    {1, 3, +1, 1, -1, 1, 0, -1, 1, 0},  {1, 4, +1, 1, 0, 0, 0, 1, -1, 0},
        {1, 4, -1, 1, -1, 0, 0, -1, -1, 0}, {1, 5, -1, 2, 0, 0, 1, 1, 0, 1},    {1, 5, +1, 2, -1, 0, 1, -1, 0, 1},
        {1, 6, +1, 2, 0, 0, 0, 1, 0, -1},   {1, 6, -1, 2, -1, 0, 0, -1, 0, -1},
    },
    {   /* v row :341-384 */
        {1, 1, -1, 1, 1, 0, 0, 2, 0, 0},    {1, 2, -1, 1, -1, 0, 0, -2, 0, 0},  {2, 3, -1, 1, 0, 1, 0, 0, 2, 0},
        {2, 4, -1, 1, 0, -1, 0, 0, -2, 0},  {1, 5, -1, 1, 0, 0, 1, 0, 0, 2},    {1, 6, -1, 1, 0, 0, -1, 0, 0, -2},
        {1, 1, -1, 0, 1, 0, 0, 1, 1, 0},    {1, 1, +1, 0, 1, -1, 0, 1, -1, 0},  {1, 2, +1, 0, 0, 0, 0, -1, 1, 0},
        {1, 2, -1, 0, 0, -1, 0, -1, -1, 0}, {1, 5, -1, 2, 0, 0, 1, 0, 1, 1},    {1, 5, +1, 2, 0, -1, 1, 0, -1, 1},
        {1, 6, +1, 2, 0, 0, 0, 0, 1, -1},   {1, 6, -1, 2, 0, -1, 0, 0, -1, -1},
    },
    {   /* w row :411-454 */
        {1, 1, -1, 2, 1, 0, 0, 2, 0, 0},    {1, 2, -1, 2, -1, 0, 0, -2, 0, 0},  {1, 3, -1, 2, 0, 1, 0, 0, 2, 0},
        {1, 4, -1, 2, 0, -1, 0, 0, -2, 0},  {2, 5, -1, 2, 0, 0, 1, 0, 0, 2},    {2, 6, -1, 2, 0, 0, -1, 0, 0, -2},
        {1, 1, -1, 0, 1, 0, 0, 1, 0, 1},    {1, 1, +1, 0, 1, 0, -1, 1, 0, -1},  {1, 2, +1, 0, 0, 0, 0, -1, 0, 1},
        {1, 2, -1, 0, 0, 0, -1, -1, 0, -1}, {1, 3, -1, 1, 0, 1, 0, 0, 1, 1},    {1, 3, +1, 1, 0, 1, -1, 0, 1, -1},
        {1, 4, +1, 1, 0, 0, 0, 0, -1, 1},   {1, 4, -1, 1, 0, 0, -1, 0, -1, -1},
    },
};

/* out_c = (A v)_c on the interior faces of the three components; solid faces -> 0; array-boundary faces untouched */
void mfs_oracle_visc_apply3d(const int64_t g[3], double scale, double mu, const double* vx, const double* vy,
                             const double* vz, double* ox, double* oy, double* oz, const double* sphi,
                             const double* vol) {
  const double* v[3] = {vx, vy, vz};
  double* o[3] = {ox, oy, oz};
  const int64_t d1 = 2 * g[1] + 1, d2 = 2 * g[2] + 1;
  for (int c = 0; c < 3; ++c) {
    int64_t sh[3][3];                                  /* shapes of the three component arrays */
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) sh[a][b] = g[b] + (a == b ? 1 : 0);
    const int64_t s0 = sh[c][0], s1 = sh[c][1], s2 = sh[c][2];
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t x = 1; x < s0 - 1; ++x)
      for (int64_t y = 1; y < s1 - 1; ++y)
        for (int64_t z = 1; z < s2 - 1; ++z) {
          const int64_t i = (x * s1 + y) * s2 + z;
          const int64_t Dx = 2 * x + kD0[c][0], Dy = 2 * y + kD0[c][1], Dz = 2 * z + kD0[c][2];
#define DG(a, b, cc) (((a) * d1 + (b)) * d2 + (cc))
          if (sphi[DG(Dx, Dy, Dz)] < 0) { o[c][i] = 0.0; continue; }          /* solid face */
          double vs[7];
          for (int k = 0; k < 7; ++k) vs[k] = vol[DG(Dx + kVolOff[k][0], Dy + kVolOff[k][1], Dz + kVolOff[k][2])];
          double s = 0.0;
          for (int k = 0; k < 6; ++k) {
            const double t = kDiag[c][k] == 2 ? 2 * vs[k + 1] : vs[k + 1];
            s = k == 0 ? t : s + t;
          }
          double val = (vs[0] + scale * mu * s) * v[c][i];                    /* diag * v */
          for (int t = 0; t < 14; ++t) {
            const vtap tp = kTaps[c][t];
            if (sphi[DG(Dx + tp.mx, Dy + tp.my, Dz + tp.mz)] < 0) continue;   /* neighbour face inside the solid */
            const double k = tp.fac == 2 ? 2 * scale * mu : scale * mu;
            const int64_t* ss = sh[tp.comp];
            const double nb = v[tp.comp][((x + tp.dx) * ss[1] + (y + tp.dy)) * ss[2] + (z + tp.dz)];
            val = MADD(4, tp.sgn * (k * vs[tp.vol]), nb, val);
          }
#undef DG
          o[c][i] = val;
        }
  }
}

/* the CG loop over the concatenated (x, y, z) vector: V = [n0 | n1 | n2] doubles, component c at V + off[c].
 * x holds the initial guess (the extrapolated velocity, :569-573); returns the iteration count.            */
int64_t mfs_oracle_visc_cg3d(const int64_t g[3], double scale, double mu, const double* b, double* x, double* d,
                             double* r, double* q, const double* sphi, const double* vol, double tol, int64_t max_iter,
                             double* history, int64_t hist_cap, double* delta_out, int* converged) {
  int64_t off[4] = {0, 0, 0, 0};
  for (int c = 0; c < 3; ++c) {
    int64_t n = 1;
    for (int a = 0; a < 3; ++a) n *= g[a] + (a == c ? 1 : 0);
    off[c + 1] = off[c] + n;
  }
  const int64_t n = off[3];
  int64_t hn = 0;
#define APPLY(V, O) mfs_oracle_visc_apply3d(g, scale, mu, (V) + off[0], (V) + off[1], (V) + off[2], (O) + off[0], (O) + off[1], (O) + off[2], sphi, vol)
  /* per-component sums added left to right, as `cp.sum(r_x**2) + cp.sum(r_y**2) + cp.sum(r_z**2)` (:585) */
#define DOT3(A, B) (dot((A) + off[0], (B) + off[0], off[1] - off[0]) + dot((A) + off[1], (B) + off[1], off[2] - off[1]) + dot((A) + off[2], (B) + off[2], off[3] - off[2]))
  APPLY(x, q);                                                                  /* :575 */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { d[i] = b[i] - q[i]; r[i] = d[i]; }          /* :577-583 */
  double delta = DOT3(r, r);                                                    /* :585 */
  if (history && hn < hist_cap) history[hn++] = delta;
  int64_t it = 0;
  int conv = delta < tol * tol;
  if (!conv) {
    for (it = 1; it <= max_iter; ++it) {                                        /* :588 */
      APPLY(d, q);                                                              /* :589 */
      const double dq = DOT3(d, q);                                             /* :592 */
      const double alpha = delta / dq;                                          /* :594 */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) { x[i] = MADD(2, alpha, d[i], x[i]); r[i] = MADD(2, -alpha, q[i], r[i]); }   /* :595-601 */
      const double old_delta = delta;
      delta = DOT3(r, r);                                                       /* :604 */
      if (history && hn + 1 < hist_cap) { history[hn++] = dq; history[hn++] = delta; }
      if (delta < tol * tol) { conv = 1; break; }                               /* :605 */
      const double beta = delta / old_delta;                                    /* :607 */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) d[i] = MADD(2, beta, d[i], r[i]);                /* :608-610 */
    }
    if (!conv) it = max_iter;
  }
#undef APPLY
#undef DOT3
  if (delta_out) *delta_out = delta;
  if (converged) *converged = conv;
  return it;
}
