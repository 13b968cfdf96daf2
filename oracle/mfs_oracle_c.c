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
 */
#include <math.h>
#include <stdint.h>
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
  if (nphi < 0) { val -= w * v[NB]; diag += w; }     \
  else { diag += w / theta(phi, nphi); }
        TAP(c + Ny * Nz, wx[((x + 1) * Ny + y) * Nz + z])          /* +x :68-76 */
        TAP(c - Ny * Nz, wx[(x * Ny + y) * Nz + z])                /* -x */
        TAP(c + Nz, wy[(x * (Ny + 1) + y + 1) * Nz + z])           /* +y */
        TAP(c - Nz, wy[(x * (Ny + 1) + y) * Nz + z])               /* -y */
        TAP(c + 1, wz[(x * Ny + y) * (Nz + 1) + z + 1])            /* +z */
        TAP(c - 1, wz[(x * Ny + y) * (Nz + 1) + z])                /* -z */
#undef TAP
        val += diag * v[c];                                        /* :128 */
        out[c] = val;
      }
}

static double dot(const double* a, const double* b, int64_t n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
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
      for (int64_t i = 0; i < n; ++i) { x[i] += alpha * d[i]; r[i] -= alpha * q[i]; }  /* :212-213 */
      const double old_delta = delta;
      delta = dot(r, r, n);                                        /* :216 */
      if (history && hn + 1 < hist_cap) { history[hn++] = dq; history[hn++] = delta; }
      if (delta < tol * tol) { conv = 1; break; }                  /* :218 */
      const double beta = delta / old_delta;                       /* :220 */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) d[i] = r[i] + beta * d[i];   /* :221 */
    }
    if (!conv) it = max_iter;
  }
  if (delta_out) *delta_out = delta;
  if (converged) *converged = conv;
  return it;
}
