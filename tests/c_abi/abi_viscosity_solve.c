/* A plain C99 caller of the C ABI for the second hot path: ViscosityCGSolver3D.solve (solver/ViscosityCGSolver3D.py:532-613)
 * through libmfs_hip.so, no Python / torch in the process --
 *   mfs_visc_extrapolate3d -> mfs_visc_rhs3d -> mfs_vcg3d_{create, setup, bind, solve, history} -> mfs_visc_writeback3d
 * on flat [x-faces | y-faces | z-faces] vectors -- checked against the oracle's C restatement (oracle/mfs_oracle_c.c, TEST
 * INFRASTRUCTURE linked into this program only) started from the SAME right-hand side and initial guess.
 * Exit code 0 = agreement; prints one line.   Built and run by tests/test_c_abi_gpu.py.                                  */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mfs.h"

int64_t mfs_oracle_visc_cg3d(const int64_t g[3], double scale, double mu, const double* b, double* x, double* d, double* r,
                             double* q, const double* sphi, const double* vol, double tol, int64_t max_iter, double* history,
                             int64_t hist_cap, double* delta_out, int* converged);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MFS(x) do { int s_ = (x); if (s_ < 0) { fprintf(stderr, "%s: status %d: %s\n", #x, s_, mfs_last_error()); return 3; } } while (0)

static uint64_t lcg_state = 777;
static double lcg(void) {
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg_state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}
static double clamp01(double v) { return v < 0 ? 0 : (v > 1 ? 1 : v); }
/* fraction of the sub-cell [c - h, c + h] covered by [lo, hi] */
static double cover(double c, double h, double lo, double hi) { return clamp01((fmin(c + h, hi) - fmax(c - h, lo)) / (2 * h)); }

int main(int argc, char** argv) {
  const int N0 = argc > 1 ? atoi(argv[1]) : 12, N1 = argc > 2 ? atoi(argv[2]) : 16, N2 = argc > 3 ? atoi(argv[3]) : 20;
  const int64_t g[3] = {N0, N1, N2};
  const double cs[3] = {1.0 / N0, 1.0 / N1, 1.0 / N2};
  const int64_t D0 = 2 * N0 + 1, D1 = 2 * N1 + 1, D2 = 2 * N2 + 1, nd = D0 * D1 * D2;
  const int64_t nf[3] = {(int64_t)(N0 + 1) * N1 * N2, (int64_t)N0 * (N1 + 1) * N2, (int64_t)N0 * N1 * (N2 + 1)};
  const int64_t off[4] = {0, nf[0], nf[0] + nf[1], nf[0] + nf[1] + nf[2]};
  const int64_t n = off[3];
  const double dt = 1.0 / 300.0, rho = 1000.0, mu = 5.0, cell_vol = cs[0] * cs[1] * cs[2], scale = dt / cell_vol / rho;

  /* scene: a container with 1.6-cell walls (solid where sphi < 0) and a block of liquid: vol = covered fraction of each sub-cell */
  double* sphi = (double*)malloc(nd * 8);
  double* vol = (double*)malloc(nd * 8);
  for (int64_t i = 0; i < D0; ++i) for (int64_t j = 0; j < D1; ++j) for (int64_t k = 0; k < D2; ++k) {
    const double X = 0.5 * i * cs[0], Y = 0.5 * j * cs[1], Z = 0.5 * k * cs[2];
    sphi[(i * D1 + j) * D2 + k] = fmin(fmin(fmin(X, 1 - X) - 1.6 * cs[0], fmin(Y, 1 - Y) - 1.6 * cs[1]), fmin(Z, 1 - Z) - 1.6 * cs[2]);
    vol[(i * D1 + j) * D2 + k] = cover(X, 0.25 * cs[0], 0.22, 0.71) * cover(Y, 0.25 * cs[1], 0.18, 0.66) * cover(Z, 0.25 * cs[2], 0.3, 0.8);
  }
  double* v0 = (double*)malloc(n * 8);          /* the velocities handed to solve(): flat [vx | vy | vz] */
  for (int64_t i = 0; i < n; ++i) v0[i] = lcg();

  HIP(hipSetDevice(0));
  void *d_sphi = NULL, *d_vol = NULL, *d_v = NULL, *d_x = NULL, *d_b = NULL, *d_d = NULL, *d_r = NULL, *d_q = NULL, *ws = NULL, *ews = NULL;
  HIP(hipMalloc(&d_sphi, nd * 8)); HIP(hipMalloc(&d_vol, nd * 8));
  HIP(hipMemcpy(d_sphi, sphi, nd * 8, hipMemcpyHostToDevice)); HIP(hipMemcpy(d_vol, vol, nd * 8, hipMemcpyHostToDevice));
  HIP(hipMalloc(&d_v, n * 8)); HIP(hipMemcpy(d_v, v0, n * 8, hipMemcpyHostToDevice));
  HIP(hipMalloc(&d_x, n * 8)); HIP(hipMemcpy(d_x, v0, n * 8, hipMemcpyHostToDevice));       /* x = a copy of the velocities (:569-571) */
  HIP(hipMalloc(&d_b, n * 8)); HIP(hipMemset(d_b, 0, n * 8));
  HIP(hipMalloc(&d_d, n * 8)); HIP(hipMemset(d_d, 0, n * 8));
  HIP(hipMalloc(&d_r, n * 8)); HIP(hipMemset(d_r, 0, n * 8));
  HIP(hipMalloc(&d_q, n * 8)); HIP(hipMemset(d_q, 0, n * 8));
  hipStream_t st = NULL;
#define C3(p) (char*)(p) + off[0] * 8, (char*)(p) + off[1] * 8, (char*)(p) + off[2] * 8
  const size_t ewb = mfs_visc_extrapolate3d_workspace_bytes(g, MFS_F64);
  HIP(hipMalloc(&ews, ewb));
  MFS(mfs_visc_extrapolate3d(g, 3, C3(d_x), MFS_F64, d_sphi, MFS_F64, ews, ewb, st));                       /* :573 */
  MFS(mfs_visc_rhs3d(g, scale, mu, C3(d_x), MFS_F64, d_sphi, MFS_F64, d_vol, MFS_F64, C3(d_b), MFS_F64, st)); /* :574 */
  if (mfs_vcg3d_dofs(g) != n) { fprintf(stderr, "mfs_vcg3d_dofs %lld != %lld\n", (long long)mfs_vcg3d_dofs(g), (long long)n); return 4; }
  const size_t wsb = mfs_vcg3d_workspace_bytes(g, MFS_F64);
  HIP(hipMalloc(&ws, wsb)); HIP(hipMemset(ws, 0, wsb));
  mfs_vcg3d* h = NULL;
  MFS(mfs_vcg3d_create(&h, g, MFS_F64, ws, wsb, st));
  MFS(mfs_vcg3d_setup(h, scale, mu, d_sphi, MFS_F64, d_vol, MFS_F64, st));
  MFS(mfs_vcg3d_bind(h, d_b, d_x, d_d, d_r, d_q));
  /* the host-side checker needs what the device loop starts from: b and the extrapolated x */
  double *b = (double*)malloc(n * 8), *x0 = (double*)malloc(n * 8), *x = (double*)malloc(n * 8), *vout = (double*)malloc(n * 8);
  HIP(hipMemcpy(b, d_b, n * 8, hipMemcpyDeviceToHost)); HIP(hipMemcpy(x0, d_x, n * 8, hipMemcpyDeviceToHost));
  const double tol = 1e-8;
  int64_t iters = -1;
  const int status = mfs_vcg3d_solve(h, tol, n, 16, st, &iters);
  if (status != MFS_OK) { fprintf(stderr, "mfs_vcg3d_solve: status %d (%s)\n", status, status < 0 ? mfs_last_error() : "not converged"); return 3; }
  enum { HCAP = 4096 };
  static double hist[HCAP], ohist[HCAP];
  const int64_t hn = mfs_vcg3d_history(h, hist, HCAP, st);
  if (hn < 0) { fprintf(stderr, "mfs_vcg3d_history: %s\n", mfs_last_error()); return 3; }
  MFS(mfs_visc_writeback3d(g, C3(d_v), MFS_F64, C3(d_x), MFS_F64, d_sphi, MFS_F64, st));                      /* :612 */
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(x, d_x, n * 8, hipMemcpyDeviceToHost)); HIP(hipMemcpy(vout, d_v, n * 8, hipMemcpyDeviceToHost));

  double *od = (double*)calloc(n, 8), *orr = (double*)calloc(n, 8), *oq = (double*)calloc(n, 8);
  int oconv = 0;
  const int64_t oit = mfs_oracle_visc_cg3d(g, scale, mu, b, x0, od, orr, oq, sphi, vol, tol, n, ohist, HCAP, NULL, &oconv);   /* x0 -> solution */
  double bmax = 0, xmax = 0, xdev = 0, hdev = 0, vchg = 0;
  for (int64_t i = 0; i < n; ++i) { bmax = fmax(bmax, fabs(b[i])); xmax = fmax(xmax, fabs(x0[i])); xdev = fmax(xdev, fabs(x[i] - x0[i])); vchg = fmax(vchg, fabs(vout[i] - v0[i])); }
  const int64_t hcmp = hn < 21 ? hn : 21;
  for (int64_t k = 0; k < hcmp; ++k) hdev = fmax(hdev, fabs(hist[k] - ohist[k]) / fabs(ohist[k]));
  const long long itol = oit / 20 > 3 ? oit / 20 : 3;
  const int ok = oconv && iters > 3 && llabs((long long)(iters - oit)) <= itol && hdev < 1e-9 && xdev <= 1e-8 * xmax && bmax > 0 && vchg > 0;
  printf("%s grid %dx%dx%d, %lld unknowns | library: %lld iterations, history %lld entries | oracle: %lld iterations | history dev (first %lld) %.2e  "
         "x dev / max %.2e | velocity changed by up to %.3f\n", ok ? "OK" : "MISMATCH", N0, N1, N2, (long long)n, (long long)iters, (long long)hn,
         (long long)oit, (long long)hcmp, hdev, xdev / xmax, vchg);
  MFS(mfs_vcg3d_destroy(h));
  return ok ? 0 : 1;
}
