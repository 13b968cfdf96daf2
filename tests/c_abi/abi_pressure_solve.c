/* A plain C99 caller of the C ABI (include/mfs.h) -- no Python, no torch: what a maintainer binding libmfs_hip.so from another
 * host language gets.  Builds a small pool scene on the host, runs the pressure path of the reference's
 * PressureCGSolver3D.solve (solver/PressureCGSolver3D.py:192-226) through the library on the GPU --
 *   mfs_solid_frac3d -> mfs_pressure_rhs3d -> mfs_pcg3d_{create, setup, bind, solve, history} -> mfs_pressure_update3d --
 * and checks the CG part against the oracle's C restatement (oracle/mfs_oracle_c.c: TEST INFRASTRUCTURE, linked into
 * this test program only) on the SAME right-hand side and weights: iteration count, residual history, solution.
 * Exit code 0 = agreement; prints one line.   Built and run by tests/test_c_abi_gpu.py.                                    */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfs.h"

/* oracle/mfs_oracle_c.c (the checker) */
int64_t mfs_oracle_pressure_cg3d(const int64_t g[3], const double* b, double* x, double* d, double* r, double* q,
                                 const double* wx, const double* wy, const double* wz, const double* lphi, double tol,
                                 int64_t max_iter, double* history, int64_t hist_cap, double* delta_out, int* converged);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MFS(x) do { int s_ = (x); if (s_ < 0) { fprintf(stderr, "%s: status %d: %s\n", #x, s_, mfs_last_error()); return 3; } } while (0)

static uint64_t lcg_state = 12345;
static double lcg(void) {   /* uniform in [-1, 1) */
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg_state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

static void* dev_copy(const void* host, size_t bytes) {
  void* p = NULL;
  if (hipMalloc(&p, bytes) != hipSuccess) return NULL;
  if (host) { if (hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL; }
  else if (hipMemset(p, 0, bytes) != hipSuccess) return NULL;
  return p;
}

int main(int argc, char** argv) {
  const int N0 = argc > 1 ? atoi(argv[1]) : 20, N1 = argc > 2 ? atoi(argv[2]) : 24, N2 = argc > 3 ? atoi(argv[3]) : 16;
  const int64_t g[3] = {N0, N1, N2};
  const int64_t n = (int64_t)N0 * N1 * N2;
  const double cs[3] = {1.0 / N0, 1.0 / N1, 1.0 / N2};
  const int64_t D0 = 2 * N0 + 1, D1 = 2 * N1 + 1, D2 = 2 * N2 + 1, nd = D0 * D1 * D2;
  const int64_t nx = (int64_t)(N0 + 1) * N1 * N2, ny = (int64_t)N0 * (N1 + 1) * N2, nz = (int64_t)N0 * N1 * (N2 + 1);

  /* scene: a box with walls 1.6 cells thick and a sphere (solid where sphi < 0), a pool filled to 55 % (fluid where lphi < 0) */
  double* sphi = (double*)malloc(nd * sizeof(double));
  double* sv = (double*)calloc(nd * 3, sizeof(double));
  double* lphi = (double*)malloc(n * sizeof(double));
  double *vx = (double*)malloc(nx * sizeof(double)), *vy = (double*)malloc(ny * sizeof(double)), *vz = (double*)malloc(nz * sizeof(double));
  for (int64_t i = 0; i < D0; ++i) for (int64_t j = 0; j < D1; ++j) for (int64_t k = 0; k < D2; ++k) {
    const double X = 0.5 * i * cs[0], Y = 0.5 * j * cs[1], Z = 0.5 * k * cs[2];
    double box = fmin(fmin(fmin(X, 1 - X) - 1.6 * cs[0], fmin(Y, 1 - Y) - 1.6 * cs[1]), fmin(Z, 1 - Z) - 1.6 * cs[2]);
    const double sph = sqrt((X - 0.45) * (X - 0.45) + (Y - 0.3) * (Y - 0.3) + (Z - 0.5) * (Z - 0.5)) - 0.16;
    sphi[(i * D1 + j) * D2 + k] = fmin(box, sph);
  }
  for (int64_t i = 0; i < N0; ++i) for (int64_t j = 0; j < N1; ++j) for (int64_t k = 0; k < N2; ++k)
    lphi[(i * N1 + j) * N2 + k] = (j + 0.5) * cs[1] - 0.55 + 0.03 * sin(7.0 * (i + 0.5) * cs[0]) * cos(5.0 * (k + 0.5) * cs[2]);
  for (int64_t i = 0; i < nx; ++i) vx[i] = lcg();
  for (int64_t i = 0; i < ny; ++i) vy[i] = lcg();
  for (int64_t i = 0; i < nz; ++i) vz[i] = lcg();

  HIP(hipSetDevice(0));
  void *d_sphi = dev_copy(sphi, nd * 8), *d_sv = dev_copy(sv, nd * 3 * 8), *d_lphi = dev_copy(lphi, n * 8);
  void *d_vx = dev_copy(vx, nx * 8), *d_vy = dev_copy(vy, ny * 8), *d_vz = dev_copy(vz, nz * 8);
  void *d_wx = dev_copy(NULL, nx * 8), *d_wy = dev_copy(NULL, ny * 8), *d_wz = dev_copy(NULL, nz * 8);
  void *d_b = dev_copy(NULL, n * 8), *d_x = dev_copy(NULL, n * 8), *d_d = dev_copy(NULL, n * 8), *d_r = dev_copy(NULL, n * 8), *d_q = dev_copy(NULL, n * 8);
  if (!d_sphi || !d_sv || !d_lphi || !d_vx || !d_vy || !d_vz || !d_wx || !d_wy || !d_wz || !d_b || !d_x || !d_d || !d_r || !d_q) { fprintf(stderr, "device allocation failed\n"); return 2; }
  hipStream_t st = NULL;      /* the default stream */

  if (mfs_abi_version() != MFS_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", mfs_abi_version(), MFS_ABI_VERSION); return 4; }
  MFS(mfs_solid_frac3d(g, d_sphi, MFS_F64, d_wx, d_wy, d_wz, MFS_F64, st));
  MFS(mfs_pressure_rhs3d(g, cs, d_vx, d_vy, d_vz, MFS_F64, d_sv, MFS_F64, d_lphi, MFS_F64, d_wx, d_wy, d_wz, MFS_F64, d_b, MFS_F64, st));
  const size_t wsb = mfs_pcg3d_workspace_bytes(g, MFS_F64);
  void* ws = dev_copy(NULL, wsb);
  if (!ws) { fprintf(stderr, "workspace allocation failed\n"); return 2; }
  mfs_pcg3d* h = NULL;
  MFS(mfs_pcg3d_create(&h, g, MFS_F64, ws, wsb, st));
  MFS(mfs_pcg3d_setup(h, d_lphi, MFS_F64, d_wx, d_wy, d_wz, MFS_F64, st));
  MFS(mfs_pcg3d_bind(h, d_b, d_x, d_d, d_r, d_q));
  const double tol = 1e-6;
  int64_t iters = -1;
  const int status = mfs_pcg3d_solve(h, tol, n, 16, st, &iters);
  if (status != MFS_OK) { fprintf(stderr, "mfs_pcg3d_solve: status %d (%s)\n", status, status < 0 ? mfs_last_error() : "not converged"); return 3; }
  enum { HCAP = 4096 };
  static double hist[HCAP], ohist[HCAP];
  const int64_t hn = mfs_pcg3d_history(h, hist, HCAP, st);
  if (hn < 0) { fprintf(stderr, "mfs_pcg3d_history: %s\n", mfs_last_error()); return 3; }
  MFS(mfs_pressure_update3d(g, cs, d_vx, d_vy, d_vz, MFS_F64, d_x, MFS_F64, d_wx, d_wy, d_wz, MFS_F64, d_sv, MFS_F64, d_lphi, MFS_F64, st));
  HIP(hipDeviceSynchronize());

  /* the checker: the same CG on the host from the SAME b and weights (brought back from the device) */
  double *b = (double*)malloc(n * 8), *x = (double*)malloc(n * 8), *wx = (double*)malloc(nx * 8), *wy = (double*)malloc(ny * 8), *wz = (double*)malloc(nz * 8);
  double *vxo = (double*)malloc(nx * 8);
  HIP(hipMemcpy(b, d_b, n * 8, hipMemcpyDeviceToHost)); HIP(hipMemcpy(x, d_x, n * 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(wx, d_wx, nx * 8, hipMemcpyDeviceToHost)); HIP(hipMemcpy(wy, d_wy, ny * 8, hipMemcpyDeviceToHost)); HIP(hipMemcpy(wz, d_wz, nz * 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(vxo, d_vx, nx * 8, hipMemcpyDeviceToHost));
  double *ox = (double*)calloc(n, 8), *od = (double*)calloc(n, 8), *orr = (double*)calloc(n, 8), *oq = (double*)calloc(n, 8);
  double odelta = 0.0;
  int oconv = 0;
  const int64_t oit = mfs_oracle_pressure_cg3d(g, b, ox, od, orr, oq, wx, wy, wz, lphi, tol, n, ohist, HCAP, &odelta, &oconv);

  int64_t fluid = 0, quarter = 0;
  double bmax = 0.0, xmax = 0.0, xdev = 0.0, hdev = 0.0, vchg = 0.0;
  for (int64_t i = 0; i < n; ++i) { fluid += lphi[i] < 0; bmax = fmax(bmax, fabs(b[i])); xmax = fmax(xmax, fabs(ox[i])); xdev = fmax(xdev, fabs(x[i] - ox[i])); }
  for (int64_t i = 0; i < nx; ++i) { quarter += wx[i] != 0.0 && wx[i] != 1.0; vchg = fmax(vchg, fabs(vxo[i] - vx[i])); }
  const int64_t hcmp = hn < 21 ? hn : 21;       /* a leading window entry by entry (the history is rounding-chaotic later on) */
  for (int64_t k = 0; k < hcmp; ++k) hdev = fmax(hdev, fabs(hist[k] - ohist[k]) / fabs(ohist[k]));
  /* (iteration counts: CG histories are rounding-chaotic on this operator -- DESIGN.md section 3 -- so the two loops may stop a few
   * iterations apart; the leading window of the history and the converged field are the parity statements) */
  const long long itol = oit / 20 > 3 ? oit / 20 : 3;
  const int ok = oconv && iters > 5 && llabs((long long)(iters - oit)) <= itol && hdev < 1e-9 && xdev <= 1e-6 * xmax && bmax > 0 && fluid > 0 &&
                 quarter > 0 && vchg > 0;
  printf("%s grid %dx%dx%d fluid cells %lld partial faces %lld | library: %lld iterations, history %lld entries | oracle: %lld iterations | "
         "history dev (first %lld) %.2e  x dev / max %.2e | velocity changed by up to %.3f\n", ok ? "OK" : "MISMATCH", N0, N1, N2,
         (long long)fluid, (long long)quarter, (long long)iters, (long long)hn, (long long)oit, (long long)hcmp, hdev, xdev / xmax, vchg);
  MFS(mfs_pcg3d_destroy(h));
  return ok ? 0 : 1;
}
