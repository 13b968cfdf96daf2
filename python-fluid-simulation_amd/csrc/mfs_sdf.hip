// mfs_sdf.hip -- rigid-body signed distance evaluation and particle projection on gfx950
// (SURVEY.md 8(f) rank 4).  Reference: solver/sdf3D.py -- evaluate_kernel (:218-239) and
// project_kernel (:241-258) with the sphere / box / cylinder device functions (:53-216).
//
// A rigid body is a (10,4) float64 block of `rb_d` (generate_rb :287-322): row 0 = [type code,
// parameters], rows 1-4 = translation matrix, rows 5-8 = rotation matrix, row 9 = velocity.
// type code // 2: 0 sphere, 1 box, 2 cylinder; odd = flipped (the fluid lives inside).
// Kept as written in the reference:
//  * box_project's test `rb[0,0] % 2 and ~(in_out)` (:126) -- `~` is a bitwise not, so it is true for
//    every flipped box: the position is ALWAYS mapped into the box frame, clamped and mapped back.
//  * mat_TR (:12-17) fills rows 0-2 only; inv_rigid / matvecmul4 accumulate in their loop order.
// Not reproducible: cylinder_eval (:148-172) reads `y_clip` unassigned when the point lies within the
// cylinder's height range; here y_clip is the point's own height there (what cylinder_project does, :181).
#include <math.h>

#include "mfs_common.h"

// separate multiply / add roundings, like the reference's expressions under CPython (the goldens)
#pragma clang fp contract(off)

namespace mfs {

struct Rb {                      // one body, loaded into registers
  double p[4];                   // row 0
  double T[3];                   // translation T[i,3]
  double R[3][3];                // rotation
  double vel[3];                 // row 9
};

__device__ __forceinline__ Rb rb_load(const double* __restrict__ rb_d, int i) {
  const double* b = rb_d + (int64_t)i * 40;
  Rb r;
  for (int k = 0; k < 4; ++k) r.p[k] = b[k];
  for (int k = 0; k < 3; ++k) r.T[k] = b[(1 + k) * 4 + 3];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) r.R[a][c] = b[(5 + a) * 4 + c];
  for (int k = 0; k < 3; ++k) r.vel[k] = b[9 * 4 + k];
  return r;
}

__device__ __forceinline__ bool rb_flipped(const Rb& r) { return fmod(r.p[0], 2.0) != 0.0; }

// pos_rb = inv_rigid(T, R) * position     (inv_rigid :31-40, matvecmul4 :19-29)
__device__ __forceinline__ void to_body(const Rb& r, const double pos[3], double out[3]) {
  for (int i = 0; i < 3; ++i) {
    double t3 = 0.0;
    for (int j = 0; j < 3; ++j) t3 -= r.R[j][i] * r.T[j];
    double tmp = 0.0;
    for (int j = 0; j < 3; ++j) tmp += r.R[j][i] * pos[j];
    tmp += t3;
    out[i] = tmp;
  }
}

// position = mat_TR(T, R) * pos_rb
__device__ __forceinline__ void to_world(const Rb& r, const double prb[3], double out[3]) {
  for (int i = 0; i < 3; ++i) {
    double tmp = 0.0;
    for (int j = 0; j < 3; ++j) tmp += r.R[i][j] * prb[j];
    tmp += r.T[i];
    out[i] = tmp;
  }
}

__device__ __forceinline__ double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

__device__ __forceinline__ double sphere_eval(const Rb& r, const double pos[3]) {
  const double d[3] = {pos[0] - r.T[0], pos[1] - r.T[1], pos[2] - r.T[2]};
  double sd = norm3(d) - r.p[1];
  if (rb_flipped(r)) sd = -sd;
  return sd;
}

__device__ __forceinline__ void sphere_project(const Rb& r, double pos[3]) {
  const double d[3] = {pos[0] - r.T[0], pos[1] - r.T[1], pos[2] - r.T[2]};
  const double dist = norm3(d);
  double sd = dist - r.p[1];
  if (rb_flipped(r)) sd = -sd;
  if (sd < 0)
    for (int i = 0; i < 3; ++i) pos[i] = d[i] / dist * r.p[1] + r.T[i];
}

__device__ __forceinline__ double box_eval(const Rb& r, const double pos[3]) {
  double prb[3];
  to_body(r, pos, prb);
  double tmp = 0.0, max_disp = -100.0;
  for (int i = 0; i < 3; ++i) {
    const double disp = fabs(prb[i]) - r.p[1 + i] / 2;
    if (disp > 0) tmp += disp * disp;
    if (max_disp < disp) max_disp = disp;
  }
  double sd = sqrt(tmp);
  if (max_disp < 0) sd += max_disp;
  if (rb_flipped(r)) sd = -sd;
  return sd;
}

__device__ __forceinline__ void box_project(const Rb& r, double pos[3]) {
  double prb[3];
  to_body(r, pos, prb);
  int in_out = 0;
  for (int i = 0; i < 3; ++i)
    if (prb[i] > r.p[1 + i] / 2 || prb[i] < -r.p[1 + i] / 2) ++in_out;
  if (rb_flipped(r)) {                       // `rb[0,0] % 2 and ~(in_out)`: ~ is bitwise, always true (:126)
    for (int i = 0; i < 3; ++i) {
      const double h = r.p[1 + i] / 2;
      if (prb[i] < -h) prb[i] = -h;
      else if (prb[i] > h) prb[i] = h;
    }
    to_world(r, prb, pos);
  } else if (in_out == 0) {                  // inside a solid box: out through the nearest face (:134-146)
    int index = 0;
    double dist_xyz = 100.0;
    for (int i = 0; i < 3; ++i) {
      const double h = r.p[1 + i] / 2;
      if (h - prb[i] < dist_xyz) { dist_xyz = h - prb[i]; index = i * 2; }
      if (prb[i] + h < dist_xyz) { dist_xyz = prb[i] + h; index = i * 2 + 1; }
    }
    prb[index / 2] += dist_xyz * ((index % 2) ? -1.0 : 1.0);
    to_world(r, prb, pos);
  }
}

__device__ __forceinline__ double cylinder_eval(const Rb& r, const double pos[3]) {
  double prb[3];
  to_body(r, pos, prb);
  const double hh = r.p[2] / 2;
  double y_clip = prb[1];                    // see the header: unassigned in the reference for |y| <= hh
  if (prb[1] < -hh) y_clip = -hh;
  else if (prb[1] > hh) y_clip = hh;
  double sd = sqrt(prb[0] * prb[0] + prb[2] * prb[2]) - r.p[1];
  const bool cap = y_clip == hh || y_clip == -hh;
  if (sd < 0) {
    if (cap) sd = fabs(y_clip - prb[1]);
    else sd = fmax(sd, fmax(prb[1] - hh, -(prb[1] + hh)));
  } else if (cap) {
    const double dy = fabs(y_clip - prb[1]);
    sd = sqrt(sd * sd + dy * dy);
  }
  if (rb_flipped(r)) sd = -sd;
  return sd;
}

__device__ __forceinline__ void cylinder_project(const Rb& r, double pos[3]) {
  double prb[3];
  to_body(r, pos, prb);
  const double hh = r.p[2] / 2;
  double y_clip = prb[1];
  if (prb[1] < -hh) y_clip = -hh;
  else if (prb[1] > hh) y_clip = hh;
  const double dist = sqrt(prb[0] * prb[0] + prb[2] * prb[2]);
  const double sd = dist - r.p[1];
  if (rb_flipped(r)) {
    if (fabs(y_clip) == hh || sd > 0) {
      if (sd < 0) {
        prb[1] = y_clip;
      } else {
        prb[0] = prb[0] / dist * r.p[1];
        prb[2] = prb[2] / dist * r.p[1];
        prb[1] = y_clip;
      }
    }
    to_world(r, prb, pos);
  } else if (sd < 0 && fabs(y_clip) != hh) {
    const double mv = fmax(sd, fmax(prb[1] - hh, -(prb[1] + hh)));
    if (mv == sd) {
      prb[0] = prb[0] / dist * r.p[1];
      prb[2] = prb[2] / dist * r.p[1];
    } else if (mv == prb[1] - hh) {
      prb[1] = hh;
    } else {
      prb[1] = -hh;
    }
    to_world(r, prb, pos);
  }
}

// evaluate_kernel :218-239
__global__ void __launch_bounds__(256)
k_sdf_evaluate(const double* __restrict__ rb_d, int nrb, const void* position, int pdt, int64_t P, void* sd, int sdt,
               void* vel, int vdt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const double pos[3] = {ldx(position, pdt, 3 * p), ldx(position, pdt, 3 * p + 1), ldx(position, pdt, 3 * p + 2)};
  double min_sd = 100.0;
  int idx = 0;
  for (int i = 0; i < nrb; ++i) {
    const Rb r = rb_load(rb_d, i);
    const int kind = (int)floor(r.p[0] / 2);
    double d = min_sd;                        // unknown kinds leave the minimum alone
    if (kind == 0) d = sphere_eval(r, pos);
    else if (kind == 1) d = box_eval(r, pos);
    else if (kind == 2) d = cylinder_eval(r, pos);
    if (d < min_sd) { min_sd = d; idx = i; }
  }
  stx(sd, sdt, p, min_sd);
  if (min_sd <= 0 && nrb > 0) {
    const Rb r = rb_load(rb_d, idx);
    for (int k = 0; k < 3; ++k) stx(vel, vdt, 3 * p + k, r.vel[k]);
  }
}

// project_kernel :241-258 -- every body in turn, each on the position the previous one left
__global__ void __launch_bounds__(256)
k_sdf_project(const double* __restrict__ rb_d, int nrb, void* position, int pdt, int64_t P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double pos[3] = {ldx(position, pdt, 3 * p), ldx(position, pdt, 3 * p + 1), ldx(position, pdt, 3 * p + 2)};
  for (int i = 0; i < nrb; ++i) {
    const Rb r = rb_load(rb_d, i);
    const int kind = (int)floor(r.p[0] / 2);
    if (kind == 0) sphere_project(r, pos);
    else if (kind == 1) box_project(r, pos);
    else if (kind == 2) cylinder_project(r, pos);
    if (pdt == MFS_F32)                       // the reference writes into the array row: float32 positions round per body
      for (int k = 0; k < 3; ++k) pos[k] = (double)(float)pos[k];
  }
  for (int k = 0; k < 3; ++k) stx(position, pdt, 3 * p + k, pos[k]);
}

}  // namespace mfs

using namespace mfs;

extern "C" {

int mfs_sdf_evaluate3d(const void* rb_d, int64_t num_bodies, const void* position, int pos_dt, int64_t num_positions,
                       void* sd, int sd_dt, void* vel, int vel_dt, mfs_stream stream) {
  MFS_REQUIRE(num_bodies >= 0 && num_bodies <= 4096 && (num_bodies == 0 || rb_d), "rigid bodies");
  MFS_REQUIRE(num_positions >= 0 && (num_positions == 0 || (position && sd && vel)), "position / output arrays");
  MFS_REQUIRE(dtype_ok(pos_dt) && dtype_ok(sd_dt) && dtype_ok(vel_dt), "dtype");
  if (num_positions == 0) return MFS_OK;
  hipLaunchKernelGGL(k_sdf_evaluate, dim3(cdiv(num_positions, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const double*)rb_d, (int)num_bodies, position, pos_dt, num_positions, sd, sd_dt, vel, vel_dt);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

int mfs_sdf_project3d(const void* rb_d, int64_t num_bodies, void* position, int pos_dt, int64_t num_positions,
                      mfs_stream stream) {
  MFS_REQUIRE(num_bodies >= 0 && num_bodies <= 4096 && (num_bodies == 0 || rb_d), "rigid bodies");
  MFS_REQUIRE(num_positions >= 0 && (num_positions == 0 || position), "position array");
  MFS_REQUIRE(dtype_ok(pos_dt), "dtype");
  if (num_positions == 0 || num_bodies == 0) return MFS_OK;
  hipLaunchKernelGGL(k_sdf_project, dim3(cdiv(num_positions, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const double*)rb_d, (int)num_bodies, position, pos_dt, num_positions);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

}  // extern "C"
