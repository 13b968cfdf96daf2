// mfs_common.h -- shared device/host helpers for libmfs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/mfs.h"

namespace mfs {

// ---------------------------------------------------------------- errors ----
void set_error(const char* fmt, ...);

#define MFS_HIP_TRY(expr)                                                            \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ::mfs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return MFS_E_HIP;                                                              \
    }                                                                                \
  } while (0)

#define MFS_REQUIRE(cond, msg)                                                       \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      ::mfs::set_error("%s:%d: invalid argument: %s (%s)", __FILE__, __LINE__, msg, #cond); \
      return MFS_E_INVALID;                                                          \
    }                                                                                \
  } while (0)

#define MFS_LAUNCH_CHECK() MFS_HIP_TRY(hipGetLastError())

static inline bool dtype_ok(int dt) { return dt == MFS_F32 || dt == MFS_F64; }
static inline size_t dtype_size(int dt) { return dt == MFS_F32 ? 4 : 8; }

// ------------------------------------------------------- typed load/store ---
// Once-per-solve kernels take arrays of either element type; the dtype code is
// wave-uniform, so the branch costs nothing.  Arithmetic is always fp64 there.
__device__ __forceinline__ double ldx(const void* p, int dt, int64_t i) {
  return dt == MFS_F32 ? (double)((const float*)p)[i] : ((const double*)p)[i];
}
__device__ __forceinline__ void stx(void* p, int dt, int64_t i, double v) {
  if (dt == MFS_F32) ((float*)p)[i] = (float)v; else ((double*)p)[i] = v;
}

// ------------------------------------------------------------ reductions ----
constexpr int kWave = 64;

// one DPP data move of a double (two 32-bit moves); CTRL: quad_perm 0x00-0xff, row_ror:n 0x120 + n
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long d = __builtin_bit_cast(long long, v);
  const int lo = (int)d, hi = (int)(d >> 32);
  const unsigned rlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const unsigned rhi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)rhi << 32) | rlo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long d = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)d, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(d >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
}

// Wave total in EVERY lane.  All 64 lanes must be active.  Fixed tree: pairs, quads, then the four quads of a row by two
// row rotations (DPP: register-to-register, ~10 cycles a step -- the ds_bpermute shuffles this replaces cost ~150 cycles
// a step, 0.3 us per reduction, which is most of a small-grid kernel's arithmetic time), then the four row totals read
// from lanes 0 / 16 / 32 / 48 and added in row order.  Deterministic: the same sequence everywhere.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f64<0x124>(v);       // row_ror:4
  v += dpp_f64<0x128>(v);       // row_ror:8
  return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}

// Block total in thread 0 (deterministic: fixed shuffle tree, then waves in order).
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double s_part[BLOCK / kWave];
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) s_part[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / kWave; ++w) t += s_part[w];
  }
  __syncthreads();
  return t;
}

// 16-byte vector of T: float4 / double2 without dragging in operator overloads.
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Vec {
  T v[VEC];
};

// native 16-byte vectors (float4 / double2) so the nontemporal builtins apply
template <typename T, int N> struct NativeVec { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T> struct NativeVec<T, 1> { typedef T type __attribute__((ext_vector_type(1))); };
template <typename T, int N> using vec_t = typename NativeVec<T, N>::type;

template <typename T, int VEC>
__device__ __forceinline__ vec_t<T, VEC> vload(const T* p) { return *reinterpret_cast<const vec_t<T, VEC>*>(p); }
template <typename T, int VEC>
__device__ __forceinline__ vec_t<T, VEC> vload_nt(const T* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const vec_t<T, VEC>*>(p));
}
template <typename T, int VEC>
__device__ __forceinline__ void vstore(T* p, vec_t<T, VEC> v) { *reinterpret_cast<vec_t<T, VEC>*>(p) = v; }
template <typename T, int VEC>
__device__ __forceinline__ void vstore_nt(T* p, vec_t<T, VEC> v) {
  __builtin_nontemporal_store(v, reinterpret_cast<vec_t<T, VEC>*>(p));
}

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<double> { static constexpr int N = 2; };

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// integer knob from the environment (tuning / test hooks only)
static inline int env_int(const char* name, int defv) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : defv;
}

}  // namespace mfs
