// mfs_p2p.h -- direct GPU-to-GPU exchange for the slab-decomposed CG (gfx950, xGMI).
//
// New design: the reference is single-GPU.  One process per GPU;
// every rank owns a WINDOW of uncached device memory that its peers map through
// HIP IPC.  Inside the CG loop nothing but kernels touches it:
//
//   halo planes   the rank that owns an edge plane of the direction vector stores it
//                 straight into its neighbour's window as self-validating 8-byte
//                 granules {tag, 32 bits of payload} (one per fp32 cell, two per fp64
//                 cell): no flag, no fence, no ordering assumption between stores --
//                 the neighbour's edge-plane stencil reads the granules out of its own
//                 memory and re-reads any whose tag is not yet this iteration's.  Two
//                 buffers per side (iteration parity).
//   dot products  every rank stores its partial sum into a slot of EVERY rank's
//                 window as two self-validating 8-byte granules {tag, half of the
//                 double}: one xGMI hop.  Each rank adds the world's slots in rank
//                 order, so all ranks hold the bit-identical sum and take identical
//                 convergence decisions.
//
// The only hardware property relied on is that an aligned 8-byte store is not torn.
// Every wait is a bounded spin (wall clock); a timeout raises the engine's error word
// and every later kernel of the solve returns at once, so a lost peer can never hang
// the GPU.  All cross-GPU accesses are system-scope (sc0 sc1) loads and stores on uncached memory.
#pragma once
#include "mfs_common.h"

namespace mfs {

typedef unsigned long long u64;

constexpr int kP2pMaxWorld = 16;
constexpr int kArRing = 4;                  // all-reduce slots are reused every 4th episode
constexpr size_t kP2pCtrlBytes = 8192;

// Head of every rank's window.  Written ONLY by remote ranks (and zeroed once at creation).
struct P2pCtrl {
  u64 ar[kArRing][kP2pMaxWorld][2];         // granules {tag << 32 | half} of rank r's contribution, episode ring
};
static_assert(sizeof(P2pCtrl) <= kP2pCtrlBytes, "control block too large");

// By-value kernel argument: where this rank's window is and where the peers' are.
struct P2pDev {
  P2pCtrl* self;
  P2pCtrl* peer[kP2pMaxWorld];              // peer[rank] == self
  u64* recv[2][2];                          // my receive buffers [side 0 = low ghost, 1 = high ghost][parity], granules
  u64* send[2][2];                          // [0]: left neighbour's high-ghost buffers, [1]: right neighbour's low-ghost buffers ([parity]); null = no neighbour
  int rank, world;
  u64 timeout_ticks;                        // bound of every spin, in wall-clock ticks (100 MHz)
};

__device__ __forceinline__ u64 sys_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- scalar all-reduce over the windows --------------------------------------------------
// tag != 0 identifies the episode; `ring` = episode index mod kArRing.
// send: lanes 0..world-1 of ONE wave, lane l writes this rank's value into rank l's window.
__device__ __forceinline__ void ar_send(const P2pDev& pd, int ring, unsigned tag, double v, int lane) {
  if (lane < pd.world) {
    const u64 bits = (u64)__double_as_longlong(v);
    u64* g = pd.peer[lane]->ar[ring][pd.rank];
    sys_store(g + 0, ((u64)tag << 32) | (bits & 0xffffffffull));
    sys_store(g + 1, ((u64)tag << 32) | (bits >> 32));
  }
}

// recv: lanes 0..world-1 of ONE wave poll their rank's two granules in the OWN window; returns the
// sum over ranks in rank order in lane 0 (identical bits on every rank); *ok = false on timeout.
__device__ __forceinline__ double ar_recv(const P2pDev& pd, int ring, unsigned tag, int lane, bool* ok) {
  double v = 0.0;
  bool good = true;
  if (lane < pd.world) {
    const u64* g = pd.self->ar[ring][lane];
    u64 lo = sys_load(g), hi = sys_load(g + 1);
    if ((lo >> 32) != tag || (hi >> 32) != tag) {
      const u64 t0 = wall_clock64();
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        lo = sys_load(g); hi = sys_load(g + 1);
        if ((lo >> 32) == tag && (hi >> 32) == tag) break;
        if (wall_clock64() - t0 > pd.timeout_ticks) { good = false; break; }
      }
    }
    v = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
  }
  *ok = __all(good);
  double tot = 0.0;
  for (int r = 0; r < pd.world; ++r) tot += __shfl(v, r, kWave);   // rank order, every lane computes it
  return tot;
}

// ---- plane payloads as granules ----------------------------------------------------------------
// element e of a plane occupies granule(s) [e * G, e * G + G), G = 1 (fp32: the float's bits) or
// 2 (fp64: low and high half); every granule = tag << 32 | 32 payload bits, written by ONE 8-byte store.
template <typename T> struct Gran;
template <> struct Gran<float> { static constexpr int N = 1; };
template <> struct Gran<double> { static constexpr int N = 2; };

// 16-byte write-through store of TWO granules: a store torn between its 8-byte halves is harmless (each half
// validates itself), and 16-byte lanes move ~2.7x the bytes per instruction slot of 8-byte ones.
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
// The trailing s_nop is part of the instruction's contract, not padding: a VMEM store of more than 64 bits reads its data
// VGPRs after issue, and a VALU write to them in the next slot corrupts the payload (CDNA ISA, manually inserted wait
// states).  The compiler inserts that wait state for its own stores but cannot see into an asm statement.
__device__ __forceinline__ void sys_store2(u64* p, u64 a, u64 b) {
  u64x2 v = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <typename T, int VEC>
__device__ __forceinline__ void gran_store_vec(u64* buf, int64_t elem, vec_t<T, VEC> v, unsigned tag) {
  const u64 t = (u64)tag << 32;
  u64* g = buf + elem * Gran<T>::N;
  u64 w[VEC * Gran<T>::N];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if (Gran<T>::N == 1) {
      w[j] = t | (u64)__float_as_uint((float)v[j]);
    } else {
      const u64 bits = (u64)__double_as_longlong((double)v[j]);
      w[2 * j] = t | (bits & 0xffffffffull);
      w[2 * j + 1] = t | (bits >> 32);
    }
  }
  if ((VEC * Gran<T>::N) % 2 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < VEC * Gran<T>::N; k += 2) sys_store2(g + k, w[k], w[k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < VEC * Gran<T>::N; ++k) sys_store(g + k, w[k]);
  }
}

// false on timeout (the payload never arrived); re-reads until every granule carries `tag`
template <typename T, int VEC>
__device__ __forceinline__ bool gran_load_vec(const u64* buf, int64_t elem, unsigned tag, u64 timeout_ticks,
                                              vec_t<T, VEC>* out) {
  constexpr int NG = VEC * Gran<T>::N;
  const u64* g = buf + elem * Gran<T>::N;
  u64 w[NG];
  bool all = true;
#pragma unroll
  for (int k = 0; k < NG; ++k) { w[k] = sys_load(g + k); all = all && (unsigned)(w[k] >> 32) == tag; }
  if (!all) {
    const u64 t0 = wall_clock64();
    for (;;) {
      __builtin_amdgcn_s_sleep(1);
      all = true;
#pragma unroll
      for (int k = 0; k < NG; ++k) { w[k] = sys_load(g + k); all = all && (unsigned)(w[k] >> 32) == tag; }
      if (all) break;
      if (wall_clock64() - t0 > timeout_ticks) return false;
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if (Gran<T>::N == 1) (*out)[j] = (T)__uint_as_float((unsigned)w[j]);
    else (*out)[j] = (T)__longlong_as_double((long long)((w[2 * j + 1] << 32) | (w[2 * j] & 0xffffffffull)));
  }
  return true;
}

// ------------------------------------------------------------------ host side --------------
struct P2pHost {
  int rank = 0, world = 1;
  size_t plane_bytes = 0, plane_stride = 0, window_bytes = 0;   // plane_stride: bytes of one receive buffer (granules)
  char* window = nullptr;                     // own window (hipExtMallocWithFlags, fine-grained)
  char* peer_window[kP2pMaxWorld] = {};       // IPC-mapped peers ([rank] = own)
  bool opened[kP2pMaxWorld] = {};
  unsigned* local = nullptr;                  // ordinary device memory: ticket counters, self-test result
  P2pDev dev = {};
  bool connected = false;
  int alloc_kind = 0;                         // 1 uncached, 2 fine-grained
  unsigned epoch = 0;                         // solves begun through this window (same count on every rank); tags carry it
};

}  // namespace mfs

// the opaque handle of include/mfs.h
struct mfs_p2p : mfs::P2pHost {};
