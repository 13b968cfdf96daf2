// mfs_p2p.h -- direct GPU-to-GPU exchange for the slab-decomposed CG (gfx950, xGMI).
//
// New design: the reference is single-GPU.  One process per GPU;
// every rank owns a WINDOW of fine-grained device memory that its peers map through
// HIP IPC.  Inside the CG loop nothing but kernels touches it:
//
//   halo planes   the rank that owns an edge plane of the direction vector stores it
//                 straight into its neighbour's window (write-through stores over
//                 xGMI), drains, and one lane raises a flag there; the neighbour's
//                 edge-plane stencil polls that flag, acquires, and reads the plane
//                 out of its own memory.  Two buffers per side (iteration parity).
//   dot products  every rank stores its partial sum into a slot of EVERY rank's
//                 window as two self-validating 8-byte granules {tag, half of the
//                 double}: no flag, no fence, one xGMI hop.  Each rank adds the world's
//                 slots in rank order, so all ranks hold the bit-identical sum and
//                 take identical convergence decisions.
//
// Every wait is a bounded spin (wall clock); a timeout raises the engine's error word
// and every later kernel of the solve returns at once, so a lost peer can never hang
// the GPU.  All cross-GPU accesses are system-scope atomics / fences.
#pragma once
#include "mfs_common.h"

namespace mfs {

typedef unsigned long long u64;

constexpr int kP2pMaxWorld = 16;
constexpr int kArRing = 4;                  // all-reduce slots are reused every 4th episode
constexpr size_t kP2pCtrlBytes = 8192;

// Head of every rank's window.  Written ONLY by remote ranks (and zeroed once at creation).
struct P2pCtrl {
  u64 halo_flag[2][2];                      // [side 0 = low ghost, 1 = high ghost][parity]: tag of the plane in recv buffer
  u64 ar[kArRing][kP2pMaxWorld][2];         // granules {tag << 32 | half} of rank r's contribution, episode ring
  u64 test_flag[kP2pMaxWorld];              // self-test: token from each rank
};
static_assert(sizeof(P2pCtrl) <= kP2pCtrlBytes, "control block too large");

// By-value kernel argument: where this rank's window is and where the peers' are.
struct P2pDev {
  P2pCtrl* self;
  P2pCtrl* peer[kP2pMaxWorld];              // peer[rank] == self
  char* recv[2][2];                         // my receive buffers [side][parity]
  char* send[2][2];                         // [0]: left neighbour's high-ghost buffers, [1]: right neighbour's low-ghost buffers ([parity]); null = no neighbour
  u64* send_flag[2][2];                     // the flags that go with them
  int rank, world;
  u64 timeout_ticks;                        // bound of every spin, in wall-clock ticks (100 MHz)
};

__device__ __forceinline__ u64 sys_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane: poll *p until it equals `want` (relaxed, system scope); false on timeout
__device__ __forceinline__ bool spin_eq(const u64* p, u64 want, u64 timeout_ticks) {
  if (sys_load(p) == want) return true;
  const u64 t0 = wall_clock64();
  for (;;) {
    __builtin_amdgcn_s_sleep(2);
    if (sys_load(p) == want) return true;
    if (wall_clock64() - t0 > timeout_ticks) return false;
  }
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

// ---- 16-byte write-through / cache-bypassing accesses for plane payloads ---------------------
template <typename T, int VEC>
__device__ __forceinline__ void vstore_sys(T* p, vec_t<T, VEC> v) {
  static_assert(sizeof(T) * VEC == 16, "payload vectors are 16 bytes");
  union { vec_t<T, VEC> v; u64 w[2]; } u;
  u.v = v;
  sys_store(reinterpret_cast<u64*>(p), u.w[0]);
  sys_store(reinterpret_cast<u64*>(p) + 1, u.w[1]);
}
template <typename T, int VEC>
__device__ __forceinline__ vec_t<T, VEC> vload_sys(const T* p) {
  static_assert(sizeof(T) * VEC == 16, "payload vectors are 16 bytes");
  union { vec_t<T, VEC> v; u64 w[2]; } u;
  u.w[0] = sys_load(reinterpret_cast<const u64*>(p));
  u.w[1] = sys_load(reinterpret_cast<const u64*>(p) + 1);
  return u.v;
}

// Publish: call from EVERY thread of the block after its payload stores.  Drains every wave,
// joins the block, then ONE lane releases at system scope and draws a ticket; the block that
// draws the last ticket raises the (up to two) remote flags and re-arms the ticket counter.
__device__ __forceinline__ void publish_planes(unsigned* ticket, unsigned nblocks, u64* flag_a, u64* flag_b, u64 tag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nblocks - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (flag_a) sys_store(flag_a, tag);
      if (flag_b) sys_store(flag_b, tag);
    }
  }
}

// Consume: ONE lane polls the (up to two) local flags, then acquires at system scope; the
// block joins behind it.  Returns false (to every thread) on timeout.
__device__ __forceinline__ bool await_planes(const u64* flag_a, const u64* flag_b, u64 tag, u64 timeout_ticks) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    bool ok = true;
    if (flag_a) ok = spin_eq(flag_a, tag, timeout_ticks);
    if (ok && flag_b) ok = spin_eq(flag_b, tag, timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return s_ok != 0;
}

// ------------------------------------------------------------------ host side --------------
struct P2pHost {
  int rank = 0, world = 1;
  size_t plane_bytes = 0, plane_stride = 0, window_bytes = 0;
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
