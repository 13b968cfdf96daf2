// mfs_cg_core.h -- the solver-independent part of the device-resident CG loop.
//
// Shared by the pressure engine (mfs_pcg.hip) and the viscosity engine
// (mfs_visc.hip): the vector phases, the deterministic reductions, and the
// device-resident control block.  The reference's loop is the same in both solvers
// (solver/PressureCGSolver3D.py:198-223, solver/ViscosityCGSolver3D.py:575-612);
// only `apply` differs.  One iteration:
//
//   <apply>          q = A d, per-block partials of d.q
//   k_update_xr      a = delta / d.q ; x += a d ; r -= a q ; partials of r.r   6 scalars/DOF
//   k_update_d       convergence test, history, b, d = r + b d                 3 scalars/DOF
//
// The two dot products have no kernel of their own on one GPU: every block of the
// consuming kernel sums the producer's per-block partials itself (<= 8192 doubles
// out of L2, same fixed order in every block => identical value everywhere).  The
// multi-GPU driver needs the sums as scalars to all-reduce, so there the 1-block
// k_reduce runs between producer and consumer and the consumers read the scalars.
// delta is kept in a 2-slot ring indexed by iteration parity (a kernel argument), so
// no kernel reads a scalar that the same kernel's bookkeeping thread rewrites.
//
// alpha, beta, delta, the iteration count and a `done` flag live in device
// memory; once `done` is set every later kernel is a no-op, so the iteration
// count and the final state equal the reference's even though the host only
// looks every `check_every` iterations.  Reductions are deterministic: fixed
// shuffle tree per wave, waves in order, blocks in order -- no float atomics.
// Storage dtype T is fp32 or fp64; all arithmetic is fp64 in registers.
#pragma once
#include <stdlib.h>

#include <algorithm>

#include "mfs_common.h"
#include "mfs_p2p.h"

namespace mfs {

constexpr int kBlock = 256;
constexpr int kMaxPartials = 8192;
constexpr int64_t kHistCap = 16384;

enum { S_DQ = MFS_PCG_S_DQ, S_RR = MFS_PCG_S_RR, S_DELTA = MFS_PCG_S_DELTA, S_TOL2 = MFS_PCG_S_TOL2,
       S_DONE = MFS_PCG_S_DONE, S_ITERS = MFS_PCG_S_ITERS, S_ALPHA = MFS_PCG_S_ALPHA, S_BETA = MFS_PCG_S_BETA,
       S_LASTRR = MFS_PCG_S_LASTRR, S_RING = 9 /* 2 slots: delta by iteration parity */,
       S_RZ = 12 /* Jacobi loop: r.z of the latest update */,
       S_LANE = MFS_PCG_S_LANE /* pressure engine: 1 when lane-level masking of the listed launches pays on this solve (pcg_build_live) */,
       S_ERR = MFS_PCG_S_ERR /* != 0: the solve was stopped -- 1 / 2 a peer-to-peer wait timed out (slab loop: all-reduce /
                                halo plane), 3 d.q == 0 (the reference's ZeroDivisionError, PressureCGSolver3D.py:211),
                                4 a non-finite d.q or r.r (the reference would spin to max_iter on `nan < tol**2`) */ };
enum { kErrArTimeout = 1, kErrHaloTimeout = 2, kErrZeroDq = 3, kErrNonFinite = 4,
       kErrNotResident = 5 };   // resident loop: its workgroups did not all show up for the FIRST dot product of a launch (nothing written)

// health of the two dot products that close an iteration: 0 fine, else the S_ERR code.  Detected on the device so that
// a poisoned solve stops within one `check_every` instead of iterating to max_iter = prod(gres).
__device__ __forceinline__ int cg_health(double dq, double rr) {
  const double big = 1.7976931348623157e308;
  if (dq == 0.0) return kErrZeroDq;
  if (!(fabs(dq) <= big) || !(fabs(rr) <= big)) return kErrNonFinite;
  return 0;
}


template <typename T, int VEC>
__device__ __forceinline__ Vec<T, VEC> ldv(const T* p) { return *reinterpret_cast<const Vec<T, VEC>*>(p); }

// ---------------------------------------------------------- vector phases ---
// LIVE CHUNKS (round 3, viscosity): a flat CG vector cut into chunks of 2^shift vectors; `list[0 .. *count)` names the
// chunks that hold at least one unknown that is not identically zero for the whole solve (its operator row is empty and
// its right-hand side 0: a face in the air).  The vector phases then sweep only those -- r, d and q ARE zero elsewhere and
// x does not change there -- which is most of the work in a liquid scene.  list == nullptr: every chunk (the default).
struct LiveMap {
  const int* list;
  const int* count;
  int shift;
};
__device__ __forceinline__ int64_t live_nv(const LiveMap& lm, int64_t nv) { return lm.list ? ((int64_t)*lm.count << lm.shift) : nv; }
__device__ __forceinline__ int64_t live_vec(const LiveMap& lm, int64_t k) {
  return lm.list ? (((int64_t)lm.list[k >> lm.shift] << lm.shift) | (k & ((1 << lm.shift) - 1))) : k;
}

// unknowns per chunk of a live map: one 128-byte line of fp32 state (8 16-byte vectors).  Much smaller than a z-row of a
// production grid on purpose -- a liquid body covers PART of a row: with 1024-unknown chunks (four whole rows at Nz = 256) the
// vector phases of the 256^3 viscosity bench swept 18.7 % of the unknowns for 7 % of liquid, with 128 (half a row) 17.4 %
#ifndef MFS_LIVE_CHUNK
#define MFS_LIVE_CHUNK 32        // A/B knob (tools/build_variant.sh chunk64 "-DMFS_LIVE_CHUNK=64")
#endif
constexpr int kLiveChunk = MFS_LIVE_CHUNK;

// flags -> ascending list of the indices whose flag is set + their count, once per solve: per-block counts (coalesced
// reads, ballots), a one-block scan of the block counts, then an ordered per-block compaction.  (A single block walking
// the flags thread by thread -- the first version -- takes milliseconds at the 1.6 M chunks of a 256^3 viscosity solve.)
constexpr int kCompactTile = 8192;        // flags per block: 1024 threads x 8
template <typename F>
static __global__ void __launch_bounds__(1024)
k_compact_count(const F* __restrict__ flags, int n, int* __restrict__ bsum) {
  const int base = blockIdx.x * kCompactTile;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = base + j * 1024 + (int)threadIdx.x;
    c += __popcll(__builtin_amdgcn_ballot_w64(i < n && flags[i] != 0));      // (every lane of a wave holds the wave's count)
  }
  __shared__ int s_w[16];
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += s_w[w];
    bsum[blockIdx.x] = t;
  }
}

// exclusive scan of the block counts in place, total -> *count (ONE block; nb is n / 8192)
static __global__ void __launch_bounds__(1024)
k_compact_scan(int* __restrict__ bsum, int nb, int* __restrict__ count) {
  __shared__ int s_pre[1024];
  __shared__ int s_carry;
  const int t = threadIdx.x;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int v = b0 + t < nb ? bsum[b0 + t] : 0;
    s_pre[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int u = t >= o ? s_pre[t - o] : 0;
      __syncthreads();
      s_pre[t] += u;
      __syncthreads();
    }
    const int carry = s_carry;
    if (b0 + t < nb) bsum[b0 + t] = carry + s_pre[t] - v;
    __syncthreads();
    if (t == 1023) s_carry = carry + s_pre[1023];
    __syncthreads();
  }
  if (t == 0) *count = s_carry;
}

template <typename F>
static __global__ void __launch_bounds__(1024)
k_compact_write(const F* __restrict__ flags, int n, const int* __restrict__ bsum, int* __restrict__ list) {
  const int base = blockIdx.x * kCompactTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int s_w[2][16];
  int run = bsum[blockIdx.x];
#pragma unroll 1
  for (int j = 0; j < 8; ++j) {
    const int i = base + j * 1024 + (int)threadIdx.x;
    const bool f = i < n && flags[i] != 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
    if (lane == 0) s_w[j & 1][wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = s_w[j & 1][w]; before += w < wave ? c : 0; total += c; }
    if (f) list[run + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
    run += total;
  }
}

// scratch: kCompactScratch(n) ints
static inline size_t core_compact_scratch_ints(int64_t n) { return (size_t)((n + kCompactTile - 1) / kCompactTile) + 2; }
template <typename F>
static inline int core_compact_flags(const F* flags, int n, int* list, int* count, int* scratch, hipStream_t st) {
  const int nb = (n + kCompactTile - 1) / kCompactTile;
  if (nb <= 0) { MFS_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int), st)); return MFS_OK; }
  hipLaunchKernelGGL(k_compact_count<F>, dim3(nb), dim3(1024), 0, st, flags, n, scratch);
  hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, scratch, nb, count);
  hipLaunchKernelGGL(k_compact_write<F>, dim3(nb), dim3(1024), 0, st, flags, n, (const int*)scratch, list);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

// work lists of the marching kernels: runrem[k] = how many consecutive list entries from k on are consecutive planes of one
// tile (one march); `np` = planes per tile in the numbering of the entries
static __global__ void __launch_bounds__(256)
k_list_runs(const int* __restrict__ items, const int* __restrict__ count, int np, int* __restrict__ runrem) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = *count;
  if (k >= n) return;
  const int it = items[k], tile = it / np;
  int r = 1;
  while (k + r < n && items[k + r] == it + r && (it + r) / np == tile) ++r;
  runrem[k] = r;
}

// flags | list | count (64 ints) | scratch of the compaction
static inline size_t core_live_ws_bytes(int64_t n) {
  const int64_t nch = (n + kLiveChunk - 1) / kLiveChunk;
  return align_up((size_t)(2 * nch + 64 + core_compact_scratch_ints(nch)) * sizeof(int), 4096);
}

template <typename T, int VEC, typename F>
__device__ __forceinline__ void for_each_vec(int64_t n, F&& f, bool reverse = false, bool blocked = false, LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  // f(i, vec): process elements [i, i+VEC) (vec) or the single element i (tail).
  // reverse: sweep from the end of the array to its start -- consecutive CG phases
  // alternate direction so each one starts on the bytes the previous one touched
  // last (still resident in the Infinity Cache) instead of the ones it evicted first.
  const int64_t nv = n / VEC;
  if (blocked) {
    // blocked (A/B knob MFS_REV_D=2, k_update_d only): workgroup b sweeps ONE contiguous chunk of the array instead of striding
    // through all of it.  Same-engine A/B, viscosity CG iteration: 256^3 fp64 988.6 (strided) vs 1002.8 us, fp32 478.6 vs 500.6,
    // 128^3 fp64 113.3 vs 114.9 -- the strided sweep stays
    const int64_t per = (nv + gridDim.x - 1) / gridDim.x;
    const int64_t k0 = (int64_t)blockIdx.x * per, k1 = k0 + per < nv ? k0 + per : nv;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += blockDim.x) f(k * VEC, true);
  } else {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (lm.list) {      // live chunks only (the last chunk of the vector may reach past its end)
      const int64_t nvl = live_nv(lm, nv);
      for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nvl; k += stride) {
        const int64_t kr = live_vec(lm, reverse ? nvl - 1 - k : k);
        if (kr < nv) f(kr * VEC, true);
      }
    } else {
      for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nv; k += stride)
        f((reverse ? nv - 1 - k : k) * VEC, true);
    }
  }
  // scalar tail (n % VEC elements) handled by the first threads of block 0
  const int64_t tail = n - nv * VEC;
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < tail) f(nv * VEC + threadIdx.x, false);
}

// d = b - q ; r = d ; partial sum r^2          (PressureCGSolver3D.py:202-204)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_cg_init(const T* __restrict__ b, const T* __restrict__ q, T* __restrict__ d, T* __restrict__ r, int64_t n,
          double* __restrict__ partial) {
  double acc = 0.0;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      const Vec<T, VEC> bv = ldv<T, VEC>(b + i), qv = ldv<T, VEC>(q + i);
      Vec<T, VEC> dv;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dv.v[j] = (T)((double)bv.v[j] - (double)qv.v[j]);
        acc += (double)dv.v[j] * (double)dv.v[j];
      }
      *reinterpret_cast<Vec<T, VEC>*>(d + i) = dv;
      *reinterpret_cast<Vec<T, VEC>*>(r + i) = dv;
    } else {
      const T dv = (T)((double)b[i] - (double)q[i]);
      d[i] = dv; r[i] = dv;
      acc += (double)dv * (double)dv;
    }
  });
  const double tot = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// sum of partial[0..count) known to EVERY thread of the block (fixed order: deterministic)
__device__ __forceinline__ double block_total_of(const double* __restrict__ partial, int count) {
  __shared__ double s_tot;
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
  const double t = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) s_tot = t;
  __syncthreads();
  return s_tot;
}

// In-launch reduction tail.  Every block calls it with its partial sum (thread 0's `my_val`); it
// returns true -- to every thread -- in exactly ONE block, the last to arrive, where *total is the sum
// of partial[0..count) in block_total_of's order (so the value is the one a following kernel would
// compute).  Hand-off: write-through (sc1) partial store, drained, then a relaxed agent-scope ticket;
// the lane that drew the last ticket performs ONE agent-scope acquire (buffer_inv sc1) and waits for it before the
// workgroup barrier that releases the other waves' loads of the partials (the consumer form of
// cdna_hip_programming.md Guideline 16; the guide validates dropping the acquire for sc1 loads only on a single
// unsharded last-arriver counter, which this two-level ticket is not -- so the acquire stays, in one block per launch).
// Arrival tickets are SHARDED: one returning atomic costs ~12 ns on its word, so 2048 arrivals on one counter would
// serialise for ~25 us -- longer than a small grid's whole update kernel.  Blocks b with equal b % 8 (the blocks of one
// XCD) share a shard counter on a cache line of its own; the last arriver of each shard draws a top-level ticket.
#ifndef MFS_TAIL_ACQUIRE
#define MFS_TAIL_ACQUIRE 1      // 0: A/B builds only (tools/iter_sweep.py with MFS_LIB)
#endif
constexpr int kTicketStride = 32;                       // unsigned words between counters (128 bytes)
constexpr int kTicketWords = 9 * kTicketStride;         // 8 shards + the top-level counter

__device__ __forceinline__ bool last_block_total(double* __restrict__ partial, int my_slot, double my_val, int count,
                                                 unsigned* ticket, unsigned nblocks, double* total) {
  __shared__ int s_last;
  __shared__ double s_total;
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + my_slot, my_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned nsh = nblocks < 8u ? nblocks : 8u;
    const unsigned shard = blockIdx.x % nsh;
    const unsigned in_shard = (nblocks - shard + nsh - 1) / nsh;      // blocks b < nblocks with b % nsh == shard
    unsigned* sc = ticket + shard * kTicketStride;
    bool last = false;
    if (__hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
      __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                       // re-arm the shard
      unsigned* top = ticket + 8 * kTicketStride;
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsh - 1) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                    // re-arm the top
        last = true;
      }
    }
    if (last && MFS_TAIL_ACQUIRE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the invalidate has completed before the barrier opens
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return false;
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock)
    acc += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double t = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) s_total = t;
  __syncthreads();
  *total = s_total;
  return true;
}

// the same tail for TWO partial-sum arrays filled by one kernel (the fused Jacobi loop's r.r and r.z): one ticket round,
// both totals in block_total_of's order
__device__ __forceinline__ bool last_block_total2(double* __restrict__ pa, double* __restrict__ pb, int my_slot, double va,
                                                  double vb, int count, unsigned* ticket, unsigned nblocks, double* ta,
                                                  double* tb) {
  __shared__ int s_last2;
  __shared__ double s_ta, s_tb;
  if (threadIdx.x == 0) {
    __hip_atomic_store(pa + my_slot, va, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pb + my_slot, vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned nsh = nblocks < 8u ? nblocks : 8u;
    const unsigned shard = blockIdx.x % nsh;
    const unsigned in_shard = (nblocks - shard + nsh - 1) / nsh;
    unsigned* sc = ticket + shard * kTicketStride;
    bool last = false;
    if (__hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
      __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned* top = ticket + 8 * kTicketStride;
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsh - 1) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = true;
      }
    }
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last2 = last;
  }
  __syncthreads();
  if (!s_last2) return false;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) {
    a += __hip_atomic_load(pa + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b += __hip_atomic_load(pb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double t1 = block_sum<kBlock>(a);
  const double t2 = block_sum<kBlock>(b);
  if (threadIdx.x == 0) { s_ta = t1; s_tb = t2; }
  __syncthreads();
  *ta = s_ta; *tb = s_tb;
  return true;
}

// the bookkeeping that closes an iteration (ONE thread): convergence test (:218), history, iteration
// count, delta ring, beta (:220)
__device__ __forceinline__ void cg_book(double* __restrict__ scal, double* __restrict__ hist, int64_t hist_cap, int par,
                                        double dq, double rr) {
  const double delta = scal[S_RING + par];
  const int64_t it = (int64_t)scal[S_ITERS];
  if (2 * it + 2 < hist_cap) { hist[2 * it + 1] = dq; hist[2 * it + 2] = rr; }
  scal[S_ITERS] = (double)(it + 1);
  scal[S_RING + (par ^ 1)] = rr;
  scal[S_RR] = rr;
  scal[S_DELTA] = delta;
  scal[S_LASTRR] = rr;
  scal[S_ALPHA] = delta / dq;
  if (const int bad = cg_health(dq, rr)) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; return; }
  if (rr < scal[S_TOL2]) scal[S_DONE] = 1.0; else scal[S_BETA] = rr / delta;
}

// one wave: this rank's value -> every rank's window -> the world's total in rank order (lane 0); see mfs_p2p.h
__device__ __forceinline__ double slab_allreduce_wave(const P2pDev& pd, int ring, unsigned tag, double v, bool* ok) {
  ar_send(pd, ring, tag, v, threadIdx.x);
  return ar_recv(pd, ring, tag, threadIdx.x, ok);
}

__device__ __forceinline__ void slab_fail(double* scal, int code) {
  // a peer did not answer: raise the error word and stop the solve (every later kernel returns at its top)
  __hip_atomic_store(scal + S_ERR, (double)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(scal + S_DONE, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what the LAST block of k_update_xr does after the r.r partials are complete
struct XrTail {
  int kind;                 // 0 nothing; 1 bookkeeping (one GPU); 2 all-reduce over the windows, then bookkeeping (slab loop);
                            // 3 the rank's total into scalars[RR] only (collective transport: all-reduce + k_cg_book follow)
  double* hist;
  int64_t hist_cap;
  unsigned* ticket;
  int ring;
  unsigned tag;
};

// alpha = delta / dq ; x += alpha d ; r -= alpha q ; partial sum r^2   (:211-216)
// NTX: x is touched once per iteration and by no other kernel -> stream it past the
// caches (nontemporal load + store) so that d, r, q keep their Infinity-Cache lines.
// MODE 0: both updates in one pass (one GPU).  MODE 1: r only (+ partials), MODE 2: x only --
// the multi-GPU driver runs them as two kernels so that the x update overlaps the r.r all-reduce.
template <typename T, int VEC, bool NTX, int MODE>
__global__ void __launch_bounds__(kBlock)
k_update_xr(T* __restrict__ x, const T* __restrict__ d, T* __restrict__ r, const T* __restrict__ q, int64_t n,
            double* __restrict__ scal, double* __restrict__ partial, int rev, int par,
            const double* __restrict__ part_dq, int npart, XrTail tail, P2pDev pd, LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  // Everything this block needs first is requested in ONE batch -- the done flag, the delta ring, the d.q partials and
  // the first vector of every lane -- and only then waited for: on a small grid the kernel is a chain of memory round
  // trips, and flag -> partials -> vectors in sequence were three of them.  (Loads past a raised flag are harmless.)
  const double dn = scal[S_DONE];
  const double delta = scal[S_RING + par];
  const int64_t nv_all = n / VEC;
  const int64_t nv = live_nv(lm, nv_all);                 // vectors swept: all of them, or those of the live chunks
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool rv_ = rev != 0;
  auto vec_at = [&](int64_t k) { return live_vec(lm, rv_ ? nv - 1 - k : k); };      // (may reach past the end in the last live chunk)
  vec_t<T, VEC> xv = {}, dv = {}, rv = {}, qv = {};
  auto fetch = [&](int64_t i) {
    if (MODE != 1) {
      xv = NTX ? vload_nt<T, VEC>(x + i) : vload<T, VEC>(x + i);
      dv = vload<T, VEC>(d + i);
    }
    if (MODE != 2) {
      rv = vload<T, VEC>(r + i);     // (streaming r here as well: no gain for viscosity, -1 % for pressure)
      // MODE 1 with NTX (x is not touched here): q -- written by the apply, read only here -- is streamed instead
      qv = (MODE == 1 && NTX) ? vload_nt<T, VEC>(q + i) : vload<T, VEC>(q + i);
    }
  };
  if (k0 < nv && vec_at(k0) < nv_all) fetch(vec_at(k0) * VEC);
  // d.q: folded reduction of the apply's partials (npart > 0) or the all-reduced scalar
  const double dq = npart > 0 ? block_total_of(part_dq, npart) : scal[S_DQ];
  if (dn != 0.0) return;
  if (npart > 0 && blockIdx.x == 0 && threadIdx.x == 0) scal[S_DQ] = dq;   // kept for the history entry
  const double alpha = delta / dq;
  double acc = 0.0;
  // sweep direction: see for_each_vec (same order of the vectors per lane, so the r.r partials group identically)
  for (int64_t k = k0; k < nv; k += stride) {
    const int64_t kr = vec_at(k);
    if (kr >= nv_all) continue;
    const int64_t i = kr * VEC;
    if (k != k0) fetch(i);
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) xv[j] = (T)((double)xv[j] + alpha * (double)dv[j]);
      if (NTX) vstore_nt<T, VEC>(x + i, xv); else vstore<T, VEC>(x + i, xv);
    }
    if (MODE != 2) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        rv[j] = (T)((double)rv[j] - alpha * (double)qv[j]);
        acc += (double)rv[j] * (double)rv[j];
      }
      vstore<T, VEC>(r + i, rv);
    }
  }
  {   // scalar tail (n % VEC elements): the first threads of block 0, after their vectors
    const int64_t rest = n - nv_all * VEC;
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < rest) {
      const int64_t i = nv_all * VEC + threadIdx.x;
      if (MODE != 1) x[i] = (T)((double)x[i] + alpha * (double)d[i]);
      if (MODE != 2) {
        const T rn = (T)((double)r[i] - alpha * (double)q[i]);
        r[i] = rn;
        acc += (double)rn * (double)rn;
      }
    }
  }
  if (MODE != 2) {
    const double tot = block_sum<kBlock>(acc);
    if (tail.kind == 0) {
      if (threadIdx.x == 0) partial[blockIdx.x] = tot;
      return;
    }
    // the iteration is closed by the last block to get here instead of by a kernel of its own
    double rr;
    if (!last_block_total(partial, blockIdx.x, tot, gridDim.x, tail.ticket, gridDim.x, &rr)) return;
    if (threadIdx.x >= kWave) return;
    if (tail.kind == 2) {
      bool ok;
      rr = slab_allreduce_wave(pd, tail.ring, tail.tag, rr, &ok);
      if (!ok) { if (threadIdx.x == 0) slab_fail(scal, 1); return; }
    }
    if (tail.kind == 3) {      // collective transport: this rank's r.r into the scalar block; an all-reduce and k_cg_book follow
      if (threadIdx.x == 0) scal[S_RR] = rr;
      return;
    }
    if (threadIdx.x == 0) cg_book(scal, tail.hist, tail.hist_cap, par, dq, rr);
  }
}

// convergence test (:218), bookkeeping, beta (:220), d = r + beta d (:221)
// XUPD: this kernel also does `x += alpha d` (:212) -- the x update rides here instead of in k_update_xr (MODE 1
// there), because d is already being read: 3 + 5 instead of 6 + 3 scalars per DOF for the two vector phases.
// The values are the same (x_j + alpha_j d_j with the same alpha); on the converging iteration only x is updated.
template <typename T, int VEC, bool XUPD = false, bool NTX = false>
__global__ void __launch_bounds__(kBlock)
k_update_d(T* __restrict__ d, const T* __restrict__ r, int64_t n, double* __restrict__ scal,
           double* __restrict__ hist, int64_t hist_cap, int rev, int par, const double* __restrict__ part_rr,
           int npart, T* __restrict__ x = nullptr, int nt_r = 0, double own_mark = 1.0, LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  // `done` is raised by THIS kernel's bookkeeping thread on the converging iteration, and with XUPD the other blocks
  // still owe the final x += alpha d: a block that starts after that store must not mistake it for an older one.
  // The marker written here is unique to the launch (own_mark = -(iteration + 1); every other writer stores +1), a
  // block returns at the top only for a marker that is not its own.  (A kernel must not test a flag it writes itself.)
  {
    const double dn = scal[S_DONE];
    if (dn != 0.0 && !(XUPD && dn == own_mark)) return;
  }
  const double rr = npart > 0 ? block_total_of(part_rr, npart) : scal[S_RR];
  const double delta = scal[S_RING + par], tol2 = scal[S_TOL2];
  const bool conv = rr < tol2;
  const double beta = rr / delta;
  const double alpha_x = XUPD ? delta / scal[S_DQ] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double dq = scal[S_DQ];
    const int64_t it = (int64_t)scal[S_ITERS];     // only this thread ever writes ITERS
    if (2 * it + 2 < hist_cap) { hist[2 * it + 1] = dq; hist[2 * it + 2] = rr; }
    scal[S_ITERS] = (double)(it + 1);
    scal[S_RING + (par ^ 1)] = rr;                 // delta of the next iteration (other ring slot)
    scal[S_RR] = rr;
    scal[S_DELTA] = delta;
    scal[S_LASTRR] = rr;
    scal[S_ALPHA] = delta / dq;
    if (const int bad = cg_health(dq, rr)) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; }
    else if (conv) scal[S_DONE] = XUPD ? own_mark : 1.0; else scal[S_BETA] = beta;
  }
  if (conv && !XUPD) return;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      Vec<T, VEC> dv = ldv<T, VEC>(d + i);
      if (XUPD) {
        if (NTX) {      // x is touched once per iteration and by no other kernel: stream it past the caches
          vec_t<T, VEC> xn = vload_nt<T, VEC>(x + i);
#pragma unroll
          for (int j = 0; j < VEC; ++j) xn[j] = (T)((double)xn[j] + alpha_x * (double)dv.v[j]);
          vstore_nt<T, VEC>(x + i, xn);
        } else {
          Vec<T, VEC> xv = ldv<T, VEC>(x + i);
#pragma unroll
          for (int j = 0; j < VEC; ++j) xv.v[j] = (T)((double)xv.v[j] + alpha_x * (double)dv.v[j]);
          *reinterpret_cast<Vec<T, VEC>*>(x + i) = xv;
        }
        if (conv) return;
      }
      Vec<T, VEC> rv;
      if (XUPD && nt_r) {      // r streamed too (MFS_NT_RD; auto by size)
        const vec_t<T, VEC> rn = vload_nt<T, VEC>(r + i);
#pragma unroll
        for (int j = 0; j < VEC; ++j) rv.v[j] = rn[j];
      } else {
        rv = ldv<T, VEC>(r + i);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) dv.v[j] = (T)((double)rv.v[j] + beta * (double)dv.v[j]);
      *reinterpret_cast<Vec<T, VEC>*>(d + i) = dv;
    } else {
      if (XUPD) x[i] = (T)((double)x[i] + alpha_x * (double)d[i]);
      if (conv) return;
      d[i] = (T)((double)r[i] + beta * (double)d[i]);
    }
  }, rev == 1, rev == 2, lm);
}

// ---- small problems: the two vector phases of an iteration in ONE launch (k_update_rdx) -----------------------------
// r -= alpha q ; r.r ; [test, beta] ; x += alpha d ; d = r + beta d  -- the r update, the reduction and the direction + x
// update of k_update_xr<MODE 1> / k_update_d<XUPD>, with the r.r exchanged between the RESIDENT workgroups of the launch
// instead of across a kernel boundary: every workgroup publishes its partial sum as a self-validating 16-byte record
// {tag|lo, tag|hi} (agent scope, the granule scheme of mfs_p2p.h / mfs_pcg_resident.h), polls all of them and adds them in
// one fixed order -- identical bits everywhere, no counter, no flag, no fence.  r_new stays in registers across the
// exchange (one read of r less) and nothing is written before the total is known, so a launch whose workgroups are not
// all resident (a GPU shared with other work) times out CLEAN: error word kErrNotResident, the poll switches the engine
// back to the launch-per-phase loop and the iteration is redone.  One launch and one reduction tail less per iteration:
// 48 x 80 x 48 fp64 viscosity 24.4 -> see DESIGN.md section 4.  Same arithmetic per element; the partial sums group by
// kRdxW workgroups instead of the update kernel's grid, so results agree with the three-launch loop to rounding.
constexpr int kRdxBlock = 512;
constexpr int kRdxW = 128;                  // workgroups (all must be co-resident: 128 of 256 CUs)
constexpr int kRdxMaxKV = 8;                // 16-byte vectors per thread held in registers
typedef unsigned long long rdx_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rdx_u64x2 rdx_load2(const unsigned long long* p) {
  rdx_u64x2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct RdxArgs {
  unsigned long long* rec;       // kRdxW records of two u64, kRdxRecStride u64 apart
  unsigned tag;                  // != 0, unique per launch
  unsigned long long timeout_ticks;   // wall clock, 100 MHz
  int drop_block;                // fault injection (MFS_RDX_TEST_DROP_WG): this workgroup never publishes; -1 none
  double* hist;
  long long hist_cap;
};
constexpr int kRdxRecStride = 16;           // u64 words between records (128 bytes)

template <typename T, int VEC, int KV>
__global__ void __launch_bounds__(kRdxBlock)
k_update_rdx(T* __restrict__ x, T* __restrict__ d, T* __restrict__ r, const T* __restrict__ q, int64_t n,
             double* __restrict__ scal, int par, const double* __restrict__ part_dq, int npart, RdxArgs a) {
  typedef vec_t<T, VEC> V;
  // agent-scope load (sc1, from L2): a workgroup scheduled LATE -- after the others of this launch timed out on its
  // missing record and raised the flag -- must see it; a plain load may be served by a line another XCD's L2 still holds
  const double dn = __hip_atomic_load(scal + S_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double delta = scal[S_RING + par], tol2 = scal[S_TOL2];
  const int64_t nv = n / VEC;
  const int64_t stride = (int64_t)kRdxW * kRdxBlock;
  const int64_t k0 = (int64_t)blockIdx.x * kRdxBlock + threadIdx.x;
  V rn[KV], qv[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int64_t i = (k0 + k * stride) * VEC;
    rn[k] = V{}; qv[k] = V{};
    if (k0 + k * stride < nv) { rn[k] = vload<T, VEC>(r + i); qv[k] = vload<T, VEC>(q + i); }
  }
  const int64_t rest = n - nv * VEC;
  const bool tail = blockIdx.x == 0 && (int64_t)threadIdx.x < rest;        // scalar tail (n % VEC elements)
  const int64_t it_ = nv * VEC + threadIdx.x;
  T rt = tail ? r[it_] : (T)0;
  const T qt = tail ? q[it_] : (T)0;
  // d.q: block_total_of's order (every workgroup: the same value)
  double dq;
  {
    __shared__ double s_dq;
    double acc = 0.0;
    for (int i = threadIdx.x; i < npart; i += kRdxBlock) acc += part_dq[i];
    const double t = block_sum<kRdxBlock>(acc);
    if (threadIdx.x == 0) s_dq = t;
    __syncthreads();
    dq = s_dq;
  }
  if (dn != 0.0) return;
  const double alpha = delta / dq;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      rn[k][j] = (T)((double)rn[k][j] - alpha * (double)qv[k][j]);
      acc += (double)rn[k][j] * (double)rn[k][j];          // (vectors past the end hold 0: they add nothing)
    }
  }
  if (tail) { rt = (T)((double)rt - alpha * (double)qt); acc += (double)rt * (double)rt; }
  // the operands of the second half, requested before the exchange (they are consumed after it)
  V dv[KV], xv[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int64_t i = (k0 + k * stride) * VEC;
    dv[k] = V{}; xv[k] = V{};
    if (k0 + k * stride < nv) { dv[k] = vload<T, VEC>(d + i); xv[k] = vload<T, VEC>(x + i); }
  }
  T dt_ = tail ? d[it_] : (T)0, xt = tail ? x[it_] : (T)0;
  // ---- r.r over the launch: publish, poll, add in one fixed order
  const double mine = block_sum<kRdxBlock>(acc);
  if (threadIdx.x == 0 && (int)blockIdx.x != a.drop_block) {
    unsigned long long* g = a.rec + (size_t)blockIdx.x * kRdxRecStride;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
    rdx_u64x2 w;
    w[0] = ((unsigned long long)a.tag << 32) | (bits & 0xffffffffull);
    w[1] = ((unsigned long long)a.tag << 32) | (bits >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(g), "v"(w) : "memory");
  }
  __shared__ double s_rr;
  __shared__ int s_ok;
  if (threadIdx.x < kWave) {
    bool good = true;
    double c[kRdxW / kWave];
#pragma unroll
    for (int m = 0; m < kRdxW / kWave; ++m) {
      const unsigned long long* g = a.rec + (size_t)(m * kWave + threadIdx.x) * kRdxRecStride;
      rdx_u64x2 w = rdx_load2(g);
      bool got = (w[0] >> 32) == a.tag && (w[1] >> 32) == a.tag;
      for (int spin = 0; spin < 64 && !got; ++spin) { w = rdx_load2(g); got = (w[0] >> 32) == a.tag && (w[1] >> 32) == a.tag; }
      if (!got && good) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          __builtin_amdgcn_s_sleep(1);
          w = rdx_load2(g);
          if ((w[0] >> 32) == a.tag && (w[1] >> 32) == a.tag) break;
          if (wall_clock64() - t0 > a.timeout_ticks) { good = false; break; }
        }
      }
      c[m] = __longlong_as_double((long long)((w[1] << 32) | (w[0] & 0xffffffffull)));
    }
    good = __all(good);
    double tot = 0.0;
#pragma unroll
    for (int m = 0; m < kRdxW / kWave; ++m) tot += wave_sum(c[m]);      // records in index order, one fixed tree
    if (threadIdx.x == 0) { s_rr = tot; s_ok = good ? 1 : 0; }
  }
  __syncthreads();
  if (!s_ok) {       // not all workgroups are resident: nothing has been written; say so and stop the batch
    if (threadIdx.x == 0) {
      // withdraw the own record (tag 0 is never a launch's tag): a workgroup that arrives after this one has left must
      // not find a complete table and carry the episode through on its own chunk -- it times out clean like everybody else
      unsigned long long* g = a.rec + (size_t)blockIdx.x * kRdxRecStride;
      rdx_u64x2 w;
      w[0] = 0ull; w[1] = 0ull;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(g), "v"(w) : "memory");
      __hip_atomic_store(scal + S_ERR, (double)kErrNotResident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(scal + S_DONE, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // a complete table and a raised flag cannot both be this launch's (whoever raised it withdrew a record first); checked
  // once more before the first store all the same: the flag is the contract "nothing written" rests on
  if (__hip_atomic_load(scal + S_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) return;
  const double rr = s_rr;
  const bool conv = rr < tol2;
  const double beta = rr / delta;
  const int bad = cg_health(dq, rr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {      // k_update_d's bookkeeping
    const int64_t it = (int64_t)scal[S_ITERS];
    if (2 * it + 2 < a.hist_cap) { a.hist[2 * it + 1] = dq; a.hist[2 * it + 2] = rr; }
    scal[S_ITERS] = (double)(it + 1);
    scal[S_RING + (par ^ 1)] = rr;
    scal[S_DQ] = dq;
    scal[S_RR] = rr;
    scal[S_DELTA] = delta;
    scal[S_LASTRR] = rr;
    scal[S_ALPHA] = alpha;
    if (bad) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; }
    else if (conv) scal[S_DONE] = 1.0; else scal[S_BETA] = beta;
  }
  // ---- second half: r (new), x += alpha d, and -- unless the iteration converged -- d = r + beta d   (:595-597, :604-610)
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int64_t i = (k0 + k * stride) * VEC;
    if (k0 + k * stride < nv) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xv[k][j] = (T)((double)xv[k][j] + alpha * (double)dv[k][j]);
        dv[k][j] = (T)((double)rn[k][j] + beta * (double)dv[k][j]);
      }
      vstore<T, VEC>(r + i, rn[k]);
      vstore<T, VEC>(x + i, xv[k]);
      if (!conv && !bad) vstore<T, VEC>(d + i, dv[k]);
    }
  }
  if (tail) {
    r[it_] = rt;
    x[it_] = (T)((double)xt + alpha * (double)dt_);
    if (!conv && !bad) d[it_] = (T)((double)rt + beta * (double)dt_);
  }
}

// d = r + beta d with beta as left in the scalar block by k_cg_book: the direction update a fused
// native loop still owes when the host stops it WITHOUT convergence (the reference updates d at the
// end of every non-converged iteration, :220-221)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_d_axpy(T* __restrict__ d, const T* __restrict__ r, int64_t n, const double* __restrict__ scal, LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  if (scal[S_DONE] != 0.0) return;
  const double beta = scal[S_BETA];
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      vec_t<T, VEC> dv = vload<T, VEC>(d + i);
      const vec_t<T, VEC> rv = vload<T, VEC>(r + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dv[j] = (T)((double)rv[j] + beta * (double)dv[j]);
      vstore<T, VEC>(d + i, dv);
    } else {
      d[i] = (T)((double)r[i] + beta * (double)d[i]);
    }
  }, false, false, lm);      // (a solve's live chunks: r = d = 0 everywhere else)
}

// dst = src over a solve's live chunks (the direction vector brought home from the engine's partner buffer: both hold 0
// everywhere else)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_copy_live(T* __restrict__ dst, const T* __restrict__ src, int64_t n, LiveMap lm) {
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) vstore<T, VEC>(dst + i, vload<T, VEC>(src + i)); else dst[i] = src[i];
  }, false, false, lm);
}

// x += alpha d with alpha as left in the scalar block (delta / d.q of the last completed iteration): the solution
// update a loop with the deferred x update still owes when it stops (converged or not)
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_x_axpy(T* __restrict__ x, const T* __restrict__ d, int64_t n, const double* __restrict__ scal, LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  const double alpha = scal[S_ALPHA];
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      vec_t<T, VEC> xv = vload<T, VEC>(x + i);
      const vec_t<T, VEC> dv = vload<T, VEC>(d + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) xv[j] = (T)((double)xv[j] + alpha * (double)dv[j]);
      vstore<T, VEC>(x + i, xv);
    } else {
      x[i] = (T)((double)x[i] + alpha * (double)d[i]);
    }
  }, false, false, lm);
}

// The bookkeeping half of k_update_d on its own (one block): r.r from the partials, convergence
// test (:218), history, iteration count, delta ring, beta (:220).  Used when the direction update
// itself is folded into the next stencil launch (mfs_pcg_apply.h, FUSE).
static __global__ void __launch_bounds__(kBlock)
k_cg_book(double* __restrict__ scal, double* __restrict__ hist, int64_t hist_cap, int par,
          const double* __restrict__ part_rr, int npart) {
  if (scal[S_DONE] != 0.0) return;
  const double rr = npart > 0 ? block_total_of(part_rr, npart) : scal[S_RR];      // (npart == 0: the all-reduced scalar)
  if (threadIdx.x == 0) cg_book(scal, hist, hist_cap, par, scal[S_DQ], rr);
}

// x *= 0.0 (:198) -- a multiply, not a memset, so NaN/inf survive as in the reference.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_scale0(T* __restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = (T)((double)x[i] * 0.0);
}

// one block: partials[0..count) -> scal[which]; fixed order => deterministic.
static __global__ void __launch_bounds__(kBlock)
k_reduce(const double* __restrict__ partial, int count, double* __restrict__ scal, int which, int check_done) {
  if (check_done && scal[S_DONE] != 0.0) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
  const double tot = block_sum<kBlock>(acc);
  if (threadIdx.x == 0) {
    scal[which] = tot;
  }
}

static __global__ void k_begin_init(double* scal, double tol2) {
  if (threadIdx.x == 0) {
    for (int i = 0; i < MFS_PCG_NSCALARS; ++i) scal[i] = 0.0;
    scal[S_TOL2] = tol2;
  }
}

static __global__ void k_begin_finish(double* scal, double* hist) {
  if (threadIdx.x == 0) {
    const double rr = scal[S_RR];
    scal[S_DELTA] = rr;
    scal[S_RING + 0] = rr;                       // iteration 0 starts from delta0 (parity 0)
    scal[S_LASTRR] = rr;
    hist[0] = rr;
    if (rr < scal[S_TOL2]) scal[S_DONE] = 1.0;  // `if not self.delta < tol ** 2` (:206)
  }
}


// ------------------------------------------------- optional Jacobi preconditioning -----
// NOT the reference's algorithm (its CG is unpreconditioned, SURVEY.md "three facts" 3): an opt-in extra
// (mfs_pcg3d_set_jacobi, mfs_vcg3d_set_jacobi; generic over the flat CG vectors and a diagonal array) for callers that want fewer iterations and do not need the reference's residual
// history.  z = r / diag is never stored: it is formed where it is consumed, so the preconditioner is fused
// into the two vector phases (one extra read of `diag` each).  delta = r.z drives alpha and beta; the
// convergence test stays the reference's r.r < tol^2.  Three launches per iteration: stencil, x/r update,
// direction update; dot products folded into their consumers as in the plain loop.
__device__ __forceinline__ double jac_z(double r, double dg) { return dg != 0.0 ? r / dg : 0.0; }

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_jac_init(const T* __restrict__ b, const T* __restrict__ q, const T* __restrict__ diag, T* __restrict__ d,
           T* __restrict__ r, int64_t n, double* __restrict__ part_rr, double* __restrict__ part_rz) {
  double arr = 0.0, arz = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T rv = (T)((double)b[i] - (double)q[i]);
    const double z = jac_z((double)rv, (double)diag[i]);
    r[i] = rv;
    d[i] = (T)z;
    arr += (double)rv * (double)rv;
    arz += (double)rv * z;
  }
  const double t1 = block_sum<kBlock>(arr);
  const double t2 = block_sum<kBlock>(arz);
  if (threadIdx.x == 0) { part_rr[blockIdx.x] = t1; part_rz[blockIdx.x] = t2; }
}

static __global__ void k_jac_begin_finish(double* scal, double* hist) {
  if (threadIdx.x == 0) {
    const double rr = scal[S_RR], rz = scal[S_RZ];
    scal[S_DELTA] = rz;
    scal[S_RING + 0] = rz;
    scal[S_LASTRR] = rr;
    hist[0] = rr;
    if (rr < scal[S_TOL2]) scal[S_DONE] = 1.0;
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_jac_update_xr(T* __restrict__ x, const T* __restrict__ d, T* __restrict__ r, const T* __restrict__ q,
                const T* __restrict__ diag, int64_t n, double* __restrict__ scal, double* __restrict__ part_rr,
                double* __restrict__ part_rz, int par, const double* __restrict__ part_dq, int npart) {
  if (scal[S_DONE] != 0.0) return;
  // d.q: folded from the stencil launch's partials, or (npart == 0: slab loops) the all-reduced scalar
  const double dq = npart > 0 ? block_total_of(part_dq, npart) : scal[S_DQ];
  if (npart > 0 && blockIdx.x == 0 && threadIdx.x == 0) scal[S_DQ] = dq;
  const double alpha = scal[S_RING + par] / dq;
  double arr = 0.0, arz = 0.0;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      vec_t<T, VEC> xv = vload<T, VEC>(x + i), rv = vload<T, VEC>(r + i);
      const vec_t<T, VEC> dv = vload<T, VEC>(d + i), qv = vload<T, VEC>(q + i), gv = vload<T, VEC>(diag + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xv[j] = (T)((double)xv[j] + alpha * (double)dv[j]);
        rv[j] = (T)((double)rv[j] - alpha * (double)qv[j]);
        arr += (double)rv[j] * (double)rv[j];
        arz += (double)rv[j] * jac_z((double)rv[j], (double)gv[j]);
      }
      vstore<T, VEC>(x + i, xv);
      vstore<T, VEC>(r + i, rv);
    } else {
      x[i] = (T)((double)x[i] + alpha * (double)d[i]);
      const T rn = (T)((double)r[i] - alpha * (double)q[i]);
      r[i] = rn;
      arr += (double)rn * (double)rn;
      arz += (double)rn * jac_z((double)rn, (double)diag[i]);
    }
  });
  const double t1 = block_sum<kBlock>(arr);
  const double t2 = block_sum<kBlock>(arz);
  if (threadIdx.x == 0) { part_rr[blockIdx.x] = t1; part_rz[blockIdx.x] = t2; }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
k_jac_update_d(T* __restrict__ d, const T* __restrict__ r, const T* __restrict__ diag, int64_t n,
               double* __restrict__ scal, double* __restrict__ hist, int64_t hist_cap, int par,
               const double* __restrict__ part_rr, const double* __restrict__ part_rz, int npart) {
  if (scal[S_DONE] != 0.0) return;
  const double rr = npart > 0 ? block_total_of(part_rr, npart) : scal[S_RR];      // (npart == 0: all-reduced scalars)
  const double rz = npart > 0 ? block_total_of(part_rz, npart) : scal[S_RZ];
  const double delta = scal[S_RING + par];
  const bool conv = rr < scal[S_TOL2];
  const double beta = rz / delta;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double dq = scal[S_DQ];
    const int64_t it = (int64_t)scal[S_ITERS];
    if (2 * it + 2 < hist_cap) { hist[2 * it + 1] = dq; hist[2 * it + 2] = rr; }
    scal[S_ITERS] = (double)(it + 1);
    scal[S_RING + (par ^ 1)] = rz;
    scal[S_RR] = rr;
    scal[S_RZ] = rz;
    scal[S_DELTA] = delta;
    scal[S_LASTRR] = rr;
    scal[S_ALPHA] = delta / dq;
    if (const int bad = cg_health(dq, rr)) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; }
    else if (conv) scal[S_DONE] = 1.0; else scal[S_BETA] = beta;
  }
  if (conv) return;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      vec_t<T, VEC> dv = vload<T, VEC>(d + i);
      const vec_t<T, VEC> rv = vload<T, VEC>(r + i), gv = vload<T, VEC>(diag + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dv[j] = (T)(jac_z((double)rv[j], (double)gv[j]) + beta * (double)dv[j]);
      vstore<T, VEC>(d + i, dv);
    } else {
      d[i] = (T)(jac_z((double)r[i], (double)diag[i]) + beta * (double)d[i]);
    }
  });
}

// class byte per 16-byte z-vector of the pressure engine's coefficient arrays (mfs_pcg_apply.h: compressed coefficient access)
// kClsDead (round 3): a ZERO vector whose r and d were 0 when the solve's sparse lists were built -- q, r, d stay +0 and x
// untouched there for the whole solve.  Written over kClsZero by k_pcg_live_flags, honoured ONLY by launches that carry
// the solve's work list (every other consumer treats it as kClsZero: it compares against REGULAR / MIXED).
enum : unsigned char { kClsZero = 0, kClsRegular = 1, kClsMixed = 2, kClsDead = 3 };

// ---- the FUSED Jacobi loop (2 launches + a one-block bookkeeping launch per iteration): z = r / diag is STORED by the
// r update -- which reads diag anyway for r.z -- into an engine buffer, and the stencil launch of the next iteration forms
// d = z + beta d_old on the fly with the plain loop's fused kernel (k_pcg_apply_march FUSE, operand `r` := z), the
// deferred x update riding along as there.  12 + 2 scalars per cell and iteration instead of 6 + 8 + 5 in three full passes.
// XUPD: x += alpha d here (cache-resident sizes); otherwise the next stencil launch does it (XDEF).
// cls (compressed coefficient access on): the class byte of a z-vector stands for its diagonal unless the vector is MIXED
// (ZERO: 0 -> z = 0; REGULAR: 6; a class leaves the never-computed boundary cells of a vector open, where r is exactly 0 and
// z therefore 0 either way).  The LAST block to finish closes the iteration (last_block_total2 + jac_book): no third launch.
// (dq comes as a VALUE: scal[S_DQ] is a plain store of another workgroup of the same launch, possibly behind another XCD's L2)
__device__ __forceinline__ void jac_book(double* __restrict__ scal, double* __restrict__ hist, int64_t hist_cap, int par,
                                         double dq, double rr, double rz) {
  const double delta = scal[S_RING + par];
  const int64_t it = (int64_t)scal[S_ITERS];
  if (2 * it + 2 < hist_cap) { hist[2 * it + 1] = dq; hist[2 * it + 2] = rr; }
  scal[S_ITERS] = (double)(it + 1);
  scal[S_RING + (par ^ 1)] = rz;
  scal[S_RR] = rr;
  scal[S_RZ] = rz;
  scal[S_DELTA] = delta;
  scal[S_LASTRR] = rr;
  scal[S_ALPHA] = delta / dq;
  if (const int bad = cg_health(dq, rr)) { scal[S_ERR] = (double)bad; scal[S_DONE] = 1.0; }
  else if (rr < scal[S_TOL2]) scal[S_DONE] = 1.0; else scal[S_BETA] = rz / delta;
}

struct JacSlab { int on; P2pDev pd; int ring[2]; unsigned tag[2]; };

template <typename T, int VEC, bool XUPD>
__global__ void __launch_bounds__(kBlock)
k_jac_update_rz(T* __restrict__ x, const T* __restrict__ d, T* __restrict__ r, const T* __restrict__ q,
                const T* __restrict__ diag, T* __restrict__ z, int64_t n, double* __restrict__ scal,
                double* __restrict__ part_rr, double* __restrict__ part_rz, int par, const double* __restrict__ part_dq,
                int npart, const unsigned char* __restrict__ cls, double* __restrict__ hist, int64_t hist_cap,
                unsigned* __restrict__ ticket, JacSlab sl, LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  const double dn = scal[S_DONE];
  const double delta = scal[S_RING + par];
  // d.q: folded from the stencil launch's partials (one GPU) or the all-reduced scalar (slab loop: npart == 0)
  const double dq = npart > 0 ? block_total_of(part_dq, npart) : scal[S_DQ];
  if (dn != 0.0) return;
  if (npart > 0 && blockIdx.x == 0 && threadIdx.x == 0) scal[S_DQ] = dq;
  const double alpha = delta / dq;
  double arr = 0.0, arz = 0.0;
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      vec_t<T, VEC> rv = vload<T, VEC>(r + i), zv;
      const vec_t<T, VEC> qv = vload<T, VEC>(q + i);
      vec_t<T, VEC> gv;
      if (cls) {
        const unsigned char c = cls[i / VEC];
        const T g0 = c == kClsRegular ? (T)6 : (T)0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) gv[j] = g0;
        if (c == kClsMixed) gv = vload<T, VEC>(diag + i);
      } else {
        gv = vload<T, VEC>(diag + i);
      }
      if (XUPD) {
        vec_t<T, VEC> xv = vload<T, VEC>(x + i);
        const vec_t<T, VEC> dv = vload<T, VEC>(d + i);
#pragma unroll
        for (int j = 0; j < VEC; ++j) xv[j] = (T)((double)xv[j] + alpha * (double)dv[j]);
        vstore<T, VEC>(x + i, xv);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        rv[j] = (T)((double)rv[j] - alpha * (double)qv[j]);
        const double zz = jac_z((double)rv[j], (double)gv[j]);
        zv[j] = (T)zz;
        arr += (double)rv[j] * (double)rv[j];
        arz += (double)rv[j] * zz;
      }
      vstore<T, VEC>(r + i, rv);
      vstore<T, VEC>(z + i, zv);
    } else {
      if (XUPD) x[i] = (T)((double)x[i] + alpha * (double)d[i]);
      const T rn = (T)((double)r[i] - alpha * (double)q[i]);
      const double zz = jac_z((double)rn, (double)diag[i]);
      r[i] = rn;
      z[i] = (T)zz;
      arr += (double)rn * (double)rn;
      arz += (double)rn * zz;
    }
  }, false, false, lm);      // (a solve's live chunks: r = z = d = 0 everywhere else)
  const double t1 = block_sum<kBlock>(arr);
  const double t2 = block_sum<kBlock>(arz);
  double rr, rz;
  if (!last_block_total2(part_rr, part_rz, blockIdx.x, t1, t2, gridDim.x, ticket, gridDim.x, &rr, &rz)) return;
  if (sl.on) {      // slab loop: both dot products over all ranks (two episodes of the window all-reduce), then the bookkeeping
    if (threadIdx.x >= kWave) return;
    bool ok;
    rr = slab_allreduce_wave(sl.pd, sl.ring[0], sl.tag[0], rr, &ok);
    if (!ok) { if (threadIdx.x == 0) slab_fail(scal, 1); return; }
    rr = __shfl(rr, 0, kWave);
    rz = slab_allreduce_wave(sl.pd, sl.ring[1], sl.tag[1], rz, &ok);
    if (!ok) { if (threadIdx.x == 0) slab_fail(scal, 1); return; }
  }
  if (threadIdx.x == 0) jac_book(scal, hist, hist_cap, par, dq, rr, rz);
}

// x += alpha d ; d = z + beta d -- the second vector phase of the z-storing Jacobi loop on flat vectors (viscosity engine).
// The r / z update before it has already CLOSED iteration `it` (jac_book in its last block: S_ITERS = it + 1, S_ALPHA, and
// S_BETA or -- converged -- S_DONE): this kernel owes x for exactly that iteration, and d unless it converged; launches queued
// behind a converged (or failed) iteration find S_ITERS != their own it + 1 and do nothing.
template <typename T, int VEC, bool NTX>
__global__ void __launch_bounds__(kBlock)
k_jac_dx(T* __restrict__ x, T* __restrict__ d, const T* __restrict__ z, int64_t n, const double* __restrict__ scal, double it,
         LiveMap lm = LiveMap{nullptr, nullptr, 0}) {
  if (scal[S_ITERS] != it + 1.0 || scal[S_ERR] != 0.0) return;
  const bool conv = scal[S_DONE] != 0.0;
  const double alpha = scal[S_ALPHA], beta = conv ? 0.0 : scal[S_BETA];
  for_each_vec<T, VEC>(n, [&](int64_t i, bool vec) {
    if (vec) {
      vec_t<T, VEC> dv = vload<T, VEC>(d + i);
      vec_t<T, VEC> xv = NTX ? vload_nt<T, VEC>(x + i) : vload<T, VEC>(x + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) xv[j] = (T)((double)xv[j] + alpha * (double)dv[j]);
      if (NTX) vstore_nt<T, VEC>(x + i, xv); else vstore<T, VEC>(x + i, xv);
      if (conv) return;
      const vec_t<T, VEC> zv = NTX ? vload_nt<T, VEC>(z + i) : vload<T, VEC>(z + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dv[j] = (T)((double)zv[j] + beta * (double)dv[j]);
      vstore<T, VEC>(d + i, dv);
    } else {
      x[i] = (T)((double)x[i] + alpha * (double)d[i]);
      if (conv) return;
      d[i] = (T)((double)z[i] + beta * (double)d[i]);
    }
  }, false, false, lm);
}

// ------------------------------------------------------------- host side ----
// The flat CG state of one engine: n DOFs of dtype dt in five caller-owned vectors.
struct CgCore {
  int dt = MFS_F64;
  int64_t n = 0;
  size_t elt = 8;
  double *scal = nullptr, *hist = nullptr, *part_dq = nullptr, *part_rr = nullptr;
  unsigned* tickets = nullptr;   // two sets of sharded arrival counters (kTicketWords each) for the in-launch reduction tails; zero between launches
  void *b = nullptr, *x = nullptr, *d = nullptr, *r = nullptr, *q = nullptr;
  int n_part_dq = 0, n_part_rr = 0;
  int nt_xd = -1;                // direction + x update: x nontemporal (MFS_NT_XD: 0 / 1; -1 auto by size)
  int grid_vec = 2048, cus = 256;
  int64_t iter_enq = 0;        // iterations enqueued since begin (its parity selects the delta ring slot)
  int rev_xr = 0, rev_d = 0;   // sweep direction of the two vector phases (see for_each_vec)
  int xr_vpt = 8;  // vectors per thread the x/r update aims for on small problems (MFS_XR_VEC_PER_THREAD)
  int nt_x = -1;   // nontemporal x stream in k_update_xr: 1 on, 0 off, -1 auto (working set > Infinity Cache)
  double* pinned = nullptr;
  // merged vector phases for small problems (k_update_rdx): records table, monotonic tag, on / off (MFS_RDX; switched off for
  // good by a poll that finds the launch was not fully resident)
  unsigned long long* rdx_rec = nullptr;
  unsigned rdx_tag = 0;
  int rdx = 1;
  // knobs read ONCE, at engine creation (never per launch: the loop is tuned at the microsecond level, and a test hook
  // must not be switchable under a running solve)
  unsigned long long rdx_timeout_ticks = 25000000ull;   // MFS_RDX_TIMEOUT_MS (wall clock, 100 MHz)
  int rdx_drop_wg = -1;                                 // MFS_RDX_TEST_DROP_WG: fault injection, tests only
  int nt_q = -1, nt_rd = -1;                            // MFS_NT_Q, MFS_NT_RD (A/B: -1 auto by size)
  LiveMap live{nullptr, nullptr, 0};                    // live chunks of the bound vectors (single domain / a slab's owned planes); null: all
  int64_t live_off = 0, live_cnt = -1;                  // ... built over the elements [live_off, live_off + live_cnt): only a sweep of
                                                        // exactly that range takes the list (-1: the whole vectors)
};

static inline size_t core_ws_bytes() {
  return 256 + align_up((size_t)kHistCap * 8, 256) + 2 * align_up((size_t)kMaxPartials * 8, 256) +
         2 * align_up((size_t)kTicketWords * 4, 256) + align_up((size_t)kRdxW * kRdxRecStride * 8, 256);
}

// carve scalars / history / partials out of the head of the workspace; returns the first free byte
static inline char* core_carve(CgCore& c, char* p) {
  c.scal = (double*)p; p += 256;
  c.hist = (double*)p; p += align_up((size_t)kHistCap * 8, 256);
  c.part_dq = (double*)p; p += align_up((size_t)kMaxPartials * 8, 256);
  c.part_rr = (double*)p; p += align_up((size_t)kMaxPartials * 8, 256);
  c.tickets = (unsigned*)p; p += 2 * align_up((size_t)kTicketWords * 4, 256);   // two ticket sets (zeroed with the workspace)
  c.rdx_rec = (unsigned long long*)p; p += align_up((size_t)kRdxW * kRdxRecStride * 8, 256);   // zeroed too: tag 0 is never used
  return p;
}

static inline int core_init(CgCore& c, int dt, int64_t n) {
  c.dt = dt; c.n = n; c.elt = dtype_size(dt);
  c.nt_xd = env_int("MFS_NT_XD", -1);
  int dev = 0;
  c.cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c.cus = prop.multiProcessorCount;
  }
  c.grid_vec = std::min(kMaxPartials, c.cus * env_int("MFS_VEC_BLOCKS_PER_CU", 8));
  c.nt_x = env_int("MFS_NT_X", -1);
  c.xr_vpt = std::max(1, env_int("MFS_XR_VEC_PER_THREAD", 8));
  c.rev_xr = env_int("MFS_REV_XR", 0);
  c.rev_d = env_int("MFS_REV_D", 0);
  c.rdx = env_int("MFS_RDX", 1);
  c.rdx_timeout_ticks = (unsigned long long)std::max(1, env_int("MFS_RDX_TIMEOUT_MS", 250)) * 100000ull;
  c.rdx_drop_wg = env_int("MFS_RDX_TEST_DROP_WG", -1);
  c.nt_q = env_int("MFS_NT_Q", -1);
  c.nt_rd = env_int("MFS_NT_RD", -1);
  if (hipHostMalloc((void**)&c.pinned, MFS_PCG_NSCALARS * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc for the poll buffer failed");
    return MFS_E_HIP;
  }
  return MFS_OK;
}

static inline void core_free(CgCore& c) {
  if (c.pinned) (void)hipHostFree(c.pinned);
  c.pinned = nullptr;
}

static inline int core_bind(CgCore& c, void* b, void* x, void* d, void* r, void* q) {
  MFS_REQUIRE(b && x && d && r && q, "null CG vector");
  void* a[5] = {b, x, d, r, q};
  for (int i = 0; i < 5; ++i) {
    MFS_REQUIRE(((uintptr_t)a[i] % c.elt) == 0, "CG vector not aligned to its element size");
    for (int j = i + 1; j < 5; ++j) MFS_REQUIRE(a[i] != a[j], "CG vectors must be distinct arrays");
  }
  c.b = b; c.x = x; c.d = d; c.r = r; c.q = q;
  c.live = LiveMap{nullptr, nullptr, 0};      // a solve's lists describe the vectors it began with
  return MFS_OK;
}

static inline bool core_vec_ok(const CgCore& c) {
  auto al = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  return al(c.b) && al(c.x) && al(c.d) && al(c.r) && al(c.q);
}

static inline int core_vec_grid(const CgCore& c, bool vec) {
  const int64_t per = vec ? (c.dt == MFS_F32 ? 4 : 2) : 1;
  return std::max(1, (int)std::min<int64_t>(c.grid_vec, (c.n / per + kBlock - 1) / kBlock));
}

static inline int core_reduce(CgCore& c, int which, int check_done, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st, which == 0 ? c.part_dq : c.part_rr,
                     which == 0 ? c.n_part_dq : c.n_part_rr, c.scal, which == 0 ? S_DQ : S_RR, check_done);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

#define MFS_XR(TT, VV, NN, MM) \
  hipLaunchKernelGGL((k_update_xr<TT, VV, NN, MM>), dim3(grid), dim3(kBlock), 0, st, (TT*)c.x + off, (const TT*)dsrc + off, \
                     (TT*)c.r + off, (const TT*)c.q + off, cnt, c.scal, c.part_rr, c.rev_xr, (int)(c.iter_enq & 1), \
                     c.part_dq, fold ? c.n_part_dq : 0, tl, pdv, (off == c.live_off && cnt == (c.live_cnt < 0 ? c.n : c.live_cnt)) ? c.live : LiveMap{nullptr, nullptr, 0})
#define MFS_XR_MODE(MM)                                                                                              \
  if (c.dt == MFS_F32) {                                                                                             \
    if (!vec) MFS_XR(float, 1, false, MM); else if (ntx) MFS_XR(float, 4, true, MM); else MFS_XR(float, 4, false, MM); \
  } else {                                                                                                           \
    if (!vec) MFS_XR(double, 1, false, MM); else if (ntx) MFS_XR(double, 2, true, MM); else MFS_XR(double, 2, false, MM); \
  }
// mode 0: x and r together; 1: r only (+ r.r partials); 2: x only.  [off, off + cnt) restricts the
// update to a sub-range of the DOFs (the slab loop skips its ghost planes; off a multiple of 16 bytes)
// tail / pd: let the last block close the iteration (XrTail); default none.
static inline int core_update_xr(CgCore& c, bool fold, hipStream_t st, int mode = 0, const void* d_override = nullptr,
                                 int64_t off = 0, int64_t cnt = -1, const XrTail* tail = nullptr,
                                 const P2pDev* pd = nullptr) {
  MFS_REQUIRE(c.x, "engine not bound");
  MFS_REQUIRE(!tail || mode != 2, "an iteration-closing tail needs the r update (it produces r.r)");
  XrTail tl = tail ? *tail : XrTail{};
  if (tail) tl.ticket = c.tickets;
  const P2pDev pdv = pd ? *pd : P2pDev{};
  const void* dsrc = d_override ? d_override : c.d;
  if (cnt < 0) cnt = c.n - off;
  MFS_REQUIRE(off >= 0 && cnt >= 0 && off + cnt <= c.n, "update range");
  const bool vec = core_vec_ok(c) && ((size_t)off * c.elt) % 16 == 0;
  int grid = core_vec_grid(c, vec);
  {   // >= 8 vectors per thread on small problems: fewer, fatter blocks (and fewer arrival tickets for a tail);
      // the same grid with or without a tail, so that the r.r partials group identically in both loop forms
    const int64_t per = vec ? (c.dt == MFS_F32 ? 4 : 2) : 1;
    grid = std::max(1, (int)std::min<int64_t>(grid, std::max<int64_t>(c.cus, cnt / per / (kBlock * c.xr_vpt))));
  }
  bool ntx = c.nt_x < 0 ? (5.0 * (double)c.n * c.elt > 200e6) : (c.nt_x != 0);
  if (mode == 1) {
    // r-only form: the flag streams q.  Pays only when one vector is far beyond what the Infinity Cache keeps between
    // the apply and this kernel (A/B: viscosity 256^3, 201 MB vectors, 642 -> 631 us/iteration; pressure 256^3, 67 MB
    // vectors, 125.0 -> 127.9 us: q is still on-die there).  MFS_NT_Q = 0 / 1 overrides (read at engine creation).
    const int knob = c.nt_q;
    ntx = knob < 0 ? ((double)c.n * c.elt > 128e6) : (knob != 0);
  }
  if (mode == 1) { MFS_XR_MODE(1) } else if (mode == 2) { MFS_XR_MODE(2) } else { MFS_XR_MODE(0) }
  MFS_LAUNCH_CHECK();
  if (mode != 2) c.n_part_rr = grid;
  return MFS_OK;
}
#undef MFS_XR_MODE
#undef MFS_XR

// xupd: the kernel also performs x += alpha d (pair it with core_update_xr mode 1)
static inline int core_update_d(CgCore& c, bool fold, hipStream_t st, bool xupd = false) {
  MFS_REQUIRE(c.d, "engine not bound");
  const bool vec = core_vec_ok(c);
  const int grid = core_vec_grid(c, vec);
  if (xupd) {
#define MFS_UD(TT, VV, NN) \
    hipLaunchKernelGGL((k_update_d<TT, VV, true, NN>), dim3(grid), dim3(kBlock), 0, st, (TT*)c.d, (const TT*)c.r, c.n, c.scal, \
                       c.hist, kHistCap, rev_d, (int)(c.iter_enq & 1), c.part_rr, fold ? c.n_part_rr : 0, (TT*)c.x, nt_r, \
                       -(double)(c.iter_enq + 1), (c.live_off == 0 && (c.live_cnt < 0 || c.live_cnt == c.n)) ? c.live : LiveMap{nullptr, nullptr, 0})
    const int rev_d = c.rev_d;
    // ... and so is r here (same-engine A/B: viscosity 256^3 632.8 -> 596.6 us/iteration, 192^3 250.0 -> 243.6)
    const int nt_r_knob = c.nt_rd;
    const int nt_r = vec && (nt_r_knob < 0 ? 5.0 * (double)c.n * c.elt > 200e6 : nt_r_knob > 0);
    // x is touched once per iteration: streamed past the caches once the five vectors exceed the Infinity Cache
    // (same-engine A/B, tools/visc_ab.py: viscosity 256^3 614 -> 589 us/iteration, 192^3 242.6 -> 237.0, 128^3 neutral).
    const int nt_knob = c.nt_xd;
    const bool ntx = vec && (nt_knob < 0 ? 5.0 * (double)c.n * c.elt > 200e6 : nt_knob > 0);
    if (c.dt == MFS_F32) { if (!vec) MFS_UD(float, 1, false); else if (ntx) MFS_UD(float, 4, true); else MFS_UD(float, 4, false); }
    else                 { if (!vec) MFS_UD(double, 1, false); else if (ntx) MFS_UD(double, 2, true); else MFS_UD(double, 2, false); }
#undef MFS_UD
    MFS_LAUNCH_CHECK();
    ++c.iter_enq;
    return MFS_OK;
  }
  if (c.dt == MFS_F32) {
    if (vec) hipLaunchKernelGGL((k_update_d<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)c.d, (const float*)c.r, c.n, c.scal, c.hist, kHistCap, c.rev_d, (int)(c.iter_enq & 1), c.part_rr, fold ? c.n_part_rr : 0);
    else hipLaunchKernelGGL((k_update_d<float, 1>), dim3(grid), dim3(kBlock), 0, st, (float*)c.d, (const float*)c.r, c.n, c.scal, c.hist, kHistCap, c.rev_d, (int)(c.iter_enq & 1), c.part_rr, fold ? c.n_part_rr : 0);
  } else {
    if (vec) hipLaunchKernelGGL((k_update_d<double, 2>), dim3(grid), dim3(kBlock), 0, st, (double*)c.d, (const double*)c.r, c.n, c.scal, c.hist, kHistCap, c.rev_d, (int)(c.iter_enq & 1), c.part_rr, fold ? c.n_part_rr : 0);
    else hipLaunchKernelGGL((k_update_d<double, 1>), dim3(grid), dim3(kBlock), 0, st, (double*)c.d, (const double*)c.r, c.n, c.scal, c.hist, kHistCap, c.rev_d, (int)(c.iter_enq & 1), c.part_rr, fold ? c.n_part_rr : 0);
  }
  MFS_LAUNCH_CHECK();
  ++c.iter_enq;      // the d update closes an iteration
  return MFS_OK;
}

// can the merged vector phases serve this engine?  (aligned vectors, everything in kRdxW workgroups' registers)
// (Not for solves with live chunks: holding the LIVE vectors of a 128^3 viscosity solve in this kernel -- tried in round 3, sized
// from the count the host's last look brought back -- was slower than the two streaming kernels on the chunk list, 29.1 vs
// 24.1 ms per 128^3 solve: 128 workgroups are half the chip.  Those grids are larger than the limit below anyway.)
static inline bool core_rdx_ok(const CgCore& c) {
  if (!c.rdx || !c.x || !core_vec_ok(c)) return false;
  const int64_t per = c.dt == MFS_F32 ? 4 : 2;
  return c.n / per <= (int64_t)kRdxW * kRdxBlock * kRdxMaxKV && c.cus >= kRdxW;
}

// r update + r.r + bookkeeping + x / direction update of one iteration in one launch; closes the iteration
static inline int core_update_rdx(CgCore& c, hipStream_t st) {
  const int64_t per = c.dt == MFS_F32 ? 4 : 2;
  const int64_t nv = c.n / per;
  const int need = (int)((nv + (int64_t)kRdxW * kRdxBlock - 1) / ((int64_t)kRdxW * kRdxBlock));
  if (c.rdx_tag >= 0xfffffff0u) {            // tags about to wrap: start over on a clean table
    MFS_HIP_TRY(hipMemsetAsync(c.rdx_rec, 0, (size_t)kRdxW * kRdxRecStride * 8, st));
    c.rdx_tag = 0;
  }
  RdxArgs a{c.rdx_rec, ++c.rdx_tag, c.rdx_timeout_ticks, c.rdx_drop_wg, c.hist, kHistCap};
  const int par = (int)(c.iter_enq & 1);
#define MFS_RDX_GO(TT, VV, KK) \
  hipLaunchKernelGGL((k_update_rdx<TT, VV, KK>), dim3(kRdxW), dim3(kRdxBlock), 0, st, (TT*)c.x, (TT*)c.d, (TT*)c.r, (const TT*)c.q, \
                     c.n, c.scal, par, c.part_dq, c.n_part_dq, a)
#define MFS_RDX_KV(TT, VV) \
  do { if (need <= 1) MFS_RDX_GO(TT, VV, 1); else if (need <= 2) MFS_RDX_GO(TT, VV, 2); else if (need <= 4) MFS_RDX_GO(TT, VV, 4); \
       else if (need <= 6) MFS_RDX_GO(TT, VV, 6); else MFS_RDX_GO(TT, VV, 8); } while (0)
  // (the fp64 one-vector form compiles to 256 VGPRs with spills -- a compiler artefact of that instantiation: two vectors instead)
  const int need_d = need < 2 ? 2 : need;
  if (c.dt == MFS_F32) { MFS_RDX_KV(float, 4); } else { const int need = need_d; MFS_RDX_KV(double, 2); }
#undef MFS_RDX_KV
#undef MFS_RDX_GO
  MFS_LAUNCH_CHECK();
  ++c.iter_enq;
  return MFS_OK;
}

// update_xr whose last block also closes the iteration (kind 1: bookkeeping; 2: all-reduce + bookkeeping)
static inline int core_update_xr_close(CgCore& c, bool fold, hipStream_t st, const void* d_src, int kind,
                                       int64_t off = 0, int64_t cnt = -1, const P2pDev* pd = nullptr, int ring = 0,
                                       unsigned tag = 0) {
  XrTail tl{kind, c.hist, kHistCap, nullptr, ring, tag};
  if (int e = core_update_xr(c, fold, st, 0, d_src, off, cnt, &tl, pd)) return e;
  ++c.iter_enq;
  return MFS_OK;
}

static inline int core_book(CgCore& c, hipStream_t st) {
  hipLaunchKernelGGL(k_cg_book, dim3(1), dim3(kBlock), 0, st, c.scal, c.hist, kHistCap, (int)(c.iter_enq & 1), c.part_rr,
                     c.n_part_rr);
  MFS_LAUNCH_CHECK();
  ++c.iter_enq;      // closes an iteration (like core_update_d)
  return MFS_OK;
}

// begin, part 1: scalars <- 0, tol^2; x *= 0 (only if zero_x: the pressure solver's `self.x *= 0.0`)
static inline int core_begin_pre(CgCore& c, double tol, bool zero_x, hipStream_t st) {
  MFS_REQUIRE(c.x, "engine not bound");
  hipLaunchKernelGGL(k_begin_init, dim3(1), dim3(64), 0, st, c.scal, tol * tol);
  MFS_LAUNCH_CHECK();
  c.iter_enq = 0;
  if (zero_x) {
    const int gs = std::max(1, (int)std::min<int64_t>(c.grid_vec, (c.n + kBlock - 1) / kBlock));
    if (c.dt == MFS_F32) hipLaunchKernelGGL((k_scale0<float>), dim3(gs), dim3(kBlock), 0, st, (float*)c.x, c.n);
    else hipLaunchKernelGGL((k_scale0<double>), dim3(gs), dim3(kBlock), 0, st, (double*)c.x, c.n);
    MFS_LAUNCH_CHECK();
  }
  return MFS_OK;
}

// begin, part 2 (after q = A x): d = r = b - q, partial r.r, reduce -> scalars[RR]
static inline int core_begin_post(CgCore& c, hipStream_t st, bool reduce = true) {
  const bool vec = core_vec_ok(c);
  const int g2 = core_vec_grid(c, vec);
  if (c.dt == MFS_F32) {
    if (vec) hipLaunchKernelGGL((k_cg_init<float, 4>), dim3(g2), dim3(kBlock), 0, st, (const float*)c.b, (const float*)c.q, (float*)c.d, (float*)c.r, c.n, c.part_rr);
    else hipLaunchKernelGGL((k_cg_init<float, 1>), dim3(g2), dim3(kBlock), 0, st, (const float*)c.b, (const float*)c.q, (float*)c.d, (float*)c.r, c.n, c.part_rr);
  } else {
    if (vec) hipLaunchKernelGGL((k_cg_init<double, 2>), dim3(g2), dim3(kBlock), 0, st, (const double*)c.b, (const double*)c.q, (double*)c.d, (double*)c.r, c.n, c.part_rr);
    else hipLaunchKernelGGL((k_cg_init<double, 1>), dim3(g2), dim3(kBlock), 0, st, (const double*)c.b, (const double*)c.q, (double*)c.d, (double*)c.r, c.n, c.part_rr);
  }
  MFS_LAUNCH_CHECK();
  c.n_part_rr = g2;
  return reduce ? core_reduce(c, 1, 0, st) : MFS_OK;
}

static inline int core_begin_finish(CgCore& c, hipStream_t st) {
  hipLaunchKernelGGL(k_begin_finish, dim3(1), dim3(64), 0, st, c.scal, c.hist);
  MFS_LAUNCH_CHECK();
  return MFS_OK;
}

// have_pinned: the caller has just read the scalar block into c.pinned on this stream (and found nothing to repair)
static inline int core_poll(CgCore& c, hipStream_t st, int64_t* iters, int* done, double* delta, double* alpha,
                            double* beta, bool have_pinned = false) {
  if (!have_pinned) {
    MFS_HIP_TRY(hipMemcpyAsync(c.pinned, c.scal, MFS_PCG_NSCALARS * sizeof(double), hipMemcpyDeviceToHost, st));
    MFS_HIP_TRY(hipStreamSynchronize(st));
  }
  if (c.pinned[S_ERR] != 0.0) {
    const int code = (int)c.pinned[S_ERR];
    if (code == kErrZeroDq) {
      set_error("d.q == 0 in CG iteration %lld: float division by zero (the reference raises ZeroDivisionError at "
                "alpha = delta / dq)", (long long)c.pinned[S_ITERS]);
      return MFS_E_ZERODIV;
    }
    if (code == kErrNonFinite) {
      set_error("non-finite d.q or r.r in CG iteration %lld (NaN / inf in the inputs): the solve was stopped",
                (long long)c.pinned[S_ITERS]);
      return MFS_E_NONFINITE;
    }
    set_error("a bounded wait timed out inside the CG loop (code %d: 1 = dot-product exchange, 2 = halo exchange, 5 = the resident "
              "small-grid loop could not get its workgroups co-resident -- MFS_RESIDENT=0 selects the launch-per-phase loop)", code);
    return MFS_E_TIMEOUT;
  }
  if (iters) *iters = (int64_t)c.pinned[S_ITERS];
  if (done) *done = c.pinned[S_DONE] != 0.0;
  if (delta) *delta = c.pinned[S_LASTRR];
  if (alpha) *alpha = c.pinned[S_ALPHA];
  if (beta) *beta = c.pinned[S_BETA];
  return MFS_OK;
}

static inline int64_t core_history(CgCore& c, double* out_host, int64_t cap, hipStream_t st) {
  if (!out_host || cap < 0) { set_error("history: bad argument"); return MFS_E_INVALID; }
  if (hipMemcpyAsync(c.pinned, c.scal, MFS_PCG_NSCALARS * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { set_error("history: scalar readback failed"); return MFS_E_HIP; }
  int64_t cnt = std::min<int64_t>(2 * (int64_t)c.pinned[S_ITERS] + 1, kHistCap);
  cnt = std::min(cnt, cap);
  if (cnt > 0) {
    if (hipMemcpyAsync(out_host, c.hist, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { set_error("history: copy failed"); return MFS_E_HIP; }
  }
  return cnt;
}

}  // namespace mfs
