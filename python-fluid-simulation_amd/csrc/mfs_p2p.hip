// mfs_p2p.hip -- the peer-to-peer window object of the slab-decomposed CG (see mfs_p2p.h):
// allocation, HIP-IPC hand-shake, and a collective self-test that moves a real plane
// payload and an all-reduce through the mapped windows before a solver trusts them.
#include "mfs_p2p.h"

namespace mfs {

constexpr unsigned kTestHaloTag = 0x20000000u;
constexpr unsigned kTestArTag = 0x40000000u;

__device__ __forceinline__ float test_pattern(int rank, int round, int side, int64_t i) {
  return (float)((i * 31 + rank * 7 + round * 3 + side) % 8191) - 4095.0f;
}

// ONE block.  result[0] = 1 ok / 0 failed, result[1] = payload vectors that never arrived or did not
// match, result[2] = all-reduce timed out, result[3] = all-reduce sum as float bits
static __global__ void __launch_bounds__(256)
k_p2p_selftest(P2pDev pd, int64_t plane_floats, int round, unsigned* result) {
  const int par = round & 1, tid = threadIdx.x;
  const int64_t nv = plane_floats / 4;
  const unsigned htag = kTestHaloTag | (unsigned)(round + 1);
  // a patterned fp32 plane to both neighbours
  for (int s = 0; s < 2; ++s) {
    if (!pd.send[s][par]) continue;
    for (int64_t i = tid; i < nv; i += blockDim.x) {
      vec_t<float, 4> v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = test_pattern(pd.rank, round, s, 4 * i + j);
      gran_store_vec<float, 4>(pd.send[s][par], 4 * i, v, htag);
    }
  }
  // all-reduce
  __shared__ double s_sum;
  __shared__ int s_arok;
  if (tid < kWave) {
    ar_send(pd, round & (kArRing - 1), kTestArTag | (unsigned)(round + 1), 1.5 * (pd.rank + 1) + round, tid);
    bool ok;
    const double tot = ar_recv(pd, round & (kArRing - 1), kTestArTag | (unsigned)(round + 1), tid, &ok);
    if (tid == 0) { s_sum = tot; s_arok = ok ? 1 : 0; }
  }
  __syncthreads();
  // planes from the neighbours: my low ghost comes from rank-1 (its side-1 send), my high ghost from rank+1
  unsigned bad = 0;
  for (int s = 0; s < 2; ++s) {
    const bool have = s == 0 ? pd.rank > 0 : pd.rank < pd.world - 1;
    if (!have) continue;
    const int from = s == 0 ? pd.rank - 1 : pd.rank + 1, from_side = s == 0 ? 1 : 0;
    for (int64_t i = tid; i < nv; i += blockDim.x) {
      vec_t<float, 4> v;
      if (!gran_load_vec<float, 4>(pd.recv[s][par], 4 * i, htag, pd.timeout_ticks, &v)) { ++bad; break; }
#pragma unroll
      for (int j = 0; j < 4; ++j) bad += v[j] != test_pattern(from, round, from_side, 4 * i + j);
    }
  }
  __shared__ unsigned s_bad;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  if (bad) atomicAdd(&s_bad, bad);
  __syncthreads();
  if (tid == 0) {
    const double want = 1.5 * pd.world * (pd.world + 1) / 2.0 + (double)round * pd.world;
    const bool ok = s_arok && s_bad == 0 && s_sum == want;
    result[0] = ok ? 1u : 0u;
    result[1] = s_bad;
    result[2] = s_arok ? 0u : 1u;
    result[3] = __float_as_uint((float)s_sum);
  }
}

static void p2p_fill_dev(P2pHost& p) {
  P2pDev& d = p.dev;
  d = P2pDev{};
  d.rank = p.rank; d.world = p.world;
  d.self = reinterpret_cast<P2pCtrl*>(p.window);
  for (int r = 0; r < p.world; ++r) d.peer[r] = reinterpret_cast<P2pCtrl*>(p.peer_window[r]);
  auto buf = [&](char* w, int side, int par) {
    return reinterpret_cast<u64*>(w + kP2pCtrlBytes + (size_t)(side * 2 + par) * p.plane_stride);
  };
  for (int s = 0; s < 2; ++s)
    for (int q = 0; q < 2; ++q) d.recv[s][q] = buf(p.window, s, q);
  for (int q = 0; q < 2; ++q) {
    if (p.rank > 0) d.send[0][q] = buf(p.peer_window[p.rank - 1], 1, q);            // my plane 1 = left neighbour's HIGH ghost
    if (p.rank < p.world - 1) d.send[1][q] = buf(p.peer_window[p.rank + 1], 0, q);  // my plane L-2 = right neighbour's LOW ghost
  }
  const int ms = std::max(1, env_int("MFS_P2P_TIMEOUT_MS", 10000));
  int dev = 0, khz = 0;                     // wall_clock64() rate of this device (100 MHz on MI300-class parts)
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess ||
      khz <= 0) {
    (void)hipGetLastError();
    khz = 100000;
  }
  d.timeout_ticks = (u64)ms * (u64)khz;
}

}  // namespace mfs

using namespace mfs;

extern "C" {

size_t mfs_p2p_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int mfs_p2p_create(mfs_p2p** out, int rank, int world, size_t plane_bytes, void* handle_out_host) {
  MFS_REQUIRE(out && handle_out_host, "null argument");
  MFS_REQUIRE(world >= 1 && world <= kP2pMaxWorld && rank >= 0 && rank < world, "rank / world");
  MFS_REQUIRE(plane_bytes > 0 && plane_bytes % 16 == 0, "plane_bytes must be a positive multiple of 16");
  mfs_p2p* p = new mfs_p2p();
  p->rank = rank; p->world = world;
  p->plane_bytes = plane_bytes;
  p->plane_stride = (2 * plane_bytes + 4095) / 4096 * 4096;   // granules: 8 bytes per 32 payload bits
  p->window_bytes = kP2pCtrlBytes + 4 * p->plane_stride;
  void* w = nullptr;
  // memory that peers write while local kernels read must not sit in a non-coherent cache:
  // uncached first (what RCCL uses on this family), fine-grained second; ordinary hipMalloc is NOT acceptable
  if (hipExtMallocWithFlags(&w, p->window_bytes, hipDeviceMallocUncached) == hipSuccess) p->alloc_kind = 1;
  else {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&w, p->window_bytes, hipDeviceMallocFinegrained) == hipSuccess) p->alloc_kind = 2;
  }
  if (!w) {
    (void)hipGetLastError();
    set_error("mfs_p2p_create: no uncached / fine-grained device memory for the window (%zu bytes)", p->window_bytes);
    delete p;
    return MFS_E_HIP;
  }
  p->window = (char*)w;
  p->peer_window[rank] = p->window;
  hipIpcMemHandle_t hd;
  if (hipMemset(w, 0, p->window_bytes) != hipSuccess || hipMalloc((void**)&p->local, 256) != hipSuccess ||
      hipMemset(p->local, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipIpcGetMemHandle(&hd, w) != hipSuccess) {
    set_error("mfs_p2p_create: window initialisation / hipIpcGetMemHandle failed: %s", hipGetErrorString(hipGetLastError()));
    if (p->local) (void)hipFree(p->local);
    (void)hipFree(w);
    delete p;
    return MFS_E_HIP;
  }
  memcpy(handle_out_host, &hd, sizeof(hd));
  *out = p;
  return MFS_OK;
}

int mfs_p2p_connect(mfs_p2p* p, const void* handles_host) {
  MFS_REQUIRE(p && handles_host, "null argument");
  MFS_REQUIRE(!p->connected, "already connected");
  const char* hs = (const char*)handles_host;
  for (int r = 0; r < p->world; ++r) {
    if (r == p->rank) continue;
    hipIpcMemHandle_t hd;
    memcpy(&hd, hs + (size_t)r * sizeof(hd), sizeof(hd));
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || !ptr) {
      (void)hipGetLastError();
      set_error("mfs_p2p_connect: hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
      return MFS_E_HIP;
    }
    p->peer_window[r] = (char*)ptr;
    p->opened[r] = true;
  }
  p2p_fill_dev(*p);
  p->connected = true;
  return MFS_OK;
}

int mfs_p2p_selftest(mfs_p2p* p, int round, mfs_stream stream, int* ok_host, unsigned* detail_host) {
  MFS_REQUIRE(p && ok_host, "null argument");
  MFS_REQUIRE(p->connected, "mfs_p2p_connect has not been called");
  MFS_REQUIRE(round >= 0 && round < 1000000, "round");
  hipStream_t st = (hipStream_t)stream;
  unsigned* result = p->local + 32;
  MFS_HIP_TRY(hipMemsetAsync(result, 0, 16, st));
  hipLaunchKernelGGL(k_p2p_selftest, dim3(1), dim3(256), 0, st, p->dev, (int64_t)(p->plane_bytes / 4), round, result);
  MFS_LAUNCH_CHECK();
  unsigned host[4] = {0, 0, 0, 0};
  MFS_HIP_TRY(hipMemcpyAsync(host, result, sizeof(host), hipMemcpyDeviceToHost, st));
  MFS_HIP_TRY(hipStreamSynchronize(st));
  *ok_host = host[0] == 1u;
  if (detail_host) memcpy(detail_host, host, sizeof(host));
  return MFS_OK;
}

int mfs_p2p_info(mfs_p2p* p, int* alloc_kind_host, size_t* window_bytes_host) {
  MFS_REQUIRE(p, "null handle");
  if (alloc_kind_host) *alloc_kind_host = p->alloc_kind;
  if (window_bytes_host) *window_bytes_host = p->window_bytes;
  return MFS_OK;
}

int mfs_p2p_destroy(mfs_p2p* p) {
  if (!p) return MFS_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < p->world; ++r)
    if (p->opened[r] && p->peer_window[r]) (void)hipIpcCloseMemHandle(p->peer_window[r]);
  if (p->local) (void)hipFree(p->local);
  if (p->window) (void)hipFree(p->window);
  delete p;
  return MFS_OK;
}

}  // extern "C"
