// mfs_rccl.h -- RCCL through its C API, resolved at run time (dlopen of the librccl the process already carries: the one
// PyTorch-ROCm links): the "collective" transport of the slab loops, for nodes where the HIP-IPC windows cannot be used.
// Halo planes move by ncclSend / ncclRecv on the solver's second stream while the interior launch runs; each dot product
// is ONE ncclAllReduce on the engine's scalar block, in stream order -- no host synchronisation, no Python, inside a batch.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mfs_common.h"

struct mfs_rccl {
  void* lib = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

namespace mfs {

#define MFS_RCCL_TRY(rc, expr)                                                                     \
  do {                                                                                             \
    const ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess) {                                                                       \
      ::mfs::set_error("RCCL: %s failed: %s", #expr, (rc)->GetErrorString ? (rc)->GetErrorString(r_) : "?");       \
      return MFS_E_HIP;                                                                            \
    }                                                                                              \
  } while (0)

static inline int rccl_load(mfs_rccl* r, const char* path) {
  // the library the process already has mapped (same path -> same handle); never a second copy of RCCL beside PyTorch's
  r->lib = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
  if (!r->lib) r->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!r->lib) { set_error("dlopen(%s) failed: %s", path, dlerror()); return MFS_E_INVALID; }
#define MFS_RCCL_SYM(field, name)                                                       \
  *(void**)(&r->field) = dlsym(r->lib, name);                                           \
  if (!r->field) { set_error("RCCL symbol " name " not found"); return MFS_E_INVALID; }
  MFS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
  MFS_RCCL_SYM(CommInitRank, "ncclCommInitRank")
  MFS_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  MFS_RCCL_SYM(AllReduce, "ncclAllReduce")
  MFS_RCCL_SYM(Send, "ncclSend")
  MFS_RCCL_SYM(Recv, "ncclRecv")
  MFS_RCCL_SYM(GroupStart, "ncclGroupStart")
  MFS_RCCL_SYM(GroupEnd, "ncclGroupEnd")
  MFS_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MFS_RCCL_SYM
  return MFS_OK;
}

// the four edge-plane transfers of one iteration (planes of `count` elements of `dt`): plane 1 -> left neighbour's
// ghost L-1, plane L-2 -> right neighbour's ghost 0, and the two ghosts of this rank back, in ONE group on `st`
static inline int rccl_halo(mfs_rccl* r, char* d, size_t plane_bytes, int L, hipStream_t st) {
  const bool left = r->rank > 0, right = r->rank < r->world - 1;
  if (!left && !right) return MFS_OK;
  MFS_RCCL_TRY(r, r->GroupStart());
  if (left) {
    MFS_RCCL_TRY(r, r->Send(d + 1 * plane_bytes, plane_bytes, ncclChar, r->rank - 1, r->comm, st));
    MFS_RCCL_TRY(r, r->Recv(d + 0 * plane_bytes, plane_bytes, ncclChar, r->rank - 1, r->comm, st));
  }
  if (right) {
    MFS_RCCL_TRY(r, r->Send(d + (size_t)(L - 2) * plane_bytes, plane_bytes, ncclChar, r->rank + 1, r->comm, st));
    MFS_RCCL_TRY(r, r->Recv(d + (size_t)(L - 1) * plane_bytes, plane_bytes, ncclChar, r->rank + 1, r->comm, st));
  }
  MFS_RCCL_TRY(r, r->GroupEnd());
  return MFS_OK;
}

// `count` doubles of the scalar block summed over the ranks, in place, in stream order
static inline int rccl_sum(mfs_rccl* r, double* p, int count, hipStream_t st) {
  MFS_RCCL_TRY(r, r->AllReduce(p, p, (size_t)count, ncclDouble, ncclSum, r->comm, st));
  return MFS_OK;
}

}  // namespace mfs
