"""The collective transport of the slab loops, native form (csrc/mfs_rccl.h, round 3): an RCCL communicator of the build's
own over the slab ranks, and the rank's OWN one-rank window.  With both attached (PcgEngine.attach_rccl) the engine's slab
loop runs the window loop's launches with no in-kernel exchange: the two edge planes of `d` travel by ncclSend / ncclRecv on
the solver's second stream beside the interior launch, each dot product is one ncclAllReduce on the device scalar block --
enqueued from C, nothing between the launches of a batch goes through Python.

`torch.distributed` is used ONLY to bootstrap (broadcast of the 128-byte ncclUniqueId).  The RCCL entry points are resolved
at run time from the librccl the process already maps (PyTorch-ROCm's), never a second copy.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, tensors as T


def rccl_library_path():
    """the librccl this process has mapped (PyTorch links it); /opt/rocm's as the last resort"""
    import torch.distributed  # noqa: F401  (makes sure libtorch_hip and its RCCL are loaded)
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"):
        if os.path.exists(cand):
            return cand
    raise _lib.MfsError("no librccl found (neither mapped by this process nor under torch/lib or /opt/rocm/lib)")


class RcclComm:
    """COLLECTIVE constructor over `group`: rank 0 draws the unique id, every rank joins the communicator."""

    def __init__(self, dist, device, group=None):
        self.lib = _lib.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        self.path = rccl_library_path()
        self.h = None
        nb = int(self.lib.mfs_rccl_unique_id_bytes())
        ids = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(nb)
            _lib.check(self.lib.mfs_rccl_unique_id(self.path.encode(), buf), "mfs_rccl_unique_id")
            ids = [bytes(buf.raw)]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(ids, src=src, group=group)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mfs_rccl_create(C.byref(h), self.path.encode(), C.create_string_buffer(ids[0], nb), self.rank,
                                                self.world), "mfs_rccl_create")
        self.h = h

    def close(self):
        h, self.h = self.h, None
        if h:
            self.lib.mfs_rccl_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalWindow:
    """this rank's OWN one-rank window (no IPC, no peer): what the collective loop's kernels use for their in-launch
    reductions; plane_bytes = Ny * Nz * sizeof(element) as for the shared windows"""

    def __init__(self, plane_bytes, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.h = None
        self.ok = False
        self.why = ""
        hb = int(self.lib.mfs_p2p_handle_bytes())
        handle = C.create_string_buffer(hb)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mfs_p2p_create(C.byref(h), 0, 1, int(plane_bytes), handle), "mfs_p2p_create")
            self.h = h
            _lib.check(self.lib.mfs_p2p_connect(self.h, bytes(handle.raw)), "mfs_p2p_connect")
        self.ok = True

    def close(self):
        h, self.h = self.h, None
        if h:
            self.lib.mfs_p2p_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
