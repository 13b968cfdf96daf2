"""Python owner of one `mfs_p2p` window (include/mfs.h, csrc/mfs_p2p.h): the block of
uncached device memory through which the ranks of a slab-decomposed solve exchange
halo planes and dot products with plain xGMI stores.

`torch.distributed` is used ONLY to bootstrap (all-gather of the 64-byte HIP-IPC
handles, agreement on the self-test's verdict); any backend works ("nccl" = RCCL in
bench.py, "gloo" in the tests).  Inside the CG loop no collective is called.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, tensors as T


class P2PWindow:
    """COLLECTIVE constructor: every rank of `group` creates its window, the IPC handles are
    all-gathered, the peers' windows are mapped and a self-test moves real plane payloads
    and an all-reduce through them.  `self.ok` is True only if EVERY rank passed;
    the caller falls back to the RCCL loop (mfs.dist.SlabCG mode "rccl") otherwise."""

    def __init__(self, dist, plane_bytes, device, group=None, rounds=3):
        self.lib = _lib.load()
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        self.h = None
        self.ok = False
        self.why = ""
        self.detail = None
        hb = int(self.lib.mfs_p2p_handle_bytes())
        handle = C.create_string_buffer(hb)
        h = C.c_void_p()
        mine, err = b"", ""
        with torch.cuda.device(self.device):
            st = self.lib.mfs_p2p_create(C.byref(h), self.rank, self.world, int(plane_bytes), handle)
        if st == 0:
            self.h = h
            mine = bytes(handle.raw)
        else:
            err = self.lib.mfs_last_error().decode(errors="replace")
        # every rank must reach every collective below, whatever happened locally
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (mine, err), group=group)
        if any(not g[0] for g in gathered):
            self.why = "window allocation failed: " + "; ".join(g[1] for g in gathered if g[1])
            return
        with torch.cuda.device(self.device):
            st = self.lib.mfs_p2p_connect(self.h, b"".join(g[0] for g in gathered))
        err = "" if st == 0 else self.lib.mfs_last_error().decode(errors="replace")
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, err, group=group)
        if any(verdicts):
            self.why = "IPC mapping failed: " + "; ".join(v for v in verdicts if v)
            return
        good = True
        for rnd in range(int(rounds)):
            ok = C.c_int(0)
            det = (C.c_uint * 4)()
            with torch.cuda.device(self.device):
                st = self.lib.mfs_p2p_selftest(self.h, rnd, T.stream(), C.byref(ok), det)
            good = good and st == 0 and ok.value == 1
            self.detail = list(det)
        # fault injection (tests / rehearsals of the downgrade path): MFS_P2P_SELFTEST_FAIL=1 fails it on every rank,
        # =r<k> on rank k only -- the window then stays unused and the callers must fall back to the collective loop
        inj = os.environ.get("MFS_P2P_SELFTEST_FAIL", "")
        injected = inj == "1" or inj == f"r{self.rank}"
        if injected:
            good = False
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, bool(good), group=group)
        self.ok = all(verdicts)
        if not self.ok:
            self.why = (f"self-test failed on ranks {[r for r, v in enumerate(verdicts) if not v]} (detail {self.detail})"
                        + (" [injected: MFS_P2P_SELFTEST_FAIL]" if inj else ""))

    @property
    def alloc_kind(self):
        k = C.c_int(0)
        if self.h:
            self.lib.mfs_p2p_info(self.h, C.byref(k), None)
        return {1: "uncached", 2: "fine-grained"}.get(k.value, "none")

    def close(self):
        """COLLECTIVE: barrier (no rank may still be writing into a window), then free."""
        h, self.h = self.h, None
        try:
            self.dist.barrier(group=self.group)
        finally:
            if h:
                self.lib.mfs_p2p_destroy(h)
