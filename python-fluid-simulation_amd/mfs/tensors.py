"""Host-side plumbing between torch (ROCm) tensors and the C ABI: dtype codes,
device pointers, the current HIP stream, grid-resolution normalisation."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib

_CODES = {torch.float32: _lib.MFS_F32, torch.float64: _lib.MFS_F64}
_PRECISIONS = {"fp64": torch.float64, "f64": torch.float64, "float64": torch.float64, "double": torch.float64,
               "fp32": torch.float32, "f32": torch.float32, "float32": torch.float32, "single": torch.float32}


def state_dtype(precision=None):
    """Solver-state dtype.  Default fp64 = the reference's state precision
    (solver/CGSolverBuffer.py:5-8); MFS_PRECISION=fp32 selects fp32 storage
    (arithmetic stays fp64 in registers, see csrc/mfs_pcg.hip)."""
    if isinstance(precision, torch.dtype):
        if precision not in _CODES:
            raise TypeError(f"unsupported solver dtype {precision}")
        return precision
    key = (precision or os.environ.get("MFS_PRECISION", "fp64")).lower()
    if key not in _PRECISIONS:
        raise ValueError(f"unknown precision {key!r}; use fp32 or fp64")
    return _PRECISIONS[key]


def as_gres(gres):
    """Reference passes a cupy int64 device array (uses .get()); accept that shape of
    thing plus torch tensors, numpy arrays and plain sequences."""
    if isinstance(gres, torch.Tensor):
        vals = gres.detach().cpu().tolist()
    elif hasattr(gres, "get") and not isinstance(gres, dict):
        vals = np.asarray(gres.get()).tolist()
    else:
        vals = np.asarray(gres).tolist()
    out = tuple(int(v) for v in vals)
    if any(v < 1 for v in out):
        raise ValueError(f"bad grid resolution {out}")
    return out


def as_f64_list(a, n):
    """bound_size / cell_size style argument -> list of n python floats (broadcast scalar)."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().to(torch.float64).numpy()
    elif hasattr(a, "get") and not isinstance(a, dict):
        a = a.get()
    arr = np.broadcast_to(np.asarray(a, dtype=np.float64), (n,))
    return [float(v) for v in arr]


def dev(a, name, shape=None):
    """Validate a device array argument: torch CUDA(HIP) tensor, fp32/fp64, C-contiguous.
    Arrays exporting DLPack (e.g. cupy-rocm) are imported zero-copy.  Never copies:
    the solver mutates caller arrays in place."""
    if not isinstance(a, torch.Tensor):
        if hasattr(a, "__dlpack__"):
            a = torch.from_dlpack(a)
        else:
            raise TypeError(f"{name}: expected a torch tensor on the GPU (or a DLPack exporter), got {type(a).__name__}")
    if not a.is_cuda:
        raise TypeError(f"{name}: tensor is on {a.device}; the solver path runs on the GPU only (no CPU fallback)")
    if a.dtype not in _CODES:
        raise TypeError(f"{name}: dtype {a.dtype} unsupported (float32 / float64)")
    if not a.is_contiguous():
        raise ValueError(f"{name}: tensor must be C-contiguous")
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(a.shape)} != expected {tuple(shape)}")
    return a


def code(t):
    return _CODES[t.dtype]


def ptr(t):
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def face_shape(g, axis):
    s = list(g)
    s[axis] += 1
    return tuple(s)


def doubled_shape(g):
    return tuple(2 * v + 1 for v in g)
