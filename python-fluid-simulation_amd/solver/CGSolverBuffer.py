"""Drop-in for the reference's solver/CGSolverBuffer.py:3-8 on PyTorch-ROCm tensors."""
import torch

from mfs import tensors as T


class CGSolverBuffer:
    """Four zero-initialised cell arrays d, r, q, b of shape `gres`, shared by the
    pressure (and, in the reference, density) solvers.  fp64 like the reference
    unless MFS_PRECISION / `precision` says fp32."""

    def __init__(self, gres, precision=None, device=None):
        shape = T.as_gres(gres)
        dt = T.state_dtype(precision)
        device = torch.device("cuda" if device is None else device)
        if device.type != "cuda":
            raise TypeError("CGSolverBuffer lives on the GPU; there is no CPU solver path")
        self.d = torch.zeros(shape, dtype=dt, device=device)
        self.r = torch.zeros(shape, dtype=dt, device=device)
        self.q = torch.zeros(shape, dtype=dt, device=device)
        self.b = torch.zeros(shape, dtype=dt, device=device)
