"""Drop-in for the reference's solver/sdf3D.py on MI355X: rigid-body scene
description (`generate_rb`, `transform_rb`, `set_vel_rb`, `get_T`, `get_R` -- host side, same packed
(n, 10, 4) float64 layout) and the two kernels the notebook calls: `evaluate` (signed distance +
body velocity at a set of points; ipynb scene set-up) and `project` (push particles out of / into
the bodies; ipynb:4584, every step).  PyTorch-ROCm tensors, HIP kernels behind the C ABI; no CPU path.
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from mfs import _lib, tensors as T


def get_T(position):
    """4x4 translation matrix (reference :264-267), float64 numpy."""
    t = np.identity(4)
    t[0:3, 3] = np.asarray(T.as_f64_list(position, 3))
    return t


def get_R(axis, angle):
    """4x4 rotation matrix about `axis` by `angle` degrees (reference :269-274), float64 numpy."""
    r = np.identity(4)
    if angle:
        ax = np.asarray(T.as_f64_list(axis, 3))
        r[:3, :3] = R.from_rotvec(ax / np.linalg.norm(ax) * angle * np.pi / 180).as_matrix()
    return r


def _empty(device):
    return torch.zeros((0, 10, 4), dtype=torch.float64, device=device)


def generate_rb(rb_d, rb_map, name, rbparam, flip=False, center=[0, 0, 0], axis=[0, 1, 0], angle=0, device=None):
    """Append one body (reference :277-305).  rbparam: ['sphere', radius] | ['box', sx, sy, sz] |
    ['cylinder', radius, height].  Row 0 = [type code (+1 if flipped), parameters], rows 1-4 translation,
    rows 5-8 rotation, row 9 velocity.  `rb_d` may be None / empty for the first body.  Returns (rb_d, rb_map)."""
    if rb_d is None:
        rb_d = _empty(torch.device("cuda" if device is None else device))
    rb = np.zeros((1, 10, 4))
    if rbparam[0] == 'sphere':
        rb[:, 0, 0] = 1 if flip else 0
        rb[:, 0, 1] = rbparam[1]
    elif rbparam[0] == 'box':
        rb[:, 0, 0] = 3 if flip else 2
        rb[:, 0, 1:] = np.asarray(rbparam[1:], dtype=np.float64)
    elif rbparam[0] == 'cylinder':
        rb[:, 0, 0] = 5 if flip else 4
        rb[:, 0, 1:3] = np.asarray(rbparam[1:], dtype=np.float64)
    else:
        return rb_d
    rb[:, 1:5, :] = get_T(center)
    rb[:, 5:9, :] = get_R(axis, angle)
    index = rb_d.shape[0]
    rb_map[name] = index
    rbt = torch.as_tensor(rb, dtype=torch.float64, device=rb_d.device)
    rb_d = rbt if index == 0 else torch.cat([rb_d, rbt], dim=0)
    return rb_d, rb_map


def transform_rb(rb_d, index, center=None, axis=None, angle=None):
    """reference :307-311"""
    if center:
        rb_d[index, 1:5, :] = torch.as_tensor(get_T(center), dtype=rb_d.dtype, device=rb_d.device)
    if axis and angle:
        rb_d[index, 5:9, :] = torch.as_tensor(get_R(axis, angle), dtype=rb_d.dtype, device=rb_d.device)


def set_vel_rb(rb_d, index, vel):
    """reference :313-314"""
    rb_d[index, -1, :3] = torch.as_tensor(np.asarray(T.as_f64_list(vel, 3)), dtype=rb_d.dtype, device=rb_d.device)


def _bodies(rb_d):
    rb_d = T.dev(rb_d, "rb_d")
    if rb_d.dim() != 3 or tuple(rb_d.shape[1:]) != (10, 4) or rb_d.dtype != torch.float64:
        raise ValueError("rb_d: expected a float64 tensor of shape (n, 10, 4)")
    return rb_d


def evaluate(rb_d, sd, vel, position):
    """sd[...] = min over bodies of the signed distance at position[..., :]; vel[..., :] = velocity of the
    closest body where sd <= 0, else 0 (reference :260-270 -> kernel :218-239)."""
    rb_d = _bodies(rb_d)
    position, sd, vel = T.dev(position, "position"), T.dev(sd, "sd"), T.dev(vel, "vel")
    assert tuple(sd.shape) == tuple(position.shape[:-1])
    assert position.shape[-1] == 3
    assert vel.shape[-1] == 3
    vel *= 0
    n = int(sd.numel())
    lib = _lib.load()
    _lib.check(lib.mfs_sdf_evaluate3d(T.ptr(rb_d), int(rb_d.shape[0]), T.ptr(position), T.code(position), n, T.ptr(sd),
                                      T.code(sd), T.ptr(vel), T.code(vel), T.stream()), "mfs_sdf_evaluate3d")


def project(rb_d, position):
    """In place: every body in turn moves the points it owns to its surface / into itself
    (reference :272-278 -> kernel :241-258)."""
    rb_d = _bodies(rb_d)
    position = T.dev(position, "position")
    assert position.shape[-1] == 3
    lib = _lib.load()
    _lib.check(lib.mfs_sdf_project3d(T.ptr(rb_d), int(rb_d.shape[0]), T.ptr(position), T.code(position),
                                     int(position.numel() // 3), T.stream()), "mfs_sdf_project3d")
