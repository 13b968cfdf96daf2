"""Container-only array plumbing used by tests/golden/make_goldens.py.

TEST INFRASTRUCTURE, never shipped on a product path.  The reference solver
files (`/root/reference/solver/*.py`) are written against `cupy` arrays, which
are not installable in the build container.  SURVEY.md section 8(c) pins the
oracle by executing the reference's own kernel bodies and CG loops under
CPython with the array container supplied by numpy.  This module supplies that
container: the numpy namespace, plus an ndarray subclass that has the one
cupy-only method the path calls (`.get()`, e.g. `gres.get()` at
solver/CGSolverBuffer.py:5).  It contains no solver arithmetic.
"""
import numpy as _np
from numpy import *  # noqa: F401,F403  (cupy mirrors the numpy namespace)

float64 = _np.float64
float32 = _np.float32
int64 = _np.int64
int32 = _np.int32
bool_ = _np.bool_


class _f32(_np.float32):
    """float32 scalar read out of an array, with numba's typing of mixed expressions: a Python
    float literal is a float64 (`0.0 + f32 -> f64`), whereas numpy >= 2 (NEP 50) treats Python
    scalars as weak (`0.0 + f32 -> f32`).  f32 (op) f32 stays f32, as in numba.  Container-only
    plumbing: no solver arithmetic."""

    def _wide(self, other):
        return isinstance(other, (float, int)) and not isinstance(other, (_np.generic, bool))

    def __add__(self, o):
        return _np.float64(self) + o if self._wide(o) else _wrap_scalar(_np.float32.__add__(self, o))

    def __radd__(self, o):
        return o + _np.float64(self) if self._wide(o) else _wrap_scalar(_np.float32.__radd__(self, o))

    def __sub__(self, o):
        return _np.float64(self) - o if self._wide(o) else _wrap_scalar(_np.float32.__sub__(self, o))

    def __rsub__(self, o):
        return o - _np.float64(self) if self._wide(o) else _wrap_scalar(_np.float32.__rsub__(self, o))

    def __mul__(self, o):
        return _np.float64(self) * o if self._wide(o) else _wrap_scalar(_np.float32.__mul__(self, o))

    def __rmul__(self, o):
        return o * _np.float64(self) if self._wide(o) else _wrap_scalar(_np.float32.__rmul__(self, o))

    def __truediv__(self, o):
        return _np.float64(self) / o if self._wide(o) else _wrap_scalar(_np.float32.__truediv__(self, o))

    def __rtruediv__(self, o):
        return o / _np.float64(self) if self._wide(o) else _wrap_scalar(_np.float32.__rtruediv__(self, o))


def _wrap_scalar(v):
    return _f32(v) if type(v) is _np.float32 else v


class ndarray(_np.ndarray):
    """numpy array with cupy's device->host accessor."""

    def get(self):
        return _np.asarray(self)

    def item(self, *a):
        return _np.asarray(self).item(*a)

    def __getitem__(self, idx):
        v = _np.ndarray.__getitem__(self, idx)
        return _f32(v) if type(v) is _np.float32 else v


def _wrap(a):
    return _np.asarray(a).view(ndarray)


def array(obj, dtype=None, copy=True):
    return _wrap(_np.array(obj, dtype=dtype, copy=copy))


def asarray(obj, dtype=None):
    return _wrap(_np.asarray(obj, dtype=dtype))


def _shape(shape):
    if isinstance(shape, _np.ndarray):
        return tuple(int(s) for s in shape.reshape(-1))
    if isinstance(shape, (tuple, list)):
        return tuple(int(s) for s in shape)
    return int(shape)


def zeros(shape, dtype=float64):
    return _wrap(_np.zeros(_shape(shape), dtype=dtype))


def ones(shape, dtype=float64):
    return _wrap(_np.ones(_shape(shape), dtype=dtype))


def zeros_like(a, dtype=None):
    return _wrap(_np.zeros_like(_np.asarray(a), dtype=dtype))


def sum(a, *args, **kw):  # noqa: A001
    return _wrap(_np.sum(_np.asarray(a), *args, **kw))


def prod(a, *args, **kw):
    return _wrap(_np.prod(_np.asarray(a), *args, **kw))
