"""Container-only launch plumbing used by tests/golden/make_goldens.py.

TEST INFRASTRUCTURE, never shipped on a product path.  The reference kernels
are `@cuda.jit` Python functions.  This module runs such a function's
*unmodified Python body* once per thread index of the launch
(`kernel[blocks, threads](*args)`), sequentially under CPython, so every
arithmetic statement executed is the reference's own source
(SURVEY.md section 8(c), Appendix B).  It contains no solver arithmetic.

Supported surface = exactly what the hot-path files touch:
`jit` (bare and `device=True`), `grid(1|2|3)`, `local.array`, `atomic.add/min`, `synchronize`.
"""
import itertools

import numpy as _np

_tid = None  # thread index of the body currently being executed
# Some notebook kernels store to `dv[x,y,z]` BEFORE their bounds check, for every thread of a launch
# grid that is rounded up to multiples of 8 -- an out-of-bounds store on the GPU (numba does not check),
# which never lands inside the logical array.  With ignore_oob set, such a thread is a no-op here.
ignore_oob = False


class _Kernel:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *args):
        # a @cuda.jit function called from inside another kernel's body is a plain device-function call
        # (solver/sdf3D.py decorates its helpers with a bare @cuda.jit)
        return self.fn(*args)

    def __getitem__(self, cfg):
        blocks, threads = cfg[0], cfg[1]
        if not isinstance(blocks, (tuple, list)):
            blocks = (blocks,)
        if not isinstance(threads, (tuple, list)):
            threads = (threads,)
        extent = tuple(int(b) * int(t) for b, t in zip(blocks, threads))

        def launch(*args):
            global _tid
            fn = self.fn
            for idx in itertools.product(*(range(e) for e in extent)):
                _tid = idx
                if ignore_oob:
                    try:
                        fn(*args)
                    except IndexError:
                        pass
                else:
                    fn(*args)
            _tid = None

        return launch


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _Kernel(args[0])
    if kwargs.get("device", False):
        return lambda fn: fn          # device function == plain call
    return lambda fn: _Kernel(fn)


def grid(ndim):
    assert _tid is not None and len(_tid) == ndim
    return _tid if ndim > 1 else _tid[0]


class _Local:
    @staticmethod
    def array(n, dtype=_np.float64):
        # the shim's array type, so that float32 elements read back with numba's typing of mixed
        # expressions (refshim/cupy.py:_f32) -- the notebook's particle kernels keep float32 locals
        import cupy as _cp
        return _cp.zeros(n, dtype=dtype)


local = _Local()


class _Atomic:
    """cuda.atomic.add / min on a sequential launcher: a plain read-modify-write."""

    @staticmethod
    def add(arr, idx, val):
        old = arr[idx]
        arr[idx] = old + val
        return old

    @staticmethod
    def min(arr, idx, val):
        old = arr[idx]
        if val < old:
            arr[idx] = val
        return old


atomic = _Atomic()


def synchronize():
    pass
