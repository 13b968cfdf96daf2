"""Container-only stand-in namespace so `from numba import cuda` resolves to
tests/golden/refshim/numba/cuda.py (see that file).  Test infrastructure only."""
from . import cuda  # noqa: F401
