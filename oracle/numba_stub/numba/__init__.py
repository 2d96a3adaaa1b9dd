"""No-op stand-in for ``numba`` — TEST INFRASTRUCTURE ONLY (lives under oracle/).

The pydata/sparse reference decorates its kernels with ``@numba.jit``; numba is not
installed in the build container.  Because those kernel bodies are plain Python/NumPy,
making ``jit`` return the function unchanged lets the reference run (slowly) in the
interpreter with *identical* integer results and accumulation order.  This package is
put on ``sys.path`` only by ``oracle/ref_loader.py`` (golden generation / oracle
pinning in the build container).  It is never imported by ``sparse_amd``.
"""
from . import types, typed  # noqa: F401
from . import np  # noqa: F401

__version__ = "0.0.stub"


def _passthrough(*args, **kwargs):
    # @jit            -> args == (fn,)
    # @jit(nopython=) -> returns decorator
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit = vectorize = guvectorize = generated_jit = _passthrough


def literal_unroll(x):
    return x


def from_dtype(dt):
    return dt


def prange(*a):
    return range(*a)
