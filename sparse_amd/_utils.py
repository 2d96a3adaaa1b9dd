"""Host-side argument contracts shared by the containers and `_dot`.

These mirror the reference's Python-level validation, which always runs *before* a kernel is
launched (SURVEY.md §8b "Errors"): same exception types and messages.
"""
import warnings
from collections.abc import Iterable
from numbers import Integral

import numpy as np


def zero_of_dtype(dtype):
    """The zero scalar of `dtype` (reference `_zero_of_dtype`, numba_backend/_utils.py)."""
    return np.zeros((), dtype=dtype)[()]


def normalize_axis(axis, ndim):
    """Negative axes -> positive; ValueError on out-of-range / non-integers
    (reference numba_backend/_utils.py:367-403)."""
    if axis is None:
        return None
    if isinstance(axis, Integral):
        a = int(axis)
        if a < 0:
            a += ndim
        if not 0 <= a < ndim:
            raise ValueError(f"Invalid axis index {int(axis)} for ndim={ndim}")
        return a
    if isinstance(axis, Iterable):
        axis = tuple(axis)
        if not all(isinstance(a, Integral) for a in axis):
            raise ValueError(f"axis {axis} not understood")
        return tuple(normalize_axis(a, ndim) for a in axis)
    raise ValueError(f"axis {axis} not understood")


def equivalent(x, y, loose=False):
    """Scalar/array equivalence (reference numba_backend/_utils.py:406-452).

    Non-float dtypes compare with ==.  `loose=True`: NaN == NaN and -0.0 == 0.0.  Otherwise the
    comparison is BIT-WISE (so -0.0 is not the fill value 0.0 — Appendix C.1)."""
    x = np.asarray(x)
    y = np.asarray(y)
    dt = np.result_type(x.dtype, y.dtype)
    if dt.kind not in "fc":
        return x == y
    if loose:
        if dt.kind == "c":
            return equivalent(x.real, y.real, loose=True) & equivalent(x.imag, y.imag, loose=True)
        return (x == y) | ((x != x) & (y != y))
    xb, yb = np.broadcast_arrays(x.astype(dt)[..., None], y.astype(dt)[..., None])
    return (np.ascontiguousarray(xb).view(np.uint8) == np.ascontiguousarray(yb).view(np.uint8)).all(axis=-1)


def check_zero_fill_value(*args, loose=True):
    """ValueError unless every sparse argument has a zero fill value
    (reference numba_backend/_utils.py:562-596; message is part of the contract)."""
    for i, arg in enumerate(args):
        if getattr(arg, "size", 1) == 0:
            continue
        if hasattr(arg, "fill_value") and not equivalent(arg.fill_value, zero_of_dtype(arg.dtype), loose=loose):
            raise ValueError(
                f"This operation requires zero fill values, but argument {i:d} had a fill value of {arg.fill_value!s}."
            )


def check_fill_value(x, /, *, accept_fv=None):
    """ValueError unless `x.fill_value` is one of `accept_fv` (default: zero only) — the export guard of
    `to_scipy_sparse` (reference numba_backend/_utils.py:537-559; message is part of the contract)."""
    if accept_fv is None:
        accept_fv = [0]
    if not isinstance(accept_fv, Iterable):
        accept_fv = [accept_fv]
    if not any(equivalent(fv, x.fill_value, loose=True) for fv in accept_fv):
        raise ValueError(f"x.fill_value={x.fill_value!r} but should be in {accept_fv}.")


def check_compressed_axes(ndim, compressed_axes):
    """GCXS `compressed_axes` must be a strictly increasing tuple of in-range integer axes that
    leaves at least one axis uncompressed; same ValueError messages as the reference
    (numba_backend/_utils.py:507-534)."""
    if compressed_axes is None:
        return
    if isinstance(ndim, Iterable):
        ndim = len(ndim)
    if not isinstance(compressed_axes, Iterable):
        raise ValueError("compressed_axes must be an iterable")
    axes = tuple(compressed_axes)
    if len(axes) == ndim:
        raise ValueError("cannot compress all axes")
    if any(not isinstance(a, Integral) for a in axes):
        raise ValueError("axes must be represented with integers")
    if any(b <= a for a, b in zip(axes, axes[1:])):
        raise ValueError("axes must be sorted without repeats")
    if axes and not (0 <= axes[0] and axes[-1] < ndim):
        raise ValueError("axis out of range")


def can_store(dtype, scalar):
    """Whether `dtype` can hold `scalar` exactly (reference numba_backend/_utils.py:651-658)."""
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            warnings.filterwarnings("error", "out-of-bound", DeprecationWarning)
            return bool(np.array(scalar, dtype=dtype) == np.array(scalar))
    except (ValueError, OverflowError):
        return False


def convert_format(fmt):
    from ._sparse_array import SparseArray

    if isinstance(fmt, type):
        if not issubclass(fmt, SparseArray):
            raise ValueError(f"Invalid format: {fmt}")
        return fmt.__name__.lower()
    if isinstance(fmt, str):
        return fmt
    raise ValueError(f"Invalid format: {fmt}")


def prod(xs):
    p = 1
    for x in xs:
        p *= int(x)
    return p
