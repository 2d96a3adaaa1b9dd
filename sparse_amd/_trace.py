"""Device evaluation of plain Python callables in `elemwise` (the general path of `_umath._elemwise_general`).

The reference applies `func` to NumPy arrays on the host (`_Elemwise`, sparse/numba_backend/_umath.py:602-633).  An arbitrary
callable cannot run on the GPU, but the common ones - lambdas over + - * /, comparisons, `&` `|`, `np.maximum / minimum`,
`abs`, `np.where`, `astype` - are just small expression graphs over exactly-rounded operations.  `trace` calls `func` ONCE with
symbolic operands that record that graph; every node carries a one-element NumPy "dummy" on which the very same NumPy
operation was applied, so the result dtype of each step is NumPy's own (NEP 50 scalar rules included), not a re-implementation
of its promotion table.  `run` then replays the graph with the library's elementwise kernels on the device arrays: no
nnz-sized host round trip.

Only operations whose device form is bit-identical to NumPy's are traced (IEEE add / subtract / multiply / divide without
contraction, comparisons, min / max with NumPy's NaN rule, logical and bit-wise operations, negative / absolute, `where`,
casts between float32 / float64 / int32 / int64 / bool, `x ** 2` as `x * x`).  Anything else - transcendental functions (the
device versions are within 2 ulp, not identical), keywords, data-dependent control flow (`bool(x)`), other dtypes - raises
`Untraceable` inside the trace and the caller takes the host path, exactly as before.
"""
import numpy as np
import torch
from numpy.lib.mixins import NDArrayOperatorsMixin

from . import _ffi
from . import _kernels as K
from ._device import ptr, require_hip, stream_ptr, torch_dtype

_ARITH = {"add", "subtract", "multiply", "true_divide", "divide", "maximum", "minimum", "fmax", "fmin",
          "floor_divide", "remainder", "fmod", "copysign"}      # (the last four: NumPy's own rules in the kernel, bit for bit)
_TO_BOOL = {"greater", "greater_equal", "less", "less_equal", "equal", "not_equal", "logical_and", "logical_or", "logical_xor"}
_BITWISE = {"bitwise_and", "bitwise_or", "bitwise_xor", "left_shift", "right_shift"}
_UNARY = {"negative": "negative", "absolute": "absolute", "fabs": "absolute", "positive": "positive",
          "logical_not": "logical_not", "square": None}
_DTYPES = {np.dtype("float32"), np.dtype("float64"), np.dtype("int32"), np.dtype("int64"), np.dtype("bool")}


class Untraceable(Exception):
    """`func` does something the device graph does not cover: the caller evaluates it on the host."""


class _Node:
    __slots__ = ("op", "args", "dummy")

    def __init__(self, op, args, dummy):
        dummy = np.asarray(dummy)
        if dummy.dtype not in _DTYPES:
            raise Untraceable(f"dtype {dummy.dtype}")
        self.op, self.args, self.dummy = op, args, dummy.reshape(-1)[:1].copy() if dummy.size else dummy


def _dummy_of(v):
    return v.node.dummy if isinstance(v, Sym) else v


def _is_scalar(v):
    return isinstance(v, (bool, int, float, np.generic)) or (isinstance(v, np.ndarray) and v.ndim == 0)


class Sym(NDArrayOperatorsMixin):
    """Symbolic operand: operators and NumPy ufuncs applied to it extend the graph."""

    __array_priority__ = 1000

    def __init__(self, node):
        self.node = node

    dtype = property(lambda self: self.node.dummy.dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs:
            raise Untraceable(f"{ufunc.__name__}.{method} with {sorted(kwargs)}")
        name = ufunc.__name__
        for v in inputs:
            if not (isinstance(v, Sym) or _is_scalar(v)):
                raise Untraceable(f"operand of type {type(v).__name__}")
        with np.errstate(all="ignore"):
            dummy = ufunc(*[_dummy_of(v) for v in inputs])
        if name == "power" and len(inputs) == 2 and _is_scalar(inputs[1]) and not isinstance(inputs[1], (bool, np.bool_)) \
                and inputs[1] == 2 and isinstance(inputs[0], Sym):
            return Sym(_Node("multiply", (inputs[0], inputs[0]), dummy))      # x ** 2 is x * x, exactly
        if name in _ARITH or name in _TO_BOOL or name in _BITWISE:
            if len(inputs) != 2:
                raise Untraceable(name)
            return Sym(_Node("divide" if name == "true_divide" else name, tuple(inputs), dummy))
        if name in _UNARY and len(inputs) == 1:
            if name == "square":
                return Sym(_Node("multiply", (inputs[0], inputs[0]), dummy))
            return Sym(_Node(_UNARY[name], tuple(inputs), dummy))
        raise Untraceable(name)

    def __array_function__(self, func, types, args, kwargs):
        if func is np.where and len(args) == 3 and not kwargs:
            for v in args:
                if not (isinstance(v, Sym) or _is_scalar(v)):
                    raise Untraceable("where operand")
            with np.errstate(all="ignore"):
                dummy = np.where(*[_dummy_of(v) for v in args])
            return Sym(_Node("where", tuple(args), dummy))
        raise Untraceable(getattr(func, "__name__", str(func)))

    def astype(self, dtype, **kwargs):
        if kwargs:
            raise Untraceable("astype keywords")
        return Sym(_Node("astype", (self,), self.node.dummy.astype(dtype)))

    def __bool__(self):
        raise Untraceable("data-dependent control flow")

    def __len__(self):
        raise Untraceable("len()")

    def __getitem__(self, index):
        raise Untraceable("indexing")

    def __iter__(self):
        raise Untraceable("iteration")


def build(func, args_spec):
    """args_spec: per positional argument either ("array", np dtype) or ("scalar", value).  -> root Sym, or None."""
    syms = []
    for i, (kind, x) in enumerate(args_spec):
        if kind == "array":
            if np.dtype(x) not in _DTYPES:
                return None
            syms.append(Sym(_Node("leaf", (i,), np.zeros(1, dtype=x))))
        else:
            if not _is_scalar(x):
                return None
            syms.append(x)
    try:
        root = func(*syms)
    except Untraceable:
        return None
    except Exception:   # noqa: BLE001 - whatever else goes wrong with symbolic operands: the host path decides
        return None
    return root if isinstance(root, Sym) else None


def _scalar_tensor(value, np_dtype, devi):
    # A Python / NumPy integer that the compute dtype cannot hold must not wrap around on its way to the device (NumPy 2
    # compares `int32_array < 2**40` exactly; int32(2**40) is 0): such a graph goes back to the host path.
    dt = np.dtype(np_dtype)
    if dt.kind in "iu" and isinstance(value, (int, np.integer)) and not isinstance(value, (bool, np.bool_)):
        info = np.iinfo(dt)
        if not (info.min <= int(value) <= info.max):
            raise Untraceable(f"integer scalar {value} does not fit {dt}")
    with np.errstate(all="ignore"):
        v = np.asarray(value).astype(np_dtype)
    t = torch.from_numpy(v.reshape(1).copy())
    return t.to(devi)


def _as(t, np_dtype):
    return K.convert(t, torch_dtype(np_dtype))


def run(root, arrays, n, devi):
    """Replay the graph: `arrays[i]` = device tensor (n elements) of positional argument i (only array arguments are
    looked up).  Returns the device tensor of the root (n elements, root dtype)."""
    from ._umath import _BIN, _UN, binary_arrays, unary_array

    memo = {}

    def value(v, want):
        """(tensor, is_scalar) of an operand in NumPy dtype `want`"""
        if isinstance(v, Sym):
            return _as(ev(v.node), want), False
        return _scalar_tensor(v, want, devi), True

    def ev(node):
        key = id(node)
        if key in memo:
            return memo[key]
        op, out_dt = node.op, node.dummy.dtype
        if op == "leaf":
            r = arrays[node.args[0]]
        elif op == "astype":
            r = _as(ev(node.args[0].node), out_dt)
        elif op == "where":
            c, a, b = node.args
            mask, m_sc = value(c, np.dtype(bool))
            if m_sc:
                mask = mask.expand(n).contiguous()
            ta, sa = value(a, out_dt)
            tb, sb = value(b, out_dt)
            r = torch.empty(n, dtype=torch_dtype(out_dt), device=devi)
            _ffi.call("spamd_ewise_select", r.element_size(), n, ptr(mask.contiguous().view(torch.uint8)), ptr(ta.contiguous()),
                      int(sa), ptr(tb.contiguous()), int(sb), ptr(r), stream_ptr(devi))
        elif op in ("negative", "absolute", "positive", "logical_not"):
            (a,) = node.args
            src_dt = _dummy_of(a).dtype if op == "logical_not" else out_dt
            t, _ = value(a, src_dt)
            if op == "logical_not" and t.dtype != torch.bool:
                t = _as(t, np.dtype(bool))
            if op != "positive" and op not in _UN:
                raise Untraceable(f"no device kernel for {op}")
            r = t if op == "positive" else unary_array(op, t)
        else:   # binary
            a, b = node.args
            if op in _TO_BOOL:
                with np.errstate(all="ignore"):
                    comp = np.result_type(_dummy_of(a), _dummy_of(b))
                if op in ("logical_and", "logical_or", "logical_xor"):
                    comp = np.dtype(bool) if comp not in _DTYPES else comp
            else:
                comp = out_dt
            if comp not in _DTYPES:
                raise Untraceable(f"compute dtype {comp}")
            ta, sa = value(a, comp)
            tb, sb = value(b, comp)
            if sa and sb:      # two scalars: NumPy already folded them into the dummy
                r = _scalar_tensor(node.dummy[0], out_dt, devi).expand(n).contiguous()
            else:
                name = op
                if comp == np.dtype(bool) and op in ("add", "maximum", "fmax"):
                    name = "logical_or"           # NumPy's boolean arithmetic is logical
                elif comp == np.dtype(bool) and op in ("multiply", "minimum", "fmin"):
                    name = "logical_and"
                elif comp == np.dtype(bool) and op in ("subtract", "divide"):
                    raise Untraceable("boolean subtract / divide")
                if name not in _BIN:
                    raise Untraceable(f"no device kernel for {name}")
                r = binary_arrays(name, ta, tb, a_scalar=sa, b_scalar=sb)
                if r.dtype == torch.uint8 and out_dt == np.dtype(bool):
                    r = r.view(torch.bool)
                elif r.dtype != torch_dtype(out_dt):
                    r = _as(r, out_dt)
        if r.numel() != n:     # a scalar leaf that reached the root
            r = r.expand(n).contiguous()
        memo[key] = r
        return r

    require_hip(*[t for t in arrays if isinstance(t, torch.Tensor)])
    return ev(root.node)
