"""Every NumPy ufunc (and the array methods built on them) a user of the reference can call on a sparse array, at 10^7 stored
elements: ms per call and whether the call left the device (`fallback_stats`).  Looking for host evaluations."""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp

g = torch.Generator(device="cuda").manual_seed(1)
lin = torch.unique(torch.randint(0, 10 ** 9, (10_000_000,), device="cuda", generator=g))
x = sp.COO._from_sorted_keys(lin, torch.rand(lin.numel(), device="cuda", dtype=torch.float64) + 0.1, (1000, 1000, 1000), 0.0, torch.int64)
lin2 = torch.unique(torch.randint(0, 10 ** 9, (10_000_000,), device="cuda", generator=g))
y = sp.COO._from_sorted_keys(lin2, torch.rand(lin2.numel(), device="cuda", dtype=torch.float64) + 0.1, (1000, 1000, 1000), 0.0, torch.int64)
xi = (x * 100).astype(np.int64)
yi = (y * 100).astype(np.int64)


def t(f, reps=3):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


unary = ["negative", "positive", "absolute", "fabs", "rint", "sign", "conj", "exp2", "expm1", "log1p", "sqrt", "square", "cbrt", "sin",
         "tan", "arcsin", "arctan", "sinh", "tanh", "arcsinh", "arctanh", "deg2rad", "rad2deg", "floor", "ceil", "trunc", "isfinite",
         "isinf", "isnan", "signbit", "logical_not", "spacing", "reciprocal", "invert", "real", "imag", "angle", "nan_to_num", "modf", "frexp"]
binary = ["add", "subtract", "multiply", "divide", "true_divide", "floor_divide", "power", "float_power", "remainder", "mod", "fmod",
          "maximum", "minimum", "fmax", "fmin", "hypot", "arctan2", "copysign", "nextafter", "logaddexp", "heaviside", "greater", "less",
          "equal", "not_equal", "logical_and", "logical_or", "logical_xor", "isclose", "ldexp"]
ibinary = ["bitwise_and", "bitwise_or", "bitwise_xor", "left_shift", "right_shift", "gcd", "lcm", "floor_divide", "remainder"]
sp.fallback_stats(reset=True)
cases = []
for n in unary:
    fn = getattr(np, n, None)
    if fn is None:
        continue
    cases.append((n + "(x)", (lambda fn=fn: fn(xi if fn is np.invert else x))))
for n in binary:
    fn = getattr(np, n)
    cases.append((n + "(x,y)", (lambda fn=fn: fn(x, y if fn is not np.ldexp else yi))))
    cases.append((n + "(x,2.5)", (lambda fn=fn: fn(x, 2.5 if fn is not np.ldexp else 2))))
for n in ibinary:
    fn = getattr(np, n)
    cases.append((n + "(xi,yi)", (lambda fn=fn: fn(xi, yi))))
    cases.append((n + "(xi,3)", (lambda fn=fn: fn(xi, 3))))
cases += [("x**3", lambda: x ** 3), ("x**0.5", lambda: x ** 0.5), ("2**x", lambda: 2 ** x), ("x//2", lambda: x // 2), ("x%3", lambda: x % 3),
          ("divmod(x,3)", lambda: divmod(x, 3)), ("abs(x)", lambda: abs(x)), ("x.conj()", lambda: x.conj()), ("x.real", lambda: x.real),
          ("x.imag", lambda: x.imag), ("x.astype(f32)", lambda: x.astype(np.float32)), ("xi.astype(f64)", lambda: xi.astype(np.float64)),
          ("x.astype(bool)", lambda: x.astype(bool)), ("isclose kw", lambda: np.isclose(x, y, rtol=1e-3)), ("x.round(2)", lambda: x.round(2)),
          ("x.clip(.2,.8)", lambda: x.clip(0.2, 0.8)), ("np.round(x,2)", lambda: np.round(x, 2)), ("np.clip(x,.2,.8)", lambda: np.clip(x, 0.2, 0.8)),
          ("np.around(x,2)", lambda: np.around(x, 2)), ("np.nan_to_num(x)", lambda: np.nan_to_num(x)), ("np.sinc?", lambda: np.sinc(x)),
          ("np.square(x)+1", lambda: np.square(x) + 1), ("np.where(x>.5,x,y)", lambda: np.where(x > 0.5, x, y))]
for name, f in cases:
    try:
        before = {k: v for k, v in sp.fallback_stats().items() if k != "recent"}
        ms = t(f)
        after = {k: v for k, v in sp.fallback_stats().items() if k != "recent"}
        fb = {k: after[k] - before.get(k, 0) for k in after if after[k] != before.get(k, 0)}
        if fb or ms > 3.0:
            print(f"{name:24s} {ms:9.2f} ms {('  HOST ' + str(fb)) if fb else ''}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:24s} {type(e).__name__}: {str(e)[:100]}", flush=True)
print("done", len(cases), "cases")
