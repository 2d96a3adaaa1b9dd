"""Cases of the general elementwise / reduction / N-D matmul surface, written once and
evaluated twice: by oracle/gen_golden.py against the REAL reference (-> tests/golden/general.npz) and by
tests/test_general_gpu.py against sparse_amd.  A case is `(name, fn)` with `fn(sp, np_inputs) -> result`; the result may
be a sparse array, an ndarray, or an exception (recorded by type)."""
import numpy as np

SHAPE = (5, 6, 7)


def inputs():
    def arr(seed, density, shape=SHAPE, lo=-0.4):
        r = np.random.default_rng(seed)
        d = np.zeros(shape)
        m = r.random(shape) < density
        d[m] = r.random(int(m.sum())) + lo
        return d

    x, y, z = arr(11, 0.4), arr(12, 0.5), arr(13, 0.3)
    xn = arr(14, 0.5)
    xn[np.random.default_rng(15).random(SHAPE) < 0.15] = np.nan
    return dict(x=x, y=y, z=z, xn=xn, yb=arr(16, 0.6, (6, 1)), zb=arr(17, 0.7, (7,)), dense=arr(18, 1.0, SHAPE, lo=0.5),
                dense_row=arr(19, 1.0, (7,), lo=0.5), big=arr(20, 1.0, (2,) + SHAPE),
                a4=arr(21, 0.5, (3, 1, 4, 5)), b4=arr(22, 0.5, (1, 2, 5, 6)), a3=arr(23, 0.5, (1, 4, 5)), b3=arr(24, 0.5, (3, 5, 6)),
                c3=arr(25, 0.5, (3, 4, 5)), d3=arr(26, 0.5, (1, 5, 6)), dn=arr(27, 1.0, (3, 4, 5)), d4=arr(28, 1.0, (1, 2, 5, 6)),
                xi=(arr(29, 0.5, lo=0.1) * 100).astype(np.int64), xc=arr(30, 0.4) + 1j * arr(31, 0.4),
                o1=arr(32, 0.5, (3, 4)), o2=arr(33, 0.6, (5,)))


def _coo(sp, a):
    return sp.COO.from_numpy(a)


def _fma3(a, b, c):
    return a * b + c


def _mix4(a, b, c, d):
    return np.maximum(a, b) - c * d


def _cmp(a, b):
    return (a > b) & (a != 0)


def _sin2(a):
    return np.sin(a) ** 2


CASES = [
    ("ternary a*b+c", lambda sp, i: sp.elemwise(_fma3, _coo(sp, i["x"]), _coo(sp, i["y"]), _coo(sp, i["z"]))),
    ("four operands, one scalar", lambda sp, i: sp.elemwise(_mix4, _coo(sp, i["x"]), _coo(sp, i["y"]), _coo(sp, i["z"]), 2.0)),
    ("clip with scalar bounds", lambda sp, i: sp.elemwise(np.clip, _coo(sp, i["x"]), -0.1, 0.2)),
    ("clip with sparse bounds", lambda sp, i: sp.elemwise(np.clip, _coo(sp, i["x"]), _coo(sp, i["y"]) - 1, _coo(sp, i["z"]) + 1)),
    ("add with dtype keyword", lambda sp, i: sp.elemwise(np.add, _coo(sp, i["x"]), _coo(sp, i["y"]), dtype=np.float32)),
    ("round with decimals keyword", lambda sp, i: sp.elemwise(np.round, _coo(sp, i["x"]) * 10, decimals=1)),
    ("three operands broadcast", lambda sp, i: sp.elemwise(_fma3, _coo(sp, i["x"]), _coo(sp, i["yb"]), _coo(sp, i["zb"]))),
    ("non-zero fills through a lambda", lambda sp, i: sp.elemwise(lambda a, b: a * b + 3, _coo(sp, i["x"]) + 1, _coo(sp, i["y"]) - 2)),
    ("comparison lambda", lambda sp, i: sp.elemwise(_cmp, _coo(sp, i["x"]), _coo(sp, i["y"]))),
    ("unary lambda", lambda sp, i: sp.elemwise(_sin2, _coo(sp, i["x"]))),
    ("nan_to_num keyword", lambda sp, i: sp.elemwise(np.nan_to_num, _coo(sp, i["xn"]), nan=7.0)),
    ("gcxs operands stay gcxs", lambda sp, i: sp.elemwise(_fma3, sp.GCXS.from_numpy(i["x"]), sp.GCXS.from_numpy(i["y"]), 0.5)),
    ("where with a dense x", lambda sp, i: sp.where(_coo(sp, i["x"]) > 0, i["dense"], _coo(sp, i["y"]))),
    ("where with a dense y: dense result", lambda sp, i: sp.where(_coo(sp, i["x"]) > 0, _coo(sp, i["y"]), i["dense"])),
    ("multiply by a broadcast dense row", lambda sp, i: sp.elemwise(np.multiply, _coo(sp, i["x"]), i["dense_row"])),
    ("lambda with a dense operand", lambda sp, i: sp.elemwise(lambda a, b, c: a * b * c, _coo(sp, i["x"]), i["dense"], _coo(sp, i["y"]))),
    ("densifying mixed operation", lambda sp, i: sp.elemwise(np.add, _coo(sp, i["x"]), i["big"])),
    ("not broadcastable", lambda sp, i: sp.elemwise(_fma3, _coo(sp, i["x"]), _coo(sp, i["yb"].T), 1.0)),
    ("var with a fill value", lambda sp, i: (_coo(sp, i["x"]) + 1).var(axis=1)),
    ("std with a fill value, ddof", lambda sp, i: (_coo(sp, i["x"]) + 0.5).std(axis=(0, 2), ddof=1)),
    ("var of everything, fill value", lambda sp, i: (_coo(sp, i["y"]) - 2).var()),
    ("matmul (3,1,4,5) @ (1,2,5,6) sparse", lambda sp, i: sp.matmul(_coo(sp, i["a4"]), _coo(sp, i["b4"]))),
    ("matmul (3,1,4,5) @ dense (1,2,5,6)", lambda sp, i: sp.matmul(_coo(sp, i["a4"]), i["d4"])),
    ("matmul (1,4,5) @ (3,5,6) sparse", lambda sp, i: sp.matmul(_coo(sp, i["a3"]), _coo(sp, i["b3"]))),
    ("matmul (3,4,5) @ (1,5,6) gcxs", lambda sp, i: sp.matmul(sp.GCXS.from_numpy(i["c3"]), sp.GCXS.from_numpy(i["d3"]))),
    ("matmul dense (3,4,5) @ sparse (1,5,6)", lambda sp, i: sp.matmul(i["dn"], _coo(sp, i["d3"]))),
    ("matmul dense (3,4,5) @ sparse (3,5,6)", lambda sp, i: sp.matmul(i["dn"], _coo(sp, i["b3"]))),
    ("matmul (3,4,5) @ dense (5,6)", lambda sp, i: sp.matmul(_coo(sp, i["c3"]), i["d3"][0])),
    # round 3: the protocol surface (`ufunc.outer`, `out=`) and reductions beyond the device kernels' table
    ("subtract.outer of two COO", lambda sp, i: np.subtract.outer(_coo(sp, i["o1"]), _coo(sp, i["o2"]))),
    ("multiply.outer, second operand first", lambda sp, i: np.multiply.outer(_coo(sp, i["o2"]), _coo(sp, i["o1"]))),
    ("bitwise_or.reduce", lambda sp, i: np.bitwise_or.reduce(_coo(sp, i["xi"]), axis=1)),
    ("bitwise_xor.reduce over two axes", lambda sp, i: np.bitwise_xor.reduce(_coo(sp, i["xi"]), axis=(0, 2))),
    ("hypot.reduce", lambda sp, i: np.hypot.reduce(_coo(sp, i["x"]), axis=2)),
    ("hypot.reduce gcxs, keepdims", lambda sp, i: np.hypot.reduce(sp.GCXS.from_numpy(i["y"]), axis=0, keepdims=True)),
    ("complex sum over an axis", lambda sp, i: _coo(sp, i["xc"]).sum(axis=0)),
    ("complex sum of everything", lambda sp, i: _coo(sp, i["xc"]).sum()),
    ("complex prod with a fill value", lambda sp, i: (_coo(sp, i["xc"]) + (1 + 0.5j)).prod(axis=(1, 2))),
    ("int16 max", lambda sp, i: _coo(sp, i["xi"].astype(np.int16)).max(axis=1)),
    ("illegal out= cast", lambda sp, i: np.add(_coo(sp, i["x"]), _coo(sp, i["y"]), out=_coo(sp, i["xi"]))),
    ("legal out=", lambda sp, i: np.add(_coo(sp, i["x"]), _coo(sp, i["y"]), out=_coo(sp, i["z"]))),
    ("out= of the wrong shape", lambda sp, i: np.add(_coo(sp, i["x"]), _coo(sp, i["y"]), out=_coo(sp, i["o1"]))),
    ("reduce with out= of another dtype", lambda sp, i: np.add.reduce(_coo(sp, i["x"]), axis=0, out=_coo(sp, i["xi"][0]))),
]


def evaluate(sp, fn, inp):
    """-> dict(kind, dense, nnz, fill, cls, dtype) | dict(kind='error', error=<type name>)"""
    import warnings

    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = fn(sp, inp)
    except Exception as e:  # noqa: BLE001 - the exception type IS the recorded result
        return dict(kind="error", error=type(e).__name__)
    if hasattr(r, "todense") and hasattr(r, "nnz"):
        return dict(kind="sparse", dense=np.asarray(r.todense()), nnz=int(r.nnz), fill=np.asarray(r.fill_value),
                    cls=type(r).__name__)
    if hasattr(r, "cpu"):
        r = r.cpu().numpy()
    return dict(kind="dense", dense=np.asarray(r))
