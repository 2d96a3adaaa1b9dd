"""`where`, NaN-skipping reductions and the creation functions, checked against NumPy on
the dense arrays (the way the reference's own tests do: sparse/numba_backend/tests/test_coo.py `test_where`,
`test_nan_reductions`, `test_all_nan_reduction_warning`, `test_nanmean`; test_array_function.py for zeros/ones/eye)."""
import warnings

import numpy as np
import pytest
import torch


def _dense(rng, shape, density, dtype=np.float64, nan_frac=0.0):
    d = np.zeros(shape, dtype=dtype)
    m = rng.random(shape) < density
    d[m] = (rng.random(int(m.sum())) - 0.4).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(1, 9, int(m.sum()))
    if nan_frac:
        d[rng.random(shape) < nan_frac] = np.nan
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(7,), (5, 6), (4, 1, 9), (3, 17, 65)])
def test_where_three_operands(shape):
    import sparse_amd as sp

    rng = np.random.default_rng(11)
    c, x, y = _dense(rng, shape, 0.4) > 0, _dense(rng, shape, 0.5), _dense(rng, shape, 0.3)
    sc, sx, sy = sp.COO.from_numpy(c), sp.COO.from_numpy(x), sp.COO.from_numpy(y)
    r = sp.where(sc, sx, sy)
    assert isinstance(r, sp.COO) and r.fill_value == 0 and np.array_equal(r.todense(), np.where(c, x, y))
    assert r.nnz == np.count_nonzero(np.where(c, x, y))   # pruned, canonical
    # scalars on either side; the result fill value is where(fills)
    r = sp.where(sc, 2.5, sy)
    assert r.fill_value == 0 and np.array_equal(r.todense(), np.where(c, 2.5, y))
    r = sp.where(sc, sx, -1.0)
    assert r.fill_value == -1.0 and np.array_equal(r.todense(), np.where(c, x, -1.0))
    # np.where dispatches here; integer operands keep their dtype
    xi = _dense(rng, shape, 0.5, np.int64)
    r = np.where(sc, sp.COO.from_numpy(xi), 0)
    assert r.dtype == np.int64 and np.array_equal(r.todense(), np.where(c, xi, 0))
    # a float condition counts as "non-zero", NaN included
    cf = _dense(rng, shape, 0.5, nan_frac=0.1)
    assert np.array_equal(sp.where(sp.COO.from_numpy(cf), sx, sy).todense(), np.where(cf, x, y))


@pytest.mark.gpu
def test_where_broadcasts_and_one_argument_form():
    import sparse_amd as sp

    rng = np.random.default_rng(12)
    c, x, y = _dense(rng, (6, 1), 0.5) > 0, _dense(rng, (1, 8), 0.6), _dense(rng, (6, 8), 0.4)
    r = sp.where(sp.COO.from_numpy(c), sp.COO.from_numpy(x), sp.COO.from_numpy(y))
    assert np.array_equal(r.todense(), np.where(c, x, y))
    coords = sp.where(sp.COO.from_numpy(y))
    want = np.where(y)
    assert len(coords) == 2 and all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(coords, want))
    assert np.array_equal(sp.argwhere(sp.COO.from_numpy(y)).cpu().numpy(), np.argwhere(y))
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(sp.nonzero(sp.COO.from_numpy(y)), np.nonzero(y)))
    with pytest.raises(ValueError):
        sp.where(sp.COO.from_numpy(c), sp.COO.from_numpy(x))
    with pytest.raises(ValueError):   # non-zero fill value has no finite set of "true" coordinates
        sp.where(sp.COO.from_numpy(y) + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("axis", [None, 0, 1, (0, 2), -1])
@pytest.mark.parametrize("name", ["nansum", "nanprod", "nanmax", "nanmin", "nanmean"])
def test_nan_reductions(name, axis):
    import sparse_amd as sp

    rng = np.random.default_rng(13)
    d = _dense(rng, (6, 9, 31), 0.5, nan_frac=0.15)
    d[2, 3, :] = np.nan
    d[2, 3, 5] = 0.25     # a row with one non-NaN element
    x = sp.COO.from_numpy(d)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        got = getattr(sp, name)(x, axis=axis)
        want = getattr(np, name)(d, axis=axis)
    got = got.todense() if hasattr(got, "todense") else np.asarray(got)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14, equal_nan=True)
    if name in ("nanmax", "nanmin"):
        assert np.array_equal(got, want, equal_nan=True)   # no arithmetic: exact


@pytest.mark.gpu
def test_all_nan_slice_warns_and_fmax_reduce_is_exposed():
    import sparse_amd as sp

    d = np.array([[np.nan, np.nan], [1.0, np.nan], [0.0, -2.0]])
    x = sp.COO.from_numpy(d)
    with pytest.warns(RuntimeWarning, match="All-NaN slice"):
        r = sp.nanmax(x, axis=1)
    assert np.array_equal(r.todense(), np.array([np.nan, 1.0, 0.0]), equal_nan=True)
    with pytest.warns(RuntimeWarning, match="Mean of empty slice"):
        m = sp.nanmean(x, axis=1)
    assert np.array_equal(m.todense(), np.array([np.nan, 1.0, -1.0]), equal_nan=True)
    assert np.array_equal(x.reduce(np.fmin, axis=0).todense(), np.fmin.reduce(d, axis=0))
    # integer arrays have no NaN: the plain reductions
    xi = sp.COO.from_numpy(np.array([[1, 0, 3], [0, 0, 5]]))
    assert np.array_equal(sp.nansum(xi, axis=0).todense(), [1, 0, 8]) and sp.nanmax(xi).todense() == 5
    assert np.array_equal(sp.nanreduce(x, np.add, axis=0).todense(), np.nansum(d, axis=0))


@pytest.mark.gpu
def test_creation_functions():
    import sparse_amd as sp

    z = sp.zeros((3, 4), dtype=np.float32)
    assert isinstance(z, sp.COO) and z.nnz == 0 and z.dtype == np.float32 and np.array_equal(z.todense(), np.zeros((3, 4), np.float32))
    o = sp.ones(5, dtype=int)
    assert o.nnz == 0 and o.fill_value == 1 and np.array_equal(o.todense(), np.ones(5, int))
    f = sp.full((2, 3), 7.5)
    assert f.dtype == np.float64 and np.array_equal(f.todense(), np.full((2, 3), 7.5))
    g = sp.zeros((4, 6), format="gcxs", compressed_axes=(1,))
    assert isinstance(g, sp.GCXS) and g.compressed_axes == (1,) and g.nnz == 0
    for n, m, k in ((4, None, 0), (3, 5, 1), (5, 3, -2), (3, 3, 7), (4, 6, -1)):
        e = sp.eye(n, m, k=k, dtype=np.float32)
        assert np.array_equal(e.todense(), np.eye(n, m, k=k, dtype=np.float32)) and e.nnz == int(np.eye(n, m, k=k).sum())
    x = sp.random((4, 5), density=0.5, random_state=1, format="gcxs", compressed_axes=(0,))
    zl, ol, fl = sp.zeros_like(x), sp.ones_like(x), sp.full_like(x, 3, dtype=np.int32)
    assert isinstance(zl, sp.GCXS) and zl.shape == x.shape and zl.dtype == x.dtype and zl.nnz == 0
    assert np.array_equal(ol.todense(), np.ones((4, 5))) and fl.dtype == np.int32 and np.array_equal(fl.todense(), np.full((4, 5), 3))
    assert np.array_equal((x.tocoo() + sp.ones((4, 5))).todense(), x.todense() + 1)   # fills combine in elemwise
    assert np.array_equal(np.zeros_like(x.tocoo()).todense(), np.zeros((4, 5)))       # __array_function__ route
    d = np.arange(24.0).reshape(2, 3, 4)
    c = sp.COO.from_numpy(d)
    assert np.array_equal(sp.moveaxis(c, 0, -1).todense(), np.moveaxis(d, 0, -1))
    assert np.array_equal(sp.moveaxis(c, (0, 1), (2, 0)).todense(), np.moveaxis(d, (0, 1), (2, 0)))
    assert sp.expand_dims(c, axis=1).shape == (2, 1, 3, 4) and sp.squeeze(sp.expand_dims(c, axis=(0, 4))).shape == (2, 3, 4)
    with pytest.raises(ValueError):
        sp.squeeze(c, axis=0)


@pytest.mark.gpu
def test_against_reference_fixture():
    """tests/golden/select.npz holds what the real reference returned for the same inputs (oracle/gen_golden.py
    `gen_select`): dense values, nnz after pruning and fill values must agree."""
    import os

    import sparse_amd as sp

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "select.npz"))
    sc, sx, sy, sxn = (sp.COO.from_numpy(g[k]) for k in ("c", "x", "y", "xn"))
    sc = sc > 0
    got = [sp.where(sc, sx, sy), sp.where(sc, 2.5, sy), sp.where(sc, sx, -1.0), sp.where(np.isnan(sxn), 0, sxn),
           sp.where(sxn, sx, sy)]
    assert len(got) == int(g["n_where"])
    for k, r in enumerate(got):
        label = str(g[f"w{k}_label"])
        assert np.array_equal(r.todense(), g[f"w{k}_dense"], equal_nan=True), label
        assert r.nnz == int(g[f"w{k}_nnz"]) and r.fill_value == g[f"w{k}_fill"] and r.dtype == g[f"w{k}_dense"].dtype, label
    for j, a in enumerate(sp.where(sy)):
        assert np.array_equal(a.cpu().numpy(), g[f"w1arg_{j}"])
    for k in range(int(g["n_nan"])):
        name, axis = str(g[f"n{k}_name"]), g[f"n{k}_axis"]
        axis = None if axis.ndim == 0 and int(axis) == -99 else (int(axis) if axis.ndim == 0 else tuple(int(a) for a in axis))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            r = getattr(sp, name)(sxn, axis=axis)
        want = g[f"n{k}_dense"]
        d = r.todense()
        assert d.shape == want.shape and d.dtype == want.dtype, (name, axis)
        if name in ("nanmax", "nanmin"):
            assert np.array_equal(d, want, equal_nan=True), (name, axis)
        else:
            assert np.allclose(d, want, rtol=1e-12, atol=1e-15, equal_nan=True), (name, axis)
        if r.ndim:
            assert r.nnz == int(g[f"n{k}_nnz"]), (name, axis)
        assert np.array_equal(np.asarray(r.fill_value, dtype=np.float64), np.asarray(g[f"n{k}_fill"], dtype=np.float64),
                              equal_nan=True), (name, axis)
    for j in range(4):
        n, m, kk = (int(v) for v in g[f"eye{j}_args"])
        e = sp.eye(n, None if m < 0 else m, k=kk, dtype=np.float32)
        assert np.array_equal(e.coords.cpu().numpy(), g[f"eye{j}_coords"]) and np.array_equal(e.data.cpu().numpy(), g[f"eye{j}_data"])
    f = sp.full((2, 3), 7, dtype=np.int32)
    assert f.fill_value == g["full_fill"] and f.nnz == int(g["full_nnz"]) and np.array_equal(f.todense(), g["full_dense"])
    assert np.array_equal((sx + sp.ones(tuple(g["shape"]))).todense(), g["ones_plus"])
