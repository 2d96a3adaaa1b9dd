"""The array-namespace functions (`sparse_amd/_array_api.py`) against NumPy on the dense arrays, over random shapes,
densities, dtypes and both containers - including arrays without stored elements, 1-D and 4-D arrays.  (The reference's own
tests check these functions the same way, against NumPy: SURVEY.md section 4; the quirks where the reference departs from
NumPy are pinned by tests/golden/array_api.npz instead.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dense(rng, shape, density, dtype):
    d = np.zeros(shape, dtype=np.float64)
    m = rng.random(shape) < density
    d[m] = rng.random(int(m.sum())) * 2 - 1          # distinct with probability 1: no ties for argmax / sort
    if np.dtype(dtype).kind in "iu":
        return (d * 1000).astype(dtype)
    return d.astype(dtype)


def _cases(seed):
    rng = np.random.default_rng(seed)
    for shape in ((7,), (5, 6), (4, 1, 5), (3, 4, 2, 3), (0, 4), (6, 6)):
        for density in (0.0, 0.3, 1.0):
            for dtype in (np.float64, np.float32, np.int64):
                yield rng, shape, _dense(rng, shape, density, dtype)


def _both(sp, d):
    yield sp.COO.from_numpy(d)
    if d.ndim >= 1 and d.size:
        yield sp.GCXS.from_numpy(d)


def _same(got, want):
    got = got.todense() if hasattr(got, "todense") else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert got.dtype == want.dtype, (got.dtype, want.dtype)
    assert np.array_equal(got, want)


def test_flip_roll_pad_against_numpy():
    import sparse_amd as sp

    for rng, shape, d in _cases(1):
        for x in _both(sp, d):
            _same(sp.flip(x), np.flip(d))
            for ax in range(d.ndim):
                _same(sp.flip(x, axis=ax), np.flip(d, axis=ax))
                if d.shape[ax]:
                    sh = int(rng.integers(-9, 9))
                    _same(sp.roll(x, sh, axis=ax), np.roll(d, sh, axis=ax))
            if d.size:
                _same(sp.roll(x, 3), np.roll(d, 3))
            width = [(int(rng.integers(0, 3)), int(rng.integers(0, 3))) for _ in shape]
            _same(sp.pad(x, width), np.pad(d, width))


def test_triangles_diagonals_kron_against_numpy():
    import sparse_amd as sp

    for rng, shape, d in _cases(2):
        if d.ndim < 2:
            continue
        x = sp.COO.from_numpy(d)
        for k in (-2, 0, 1, 5):
            _same(sp.triu(x, k), np.triu(d, k))
            _same(sp.tril(x, k), np.tril(d, k))
        if d.shape[0] == d.shape[1] and d.ndim == 2:
            for off in (0, 1, 3):
                _same(sp.diagonal(x, offset=off), np.diagonal(d, offset=off))
        small = _dense(rng, (2, 3) if d.ndim == 2 else (2,) * d.ndim, 0.7, d.dtype)
        if d.size * small.size < 50000:
            _same(sp.kron(x, sp.COO.from_numpy(small)), np.kron(d, small))
            _same(sp.kron(small, x), np.kron(small, d))


def test_argmax_argmin_sort_against_numpy():
    import sparse_amd as sp

    for rng, shape, d in _cases(4):
        if not d.size:
            continue
        for x in _both(sp, d):
            full = not (d == 0).any()
            for ax in range(d.ndim):
                _same(sp.sort(x, axis=ax), np.sort(d, axis=ax))
                _same(sp.sort(x, axis=ax, descending=True), np.flip(np.sort(d, axis=ax), axis=ax))
                if full or d.dtype.kind == "f":
                    # (with zeros in the line NumPy returns the first zero's position: so does the reference's first-gap
                    #  rule when no stored value beats the fill value; distinct values otherwise)
                    # (the reference squeezes EVERY unit axis of the result, `_arg_minmax_common`, _coo/common.py:1568)
                    _same(sp.argmax(x, axis=ax), np.squeeze(np.argmax(d, axis=ax)))
                    _same(sp.argmin(x, axis=ax), np.squeeze(np.argmin(d, axis=ax)))
                    if d.ndim > 1:     # (a vector is lifted to a column first there: keepdims gives (1, 1))
                        _same(sp.argmax(x, axis=ax, keepdims=True), np.argmax(d, axis=ax, keepdims=True))
            _same(np.asarray(sp.argmax(x).todense()).reshape(()), np.argmax(d))
