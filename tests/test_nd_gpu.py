"""N-D matmul broadcasting, stack / concatenate, x[i] — vs fixtures produced by
the real reference (tests/golden/nd.npz); plus the exception parity of the `_dot` front end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def sp():
    import sparse_amd

    return sparse_amd


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "nd.npz"))


def _dense(x):
    return x.todense() if hasattr(x, "todense") else (x.cpu().numpy() if isinstance(x, torch.Tensor) else x)


def test_matmul_broadcasting(sp, g):
    for k in range(int(g["n_matmul"])):
        a = sp.COO(g[f"m{k}_a_coords"], g[f"m{k}_a_data"], shape=tuple(g[f"m{k}_a_shape"]))
        b = sp.COO(g[f"m{k}_b_coords"], g[f"m{k}_b_data"], shape=tuple(g[f"m{k}_b_shape"]))
        r = sp.matmul(a, b)
        assert isinstance(r, sp.SparseArray) == bool(g[f"m{k}_ss_sparse"]), k
        assert np.allclose(_dense(r), g[f"m{k}_ss"], rtol=1e-13, atol=1e-15), k
        r = sp.matmul(a, g[f"m{k}_bd"])
        assert np.allclose(_dense(r), g[f"m{k}_sd"], rtol=1e-13, atol=1e-15), k
        r = sp.matmul(sp.GCXS(a), sp.GCXS(b))
        assert r.format == str(g[f"m{k}_gg_fmt"]), k
        assert np.allclose(_dense(r), g[f"m{k}_gg"], rtol=1e-13, atol=1e-15), k


def test_stack_concatenate_take(sp, g):
    xs = [sp.COO(g[f"s{i}_coords"], g[f"s{i}_data"], shape=(4, 5, 3)) for i in range(3)]
    for ax in (0, 1, 2, 3, -1):
        r = sp.stack(xs, axis=ax)
        assert r.shape == tuple(g[f"stack{ax}_shape"])
        assert np.array_equal(r.coords.cpu().numpy(), g[f"stack{ax}_coords"])
        assert np.array_equal(r.data.cpu().numpy(), g[f"stack{ax}_data"])
    for ax in (0, 1, 2):
        r = sp.concatenate(xs, axis=ax)
        assert r.shape == tuple(g[f"cat{ax}_shape"])
        assert np.array_equal(r.coords.cpu().numpy(), g[f"cat{ax}_coords"])
        assert np.array_equal(r.data.cpu().numpy(), g[f"cat{ax}_data"])
    r = sp.concatenate([sp.GCXS(x, compressed_axes=(0,)) for x in xs], axis=1)
    assert r.format == str(g["gcat_fmt"]) and np.array_equal(r.todense(), g["gcat_dense"])
    for i in (0, 2, -1):
        r = xs[0][i]
        assert np.array_equal(r.coords.cpu().numpy(), g[f"take{i}_coords"])
        assert np.array_equal(r.data.cpu().numpy(), g[f"take{i}_data"])


def test_exception_parity(sp):
    """Same exception types/messages as the reference front end."""
    x = sp.COO(np.array([[0, 1], [1, 2]]), np.array([1.0, 2.0]), shape=(3, 4))
    with pytest.raises(ValueError, match="shape-mismatch for sum"):
        sp.tensordot(x, np.ones((5, 2)), axes=1)
    with pytest.raises(ValueError, match="requires zero fill values, but argument 0 had a fill value of 1.0"):
        sp.dot(x + 1, np.ones((4, 2)))
    with pytest.raises(TypeError, match="Cannot perform dot product on types"):
        sp.matmul(x, 3)
    with pytest.raises(ValueError, match="shapes of a and b are not broadcastable"):
        sp.matmul(sp.random((2, 3, 4), density=0.5, random_state=0), sp.random((3, 4, 5), density=0.5, random_state=1))
    with pytest.raises(ValueError, match="would result in a dense array"):
        _ = x + np.arange(4.0)          # func(fill, dense) varies and the dense operand is the smaller one
    r = x + np.ones((3, 4))             # func(fill, dense) is constant: sparse result whose fill value is that constant
    assert isinstance(r, sp.COO) and r.fill_value == 1.0 and r.nnz == 2 and np.array_equal(r.todense(), x.todense() + 1)
    d = np.arange(12.0).reshape(3, 4)
    r = x + d                           # varies, same shape: the reference returns the dense result (_umath.py:463-465)
    assert not isinstance(r, sp.SparseArray) and np.array_equal(np.asarray(r.cpu()), x.todense() + d)
    r = x * -np.ones((3, 4))            # 0 * -1 = -0.0 everywhere: still the (signed) zero fill, result stays sparse
    assert isinstance(r, sp.COO) and r.nnz == 2 and np.signbit(r.fill_value) and np.array_equal(r.todense(), x.todense() * -1.0)
    with pytest.raises(ValueError, match="would produce a dense result"):
        np.subtract.reduce(x + 1, axis=0)
    with pytest.raises(RuntimeError, match="Cannot convert a sparse array to dense automatically"):
        np.asarray(x)
    with pytest.raises(ValueError, match="cannot compress all axes"):
        sp.GCXS(x, compressed_axes=(0, 1))
    with pytest.raises(ValueError, match="The data length does not match the coordinates given"):
        sp.COO(np.array([[0, 1]]), np.array([1.0]), shape=(3,))
    # NumPy protocol surface (reference tests/test_array_function.py:14-49)
    y = np.ones((4, 2))
    assert np.allclose(np.dot(x, y), x.todense() @ y)
    assert np.allclose(np.tensordot(x, y, axes=1), x.todense() @ y)
    assert np.allclose(np.matmul(x, y), x.todense() @ y)
    assert float(np.sum(x).todense()) == 3.0


def test_broadcasting_and_npz_interchange(sp, tmp_path):
    """N3: sparse (x) sparse / sparse (x) dense broadcasting vs the reference's result; N4: npz files
    written here load in the reference's layout and vice versa (layout = reference _io.py:49-62)."""
    ge = np.load(os.path.join(GOLD, "elemwise.npz"))
    shape = tuple(ge["shape"])
    x = sp.COO(ge["x_coords"], ge["x_data"], shape=shape)
    z = sp.COO(ge["bz_coords"], ge["bz_data"], shape=(8, 1))
    r = x * z
    assert r.shape == shape
    assert np.array_equal(r.coords.cpu().numpy(), ge["bmul_coords"])
    assert np.array_equal(r.data.cpu().numpy(), ge["bmul_data"])
    zb = sp.broadcast_to(z, shape)
    assert np.array_equal(zb.todense(), np.broadcast_to(z.todense(), shape))
    d = np.linspace(0.5, 2.0, shape[2])
    assert np.array_equal((x * d).todense(), x.todense() * d)
    # npz round trip + the reference's key layout
    p = tmp_path / "a.npz"
    sp.save_npz(p, x)
    with np.load(p) as f:
        assert set(f.files) == {"data", "coords", "shape", "fill_value"}
    y = sp.load_npz(p)
    assert np.array_equal(y.coords.cpu().numpy(), ge["x_coords"]) and np.array_equal(y.data.cpu().numpy(), ge["x_data"])
    gx = sp.GCXS(x, compressed_axes=(1,))
    sp.save_npz(p, gx)
    with np.load(p) as f:
        assert set(f.files) == {"data", "indices", "indptr", "compressed_axes", "shape", "fill_value"}
    gy = sp.load_npz(p)
    assert gy.compressed_axes == (1,) and np.array_equal(gy.todense(), x.todense())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int64, np.float64, np.float32])
def test_reductions_with_runs_longer_than_a_tile(dtype):
    """Grouped reduce (csrc/group_reduce.hip): runs that span many 2048-element tiles (full reduction, one long
    group among short ones) go through the chained fix-up; checked against NumPy on the dense array."""
    import sparse_amd as sp

    rng = np.random.default_rng(5)
    shape = (6, 700, 900)
    dense = np.zeros(shape, dtype=dtype)
    mask = rng.random(shape) < 0.15
    mask[2] = True  # one group of 630000 elements when reducing over the last two axes
    vals = rng.integers(-40, 40, size=int(mask.sum())) if np.dtype(dtype).kind == "i" else rng.random(int(mask.sum())) - 0.4
    dense[mask] = vals.astype(dtype)
    x = sp.COO.from_numpy(dense)
    for axis in (None, (1, 2), 2, (0, 1)):
        for name in ("sum", "max", "min"):
            got = getattr(x, name)(axis=axis)
            want = getattr(dense, name)(axis=axis)
            got = got.todense() if hasattr(got, "todense") else np.asarray(got)
            if np.dtype(dtype).kind == "i" or name != "sum":
                assert np.array_equal(got, want), (name, axis)
            else:
                assert np.allclose(got, want, rtol=1e-5 if dtype == np.float32 else 1e-12), (name, axis)


@pytest.mark.gpu
def test_prune_count_first_path(monkeypatch):
    """Large prunes first count the fill values (one read-only pass) and stop when there are none; the path is
    forced here for small arrays: nothing to prune, something to prune, -0.0 is not a fill value (bit-wise)."""
    import sparse_amd as sp
    from sparse_amd import _kernels as Kn

    monkeypatch.setattr(Kn, "PRUNE_COUNT_FIRST", 1)
    coords = np.array([[0, 0, 1, 2, 2], [1, 3, 0, 2, 4]])
    full = sp.COO(coords, np.array([1.0, 2.0, 3.0, 4.0, 5.0]), shape=(3, 5), prune=True)
    assert full.nnz == 5 and Kn.count_eq_bits(full.data, 0.0) == 0
    some = sp.COO(coords, np.array([1.0, 0.0, 3.0, -0.0, 0.0]), shape=(3, 5), prune=True)
    assert some.nnz == 3 and np.array_equal(some.todense()[[0, 1, 2], [1, 0, 2]], [1.0, 3.0, -0.0])
    assert np.signbit(some.todense()[2, 2])
    g = sp.GCXS(some, compressed_axes=(0,))
    assert g.nnz == 3
    ints = sp.COO(coords, np.array([1, 0, 0, 7, 0], dtype=np.int32), shape=(3, 5), prune=True)
    assert ints.nnz == 2 and Kn.count_eq_bits(torch.tensor([1, 0, 0, 7, 0], dtype=torch.int32, device="cuda"), 0) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("nnz", [(0, 5), (4096, 4096), (4097, 1), (100_000, 171_873), (171_873, 171_873)])
def test_merge_single_pass_matches_count_scan_fill(monkeypatch, nnz):
    """The elementwise merge exists in two forms (csrc/merge.hip): one pass with look-back offsets (default) and
    count + scan + fill.  Both must give the same canonical result, including full operands (every key shared),
    tile-boundary sizes and an empty operand; checked against the dense arithmetic as well."""
    import sparse_amd as sp
    from sparse_amd import _umath as U

    shape = (39, 39, 113)  # 171873 cells
    x = sp.random(shape, nnz=nnz[0], random_state=3)
    y = sp.random(shape, nnz=nnz[1], random_state=4)
    for f in (lambda a, b: a + b, lambda a, b: a * b, lambda a, b: np.maximum(a, b)):
        monkeypatch.setattr(U, "MERGE_SINGLE_PASS", True)
        r1 = f(x, y)
        monkeypatch.setattr(U, "MERGE_SINGLE_PASS", False)
        r2 = f(x, y)
        assert r1.nnz == r2.nnz <= x.nnz + y.nnz
        assert torch.equal(r1.linear_loc(), r2.linear_loc()) and torch.equal(r1.data, r2.data)
        assert np.array_equal(r1.todense(), f(x.todense(), y.todense()))
