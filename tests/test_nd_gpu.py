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


@pytest.mark.gpu
def test_interchange_pickle_scipy_iter():
    """pickling (host-array state, reference core.py:293-298), scipy CSR/CSC export
    (core.py:1200-1291), `from_iter` (core.py:469-560), and the small shape helpers."""
    import pickle

    import scipy.sparse

    import sparse_amd as sp

    x = sp.random((7, 9), density=0.4, random_state=2)
    g = sp.GCXS.from_coo(x, compressed_axes=(1,))
    g @ torch.ones((9, 2), device="cuda", dtype=torch.float64)   # populates caches that must not be pickled
    for a in (x, g, x + 1):
        b = pickle.loads(pickle.dumps(a))
        assert type(b) is type(a) and b.shape == a.shape and b.fill_value == a.fill_value and b.nnz == a.nnz
        assert np.array_equal(b.todense(), a.todense())
    assert pickle.loads(pickle.dumps(g)).compressed_axes == (1,) and len(pickle.dumps(g)) < 2 * g.nbytes + 2000
    csr, csc = x.tocsr(), x.tocsc()
    assert isinstance(csr, scipy.sparse.csr_array) and isinstance(csc, scipy.sparse.csc_array)
    assert np.array_equal(csr.toarray(), x.todense()) and np.array_equal(csc.toarray(), x.todense())
    assert csr.has_sorted_indices and csc.has_sorted_indices
    with pytest.raises(ValueError):
        sp.random((2, 3, 4), density=0.5, random_state=0).tocsr()
    with pytest.raises(ValueError):
        (x + 1).tocsc()
    want = np.array([[1, 0], [0, 1]])
    for it in ([((0, 0), 1), ((1, 1), 1)], {(0, 0): 1, (1, 1): 1}, ([1, 1], ([0, 1], [0, 1])), iter([((0, 0), 1), ((1, 1), 1)])):
        assert np.array_equal(sp.COO.from_iter(it, shape=(2, 2)).todense(), want)
    assert np.array_equal(sp.GCXS.from_iter({(0, 0): 1, (1, 1): 1}, shape=(2, 2), compressed_axes=(1,)).todense(), want)
    assert sp.COO.from_iter([], shape=(3, 2)).nnz == 0
    with pytest.raises(ValueError):
        sp.COO.from_iter([(1, 2, 3)], shape=(2, 2))
    d = x.todense()
    assert np.array_equal(x.swapaxes(0, 1).todense(), d.swapaxes(0, 1))
    assert np.array_equal(x[None].squeeze().todense(), d) and x.broadcast_to((2, 7, 9)).shape == (2, 7, 9)
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(x.nonzero(), d.nonzero()))
    one = sp.COO.from_numpy(np.array(3))
    assert complex(one) == 3 + 0j and [10, 20, 30, 40][one] == 40


@pytest.mark.gpu
def test_basic_indexing_matches_numpy():
    """Integers, slices with any step, None and Ellipsis on COO and GCXS (reference _coo/indexing.py:12-133),
    checked against NumPy on the dense array over a table of index expressions plus random ones."""
    import sparse_amd as sp

    rng = np.random.default_rng(21)
    d = rng.random((6, 7, 8)) * (rng.random((6, 7, 8)) < 0.4)
    x = sp.COO.from_numpy(d)
    g = sp.GCXS.from_coo(x, compressed_axes=(0,))
    s_ = np.s_
    table = [s_[2], s_[-1], s_[1:4], s_[:, 3], s_[..., 5], s_[1, 2, 3], s_[1, :, 3], s_[::2], s_[::-1], s_[5:1:-2],
             s_[:, ::3, 1:7:2], s_[None], s_[:, None, 2], s_[..., None], s_[3:3], s_[10:20], s_[-3:], s_[:, -2::-3],
             s_[1:2, 1:2, 1:2], s_[None, 2, ..., None, 4], s_[()], s_[...], s_[0, ::-1, ::-1]]
    for _ in range(40):
        ix = []
        for n in d.shape:
            kind = rng.integers(0, 4)
            if kind == 0:
                ix.append(int(rng.integers(-n, n)))
            elif kind == 1:
                ix.append(slice(*(None if rng.random() < 0.3 else int(v) for v in rng.integers(-n - 1, n + 2, 2)),
                                int(rng.choice([-3, -2, -1, 1, 2, 3]))))
            elif kind == 2:
                ix.append(slice(None))
            else:
                ix.extend([None, slice(None)])
        table.append(tuple(ix))
    for idx in table:
        want = d[idx]
        for arr in (x, g):
            got = arr[idx]
            if np.ndim(want) == 0:
                assert not isinstance(got, sp.SparseArray) and got == want, idx
            else:
                assert got.shape == want.shape and np.array_equal(got.todense(), want), idx
                assert got.nnz == np.count_nonzero(want), idx
    assert isinstance(g[1:3], sp.GCXS) and isinstance(x[1:3], sp.COO)
    y = (x + 2)[::2, 1]      # the fill value travels
    assert y.fill_value == 2 and np.array_equal(y.todense(), (d + 2)[::2, 1])
    v = sp.COO.from_numpy(np.array([0.0, 1.5, 0.0, 2.5]))
    assert v[1] == 1.5 and v[2] == 0.0 and v[-1] == 2.5
    for bad in (s_[6], s_[0, 0, 0, 0], s_[..., ...]):
        with pytest.raises(IndexError):
            x[bad]
    assert np.array_equal(x[[0, 1]].todense(), d[[0, 1]])          # one integer array: the join of tests/array_api_cases.py
    with pytest.raises(NotImplementedError):
        x[[0, 1], [1, 0]]


@pytest.mark.gpu
@pytest.mark.parametrize("lead", [(5,), (2, 3), (1,), (4, 1)])
def test_batched_matmul_is_one_block_diagonal_product(lead):
    """`a @ b` with equal leading axes runs as blockdiag(a) @ stacked(b) (sparse_amd/_batched.py): same values as
    the per-slice loop of the reference's `_matmul_recurser` (_common.py:278-293) and as NumPy on dense arrays."""
    import sparse_amd as sp
    from sparse_amd._batched import block_diagonal_csr, matmul_batched

    rng = np.random.default_rng(31)
    M, Kd, N = 37, 29, 16
    da = rng.random(lead + (M, Kd)) * (rng.random(lead + (M, Kd)) < 0.2)
    db = rng.random(lead + (Kd, N))
    dbs = db * (rng.random(db.shape) < 0.3)
    a = sp.COO.from_numpy(da)
    big = block_diagonal_csr(a)
    B = int(np.prod(lead))
    want_big = np.zeros((B * M, B * Kd))
    for i, blk in enumerate(da.reshape(B, M, Kd)):
        want_big[i * M:(i + 1) * M, i * Kd:(i + 1) * Kd] = blk
    assert big.shape == (B * M, B * Kd) and np.array_equal(big.todense(), want_big)
    r = a @ torch.from_numpy(db).cuda()
    assert tuple(r.shape) == lead + (M, N) and np.allclose(r.cpu().numpy(), da @ db, rtol=1e-13, atol=1e-15)
    loop = matmul_batched(a, torch.from_numpy(db).cuda())
    assert torch.equal(r, loop)                          # same accumulation order as slice-by-slice
    rs = a @ sp.COO.from_numpy(dbs)
    assert isinstance(rs, sp.COO) and np.allclose(rs.todense(), da @ dbs, rtol=1e-13, atol=1e-15)
    rg = sp.GCXS.from_coo(a) @ sp.GCXS.from_coo(sp.COO.from_numpy(dbs))
    assert isinstance(rg, sp.GCXS) and np.allclose(rg.todense(), da @ dbs, rtol=1e-13, atol=1e-15)
    rn = a @ db
    assert isinstance(rn, np.ndarray) and np.allclose(rn, da @ db, rtol=1e-13)   # NumPy in -> NumPy out


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1000, 1000, 1000), (65536, 65535), (3, 1, 7, 2, 5), (2 ** 31 - 1, 2), (2 ** 20, 2 ** 20, 4094),
                                   (2 ** 26, 2 ** 26 - 3), (2 ** 30, 2 ** 30), (1, 1), (7,)])
def test_key_coordinate_conversions_match_numpy(shape):
    """`delinearize` / `permute_keys` (csrc/prims.hip) take a reciprocal-division path below 2^32 and below 2^52 cells
    and the generic 64-bit division above: all three against NumPy, with keys at the ends of the range and at
    multiples of the dimensions (where a truncated quotient estimate is off by one)."""
    from sparse_amd import _kernels as Kn

    size = int(np.prod([int(s) for s in shape], dtype=object))
    rng = np.random.default_rng(41)
    special = [0, size - 1, size // 2]
    for d in shape:
        special += [min(size - 1, m * d + o) for m in (1, 2, size // max(d, 1) - 1) for o in (-1, 0, 1) if 0 <= m * d + o]
    keys = np.unique(np.concatenate([np.array([k for k in special if 0 <= k < size], dtype=np.int64),
                                     rng.integers(0, size, 5000, dtype=np.int64),
                                     size - 1 - rng.integers(0, min(size, 1000), 200, dtype=np.int64)]))
    tk = torch.from_numpy(keys).cuda()
    got = Kn.delinearize(tk, shape, torch.int64).cpu().numpy()
    want = np.stack(np.unravel_index(keys, shape))
    assert np.array_equal(got, want)
    if max(shape) < 2 ** 31:
        assert np.array_equal(Kn.delinearize(tk, shape, torch.int32).cpu().numpy(), want.astype(np.int32))
    if len(shape) > 1:
        for perm in (tuple(reversed(range(len(shape)))), tuple(np.roll(np.arange(len(shape)), 1).tolist())):
            pk = Kn.permute_keys(tk, shape, perm).cpu().numpy()
            pshape = tuple(shape[a] for a in perm)
            assert np.array_equal(pk, np.ravel_multi_index(tuple(want[a] for a in perm), pshape))
