"""Round-3 additions: config 5's per-GPU share against the oracle on sampled rows, the NumPy protocol surface
(`ufunc.outer`, `out=` casting), reductions with ufuncs outside the device kernels' table, complex values, and the
robustness fixes of the round (unsorted column indices reaching the one-pass inspector, out-of-range row ids)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def sp():
    import sparse_amd

    return sparse_amd


def test_config5_share_against_the_oracle_on_sampled_rows(orc, sp):
    """One of the 8 row blocks of config 5 — GCXS(125000 x 10^6) @ GCXS(10^6 x 10^6 @ 1e-4): 1.25e9 products, a result
    with more than 2^30 stored elements (int64 row pointers) — against the oracle's restatement of `_dot_csr_csr`
    (_common.py:639-717) on 256 sampled rows: column indices bit for bit (after the canonical per-row sort: the
    reference emits rows in reverse discovery order, Appendix C.2), values bit for bit as well (the products of an
    output element are summed left to right in the order of A's elements, the reference's `sums[j] += ...`)."""
    n, share = 1_000_000, 8
    gB = sp.random((n, n), density=1e-4, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs",
                   compressed_axes=(0,))
    rows = n // share
    p1 = int(gB.indptr[rows])
    gA = sp.GCXS((gB.data[:p1].contiguous(), gB.indices[:p1].contiguous(), gB.indptr[:rows + 1].contiguous()),
                 shape=(rows, n), compressed_axes=(0,))
    c = gA @ gB
    assert isinstance(c, sp.GCXS) and c.shape == (rows, n) and c.compressed_axes == (0,)
    assert c.indptr.dtype == torch.int64 and c.nnz > 2 ** 30
    pick = np.sort(np.random.default_rng(3).choice(rows, size=256, replace=False))
    hA = [t.cpu().numpy() for t in (gA.data, gA.indices, gA.indptr)]
    hB = [t.cpu().numpy() for t in (gB.data, gB.indices, gB.indptr)]
    segs = [np.arange(hA[2][r], hA[2][r + 1]) for r in pick]
    sub_ptr = np.zeros(len(pick) + 1, dtype=hA[2].dtype)
    sub_ptr[1:] = np.cumsum([len(s) for s in segs])
    sel = np.concatenate(segs)
    wd, wi, wp = orc.dot_csr_csr((len(pick), n), hA[0][sel], hB[0], hA[1][sel], hB[1], sub_ptr, hB[2])
    cp = c.indptr.cpu().numpy()
    for j, r in enumerate(pick):
        lo, hi = int(cp[r]), int(cp[r + 1])
        gi, gd = c.indices[lo:hi].cpu().numpy(), c.data[lo:hi].cpu().numpy()
        o = np.argsort(wi[wp[j]:wp[j + 1]], kind="stable")
        keep = wd[wp[j]:wp[j + 1]][o].view(np.uint32) != 0      # the GCXS constructor prunes explicit +0.0 results
        assert np.array_equal(gi, wi[wp[j]:wp[j + 1]][o][keep]), f"row {r}: column indices differ"
        assert np.array_equal(gd.view(np.uint32), wd[wp[j]:wp[j + 1]][o][keep].view(np.uint32)), f"row {r}: values differ"
    # the row pointers of the sampled rows' neighbours are consistent with the row lengths the oracle found
    assert np.array_equal(np.diff(cp)[pick], [int(np.count_nonzero(wd[wp[j]:wp[j + 1]].view(np.uint32))) for j in range(len(pick))])


def _dense(shape, seed, density=0.5):
    r = np.random.default_rng(seed)
    return np.where(r.random(shape) < density, r.random(shape) - 0.4, 0.0)


def test_ufunc_outer_has_numpys_axis_order(sp):
    """`np.<ufunc>.outer(a, b)`: result axes are a's, then b's (reference _sparse_array.py:343-352)."""
    a, b, c = _dense((3, 4), 1), _dense((5,), 2), _dense((2, 3), 3)
    for f in (np.subtract, np.multiply, np.maximum):
        for x, y in ((a, b), (b, a), (a, c)):
            got = f.outer(sp.COO.from_numpy(x), sp.COO.from_numpy(y))
            want = f.outer(x, y)
            assert got.shape == want.shape and np.array_equal(got.todense(), want)
    got = np.multiply.outer(sp.GCXS.from_numpy(a), sp.GCXS.from_numpy(c))
    assert isinstance(got, sp.GCXS) and np.array_equal(got.todense(), np.multiply.outer(a, c))


def test_out_casting_is_validated_before_any_work(sp):
    """An `out=` the ufunc may not cast into raises NumPy's own `UFuncTypeError` (reference _sparse_array.py:333-342);
    a legal `out=` receives the result and is returned."""
    x, y = sp.COO.from_numpy(_dense((4, 5), 4)), sp.COO.from_numpy(_dense((4, 5), 5))
    bad = sp.COO.from_numpy(np.arange(20).reshape(4, 5))
    with pytest.raises(np._core._exceptions.UFuncTypeError):
        np.add(x, y, out=bad)
    with pytest.raises(ValueError):     # the reference's dry run reduces 1-element stand-ins into a 1-element `out`:
        np.add.reduce(x, axis=None, out=bad.sum())   # NumPy refuses the shape before it looks at the cast (same here)
    good = sp.COO.from_numpy(np.zeros((4, 5), dtype=np.float32))
    r = np.add(x, y, out=good)
    assert r is good and r.dtype == np.float32
    assert np.array_equal(r.todense(), (x.todense() + y.todense()).astype(np.float32))
    assert np.add(x, y, out=(good,)) is good


@pytest.mark.parametrize("ufunc, dtype", [(np.bitwise_or, np.int64), (np.bitwise_and, np.int32), (np.bitwise_xor, np.int64),
                                          (np.hypot, np.float64), (np.gcd, np.int64), (np.logaddexp2, np.float32)])
@pytest.mark.parametrize("axis", [0, (0, 2), None])
def test_reduce_with_any_binary_ufunc(sp, ufunc, dtype, axis):
    """Ufuncs outside the grouped-reduce kernels' table take the reference's own algorithm (`ufunc.reduceat` over the
    grouped runs + fold-in of the implicit fill values, _coo/core.py:1631-1661, _sparse_array.py:398-408); the
    comparison is NumPy's dense reduction, or the `ValueError` the reference raises when the result would be dense."""
    d = _dense((4, 5, 6), 7)
    d = (d * 50).astype(dtype) if np.dtype(dtype).kind == "i" else d.astype(dtype)
    x = sp.COO.from_numpy(d)
    fv = x.fill_value
    dense_result = not np.array_equal(ufunc.reduce([fv, fv]), fv)
    if dense_result:
        with pytest.raises(ValueError, match="dense result"):
            ufunc.reduce(x, axis=axis)
        return
    got = ufunc.reduce(x, axis=axis)
    want = ufunc.reduce(d, axis=axis)
    assert got.dtype == want.dtype and got.shape == np.shape(want)
    if np.dtype(dtype).kind == "i":
        assert np.array_equal(got.todense(), want)
    else:
        assert np.allclose(got.todense(), want, rtol=1e-6 if dtype == np.float32 else 1e-13)
    g = ufunc.reduce(sp.GCXS.from_numpy(d), axis=axis)
    assert np.array_equal(g.todense(), got.todense())


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_values(sp, dtype):
    """Complex arrays: canonical form, densification, transposition, elementwise arithmetic (host-evaluated function,
    device structure) and reductions (host `reduceat`), against NumPy on the dense arrays."""
    d = (_dense((4, 5, 6), 8) + 1j * _dense((4, 5, 6), 9)).astype(dtype)
    e = (_dense((4, 5, 6), 10) + 1j * _dense((4, 5, 6), 11)).astype(dtype)
    x, y = sp.COO.from_numpy(d), sp.COO.from_numpy(e)
    assert x.dtype == dtype and x.nnz == np.count_nonzero(d) and np.array_equal(x.todense(), d)
    assert np.array_equal(x.transpose((2, 0, 1)).todense(), d.transpose(2, 0, 1))
    assert np.array_equal((x + y).todense(), d + e) and np.array_equal((x * y).todense(), d * e)
    assert np.array_equal(np.conj(x).todense(), np.conj(d))
    tol = dict(rtol=1e-5 if dtype == np.complex64 else 1e-13, atol=1e-6 if dtype == np.complex64 else 1e-14)
    for axis in (0, (1, 2), None):
        s = x.sum(axis=axis)
        assert s.dtype == dtype and np.allclose(s.todense(), d.sum(axis=axis), **tol)
    assert np.allclose((x + (1 + 2j)).prod(axis=2).todense(), (d + (1 + 2j)).prod(axis=2), **tol)
    unsorted = sp.COO(np.array([[1, 0, 0], [0, 2, 1]]), np.array([3j, 1 + 1j, 2 - 1j], dtype=dtype), shape=(2, 3),
                      has_duplicates=False, sorted=False)
    assert np.array_equal(unsorted.todense(), np.array([[0, 2 - 1j, 1 + 1j], [3j, 0, 0]], dtype=dtype))
    g = sp.GCXS.from_numpy(d)
    assert np.array_equal(g.todense(), d) and np.allclose(g.sum(axis=1).todense(), d.sum(axis=1), **tol)


def test_unsorted_columns_never_reach_the_executor_as_garbage(sp):
    """A GCXS built from raw (data, indices, indptr) with UNSORTED column indices inside the rows (the tuple constructor
    does not sort) and large enough for the inspector/executor path: the one-pass inspector must stay inside its
    allocation, and the product must be the reference's (the per-row sums do not depend on the order the row is stored
    in beyond fp re-association; here values are small integers, so exactly)."""
    from sparse_amd import _kernels

    M, K, N = 70_000, 4096, 128
    rng = np.random.default_rng(5)
    per_row = 48
    cols = np.argsort(rng.random((M, K // 8)), axis=1)[:, :per_row].astype(np.int32) * 8 + rng.integers(0, 8, (M, per_row), dtype=np.int32)
    data = rng.integers(1, 5, size=(M, per_row)).astype(np.float32)
    ptr = (np.arange(M + 1) * per_row).astype(np.int32)
    b = rng.integers(-3, 4, size=(K, N)).astype(np.float32)
    a = sp.GCXS((data.reshape(-1), cols.reshape(-1), ptr), shape=(M, K), compressed_axes=(0,))
    got = a @ torch.from_numpy(b).cuda()
    import scipy.sparse as sps

    want = sps.csr_matrix((data.reshape(-1), cols.reshape(-1), ptr), shape=(M, K)) @ b
    assert np.array_equal(got.cpu().numpy(), want)
    lay = getattr(a, "_tiled_layouts", None)
    assert lay, "this shape must take the inspector/executor path"
    assert not getattr(next(iter(lay.values())), "group_ends", False), "the layout in use must be the key-sort recipe's"
    # the one-pass inspector on its own: flags the matrix, and every group that met an unsorted row holds zero entries only
    direct = _kernels.csr_tiled_layout(a.data, a.indices, a.indptr, M, K, defer_check=True)
    assert int(direct.pending[0]) != 0
    rg, kb, gpb, epb, slack, _, _ = _kernels.tiled_params(torch.float32)
    ntiles = -(-K // kb)
    off = direct[1].view(-1, ntiles + 1).cpu().numpy()
    blocks = direct[0].cpu().numpy()
    assert not blocks[off[0, 0] * 16: off[200, ntiles] * 16].any(), "lists of groups with unsorted rows must be zero entries"


def test_rows_to_indptr_ignores_nothing_and_writes_in_bounds(sp):
    """`rows_to_indptr` on row ids that a trusting constructor let through (sorted=True, has_duplicates=False skip the
    range check): ids beyond R are clamped, never written past the R + 1 pointers."""
    from sparse_amd import _kernels

    R = 1000
    rows = torch.tensor([0, 0, 5, 999, 1500, 4000], dtype=torch.int64, device="cuda")
    guard = torch.full((R + 1 + 64,), -7, dtype=torch.int64, device="cuda")
    ptr = _kernels.rows_to_indptr(rows, R)
    assert ptr.numel() == R + 1 and int(ptr[0]) == 0 and int(ptr[1]) == 2 and int(ptr[6]) == 3 and int(ptr[R]) <= 6
    # a long empty stretch in front of the only populated rows (one thread used to fill it serially)
    big = 3_000_000
    rows = torch.full((5000,), big - 1, dtype=torch.int64, device="cuda")
    ptr = _kernels.rows_to_indptr(rows, big)
    assert int(ptr[big]) == 5000 and int(ptr[big - 1]) == 0 and int(ptr[big // 2]) == 0 and int(ptr.max()) == 5000
    del guard


def test_sddmm_with_a_misaligned_bf16_view_takes_the_sampled_kernel(sp):
    """A contiguous bf16 operand whose storage offset is not a multiple of 16 bytes cannot feed the matrix-core tile
    kernel; the product must still be computed (by the sampled kernel), not refused."""
    M, Kd = 2048, 64
    s = sp.random((M, M), density=0.02, random_state=1, dtype=np.float32)
    base = (torch.rand(M * Kd + 8, device="cuda") - 0.5).to(torch.bfloat16)
    a = base[3:3 + M * Kd].view(M, Kd)
    bt = base[5:5 + M * Kd].view(M, Kd)
    assert a.data_ptr() % 16 != 0
    got = sp.sddmm(s, a, bt.T)
    want = sp.sddmm(s, a.clone(), bt.clone().T)
    assert got.nnz == want.nnz and torch.equal(got.data, want.data)


def test_fused_merge_equals_the_two_launch_form(sp):
    """`x (op) y` through the one-launch merge (in-kernel partition, self-cleaning workspace, pinned output count) and
    through the partition kernel + merge kernel: keys and values bit for bit, for sizes that change the tile count between
    calls (the workspace is shared and must come back zeroed) and for every result density (union, intersection, empty)."""
    from sparse_amd import _umath

    rng = np.random.default_rng(11)
    shape = (300, 400, 50)
    size = int(np.prod(shape))
    for na, nb in ((5000, 7000), (200_000, 150_000), (1, 1), (3000, 0), (600_000, 900_000), (10, 40_000)):
        ka = np.sort(rng.choice(size, na, replace=False)) if na else np.zeros(0, np.int64)
        kb = np.sort(rng.choice(size, nb, replace=False)) if nb else np.zeros(0, np.int64)
        if na and nb:
            kb = np.unique(np.concatenate([kb, ka[::3][:1000]]))   # positions stored in both operands
        x = sp.COO(np.stack(np.unravel_index(ka, shape)), rng.random(len(ka)) - 0.5, shape=shape, has_duplicates=False, sorted=True)
        y = sp.COO(np.stack(np.unravel_index(kb, shape)), rng.random(len(kb)) - 0.5, shape=shape, has_duplicates=False, sorted=True)
        for f in (np.add, np.multiply, np.maximum, np.greater):
            _umath.MERGE_FUSED = True
            a = f(x, y)
            _umath.MERGE_FUSED = False
            try:
                b = f(x, y)
            finally:
                _umath.MERGE_FUSED = True
            assert a.nnz == b.nnz and torch.equal(a.linear_loc(), b.linear_loc()) and torch.equal(a.data, b.data), (na, nb, f)
            assert np.array_equal(a.fill_value, b.fill_value)
    ws = next(iter(_umath._MergeWorkspace._pool.values()))
    assert int(ws.ws.abs().sum()) == 0, "the fused merge must leave its workspace zeroed"


def test_reduction_drops_results_equal_to_the_fill_value_in_one_read(sp):
    """Sums that cancel to exactly 0.0, maxima of negative numbers against a zero fill, products with a stored zero: results
    bit-equal to the result's fill value are not stored (the count comes back with the group count in one host read)."""
    d = np.zeros((6, 8))
    d[0, :2] = (1.5, -1.5)          # cancels exactly
    d[1, 3] = 2.0
    d[2, :] = -1.0                  # max over a full row of negatives stays stored; over a partial row it is the fill 0
    d[3, 1] = -4.0
    d[4, 2:4] = (3.0, 0.25)
    x = sp.COO.from_numpy(d)
    for name, f in (("sum", np.add), ("max", np.maximum), ("min", np.minimum), ("prod", np.multiply)):
        got = f.reduce(x, axis=1)
        want = f.reduce(d, axis=1)
        assert np.array_equal(got.todense(), want), name
        assert got.nnz == np.count_nonzero(want.view(np.uint64) != np.asarray(got.fill_value).view(np.uint64)), name
    s = x.sum(axis=1)
    assert s.nnz == 4 and 0 not in s.coords[0].tolist()


@pytest.mark.parametrize("dtype, N", [(np.float32, 5), (np.float32, 6), (np.float32, 7), (np.float32, 9), (np.float32, 127), (np.float32, 129), (np.float32, 255), (np.float32, 32), (np.float32, 48), (np.float32, 160), (np.float32, 34), (np.float32, 33),
                                      (np.float64, 16), (np.float64, 80), (np.float64, 17)])
def test_narrow_results_through_the_executor(sp, dtype, N, monkeypatch):
    """Results narrower than a whole number of column panels: B is zero-padded to whole panels, C is not - the last panel
    stores its leading columns only (odd float32 widths: the straddling lane stores one column).  Bit-identical to the
    row-group kernel (same k-ascending FMA per output element), the memory just past the result untouched.
    (Round 6: results of 5-12 columns from CSR arrays take the stream kernel in several passes - switched off here: the
    executor still serves these widths for CSC operands without a CSR twin and for B too large for the LDS.)"""
    from bench import make_csr_device
    from sparse_amd import _dot, _kernels

    monkeypatch.setattr(_kernels, "STREAM_MULTI_MAX_N", 4)

    M, K = 70_000, 3000
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    data, idx, ptr = make_csr_device(M, K, 0.01, seed=21, dtype=tdt)
    b = (torch.rand((K, N), device="cuda", dtype=tdt) - 0.5)
    a = sp.GCXS((data, idx, ptr), shape=(M, K), compressed_axes=(0,))
    assert _dot._tiled_eligible(a.data, b, (M, N), K)
    got = a @ b
    assert getattr(a, "_tiled_layouts", None), "must take the inspector/executor path"
    want = _kernels.dot_csr_ndarray((M, N), data, idx, ptr, b, keep_order=True)       # (the row-group kernel)
    assert got.shape == (M, N) and got.is_contiguous() and torch.equal(got, want)
    # straight into a caller's buffer with a guard band behind it (odd float32 widths too since late round 4: the lane whose
    # column pair straddles the end of the row stores its first column alone)
    panel = 128 if dtype == np.float32 else 64
    npad = -(-N // panel) * panel
    bp = torch.zeros((K, npad), device="cuda", dtype=tdt)
    bp[:, :N] = b
    buf = torch.full((M * N + 4096,), 7.0, device="cuda", dtype=tdt)
    _kernels.dot_csr_ndarray_tiled(a._tiled_layouts[tdt], (M, N), K, bp, out=buf[: M * N].view(M, N))
    assert torch.equal(buf[: M * N].view(M, N), want) and bool((buf[M * N:] == 7.0).all())


def test_common_lambdas_run_on_the_device_and_equal_numpy_bit_for_bit(sp):
    """`elemwise` with plain callables over exactly-rounded operations: evaluated on the device from a traced graph (no
    nnz-sized host round trip - the host evaluation is made to fail loudly here), results bit-identical to NumPy on the
    dense arrays, fill values and pruning included; callables outside that set still take the host path."""
    from sparse_amd import _umath

    rng = np.random.default_rng(5)
    shape = (40, 50, 6)

    def arr(density, lo=-0.5, dtype=np.float64):
        d = np.where(rng.random(shape) < density, rng.random(shape) + lo, 0.0)
        return (d * 20).astype(dtype) if np.dtype(dtype).kind == "i" else d.astype(dtype)

    x, y, z = arr(0.3), arr(0.4), arr(0.2)
    xf, yi = arr(0.3, dtype=np.float32), arr(0.5, dtype=np.int32)
    dense = rng.random(shape) + 0.5
    cases = [
        (lambda a, b, c: a * b + c, (x, y, z)),
        (lambda a, b: a * b + 3, (x + 1, y - 2)),
        (lambda a, b, c, d: np.maximum(a, b) - c * d, (x, y, z, 2.0)),
        (lambda a, b: (a > b) & (a != 0), (x, y)),
        (lambda a, b: np.where(a > b, a, b * 2) - 1, (x, y)),
        (lambda a, b: abs(a) ** 2 / (b + 4), (x, yi)),
        (lambda a, b: (a.astype(np.float64) - b) * 0.5, (xf, yi)),
        (lambda a, b, c: a * b * c, (x, dense, y)),
        (lambda a, b: -a + (b < 0), (xf, y)),
        (lambda a: a * 2 + 1, (yi,)),
    ]
    called = []
    orig = _umath._on_device

    def spy(func, *a, **k):
        r = orig(func, *a, **k)
        called.append(r is not None)
        return r

    _umath._on_device = spy
    try:
        for k, (f, operands) in enumerate(cases):
            args = [sp.COO.from_numpy(o) if isinstance(o, np.ndarray) and o is not dense else o for o in operands]
            got = sp.elemwise(f, *args)
            with np.errstate(all="ignore"):
                want = f(*operands)
            assert called[-1] is True, f"case {k} did not take the device path"
            gd = got.todense()
            assert gd.dtype == want.dtype and np.array_equal(gd.view(np.uint8), np.ascontiguousarray(want).view(np.uint8)), k
            assert got.nnz == int(np.count_nonzero(~_same_bits(want, got.fill_value))), k
        got = sp.elemwise(lambda a: np.sin(a) ** 2, sp.COO.from_numpy(x))       # not exactly reproducible: host path
        assert called[-1] is False and np.array_equal(got.todense(), np.sin(x) ** 2)
    finally:
        _umath._on_device = orig


def _same_bits(a, fill):
    a = np.ascontiguousarray(a)
    f = np.asarray(fill).astype(a.dtype)
    if a.dtype == np.dtype(bool):
        return a == f
    u = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return a.view(u) == f.reshape(1).view(u)[0]
