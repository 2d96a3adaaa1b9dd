"""Round-5 additions: the host-side fast paths for the reference's own benchmark sizes (bench_small.py) must give what
the general paths give, bit for bit, and must notice everything the general paths notice (changed buffers, NaNs, settings);
skewed (power-law) operands through the executor against the oracle."""
import os
import sys
import warnings

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


def _bits(t):
    return t.contiguous().view(torch.uint8).cpu().numpy().tobytes() if t.numel() else b""


def _same_gcxs(x, y):
    return (type(x) is type(y) and x.shape == y.shape and x.compressed_axes == y.compressed_axes and x.data.dtype == y.data.dtype
            and x.indices.dtype == y.indices.dtype and _bits(x.data) == _bits(y.data) and _bits(x.indices) == _bits(y.indices)
            and _bits(x.indptr) == _bits(y.indptr) and np.asarray(x.fill_value).tobytes() == np.asarray(y.fill_value).tobytes())


# ---- `_SpmmPlan`: sparse @ dense of a few hundred stored elements ------------------------------------------------------------

@pytest.mark.parametrize("fmt,ca", [("gcxs", (0,)), ("gcxs", (1,)), ("coo", None)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64])
def test_planned_product_equals_the_general_path(sp, orc, fmt, ca, dtype):
    from sparse_amd import _dot

    m, n, p = 300, 200, 70
    x = sp.random((m, n), density=0.01, random_state=3, format="coo")
    x = sp.COO(x.coords, (x.data * 50).to(torch.float64).cpu().numpy().astype(dtype), shape=x.shape)
    if fmt == "gcxs":
        x = x.asformat("gcxs", compressed_axes=ca)
    tdt = {np.float64: torch.float64, np.float32: torch.float32, np.int64: torch.int64}[dtype]
    t = (torch.rand((n, p), device="cuda", dtype=torch.float64) * 9).to(tdt)
    first = x @ t                       # general path; registers the plan
    plans = x.__dict__.get("_mm_plans") or {}
    assert (t.dtype, p) in plans, "a row-group product of a 2-D operand with a device tensor registers its plan"
    calls = []
    real = _dot._SpmmPlan.run
    try:
        _dot._SpmmPlan.run = lambda self, a, b: (calls.append(1), real(self, a, b))[1]
        second = x @ t
        third = sp.matmul(x, t)
    finally:
        _dot._SpmmPlan.run = real
    assert len(calls) == 2 and _bits(first) == _bits(second) == _bits(third)
    xc = x.asformat("coo")
    want = orc.dot_coo_ndarray(xc.coords.cpu().numpy(), xc.data.cpu().numpy(), t.cpu().numpy(), (m, p))
    if np.dtype(dtype).kind == "f":
        assert np.allclose(second.cpu().numpy(), want, rtol=1e-5 if dtype == np.float32 else 1e-13, atol=0)
    else:
        assert np.array_equal(second.cpu().numpy(), want)
    # another width: its own plan, not this one's
    t2 = t[:, :13].contiguous()
    assert _bits(x @ t2) == _bits(first[:, :13].contiguous())


def test_plan_notices_changed_buffers_settings_and_operands(sp):
    from sparse_amd import _settings

    x = sp.random((400, 300), density=0.01, random_state=4, format="gcxs", compressed_axes=(0,))
    t = torch.rand((300, 40), device="cuda", dtype=torch.float64)
    r0 = x @ t
    assert x.__dict__.get("_mm_plans")
    x.data *= 2.0                                   # in place: torch's version counter moves
    r1 = x @ t
    assert torch.equal(r1, r0 * 2.0)
    x.data = x.data * 0.5                           # buffer replaced
    assert torch.equal(x @ t, r0)
    # a non-contiguous / misplaced dense operand takes the general path and gives the same numbers
    tt = torch.rand((40, 300), device="cuda", dtype=torch.float64)
    assert torch.equal(x @ tt.t(), x @ tt.t().contiguous())
    with pytest.raises(Exception):
        x @ torch.rand((299, 40), device="cuda", dtype=torch.float64)
    old = _settings.EXACT_MULADD
    try:
        _settings.EXACT_MULADD = True
        exact = x @ t
        ref = x.todense() @ t.cpu().numpy()
        assert np.allclose(exact.cpu().numpy(), ref, rtol=1e-13)
    finally:
        _settings.EXACT_MULADD = old
    x.fill_value = np.float64(1.0)
    with pytest.raises(ValueError, match="zero fill"):
        x @ t


def test_planned_product_still_warns_about_nan(sp):
    x = sp.random((200, 200), density=0.02, random_state=5, format="gcxs", compressed_axes=(0,))
    t = torch.rand((200, 16), device="cuda", dtype=torch.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        x @ t
        x @ t                                       # planned, silent
    tn = t.clone()
    tn[7, 3] = float("nan")
    with pytest.warns(RuntimeWarning, match="Nan will not be propagated"):
        x @ tn
    x.data[5] = float("nan")                        # A's values changed: the plan steps aside, the general path scans A
    with pytest.warns(RuntimeWarning, match="Nan will not be propagated"):
        x @ t
    with pytest.warns(RuntimeWarning, match="Nan will not be propagated"):
        x @ t                                       # planned again, A's verdict memoised as "has NaN"


# ---- GCXS (x) GCXS in their own layout ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape,ca", [((500,), None), ((60, 70), (0,)), ((60, 70), (1,)), ((12, 9, 14), (0, 2)), ((12, 9, 14), (1,)), ((5, 6, 7, 8), (1, 3))])
@pytest.mark.parametrize("op", ["add", "multiply", "subtract", "maximum", "greater"])
def test_gcxs_elementwise_in_place_of_the_coo_round_trip(sp, shape, ca, op):
    from sparse_amd import _umath

    f = getattr(np, op)
    x = sp.random(shape, density=0.2, random_state=10, format="gcxs", compressed_axes=ca)
    y = sp.random(shape, density=0.2, random_state=11, format="gcxs", compressed_axes=ca)
    y.data -= 0.5
    real = _umath._gcxs_same_layout
    took = []
    try:
        _umath._gcxs_same_layout = lambda *a: None
        want = f(x, y)                          # the general route (also registers the plan of this ufunc / dtype pair)
        _umath._gcxs_same_layout = lambda *a: (took.append(1), real(*a))[1]
        got = f(x, y)
    finally:
        _umath._gcxs_same_layout = real
    assert took and _same_gcxs(got, want)
    assert np.array_equal(got.todense(), f(x.todense(), y.todense()))


def test_gcxs_elementwise_declines_what_it_must(sp):
    from sparse_amd import _umath

    x = sp.random((40, 50), density=0.2, random_state=1, format="gcxs", compressed_axes=(0,))
    y = sp.random((40, 50), density=0.2, random_state=2, format="gcxs", compressed_axes=(1,))
    x + x                                               # plan for add registered
    assert _umath._gcxs_same_layout("add", x, y) is None     # different layouts
    # unsorted column indices inside a row (the constructor takes the caller's arrays as they are)
    d, i, p = x.data.clone(), x.indices.clone(), x.indptr.clone()
    r = int(torch.nonzero(p[1:] - p[:-1] >= 2)[0])
    lo = int(p[r])
    i[lo], i[lo + 1] = i[lo + 1].clone(), i[lo].clone()
    d[lo], d[lo + 1] = d[lo + 1].clone(), d[lo].clone()
    z = sp.GCXS((d, i, p), shape=x.shape, compressed_axes=(0,))
    assert _umath._gcxs_same_layout("add", z, x) is None
    assert np.array_equal((z + x).todense(), (x + x).todense())


# ---- balanced (row-mapped) executor layouts for skewed matrices ----------------------------------------------------------------

def _zipf_csr(M, K, nnz, seed, dtype=torch.float32):
    from bench import make_powerlaw_csr_device

    return make_powerlaw_csr_device(M, K, nnz, seed, dtype=dtype)


@pytest.mark.parametrize("dtype,N", [(torch.float32, 128), (torch.float32, 77), (torch.float64, 64), (torch.float64, 130), (torch.int32, 128)])
def test_balanced_layout_is_bit_identical_to_the_row_group_kernel(sp, orc, dtype, N):
    """Zipf row lengths (the longest rows are full, 30 % are empty): the inspector classes the rows, deals the groups and the
    executor reads / stores through the row map - every output element still adds its terms k-ascending, so the product is
    the row-group kernel's bit for bit in both arithmetic modes, and the oracle's within 1e-6 x sum |a||b|."""
    from sparse_amd import _kernels as K

    M, Kd = 30_011, 2_000          # (not a multiple of the 35-row groups)
    d, i, p = _zipf_csr(M, Kd, 3_000_000, 7, dtype=torch.float32 if dtype == torch.int32 else dtype)
    if dtype == torch.int32:
        d = (d * 100).to(torch.int32)
        b = (torch.rand((Kd, N), device="cuda") * 10).to(torch.int32)
    else:
        b = torch.rand((Kd, N), device="cuda", dtype=dtype)
    old = K.TILED_BALANCE_MIN_NNZ
    try:
        K.TILED_BALANCE_MIN_NNZ = 0
        lay = K.csr_tiled_layout(d, i, p, M, Kd, dtype=dtype)
    finally:
        K.TILED_BALANCE_MIN_NNZ = old
    assert lay.rowmap is not None and K.TILED_BALANCE_STATS["balanced"], K.TILED_BALANCE_STATS
    rm = lay.rowmap[: lay.groups * 35].cpu().numpy()
    used = np.sort(rm[rm >= 0])
    assert np.array_equal(used, np.arange(M)), "every row sits in exactly one slot"
    panel = 64 if dtype == torch.float64 else 128
    npad = -(-N // panel) * panel
    bp = torch.zeros((Kd, npad), dtype=dtype, device="cuda")
    bp[:, :N] = b
    for exact in (False, True):
        got = K.dot_csr_ndarray_tiled(lay, (M, N), Kd, bp, exact=exact)
        want = K.dot_csr_ndarray((M, N), d, i, p, b, exact=exact)
        assert _bits(got) == _bits(want), f"exact={exact}"
    hd, hi, hp, hb = (t.cpu().numpy() for t in (d, i, p, b))
    ref = orc.dot_csr_ndarray((M, N), hd, hi, hp, hb)
    if dtype == torch.int32:
        assert np.array_equal(got.cpu().numpy(), ref)
    else:
        bound = orc.dot_csr_ndarray((M, N), np.abs(hd), hi, hp, np.abs(hb))
        assert np.all(np.abs(K.dot_csr_ndarray_tiled(lay, (M, N), Kd, bp).cpu().numpy() - ref) <= 1e-6 * bound + 1e-30)
        assert _bits(got) == _bits(torch.from_numpy(ref).cuda())       # exact mode: the reference's own bits


def test_balanced_layout_through_the_product_api_and_the_natural_one_for_uniform_matrices(sp):
    from sparse_amd import _dot, _kernels as K
    from bench import make_csr_device

    old = K.TILED_BALANCE_MIN_NNZ
    try:
        K.TILED_BALANCE_MIN_NNZ = 0
        M, Kd, N = 80_000, 3_000, 128
        d, i, p = _zipf_csr(M, Kd, 4_000_000, 9)
        a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
        b = torch.rand((Kd, N), device="cuda")
        r = a @ b
        lay = (getattr(a, "_tiled_layouts", None) or {}).get(torch.float32)
        assert lay is not None and lay.rowmap is not None, "the product API takes the balanced layout for a skewed operand"
        assert torch.equal(r, K.dot_csr_ndarray((M, N), d, i, p, b))
        assert torch.equal(a @ b, r)
        # uniform `random` rows: the natural layout stays (its heaviest group is within TILED_BALANCE_SKEW of the mean)
        du, iu, pu = make_csr_device(M, Kd, 0.015, seed=3)
        lay_u = K.csr_tiled_layout(du, iu, pu, M, Kd)
        assert lay_u.rowmap is None and not K.TILED_BALANCE_STATS["balanced"] and K.TILED_BALANCE_STATS["skew"] < 1.6
        # unsorted column indices in a heavy row: reported like in the natural layout, the product recovers through the key sort
        heavy = int(torch.argmax(p[1:] - p[:-1]))
        lo = int(p[heavy])
        i2, d2 = i.clone(), d.clone()
        i2[lo], i2[lo + 1] = i[lo + 1], i[lo]
        d2[lo], d2[lo + 1] = d[lo + 1], d[lo]
        a2 = sp.GCXS((d2, i2, p), shape=(M, Kd), compressed_axes=(0,))
        assert torch.allclose(a2 @ b, r, rtol=1e-5, atol=1e-5)
    finally:
        K.TILED_BALANCE_MIN_NNZ = old


# ---- the small sparse @ sparse product in one launch -------------------------------------------------------------------------

@pytest.mark.parametrize("dtype,idt", [(np.float64, np.int64), (np.float32, np.int32), (np.int64, np.int64), (np.int32, np.int32)])
@pytest.mark.parametrize("m,k,n,da,db", [(200, 200, 200, 0.01, 0.01), (1000, 1000, 1000, 0.01, 0.01), (37, 500, 4000, 0.2, 0.05),
                                         (300, 64, 7000, 0.5, 0.3), (1, 10, 10, 1.0, 1.0), (500, 300, 33, 0.05, 0.5)])
def test_small_spgemm_equals_the_general_path_and_the_oracle(sp, orc, dtype, idt, m, k, n, da, db):
    from sparse_amd import _kernels as K
    from util import random_csr

    ad, ai, ap = random_csr(m, k, da, 1, dtype, idt, empty_rows=(0,) if m > 3 else ())
    bd, bi, bp = random_csr(k, n, db, 2, dtype, idt, empty_rows=(1,) if k > 3 else ())
    dev = [torch.from_numpy(x).cuda() for x in (ad, ai, ap, bd, bi, bp)]
    if n > int(__import__("sparse_amd")._ffi.lib().spamd_spgemm_small_max_cols(K.code_of(dev[0].dtype))):
        pytest.skip("wider than the kernel's accumulator")
    got = K._spgemm_small(m, n, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5])
    assert got is not None
    old = K.SPGEMM_SMALL
    try:
        K.SPGEMM_SMALL = False
        want = K.dot_csr_csr((m, n), dev[0], dev[3], dev[1], dev[4], dev[2], dev[5])
    finally:
        K.SPGEMM_SMALL = old
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and _bits(g) == _bits(w)
    wd, wi, wp = orc.dot_csr_csr((m, n), ad, bd, ai, bi, ap, bp)
    gd, gi, gp = (t.cpu().numpy() for t in got)
    assert np.array_equal(gp, wp)
    for r in range(m):
        o = np.argsort(wi[wp[r]:wp[r + 1]], kind="stable")      # the reference emits rows in reverse discovery order
        assert np.array_equal(gi[gp[r]:gp[r + 1]], wi[wp[r]:wp[r + 1]][o])
        assert gd[gp[r]:gp[r + 1]].tobytes() == wd[wp[r]:wp[r + 1]][o].tobytes()


def test_small_spgemm_through_the_api_and_its_fallback(sp):
    from sparse_amd import _kernels as K

    x = sp.random((300, 400), density=0.02, random_state=1, format="gcxs", compressed_axes=(0,))
    y = sp.random((400, 250), density=0.02, random_state=2, format="gcxs", compressed_axes=(0,))
    K.SPGEMM_STATS.clear()
    z = x @ y
    assert K.SPGEMM_STATS.get("kernel") == "small"
    assert np.allclose(z.todense(), x.todense() @ y.todense(), rtol=1e-13, atol=0)
    xc, yc = x.asformat("coo"), y.asformat("coo")
    assert np.array_equal((xc @ yc).todense(), z.todense())
    # a B row with unsorted columns: the kernel reports it, the general path (which does not depend on B's order) answers
    d, i, p = y.data.clone(), y.indices.clone(), y.indptr.clone()
    r = int(torch.nonzero(p[1:] - p[:-1] >= 2)[0])
    lo = int(p[r])
    i[lo], i[lo + 1] = i[lo + 1].clone(), i[lo].clone()
    d[lo], d[lo + 1] = d[lo + 1].clone(), d[lo].clone()
    y2 = sp.GCXS((d, i, p), shape=y.shape, compressed_axes=(0,))
    K.SPGEMM_STATS.clear()
    z2 = x @ y2
    assert K.SPGEMM_STATS.get("kernel") != "small"
    assert np.allclose(z2.todense(), z.todense(), rtol=1e-13, atol=0)


def test_tensordot_views_are_kept_and_follow_the_operand(sp):
    """small sparse operands keep the 2-D form a tensordot made of them (`_dot._permute_reshape`): the second contraction re-uses
    it, an in-place change of the operand drops it"""
    x = sp.random((20, 15, 12, 9), density=0.05, random_state=3)
    t = torch.rand((20, 15), device="cuda", dtype=torch.float64)
    r1 = sp.tensordot(x, t, axes=([0, 1], [0, 1]))
    views = x.__dict__.get("_tdot_views")
    assert views and len(views) == 1
    first = next(iter(views.values()))
    r2 = sp.tensordot(x, t, axes=([0, 1], [0, 1]))
    assert next(iter(x.__dict__["_tdot_views"].values())) is first and torch.equal(r1, r2)
    want = np.tensordot(x.todense(), t.cpu().numpy(), axes=([0, 1], [0, 1]))
    assert np.allclose(r1.cpu().numpy(), want, rtol=1e-12)
    x.data *= 3.0
    r3 = sp.tensordot(x, t, axes=([0, 1], [0, 1]))
    assert next(iter(x.__dict__["_tdot_views"].values())) is not first
    assert np.allclose(r3.cpu().numpy(), 3.0 * want, rtol=1e-12)


@pytest.mark.parametrize("shape,axis,density", [((1000, 100, 100), 0, 0.01), ((40, 50, 60, 7), (0, 1), 0.02), ((3, 5000, 40), 0, 0.05), ((70, 300, 30), 0, 0.05),
                                                ((2000, 4), 0, 0.9), ((1500, 3000), 0, 0.002), ((7, 11), 0, 0.5)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64])
def test_reductions_over_leading_axes_merge_the_slabs_instead_of_sorting(sp, shape, axis, density, dtype):
    """csrc/lead_rotate.hip: the kept-axes-first order of a reduction over the leading axes comes from merging the sorted runs
    of the leading indices; results (coordinates, values, fill value) equal the sort's bit for bit - sums are added in the same
    order -, also where a cell range overflows the kernel's arrays (the 2000 x 4 case: 4 cells for 7200 elements) and the
    host takes the sort after all."""
    from sparse_amd import _kernels as K
    rng = np.random.default_rng(hash((shape, str(dtype))) % 2 ** 32)
    x = sp.random(shape, density=density, random_state=rng, dtype=np.float64)
    tdt = {np.float32: torch.float32, np.float64: torch.float64, np.int64: torch.int64}[dtype]
    x = sp.COO(x.coords, (x.data * 100 - 50).to(tdt), shape=shape)
    for red in ("sum", "max", "prod") if dtype != np.int64 else ("sum", "min"):
        K.LEAD_LAST = False
        try:
            want = getattr(x, red)(axis=axis)
        finally:
            K.LEAD_LAST = True
        K.LEAD_LAST_STATS.clear()
        got = getattr(x, red)(axis=axis)
        ax = axis if isinstance(axis, tuple) else (axis,)
        n_slabs = int(np.prod([shape[a] for a in ax]))
        planned = K.lead_last_plan(x.nnz, n_slabs, int(np.prod(shape)) // n_slabs) is not None and x.data.element_size() in (4, 8)
        assert K.LEAD_LAST_STATS.get("calls", 0) == (1 if planned else 0), (red, K.LEAD_LAST_STATS)   # (fewer than 64 runs: the sort)
        assert got.shape == want.shape and got.nnz == want.nnz
        assert torch.equal(got.coords, want.coords)
        assert got.data.dtype == want.data.dtype and _bits(got.data) == _bits(want.data)
        assert np.array_equal(np.asarray(got.fill_value), np.asarray(want.fill_value), equal_nan=True)
    # against NumPy on the dense form (the golden fixtures cover the reference's own cases)
    d = x.todense()
    assert np.array_equal(np.asarray(x.sum(axis=axis).todense()), d.sum(axis=axis)) or dtype != np.int64


@pytest.mark.parametrize("xshape,target", [((30, 1, 40), (30, 25, 40)), ((35, 40), (20, 35, 40)), ((30, 40, 1), (30, 40, 7)),
                                           ((1, 9, 1, 11, 1), (4, 9, 5, 11, 3)), ((6, 1, 1, 7), (6, 3, 2, 7)), ((1, 1), (5, 6)),
                                           ((5, 1, 6, 1, 7), (5, 2, 6, 3, 7)), ((8,), (3, 4, 8)), ((2, 1, 3), (4, 2, 5, 3))])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int8, bool])
def test_broadcast_to_in_one_pass_equals_the_general_construction(sp, xshape, target, dtype):
    """csrc/broadcast.hip writes the replicas of a broadcast operand already sorted when the target's axes are (broadcast)(own)
    (broadcast)(own)(broadcast) groups; other patterns (the (5, 1, 6, 1, 7) case: three own groups) keep the outer-sum + sort
    construction.  Same keys, same values, and NumPy's `broadcast_to` on the dense form."""
    from sparse_amd import _broadcast as B
    rng = np.random.default_rng(len(xshape) * 1000 + sum(target))
    dense = rng.random(xshape)
    dense[rng.random(xshape) < 0.6] = 0
    dense = (dense * 100).astype(dtype)
    x = sp.COO.from_numpy(dense)
    B.BROADCAST_FUSED = False
    try:
        want = sp.broadcast_to(x, target)
    finally:
        B.BROADCAST_FUSED = True
    B.BROADCAST_STATS.clear()
    got = sp.broadcast_to(x, target)
    fused_expected = xshape != (5, 1, 6, 1, 7)
    assert bool(B.BROADCAST_STATS.get("fused")) == (fused_expected and x.nnz > 0), B.BROADCAST_STATS
    assert got.shape == want.shape == tuple(target) and got.nnz == want.nnz
    assert torch.equal(got.linear_loc(), want.linear_loc()) and torch.equal(got.coords, want.coords)
    assert got.data.dtype == want.data.dtype and _bits(got.data) == _bits(want.data)
    assert np.array_equal(np.asarray(got.todense()), np.broadcast_to(dense, target))


def test_elementwise_broadcasting_at_the_reference_benchmark_shapes(sp):
    """benchmarks/test_benchmark_coo.py:69-94: (side, 1, side) op (side, side), COO and GCXS - three C-ABI calls per operation
    (two operand expansions + the fused union) instead of 27."""
    from sparse_amd import _ffi
    side = 300
    for fmt in ("coo", "gcxs"):
        x = sp.random((side, 1, side), density=0.001, random_state=5, format=fmt)
        y = sp.random((side, side), density=0.001, random_state=6, format=fmt)
        xd, yd = np.asarray(x.todense()), np.asarray(y.todense())
        for f, uf in ((lambda a, b: a + b, np.add), (lambda a, b: a * b, np.multiply), (lambda a, b: a > b, np.greater)):
            r = f(x, y)
            assert np.array_equal(np.asarray(r.todense()), uf(xd, yd))
        if fmt == "coo":
            c0 = _ffi.CALLS
            x + y
            assert _ffi.CALLS - c0 <= 4


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64, bool])
def test_full_reductions_return_the_same_0d_array_as_before(sp, dtype):
    """`x.sum()` / `x.max()` / `x.any()`: the 0-d result (no stored element, the value as fill value) is built from one 8-byte
    read; values, dtypes and fill values as NumPy gives them on the dense form, also for a non-zero fill value and with
    `keepdims` (which keeps the general construction)."""
    rng = np.random.default_rng(11)
    dense = rng.random((40, 50))
    dense[rng.random((40, 50)) < 0.7] = 0
    dense = (dense * 100).astype(dtype)
    x = sp.COO.from_numpy(dense)
    for red, npf in (("sum", np.sum), ("max", np.max), ("min", np.min)) if dtype != bool else (("any", np.any), ("all", np.all), ("sum", np.sum)):
        r = getattr(x, red)()
        want = npf(dense)
        assert r.shape == () and r.nnz == 0
        got = np.asarray(r.todense())[()]
        assert got.dtype == np.asarray(want).dtype, (red, got.dtype, np.asarray(want).dtype)
        assert np.array_equal(got, want) or np.allclose(got, want, rtol=1e-6 if dtype == np.float32 else 1e-12), (red, got, want)
        rk = getattr(x, red)(keepdims=True)
        assert rk.shape == (1, 1) and np.allclose(np.asarray(rk.todense()).astype(np.float64), np.asarray(want, dtype=np.float64), rtol=1e-6)
    if dtype == np.float64:
        y = sp.COO.from_numpy(dense + 2.5, fill_value=2.5)
        assert np.allclose(np.asarray(y.sum().todense()), (dense + 2.5).sum(), rtol=1e-12)
        empty = sp.COO.from_numpy(np.zeros((3, 4)))
        assert empty.sum().nnz == 0 and np.asarray(empty.sum().todense()) == 0.0


@pytest.mark.parametrize("shape,ca", [((60, 70), (0,)), ((60, 70), (1,)), ((9, 10, 11), (0, 2)), ((500,), None)])
def test_functions_of_one_gcxs_stay_in_its_layout(sp, shape, ca):
    """`g * 2`, `abs(g)`, `g > 0.5`, `2 ** g`, `g.astype(...)`, `g * 0`: evaluated on the operand's compressed layout
    (`_umath._gcxs_single`) - the same GCXS, bit for bit, as the COO round trip gives, also when the fill value changes
    (`2 ** g`: fill 1) and when results are pruned (`g * 0` keeps nothing, `g > 0.5` some)."""
    from sparse_amd import _umath as U
    g = sp.random(shape, density=0.2, random_state=4, format="gcxs", **({"compressed_axes": ca} if ca is not None else {}))
    g = g - 0.3 * (g > 0.6)       # some negative-free variety: values in (0, 1) and (0.3, 0.7)
    fs = {"mul2": lambda v: v * 2, "abs": lambda v: abs(v), "gt": lambda v: v > 0.5, "pow": lambda v: 2 ** v,
          "f32": lambda v: v.astype(np.float32), "mul0": lambda v: v * 0, "sin": lambda v: np.sin(v), "rsub": lambda v: 1.5 - v}
    for name, f in fs.items():
        U.GCXS_SINGLE = False
        try:
            want = f(g)
        finally:
            U.GCXS_SINGLE = True
        got = f(g)
        assert _same_gcxs(got, want), name
        assert np.array_equal(np.asarray(got.todense()), np.asarray(want.todense()), equal_nan=True)


@pytest.mark.parametrize("ca", [(0,), (1,)])
@pytest.mark.parametrize("axis", [0, 1, (0, 1), -1])
def test_reductions_of_a_2d_gcxs_through_its_own_keys(sp, ca, axis):
    """`GCXS.sum / max(axis=...)` of a matrix: the keys of the compressed layout stand in for the COO form (no conversion per
    call); same GCXS result as through `tocoo()`, and NumPy's on the dense form."""
    g = sp.random((300, 200), density=0.05, random_state=9, format="gcxs", compressed_axes=ca)
    c = g.tocoo()
    d = np.asarray(g.todense())
    for red, npf in (("sum", np.sum), ("max", np.max)):
        for keep in (False, True):
            got = getattr(g, red)(axis=axis, keepdims=keep)
            want = getattr(c, red)(axis=axis, keepdims=keep)
            assert type(got).__name__ == ("GCXS" if got.ndim else "COO") or got.ndim == 0
            gd, wd = np.asarray(got.todense()), np.asarray(want.todense())
            # (every axis reduced: a GCXS sums its values in the order it stores them - compressed_axes=(1,): column-major -,
            # so the sum equals the COO's only up to re-association)
            assert gd.shape == wd.shape and (np.array_equal(gd, wd) or (axis == (0, 1) and np.allclose(gd, wd, rtol=1e-13, atol=0)))
            assert np.allclose(gd, npf(d, axis=axis, keepdims=keep), rtol=1e-12)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64, np.int32, np.int8, bool])
@pytest.mark.parametrize("shape", [(1,), (2047,), (2048,), (2049,), (37, 113), (5, 6, 7, 8), (3000, 700)])
def test_from_numpy_in_one_pass_equals_the_five_pass_construction(sp, dtype, shape):
    """`COO.from_numpy`: the stored elements (not bit-identical to the fill value; -0.0 is stored) and their positions come out
    of one kernel (tile counts + look-back) - same keys and values as flags + scan + iota + two compactions, for a zero and a
    non-zero fill value, for arrays that are all fill and that hold no fill at all."""
    from sparse_amd import _kernels as K
    rng = np.random.default_rng(sum(shape) + np.dtype(dtype).itemsize)
    base = rng.random(shape)
    for frac, fill in ((0.9, 0), (0.5, 3), (1.1, 0), (-1.0, 0)):
        d = (base * 100).astype(dtype)
        d[rng.random(shape) < frac] = np.asarray(fill).astype(dtype)
        if dtype in (np.float64, np.float32) and d.size > 4:
            d.reshape(-1)[1] = -0.0
        fv = np.asarray(fill).astype(dtype)[()]
        K.DENSE_NONFILL = False
        try:
            want = sp.COO.from_numpy(d, fill_value=fv)
        finally:
            K.DENSE_NONFILL = True
        got = sp.COO.from_numpy(d, fill_value=fv)
        assert got.nnz == want.nnz and torch.equal(got.linear_loc(), want.linear_loc())
        assert got.data.dtype == want.data.dtype and _bits(got.data) == _bits(want.data)
        assert np.array_equal(np.asarray(got.todense()), d)


def test_dense_times_coo_keeps_the_transposed_form_and_follows_the_operand(sp):
    """dense @ COO runs on the COO's transpose compressed by rows; that form is kept with the operand (`_csr_of_t`) and rebuilt
    when a stored buffer is written to; sparse and dense results, against NumPy."""
    rng = np.random.default_rng(3)
    a = rng.random((30, 40))
    b = sp.random((40, 50), density=0.05, random_state=7, format="coo")
    at = torch.from_numpy(a).cuda()
    for _ in range(2):
        r = sp.matmul(at, b)
        assert "_csr_of_t" in b.__dict__
        assert np.allclose(np.asarray(r.cpu() if hasattr(r, "cpu") else r), a @ np.asarray(b.todense()), rtol=1e-12)
        rs = sp.tensordot(at, b, axes=1, return_type=sp.COO)
        assert np.allclose(np.asarray(rs.todense()), a @ np.asarray(b.todense()), rtol=1e-12) and isinstance(rs, sp.COO)
    b.data[:5] = 2.0          # (in place: the derived forms must go)
    r = sp.matmul(at, b)
    assert np.allclose(np.asarray(r.cpu() if hasattr(r, "cpu") else r), a @ np.asarray(b.todense()), rtol=1e-12)
