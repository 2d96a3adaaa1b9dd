"""csrc/spgemm_bitmap.hip (round 4: bitmap-ordered rows, written in place through a look-back over the rows) against the
bucket kernels of csrc/spgemm_rows.hip and the global expand-sort-compress - bit for bit: row pointers, column indices and
values (the products of an output element are added in the order of A's elements, the reference's `sums[j] += ...`,
_common.py:690-705) - and against the oracle's restatement of `_dot_csr_csr` on small cases."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _csr(m, n, per_row, seed, dtype=np.float32, idt=np.int32, empty_every=0, hot_col=None):
    """random CSR, ~per_row (Poisson-like spread) sorted distinct columns per row"""
    rng = np.random.default_rng(seed)
    width = int(per_row * 1.3) + 4
    cols = np.sort(rng.integers(0, n, size=(m, width)), axis=1)
    keep = np.ones((m, width), dtype=bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    keep &= rng.random((m, width)) < per_row / width
    if hot_col is not None:     # a column present in EVERY row: many products of one output element
        keep &= cols != hot_col
        cols[:, 0], keep[:, 0] = hot_col, True
        o = np.argsort(cols, axis=1, kind="stable")
        cols, keep = np.take_along_axis(cols, o, 1), np.take_along_axis(keep, o, 1)
    if empty_every:
        keep[::empty_every] = False
    counts = keep.sum(axis=1)
    indptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = cols[keep]
    vals = rng.random(indices.size) - 0.3
    if np.dtype(dtype).kind == "f":
        data = vals.astype(dtype)
    else:
        data = (vals * 40).astype(dtype)
        data[data == 0] = 1
    return data, indices.astype(idt), indptr.astype(idt)


SPLIT_RUNS = []     # `parts` of every product the split form took (checked at the end: it must have run)


def _dev(t):
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in t)


def _both(shape, A, B):
    """(bitmap result or None, bucket-kernel result) for A (m x k) @ B (k x n), host triplets"""
    from sparse_amd import _kernels as K

    (ad, ai, ap), (bd, bi, bp) = _dev(A), _dev(B)
    old = K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN, K.SPGEMM_BITMAP_MAX_DUPS
    try:
        K.SPGEMM_BITMAP = False
        want = K._spgemm_rows(shape[0], shape[1], ad, ai, ap, bd, bi, bp)
        if want is None:
            want = K.dot_csr_csr(shape, ad, bd, ai, bi, ap, bp)
        K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN, K.SPGEMM_BITMAP_MAX_DUPS = True, 0, 10 ** 9
        # both forms of the kernel: whole rows (one workgroup per CU) and column ranges (two per CU; 4-byte values)
        K.SPGEMM_BITMAP_SPLIT = False
        K.SPGEMM_STATS.clear()
        got = K._spgemm_rows(shape[0], shape[1], ad, ai, ap, bd, bi, bp)
        used = K.SPGEMM_STATS.get("kernel")
        K.SPGEMM_BITMAP_SPLIT = "first"
        K.SPGEMM_STATS.clear()
        got2 = K._spgemm_rows(shape[0], shape[1], ad, ai, ap, bd, bi, bp)
        if K.SPGEMM_STATS.get("kernel") == "bitmap" and K.SPGEMM_STATS.get("parts", 1) > 1:
            SPLIT_RUNS.append(K.SPGEMM_STATS["parts"])
            _same(got2, want)
    finally:
        K.SPGEMM_BITMAP_SPLIT = True
        K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN, K.SPGEMM_BITMAP_MAX_DUPS = old
    return got, want, used


def _same(got, want):
    for g, w in zip(got, want):
        g, w = g.cpu().numpy(), w.cpu().numpy()
        assert g.shape == w.shape and g.dtype == w.dtype, (g.shape, w.shape, g.dtype, w.dtype)
        assert np.array_equal(g.view(np.uint8), w.view(np.uint8))


@pytest.mark.parametrize("dtype,idt", [(np.float32, np.int32), (np.float32, np.int64), (np.float64, np.int64),
                                       (np.int32, np.int32), (np.int64, np.int64)])
def test_bitmap_rows_equal_the_bucket_kernels(dtype, idt):
    """config 5 in small: ~100 x ~100 products per row, 10^6 columns; more rows than CUs, some of them empty"""
    m, k, n = 1500, 40_000, 1_000_000
    per_a = 100 if np.dtype(dtype).itemsize == 4 else 60        # (8-byte values: rows of at most 8192 products)
    A = _csr(m, k, per_a, 1, dtype, idt, empty_every=97)
    B = _csr(k, n, 100, 2, dtype, idt, empty_every=13)
    got, want, used = _both((m, n), A, B)
    assert used == "bitmap"
    _same(got, want)
    assert int(got[2][-1]) == got[0].numel() == got[1].numel()


def test_bitmap_rows_with_many_products_per_output_element():
    """a column of B that every row holds: ~100 products of ONE output element per row, added in the order of A's elements
    (the parked-product list at work), plus the usual handful of pairs"""
    m, k, n = 700, 5_000, 1_000_000
    A = _csr(m, k, 100, 3, np.float32, np.int32)
    B = _csr(k, n, 90, 4, np.float32, np.int32, hot_col=12345)
    got, want, used = _both((m, n), A, B)
    assert used == "bitmap"
    _same(got, want)
    B64 = _csr(k, n, 60, 5, np.float64, np.int64, hot_col=999_999)
    A64 = _csr(m, k, 90, 6, np.float64, np.int64)
    got, want, used = _both((m, n), A64, B64)
    assert used == "bitmap"
    _same(got, want)


def test_bitmap_kernel_declines_what_its_list_cannot_hold_and_the_caller_falls_back():
    """few columns: thousands of products share output elements - the kernel sets its `failed` word, the host takes the
    bucket kernels, the result is the same"""
    m, k, n = 300, 4_000, 20_000
    A = _csr(m, k, 100, 7)
    B = _csr(k, n, 100, 8)
    got, want, used = _both((m, n), A, B)
    assert used == "buckets"
    _same(got, want)


@pytest.mark.parametrize("m", [1, 2, 255, 257, 600])
def test_bitmap_rows_few_and_odd_row_counts(m):
    k, n = 3_000, 1_048_576
    A = _csr(m, k, 120, 10 + m)
    B = _csr(k, n, 110, 11)
    got, want, used = _both((m, n), A, B)
    assert used == "bitmap"
    _same(got, want)


def test_bitmap_rows_at_the_limits():
    """A rows of exactly 256 elements, rows near 16384 products, n_col = 2^20, and a matrix one step beyond each limit
    (which the host must not hand to the kernel)"""
    from sparse_amd import _ffi

    lim = _ffi.lib().spamd_spgemm_bitmap_limits
    assert (lim(_ffi.F32, 0), lim(_ffi.F32, 1), lim(_ffi.F32, 2), lim(_ffi.F32, 3)) == (16384, 256, 1 << 20, 512)
    assert lim(_ffi.F64, 0) == 8192
    m, k, n = 300, 2_000, 1 << 20
    rng = np.random.default_rng(0)
    cols = np.stack([np.sort(rng.choice(k, 256, replace=False)) for _ in range(m)]).astype(np.int32)
    A = ((rng.random(m * 256) + 0.1).astype(np.float32), cols.reshape(-1), (np.arange(m + 1) * 256).astype(np.int32))
    B = _csr(k, n, 60, 12)
    got, want, used = _both((m, n), A, B)
    assert used == "bitmap"
    _same(got, want)
    cols = np.stack([np.sort(rng.choice(k, 257, replace=False)) for _ in range(m)]).astype(np.int32)
    A2 = ((rng.random(m * 257) + 0.1).astype(np.float32), cols.reshape(-1), (np.arange(m + 1) * 257).astype(np.int32))
    got, want, used = _both((m, n), A2, B)
    assert used == "buckets"
    _same(got, want)


def test_bitmap_rows_against_the_oracle(orc):
    m, k, n = 64, 500, 70_000
    A = _csr(m, k, 40, 20, np.float64, np.int64)
    B = _csr(k, n, 50, 21, np.float64, np.int64, hot_col=7)
    got, want, used = _both((m, n), A, B)
    assert used == "bitmap"
    wd, wi, wp = orc.dot_csr_csr((m, n), A[0], B[0], A[1], B[1], A[2], B[2])
    gd, gi, gp = (t.cpu().numpy() for t in got)
    assert np.array_equal(gp, wp)
    for r in range(m):
        o = np.argsort(wi[wp[r]:wp[r + 1]], kind="stable")       # the reference emits rows in reverse discovery order
        assert np.array_equal(gi[gp[r]:gp[r + 1]], wi[wp[r]:wp[r + 1]][o])
        assert np.array_equal(gd[gp[r]:gp[r + 1]].view(np.uint64), wd[wp[r]:wp[r + 1]][o].view(np.uint64))


def test_more_column_ranges_than_two_and_the_split_form_ran():
    """n_col = 3 * 10^6: beyond the wide form's bitmap, six or more column ranges"""
    from sparse_amd import _kernels as K

    m, k, n = 900, 20_000, 3_000_000
    A = _csr(m, k, 110, 40)
    B = _csr(k, n, 100, 41, empty_every=11)
    (ad, ai, ap), (bd, bi, bp) = _dev(A), _dev(B)
    old = K.SPGEMM_BITMAP
    try:
        K.SPGEMM_BITMAP = False
        want = K._spgemm_rows(m, n, ad, ai, ap, bd, bi, bp)
        K.SPGEMM_BITMAP = True
        K.SPGEMM_STATS.clear()
        got = K._spgemm_rows(m, n, ad, ai, ap, bd, bi, bp)
        assert K.SPGEMM_STATS.get("kernel") == "bitmap" and K.SPGEMM_STATS.get("parts", 1) >= 6, K.SPGEMM_STATS
        SPLIT_RUNS.append(K.SPGEMM_STATS["parts"])
    finally:
        K.SPGEMM_BITMAP = old
    _same(got, want)
    assert SPLIT_RUNS, "the column-range form must have taken some of this module's products"


@pytest.mark.parametrize("dtype,idt", [(np.float64, np.int64), (np.int64, np.int32)])
def test_eight_byte_values_of_config5_shape_take_column_ranges(dtype, idt):
    """the reference's default value type at config 5's row shape (100 x 100 products): beyond the whole-row form's 8192
    products for 8-byte values, so the row goes in three or more column ranges"""
    from sparse_amd import _kernels as K

    m, k, n = 1200, 30_000, 1_000_000
    A = _csr(m, k, 100, 50, dtype, idt, empty_every=101)
    B = _csr(k, n, 100, 51, dtype, idt)
    (ad, ai, ap), (bd, bi, bp) = _dev(A), _dev(B)
    old = K.SPGEMM_BITMAP
    try:
        K.SPGEMM_BITMAP = False
        want = K._spgemm_rows(m, n, ad, ai, ap, bd, bi, bp)
        K.SPGEMM_BITMAP = True
        K.SPGEMM_STATS.clear()
        got = K._spgemm_rows(m, n, ad, ai, ap, bd, bi, bp)
        assert K.SPGEMM_STATS.get("kernel") == "bitmap" and K.SPGEMM_STATS.get("parts", 1) >= 3, K.SPGEMM_STATS
    finally:
        K.SPGEMM_BITMAP = old
    _same(got, want)


@pytest.mark.parametrize("seed", range(8))
def test_bitmap_forms_on_random_shapes(seed):
    """randomised cross-check of both forms against the bucket kernels: row counts, inner sizes, column counts up to 6 x 10^6,
    row lengths, value and index types, empty rows; whatever form the host policy picks must give the bucket kernels' bits"""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(1, 1500))
    k = int(rng.integers(200, 30_000))
    n = int(rng.choice([70_000, 300_000, 1_000_000, 1_048_576, 2_500_000, 6_000_000]))
    per_a = int(rng.integers(5, 140))
    per_b = int(rng.integers(5, 140))
    dtype, idt = [(np.float32, np.int32), (np.float64, np.int64), (np.int32, np.int64), (np.float32, np.int64)][seed % 4]
    A = _csr(m, k, min(per_a, k // 2), 2000 + seed, dtype, idt, empty_every=int(rng.integers(0, 9)))
    B = _csr(k, n, per_b, 3000 + seed, dtype, idt, empty_every=int(rng.integers(0, 9)))
    got, want, used = _both((m, n), A, B)
    _same(got, want)      # (`used` may be "buckets": few columns with long rows park more products than the list holds)


def test_product_api_takes_the_bitmap_kernel_and_is_reproducible():
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    n = 60_000
    g = sp.random((n, 1_000_000), density=1e-4, random_state=3, dtype=np.float32, idx_dtype=np.int32, format="gcxs",
                  compressed_axes=(0,))
    rows = 4000
    p1 = int(g.indptr[rows])
    a = sp.GCXS((g.data[:p1].contiguous(), (g.indices[:p1] % n).contiguous(), g.indptr[:rows + 1].contiguous()), shape=(rows, n),
                compressed_axes=(0,))
    # (column indices folded into B's row range; duplicates inside a row are possible but rare - rebuild canonically)
    coo = a.tocoo()
    a = sp.GCXS(sp.COO(coo.coords, coo.data, shape=coo.shape), compressed_axes=(0,))    # (duplicates summed, sorted)
    K.SPGEMM_STATS.clear()
    c1 = a @ g
    assert K.SPGEMM_STATS.get("kernel") == "bitmap", K.SPGEMM_STATS
    c2 = a @ g
    for x, y in ((c1.data, c2.data), (c1.indices, c2.indices), (c1.indptr, c2.indptr)):
        assert torch.equal(x, y)


def _dense_of(t, shape):
    d, i, p = (np.asarray(x) for x in t)
    out = np.zeros(shape, dtype=np.float64)
    np.add.at(out, (np.repeat(np.arange(shape[0]), np.diff(p)), i), d.astype(np.float64))
    return out


def _run_forms(shape, A, B, split):
    """the product through `_spgemm_rows` with the bitmap kernel forced on; returns (result, stats)"""
    from sparse_amd import _kernels as K

    (ad, ai, ap), (bd, bi, bp) = _dev(A), _dev(B)
    old = K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN, K.SPGEMM_BITMAP_MAX_DUPS, K.SPGEMM_BITMAP_SPLIT
    try:
        K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN, K.SPGEMM_BITMAP_MAX_DUPS, K.SPGEMM_BITMAP_SPLIT = True, 0, 10 ** 9, split
        K.SPGEMM_STATS.clear()
        got = K._spgemm_rows(shape[0], shape[1], ad, ai, ap, bd, bi, bp)
        if got is None:
            got = K.dot_csr_csr(shape, ad, bd, ai, bi, ap, bp)
        return got, dict(K.SPGEMM_STATS)
    finally:
        K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN, K.SPGEMM_BITMAP_MAX_DUPS, K.SPGEMM_BITMAP_SPLIT = old


@pytest.mark.parametrize("split", [False, "first"])
def test_b_rows_with_unsorted_columns(split):
    """`GCXS((data, indices, indptr))` takes the caller's arrays as they are: B rows whose columns do not ascend.  The wide
    form does not depend on B's order; the split form's binary search over a B row does - a product then lands outside its
    part's column range, is dropped before it touches the bitmap, the call fails and the next form takes the product
    (round-4 advice: an out-of-range `atomicOr` in LDS and a silently wrong result before)."""
    m, k, n = 400, 3_000, 1_000_000
    A = _csr(m, k, 80, 50)
    bd, bi, bp = _csr(k, n, 90, 51)
    rng = np.random.default_rng(5)
    bd, bi = bd.copy(), bi.copy()
    for r in range(k):                      # every B row in a random order
        lo, hi = int(bp[r]), int(bp[r + 1])
        o = rng.permutation(hi - lo)
        bd[lo:hi], bi[lo:hi] = bd[lo:hi][o], bi[lo:hi][o]
    got, stats = _run_forms((m, n), A, (bd, bi, bp), split)
    if split == "first":
        assert stats.get("bitmap_failed", 0) >= 1, stats     # the split form declined ...
    assert stats.get("kernel") == "bitmap" and stats.get("parts", 1) == 1, stats   # ... and whole rows took it
    want, _ = _run_forms((m, n), A, _csr(k, n, 90, 51), False)     # the same matrix in canonical order
    _same(got, want)


@pytest.mark.parametrize("split", [False, "first"])
def test_b_rows_holding_a_column_twice_fail_over_to_the_bucket_kernels(split):
    """a B row with the same column twice gives two products with EQUAL (column, A element) keys: the parked-product walk
    finds fewer distinct keys than entries (`__ballot` = 0; round-4 advice: `ctz(0)` + a garbage `readlane` before).  Both
    forms now fail the call; the bucket kernels sum both products, as the reference's `sums[j] += ...` does."""
    m, k, n = 300, 2_000, 1_000_000
    A = _csr(m, k, 70, 60, np.float64, np.int64)
    bd, bi, bp = _csr(k, n, 60, 61, np.float64, np.int64)
    bi = bi.copy()
    for r in range(0, k, 3):
        lo, hi = int(bp[r]), int(bp[r + 1])
        if hi - lo >= 2:
            bi[lo + 1] = bi[lo]             # (still ascending: only the duplicate makes the operand non-canonical)
    got, stats = _run_forms((m, n), A, (bd, bi, bp), split)
    assert stats.get("bitmap_failed", 0) >= 1 and stats.get("kernel") != "bitmap", stats
    gd, gi, gp = (t.cpu().numpy() for t in got)
    dense = np.zeros((m, n // 1000 + 1))  # (checked on a column sample: the dense product has 3 x 10^8 cells)
    keep = bi % 1000 == 0
    b_rows = np.repeat(np.arange(k), np.diff(bp))[keep]
    b_s = np.zeros((k, n // 1000 + 1))
    np.add.at(b_s, (b_rows, bi[keep] // 1000), bd[keep])
    want = _dense_of(A, (m, k)) @ b_s
    rows = np.repeat(np.arange(m), np.diff(gp))
    sel = gi % 1000 == 0
    np.add.at(dense, (rows[sel], gi[sel] // 1000), gd[sel])
    assert np.allclose(dense, want, rtol=1e-12, atol=1e-12)
