"""A1, results of at most 4 columns: the STREAM form (`sparse_amd/csrc/spmm_stream.hip`) of `_dot_csr_ndarray`
(reference sparse/numba_backend/_common.py:720-755) against the CPU oracle, through the C ABI (`spamd_spmm_csr_stream`
directly, so that small matrices reach it too, and `spamd_spmm_csr`'s own dispatch at a size that takes it)."""
import numpy as np
import pytest
import torch

from util import assert_within_fma_bound, random_csr, random_dense

pytestmark = pytest.mark.gpu


def _stream(M, K, N, data, idx, ptr, b, mult=0, known_nnz=True, passes=1):
    from sparse_amd import _ffi
    from sparse_amd._device import code_of, ptr as p_, stream_ptr

    d = torch.device("cuda")
    td, ti, tp, tb = (torch.from_numpy(np.ascontiguousarray(x)).to(d) for x in (data, idx, ptr, b))
    out = torch.full((M, N), -7, dtype=td.dtype, device=d)   # every element must be written
    assert _ffi.lib().spamd_spmm_csr_stream_fits(code_of(td.dtype), M, K, N, p_(td), p_(ti)) == passes
    nnz_arg = len(data) if known_nnz else -1
    _ffi.call("spamd_spmm_csr_stream", code_of(td.dtype), code_of(ti.dtype), M, K, N, p_(td), p_(ti), p_(tp), p_(tb), N,
              p_(out), N, nnz_arg, mult << 8, stream_ptr(d))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _check(orc, M, K, N, data, idx, ptr, b, **kw):
    got = _stream(M, K, N, data, idx, ptr, b, **kw)
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    if np.dtype(data.dtype).kind == "f":
        assert_within_fma_bound(got, want, data, idx, ptr, b)
    else:
        assert got.dtype == want.dtype and np.array_equal(got, want)
    return got


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("N", [1, 2, 3, 4])
def test_stream_vs_oracle(orc, dtype, idt, N):
    M, K = 5000, 900
    data, idx, ptr = random_csr(M, K, 0.02, 11 + N, dtype, idt)
    # (8-byte values: a pass holds at most 3 columns - 32-byte rows of B with 64-bit indices do not fit the kernel's 128
    # registers -, so 4 columns are two passes since round 6; declined before)
    passes = 2 if N == 4 and np.dtype(dtype).itemsize == 8 else 1
    _check(orc, M, K, N, data, idx, ptr, random_dense(K, N, 5, dtype), passes=passes)


@pytest.mark.parametrize("density,M,K", [(0.0, 300, 50), (0.0005, 40000, 300), (0.004, 30000, 700), (0.3, 3000, 1000),
                                         (1.0, 700, 1200)])
@pytest.mark.parametrize("N", [1, 4])
def test_row_lengths_from_empty_to_dense(orc, density, M, K, N):
    """rows of 0 / ~0.15 / ~3 / 300 / 1200 elements: windows of row ends exhausted several times per subtile (phase
    A's and B's window loops) up to rows that span five subtiles (the carry between subtiles)."""
    data, idx, ptr = random_csr(M, K, density, 3, np.float32, np.int32)
    got = _check(orc, M, K, N, data, idx, ptr, random_dense(K, N, 9, np.float32))
    if density == 0.0:
        assert not got.any()


def test_empty_row_runs_and_a_long_row(orc):
    M, K = 9000, 2000
    empty = tuple(range(0, 200)) + tuple(range(4000, 4300)) + tuple(range(8990, 9000)) + (777, 779)
    data, idx, ptr = random_csr(M, K, 0.01, 4, np.float64, np.int64, empty_rows=empty, long_row=5000)
    got = _check(orc, M, K, 2, data, idx, ptr, random_dense(K, 2, 2, np.float64))
    assert not got[list(empty)].any()


def test_misaligned_piece_starts_and_partial_last_vector(orc):
    """nnz % 4 != 0 (the array's last 16-byte vector is partial) and many pieces (a wave's piece starts at a row start,
    almost never on a 16-byte boundary)."""
    for seed, M in ((1, 2049), (2, 2050), (3, 2051)):
        K = 333
        data, idx, ptr = random_csr(M, K, 0.03, seed, np.float32, np.int32)
        if len(data) % 4 == 0:
            data, idx = data[:-1], idx[:-1]
            ptr = np.minimum(ptr, len(data)).astype(ptr.dtype)
        _check(orc, M, K, 1, data, idx, ptr, random_dense(K, 1, 8, np.float32))
        _check(orc, M, K, 3, data, idx, ptr, random_dense(K, 3, 8, np.float32), mult=3, known_nnz=False)


def test_special_values_do_not_leak(orc):
    """inf / nan in rows of B that a wave's masked lanes (outside its piece) would read must not reach the sums"""
    M, K = 4000, 64
    data, idx, ptr = random_csr(M, K, 0.1, 6, np.float32, np.int32)
    b = random_dense(K, 1, 7, np.float32)
    b[0, 0] = np.inf          # masked lanes gather row 0
    keep = idx != 0
    rows = np.repeat(np.arange(M), np.diff(ptr))[keep]
    data, idx = data[keep], idx[keep]
    ptr = np.zeros(M + 1, ptr.dtype)
    np.cumsum(np.bincount(rows, minlength=M), out=ptr[1:])
    got = _stream(M, K, 1, data, idx, ptr, b)
    assert np.isfinite(got).all()
    b[0, 0] = 0.0
    assert np.array_equal(got, _stream(M, K, 1, data, idx, ptr, b))


def test_run_twice_identical_and_dispatch_takes_the_stream_form(orc):
    from sparse_amd import _ffi, _kernels as Kn

    M, K, N = 70000, 500, 1
    data, idx, ptr = random_csr(M, K, 0.02, 12, np.float32, np.int32)
    b = random_dense(K, N, 13, np.float32)
    a = _stream(M, K, N, data, idx, ptr, b)
    assert np.array_equal(a, _stream(M, K, N, data, idx, ptr, b))
    d = torch.device("cuda")
    args = [torch.from_numpy(x).to(d) for x in (data, idx, ptr, b)]
    via = Kn.dot_csr_ndarray((M, N), *args).cpu().numpy()          # M >= 32768: spamd_spmm_csr routes here
    assert np.array_equal(via, a)
    rv = Kn.dot_csr_ndarray((M, N), *args, rowvec=True).cpu().numpy()  # the row-vector kernel it replaces
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert_within_fma_bound(rv, want, data, idx, ptr, b)
    assert_within_fma_bound(via, want, data, idx, ptr, b)


@pytest.mark.parametrize("N", [1, 2, 3, 4])
def test_strided_dense_operand_and_result(orc, N):
    """ldb > N (B is a column slice of a wider matrix: copied into LDS value by value) and ldo > N (the result is a column
    slice: stored value by value, the columns beside it untouched) through the C ABI"""
    from sparse_amd import _ffi
    from sparse_amd._device import code_of, ptr as p_, stream_ptr

    M, K, ldb, ldo = 3000, 500, 7, 6
    data, idx, ptr = random_csr(M, K, 0.03, 21, np.float32, np.int32)
    wide = random_dense(K, ldb, 22, np.float32)
    d = torch.device("cuda")
    td, ti, tp, tb = (torch.from_numpy(np.ascontiguousarray(x)).to(d) for x in (data, idx, ptr, wide))
    out = torch.full((M, ldo), -7.0, dtype=torch.float32, device=d)
    bview = tb[:, 2:]                       # first value of the slice: element 2 of every row (8-byte aligned only)
    _ffi.call("spamd_spmm_csr_stream", code_of(td.dtype), code_of(ti.dtype), M, K, N, p_(td), p_(ti), p_(tp),
              bview.data_ptr(), ldb, out.data_ptr() + 4, ldo, len(data), 0, stream_ptr(d))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    b = np.ascontiguousarray(wide[:, 2:2 + N])
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert_within_fma_bound(np.ascontiguousarray(got[:, 1:1 + N]), want, data, idx, ptr, b)
    assert (got[:, 0] == -7.0).all() and (got[:, 1 + N:] == -7.0).all()


# ---- results of 5-12 columns: several passes of the stream kernel over chunks of columns (round 6) ----------------------------

@pytest.mark.parametrize("dtype, itype, N", [(torch.float32, torch.int32, 5), (torch.float32, torch.int32, 8), (torch.float32, torch.int64, 12),
                                             (torch.float64, torch.int64, 4), (torch.float64, torch.int32, 6), (torch.int32, torch.int32, 7),
                                             (torch.int64, torch.int64, 5)])
def test_stream_kernel_in_several_passes(dtype, itype, N):
    """`dot_csr_ndarray` / `spamd_spmm_csr` take results of up to 12 columns (8-byte values: 6) through the stream kernel,
    4 (3) columns a pass, b and out strided: every column as the one-pass kernel computes it alone (bit-identical: a column's
    arithmetic does not depend on its neighbours... of the same chunk width) and within tolerance of the k-ascending sums"""
    from bench import make_csr_device
    from sparse_amd import _kernels as K

    M, Kd = 70_000, 3000
    data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=11, dtype=torch.float32 if not dtype.is_floating_point else dtype)
    if not dtype.is_floating_point:
        data = (data * 200 - 100).to(dtype)
    idx, ptr = idx.to(itype), ptr.to(itype)
    b = (torch.rand((Kd, N), device="cuda", dtype=torch.float64) - 0.5)
    b = b.to(dtype) if dtype.is_floating_point else (b * 50).to(dtype)
    assert K.stream_passes(M, Kd, N, dtype, data, idx) >= 2
    got = K.dot_csr_ndarray((M, N), data, idx, ptr, b)
    want = K.dot_csr_ndarray((M, N), data, idx, ptr, b, keep_order=True)       # the row-group kernel: k-ascending sums
    if dtype.is_floating_point:
        scale = K.dot_csr_ndarray((M, N), data.abs(), idx, ptr, b.abs(), keep_order=True)
        tol = 1e-6 if dtype == torch.float32 else 1e-14
        assert float(((got - want).abs() / scale.clamp_min(1e-30)).max()) < tol
    else:
        assert torch.equal(got, want)
    # through the C ABI's own dispatch (nnz unknown there) and again: the same bits
    out2 = torch.empty_like(got)
    from sparse_amd import _ffi
    from sparse_amd._device import code_of, ptr as dptr, stream_ptr
    _ffi.call("spamd_spmm_csr", code_of(dtype), code_of(itype), M, Kd, N, dptr(data), dptr(idx), dptr(ptr), dptr(b), N, dptr(out2), N, 0,
              stream_ptr(got.device))
    assert torch.equal(out2, got)


def test_products_of_a_few_columns_take_the_stream_kernel_not_the_executor():
    """`a @ b` with 8 columns at a size where the tiled executor would be eligible: no block stream is built"""
    import sparse_amd as sp
    from bench import make_csr_device

    M, Kd = 300_000, 4000
    d, i, p = make_csr_device(M, Kd, 0.01, seed=5)
    a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
    b = torch.rand((Kd, 8), device="cuda") - 0.5
    c = a @ b
    assert not getattr(a, "_tiled_layouts", None)
    b128 = torch.zeros((Kd, 128), device="cuda")
    b128[:, :8] = b
    ref = (a @ b128)[:, :8]
    assert getattr(a, "_tiled_layouts", None)
    scale = (sp.GCXS((d.abs(), i, p), shape=(M, Kd), compressed_axes=(0,)) @ b128.abs())[:, :8]
    assert float(((c - ref).abs() / scale.clamp_min(1e-30)).max()) < 1e-6
