"""A1 parity: HIP CSR x dense SpMM vs the CPU oracle (`_dot_csr_ndarray`,
reference sparse/numba_backend/_common.py:720-755), through the C ABI."""
import numpy as np
import pytest
import torch

from util import random_csr, random_dense

pytestmark = pytest.mark.gpu

F32_RTOL = 1e-6  # north_star tolerance, relative to sum_k |a_k b_k| (Appendix C.5)


def _run(M, K, N, density, dtype, idt, exact, seed=0, **kw):
    import sparse_amd
    from sparse_amd import _kernels as Kn

    data, idx, ptr = random_csr(M, K, density, seed, dtype, idt, **kw)
    b = random_dense(K, N, seed + 1, dtype)
    d = torch.device("cuda")
    out = Kn.dot_csr_ndarray((M, N), torch.from_numpy(data).to(d), torch.from_numpy(idx).to(d),
                             torch.from_numpy(ptr).to(d), torch.from_numpy(b).to(d), exact=exact)
    torch.cuda.synchronize()
    return (data, idx, ptr, b), out.cpu().numpy()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("N", [1, 7, 16, 64, 128, 130, 512])
def test_exact_mode_is_bit_identical(orc, dtype, idt, N):
    (data, idx, ptr, b), got = _run(2000, 1000, N, 0.01, dtype, idt, exact=True)
    want = orc.dot_csr_ndarray((2000, N), data, idx, ptr, b)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N", [3, 32, 128, 200])
def test_fma_mode_within_tolerance(orc, dtype, N):
    (data, idx, ptr, b), got = _run(3000, 1500, N, 0.02, dtype, np.int32, exact=False)
    want = orc.dot_csr_ndarray((3000, N), data, idx, ptr, b)
    # error is measured against sum_k |a_k b_k| computed in float64
    import scipy.sparse as sps

    absum = sps.csr_matrix((np.abs(data).astype(np.float64), idx, ptr), shape=(3000, 1500)) @ np.abs(b).astype(np.float64)
    tol = F32_RTOL if dtype == np.float32 else 1e-14
    assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= tol * absum + 1e-300)


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_integers_exact(orc, dtype):
    (data, idx, ptr, b), got = _run(1000, 700, 33, 0.03, dtype, np.int64, exact=False)
    want = orc.dot_csr_ndarray((1000, 33), data, idx, ptr, b)
    assert got.dtype == want.dtype
    assert np.array_equal(got, want)


def test_empty_rows_long_row_and_empty_matrix(orc):
    (data, idx, ptr, b), got = _run(500, 3000, 128, 0.01, np.float32, np.int32, exact=True,
                                    empty_rows=(0, 7, 8, 499), long_row=250)
    want = orc.dot_csr_ndarray((500, 128), data, idx, ptr, b)
    assert np.array_equal(got, want)
    assert not got[0].any() and not got[499].any()
    # nnz == 0: every output element must still be written (zeros)
    (data, idx, ptr, b), got = _run(64, 32, 128, 0.0, np.float32, np.int32, exact=False)
    assert got.shape == (64, 128) and not got.any()


def test_nan_inf_in_dense_only_touch_referenced_rows(orc):
    data, idx, ptr = random_csr(300, 200, 0.05, 3, np.float32, np.int32)
    b = random_dense(200, 128, 4, np.float32)
    unused = np.setdiff1d(np.arange(200), idx)
    b[5, 3] = np.inf
    if len(unused):
        b[unused[0], :] = np.nan  # never referenced: must not leak (0 * nan) into the output
    d = torch.device("cuda")
    from sparse_amd import _kernels as Kn

    got = Kn.dot_csr_ndarray((300, 128), *(torch.from_numpy(x).to(d) for x in (data, idx, ptr, b)),
                             exact=True).cpu().numpy()
    want = orc.dot_csr_ndarray((300, 128), data, idx, ptr, b)
    assert np.array_equal(got, want, equal_nan=True)


def test_container_matmul_dropin(orc):
    """`GCXS(ca=(0,)) @ ndarray -> ndarray` (Appendix B row 4), NumPy in -> NumPy out."""
    import sparse_amd

    data, idx, ptr = random_csr(400, 300, 0.05, 5, np.float64, np.int64)
    b = random_dense(300, 20, 6, np.float64)
    a = sparse_amd.GCXS((data, idx, ptr), shape=(400, 300), compressed_axes=(0,))
    got = a @ b
    assert isinstance(got, np.ndarray) and got.dtype == np.float64
    want = orc.dot_csr_ndarray((400, 20), data, idx, ptr, b)
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    tb = torch.from_numpy(b).cuda()
    got_t = sparse_amd.matmul(a, tb)
    assert isinstance(got_t, torch.Tensor) and got_t.is_cuda


def test_full_size_linearity_property():
    """Size-independent property at BASELINE config-2 scale (M=1e6, K=1e4, N=128, 1e8 nnz is
    exercised by bench.py; here 1e5 x 1e4 @ 1 % = 1e7 nnz): A @ (B1 + B2) == A @ B1 + A @ B2
    within fp32 tolerance, and A @ e_j picks column j of A exactly."""
    from sparse_amd import _kernels as Kn
    from bench import make_csr_device

    M, Kd, N = 100_000, 10_000, 128
    data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=11, idx_dtype=torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    b1 = torch.rand((Kd, N), generator=g, device="cuda")
    b2 = torch.rand((Kd, N), generator=g, device="cuda")
    lhs = Kn.dot_csr_ndarray((M, N), data, idx, ptr, b1 + b2)
    rhs = Kn.dot_csr_ndarray((M, N), data, idx, ptr, b1) + Kn.dot_csr_ndarray((M, N), data, idx, ptr, b2)
    assert torch.allclose(lhs, rhs, rtol=2e-5, atol=1e-5)
    # one-hot dense operand: out[:, j] == column (100+j) of A, exactly
    e = torch.zeros((Kd, N), device="cuda")
    e[torch.arange(N) + 100, torch.arange(N)] = 1.0
    col = Kn.dot_csr_ndarray((M, N), data, idx, ptr, e, exact=True)
    rows = torch.repeat_interleave(torch.arange(M, device="cuda"), (ptr[1:] - ptr[:-1]).long())
    sel = (idx >= 100) & (idx < 100 + N)
    want = torch.zeros((M, N), device="cuda")
    want[rows[sel], (idx[sel] - 100).long()] = data[sel]
    assert torch.equal(col, want)


VARIANTS = ["G=32,VEC=2,U=8", "G=16,VEC=1,U=4,PANEL=32", "G=64,VEC=1,U=4",
            "G=64,VEC=2,CH=2", "G=16,VEC=2,CH=4", "G=32,VEC=1,CH=4,PANEL=64",
            "KSPLIT=2", "KSPLIT=3,G=32,VEC=2", "KSPLIT=5,G=16,VEC=1"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("N", [2, 64, 128, 200, 512])
@pytest.mark.parametrize("dtype,idt", [(np.float32, np.int32), (np.float64, np.int64), (np.int64, np.int32)])
def test_kernel_variants_bit_identical(orc, monkeypatch, variant, N, dtype, idt):
    """Every kernel form keeps the reference's summation order: exact mode == oracle."""
    monkeypatch.setenv("SPAMD_SPMM_VARIANT", variant)
    (data, idx, ptr, b), got = _run(777, 900, N, 0.03, dtype, idt, exact=True, seed=3,
                                    empty_rows=(0, 1, 2, 400, 401, 776), long_row=300)
    want = orc.dot_csr_ndarray((777, N), data, idx, ptr, b)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("variant", VARIANTS[:5])
def test_kernel_variants_all_rows_empty(monkeypatch, variant):
    monkeypatch.setenv("SPAMD_SPMM_VARIANT", variant)
    (_, _, _, _), got = _run(300, 50, 128, 0.0, np.float32, np.int32, exact=False)
    assert got.shape == (300, 128) and not got.any()


# ---- results of 1..4 columns: the row-vector kernel (lanes along the row, tree order per row) ------------------------

def _fast_csr(M, K, nnz, seed, dtype, idt, empty_rows=(), long_row=None):
    """CSR with ~nnz stored elements (duplicates of the random draw dropped): sizes `util.random_csr`'s
    choice-without-replacement cannot reach."""
    rng = np.random.default_rng(seed)
    lin = np.unique(rng.integers(0, M * K, size=nnz))
    rows, cols = lin // K, lin % K
    if len(empty_rows):
        keep = ~np.isin(rows, np.asarray(empty_rows))
        rows, cols = rows[keep], cols[keep]
    if long_row is not None:
        keep = rows != long_row
        rows = np.concatenate([rows[keep], np.full(K, long_row)])
        cols = np.concatenate([cols[keep], np.arange(K)])
        o = np.lexsort((cols, rows))
        rows, cols = rows[o], cols[o]
    data = (rng.random(len(rows)) - 0.3).astype(dtype) if np.dtype(dtype).kind == "f" else \
        rng.integers(-50, 50, size=len(rows)).astype(dtype)
    ptr = np.zeros(M + 1, dtype=idt)
    np.cumsum(np.bincount(rows, minlength=M), out=ptr[1:])
    return data, cols.astype(idt), ptr


def _rowvec_check(orc, M, K, N, nnz, dtype, idt, seed=0, **kw):
    from sparse_amd import _kernels as Kn
    from util import assert_within_fma_bound

    data, idx, ptr = _fast_csr(M, K, nnz, seed, dtype, idt, **kw)
    b = random_dense(K, N, seed + 1, dtype)
    d = torch.device("cuda")
    args = [torch.from_numpy(x).to(d) for x in (data, idx, ptr, b)]
    got = Kn.dot_csr_ndarray((M, N), *args).cpu().numpy()
    kept = Kn.dot_csr_ndarray((M, N), *args, keep_order=True).cpu().numpy()
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    if np.dtype(dtype).kind == "f":
        assert_within_fma_bound(got, want, data, idx, ptr, b)
        assert_within_fma_bound(kept, want, data, idx, ptr, b)
    else:
        assert np.array_equal(got, want) and np.array_equal(kept, want)
    return got


@pytest.mark.parametrize("dtype,idt", [(np.float32, np.int32), (np.float64, np.int64), (np.int64, np.int32)])
@pytest.mark.parametrize("N", [1, 2, 3, 4])
@pytest.mark.parametrize("avg", [3, 12, 24, 48, 120])       # one per lanes-per-row choice (4, 8, 16, 32, 64)
def test_rowvec_dense_operand_in_lds(orc, dtype, idt, N, avg):
    """M >= 32768 and K * N values within the LDS budget: B is staged per block."""
    M, K = 33000, 600
    _rowvec_check(orc, M, K, N, M * avg, dtype, idt, seed=avg + N, empty_rows=(0, 5, 6, M - 1), long_row=777)


@pytest.mark.parametrize("dtype,idt", [(np.float32, np.int64), (np.float64, np.int32), (np.int32, np.int32)])
@pytest.mark.parametrize("N", [1, 2, 3, 4])
@pytest.mark.parametrize("M,K,avg", [(3000, 5000, 5), (3000, 5000, 20), (2500, 5000, 100), (33000, 40000, 30)])
def test_rowvec_dense_operand_gathered(orc, dtype, idt, N, M, K, avg):
    """Few rows, or a B beyond the LDS budget (40000 x N): gathers from global memory."""
    _rowvec_check(orc, M, K, N, M * avg, dtype, idt, seed=avg + N, empty_rows=(1, M - 1), long_row=M // 2)


@pytest.mark.parametrize("dtype,idt,N,K", [(np.float32, np.int32, 4, 10240), (np.float32, np.int32, 4, 10241),
                                           (np.float64, np.int32, 2, 10240), (np.float64, np.int64, 4, 5120),
                                           (np.float32, np.int32, 3, 13653), (np.float32, np.int32, 3, 13654)])
def test_rowvec_dense_operand_fills_the_whole_lds(orc, dtype, idt, N, K):
    """Round 4: the resident copy of B may take all 160 KB of the CU (config 2's matrix times 4 fp32 columns is 156.25 KB:
    0.81 -> 0.35 ms); one row more and the kernel gathers from global memory.  Both sides of the bound, last row of B used."""
    M = 33000
    _rowvec_check(orc, M, K, N, M * 20, dtype, idt, seed=N + K % 7, empty_rows=(0, M - 1), long_row=4242)


@pytest.mark.parametrize("N", [1, 2, 4])
@pytest.mark.parametrize("M", [1000, 40000])
def test_rowvec_leading_dimensions_and_misaligned_operand(orc, N, M):
    """Through the C ABI with ldb > N, ldo > N and a dense operand that starts 4 bytes off a 16-byte boundary."""
    from sparse_amd import _ffi
    from sparse_amd._device import ptr as p_, stream_ptr

    K, ldb, ldo = 700, 7, 6
    data, idx, ptr = _fast_csr(M, K, M * 40, 11, np.float32, np.int32)
    rng = np.random.default_rng(12)
    bbuf = rng.random(K * ldb + 1).astype(np.float32)
    d = torch.device("cuda")
    tb = torch.from_numpy(bbuf).to(d)
    b_view = tb[1:]
    out = torch.full((M, ldo), 7.0, dtype=torch.float32, device=d)
    td, ti, tp = (torch.from_numpy(x).to(d) for x in (data, idx, ptr))
    _ffi.call("spamd_spmm_csr", _ffi.F32, _ffi.I32, M, K, N, p_(td), p_(ti), p_(tp), p_(b_view), ldb, p_(out), ldo, 0,
              stream_ptr(d))
    torch.cuda.synchronize()
    b = bbuf[1:].reshape(K, ldb)[:, :N].copy()
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    got = out.cpu().numpy()
    from util import assert_within_fma_bound

    assert_within_fma_bound(got[:, :N].copy(), want, data, idx, ptr, b)
    assert np.all(got[:, N:] == 7.0)          # nothing beyond the result's columns is written


def test_rowvec_nan_inf_only_touch_referenced_rows(orc):
    data, idx, ptr = _fast_csr(40000, 300, 40000 * 3, 3, np.float32, np.int32)
    b = random_dense(300, 1, 4, np.float32)
    b[5, 0] = np.inf
    b[0, 0] = np.nan        # B's row 0 is what a masked-off lane would gather if it were not masked
    data[idx == 0] = 0.0    # ... and where row 0 IS referenced the stored value is a zero: 0 * nan must appear
    d = torch.device("cuda")
    from sparse_amd import _kernels as Kn

    got = Kn.dot_csr_ndarray((40000, 1), *(torch.from_numpy(x).to(d) for x in (data, idx, ptr, b))).cpu().numpy()
    want = orc.dot_csr_ndarray((40000, 1), data, idx, ptr, b)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    ok = np.isfinite(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-5, atol=1e-6)


def test_container_matrix_vector_product(orc):
    """`GCXS @ 1-D ndarray` and `COO @ 1-D ndarray` (numpy.dot's contraction over the last axis)."""
    import sparse_amd

    data, idx, ptr = _fast_csr(35000, 500, 35000 * 20, 21, np.float64, np.int64)
    v = random_dense(500, 1, 22, np.float64)[:, 0].copy()
    a = sparse_amd.GCXS((data, idx, ptr), shape=(35000, 500), compressed_axes=(0,))
    want = orc.dot_csr_ndarray((35000, 1), data, idx, ptr, v[:, None])[:, 0]
    for got in (a @ v, sparse_amd.dot(a, v), a.tocoo() @ v):
        assert isinstance(got, np.ndarray) and got.shape == (35000,) and got.dtype == np.float64
        assert np.allclose(got, want, rtol=1e-12, atol=1e-13)


# ---- short contracted axis: the dense operand resident in LDS (spmm_ldsb.hip) ----------------------------------------

def _ldsb_case(M, K, N, nnz, dtype, idt, seed, exact, **kw):
    from sparse_amd import _kernels as Kn

    data, idx, ptr = _fast_csr(M, K, nnz, seed, dtype, idt, **kw)
    b = random_dense(K, N, seed + 1, dtype)
    d = torch.device("cuda")
    args = [torch.from_numpy(x).to(d) for x in (data, idx, ptr, b)]
    got = Kn.dot_csr_ndarray((M, N), *args, exact=exact).cpu().numpy()
    kept = Kn.dot_csr_ndarray((M, N), *args, exact=exact, keep_order=True).cpu().numpy()      # the row-group kernel
    return (data, idx, ptr, b), got, kept


@pytest.mark.parametrize("dtype,idt", [(np.float32, np.int32), (np.float64, np.int64), (np.float32, np.int64)])
@pytest.mark.parametrize("K,N", [(512, 512), (575, 68), (639, 68), (64, 32), (300, 130 * 2), (1, 64)])
@pytest.mark.parametrize("avg", [0.5, 5, 40])
def test_ldsb_exact_mode_is_bit_identical(orc, dtype, idt, K, N, avg):
    """M >= 8192, (K + 1) rows of 256 bytes within the LDS budget, N * itemsize >= 128: rows shorter and longer than the
    16-pair chunk (and than two of them), empty rows, a dense row, partial last panels."""
    M = 9000
    if np.dtype(dtype).itemsize * N < 128:
        pytest.skip("narrower than half a panel: not this kernel's shape")
    (data, idx, ptr, b), got, kept = _ldsb_case(M, K, N, int(M * min(avg, K * 0.8)), dtype, idt, int(avg * 10) + K + N, True,
                                                empty_rows=(0, 3, M - 1), long_row=4000)
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    u = np.uint32 if dtype == np.float32 else np.uint64
    assert np.array_equal(got.view(u), want.view(u))
    assert np.array_equal(kept.view(u), want.view(u))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_ldsb_default_mode_equals_the_rowgroup_kernel(orc, dtype):
    (data, idx, ptr, b), got, kept = _ldsb_case(20000, 200, 96, 20000 * 12, dtype, np.int32, 77, False, empty_rows=(5,), long_row=9)
    want = orc.dot_csr_ndarray((20000, 96), data, idx, ptr, b)
    assert np.array_equal(got, kept)                     # same k-ascending FMA chain per output element
    if np.dtype(dtype).kind == "f":
        from util import assert_within_fma_bound

        assert_within_fma_bound(got, want, data, idx, ptr, b)
    else:
        assert np.array_equal(got, want)


def test_ldsb_signed_zeros_nan_and_inf(orc):
    """The lanes beyond a row's end multiply +0 by the neutral row of -0.0: the sign of every zero result must be the
    reference's (products of -0.0 included), rows of B that no stored element refers to may hold NaN / inf without
    leaking, and referenced ones must propagate."""
    M, K, N = 8200, 40, 64
    data, idx, ptr = _fast_csr(M, K, M * 3, 9, np.float32, np.int32, empty_rows=(7,))
    b = random_dense(K, N, 10, np.float32)
    b[0, :] = np.nan                 # row 0 of B: what a clamped / padded lane must never touch ...
    data[idx == 0] = 0.0             # ... and where it IS referenced, 0 * NaN = NaN must appear
    b[1, :8] = np.inf
    b[2, :] = 0.0
    data[idx == 2] = -np.abs(data[idx == 2])     # (-x) * 0.0 = -0.0 products
    d = torch.device("cuda")
    from sparse_amd import _kernels as Kn

    args = [torch.from_numpy(x).to(d) for x in (data, idx, ptr, b)]
    got = Kn.dot_csr_ndarray((M, N), *args, exact=True).cpu().numpy()
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert np.array_equal(got.view(np.uint32) | (np.isnan(got) * np.uint32(0x7fffffff)),
                          want.view(np.uint32) | (np.isnan(want) * np.uint32(0x7fffffff)))      # NaN payloads aside
    assert np.isnan(got).any() and np.isinf(got).any() and (got == 0).any()          # the cases are exercised


def test_ldsb_all_rows_empty_and_boundaries_of_the_policy(orc):
    from sparse_amd import _kernels as Kn

    d = torch.device("cuda")
    ptr = torch.zeros(8193, dtype=torch.int32, device=d)
    e_i = torch.zeros(0, dtype=torch.int32, device=d)
    e_v = torch.zeros(0, dtype=torch.float32, device=d)
    b = torch.rand((100, 64), device=d)
    out = torch.full((8192, 64), 3.0, device=d)
    Kn.dot_csr_ndarray((8192, 64), e_v, e_i, ptr, b, out=out)
    assert not out.any()
    # K = 576 does not fit the LDS budget (577 rows of 256 bytes), M = 8191 is below the policy: both take the row-group kernel
    for M, K in ((8192, 576), (8191, 512)):
        (data, idx, ptr_, bb), got, kept = _ldsb_case(M, K, 128, M * 4, np.float32, np.int32, K, True)
        assert np.array_equal(got, orc.dot_csr_ndarray((M, 128), data, idx, ptr_, bb))
