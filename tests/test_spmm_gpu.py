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
