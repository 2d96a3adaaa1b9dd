"""Seeded synthetic inputs shared by the tests (NumPy, host side)."""
import numpy as np


def random_csr(M, K, density, seed, dtype=np.float32, idx_dtype=np.int64, empty_rows=(), long_row=None):
    """Uniform random CSR with sorted, duplicate-free column indices per row."""
    rng = np.random.default_rng(seed)
    nnz = int(round(M * K * density))
    lin = np.sort(rng.choice(M * K, size=nnz, replace=False)) if nnz else np.zeros(0, np.int64)
    rows, cols = lin // K, lin % K
    if len(empty_rows):
        keep = ~np.isin(rows, np.asarray(empty_rows))
        rows, cols = rows[keep], cols[keep]
    if long_row is not None:  # one completely dense row
        keep = rows != long_row
        rows = np.concatenate([rows[keep], np.full(K, long_row)])
        cols = np.concatenate([cols[keep], np.arange(K)])
        o = np.lexsort((cols, rows))
        rows, cols = rows[o], cols[o]
    if np.dtype(dtype).kind == "f":
        data = (rng.random(len(rows)) - 0.3).astype(dtype)
    else:
        data = rng.integers(-50, 50, size=len(rows)).astype(dtype)
    indptr = np.zeros(M + 1, dtype=idx_dtype)
    np.cumsum(np.bincount(rows, minlength=M), out=indptr[1:])
    return data, cols.astype(idx_dtype), indptr


def random_dense(K, N, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        return (rng.random((K, N)) - 0.5).astype(dtype)
    return rng.integers(-50, 50, size=(K, N)).astype(dtype)


F32_FMA_RTOL = 1e-6   # north_star: "numerics within 1e-6 rel of reference" — measured against sum_k |a_k b_k|


def assert_within_fma_bound(got, want, data, idx, ptr, b, rtol=None):
    """|got - want| <= rtol * sum_k |a_k| |b_kj| element by element (the sum evaluated in float64): the only
    difference between one FMA per term and the reference's rounded multiply + rounded add is one rounding per term,
    so the bound scales with the magnitude of the terms, not of the (possibly cancelling) result."""
    import scipy.sparse as sps

    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype
    M, K = len(ptr) - 1, b.shape[0]
    absum = sps.csr_matrix((np.abs(data).astype(np.float64), idx, ptr), shape=(M, K)) @ np.abs(b).astype(np.float64)
    if rtol is None:
        rtol = F32_FMA_RTOL if got.dtype == np.float32 else 1e-14
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(err <= rtol * absum + 1e-300), f"max err / bound = {np.max(err / (rtol * absum + 1e-300)):.3f}"
