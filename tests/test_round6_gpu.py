"""Round 6: the advisor's findings of round 5 (each with the case that showed it) and the evidence-hygiene items of the verdict."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sparse_times_sparse_with_a_zero_width_result():
    """(n x k, stored elements) @ (k x 0): the one-launch kernel's argument check refuses n_col <= 0; the wrapper must not
    route the product there (_kernels._spgemm_small)."""
    import sparse_amd as sp

    a = sp.random((30, 20), density=0.2, random_state=1, format="gcxs")
    b = sp.GCXS.from_numpy(np.zeros((20, 0))) if hasattr(sp.GCXS, "from_numpy") else sp.asarray(np.zeros((20, 0)), format="gcxs")
    for x, y in ((a, b), (a.tocoo(), b.tocoo())):
        c = x @ y
        assert c.shape == (30, 0) and c.nnz == 0
        assert c.todense().shape == (30, 0)
    z = sp.asarray(np.zeros((0, 30)), format="gcxs")
    assert (z @ a).shape == (0, 20)


def test_a_callable_named_like_a_ufunc_is_not_that_ufunc():
    """`_gcxs_same_layout` picked its plan by __name__ alone: a plain function called `multiply` was evaluated as
    np.multiply on two GCXS of one layout."""
    import sparse_amd as sp

    def multiply(a, b):
        return a * b + 1

    g1 = sp.random((40, 30), density=0.1, random_state=2, format="gcxs")
    g2 = sp.random((40, 30), density=0.1, random_state=3, format="gcxs")
    sp.elemwise(np.multiply, g1, g2)            # plans np.multiply for this dtype / fill combination
    got = sp.elemwise(multiply, g1, g2)
    want = g1.todense() * g2.todense() + 1
    assert np.array_equal(got.todense(), want)
    assert got.fill_value == 1


def test_reduced_gcxs_keeps_the_operand_index_width():
    """ADVICE r05 (_reduce.py): the 2-D stand-in carried int64 indices whatever the operand's"""
    import sparse_amd as sp

    g = sp.random((300, 200), density=0.05, random_state=4, format="gcxs", idx_dtype=np.int32)
    assert g.indices.dtype == torch.int32
    for ax in (0, 1):
        r = g.sum(axis=ax)
        assert np.allclose(r.todense(), g.todense().sum(axis=ax))
        assert r.indices.dtype == g.tocoo().sum(axis=ax).asformat("gcxs").indices.dtype


def test_fallback_stats_counts_host_evaluations():
    import sparse_amd as sp

    x = sp.random((50, 40), density=0.1, random_state=5)
    sp.fallback_stats(reset=True)
    sp.elemwise(lambda a: a * 2 + a, x)                # traced: stays on the device
    assert sum(v for k, v in sp.fallback_stats().items() if k != "recent") == 0
    got = sp.elemwise(lambda a: np.sin(a) * 0 + a, x)   # sin has no exactly-rounded device kernel
    assert np.allclose(got.todense(), x.todense())
    st = sp.fallback_stats()
    assert st["untraceable"] + st["not_traced"] >= 1 and st["recent"]
    assert sp.fallback_stats(reset=True)["recent"] and not sp.fallback_stats()["recent"]


def test_config2_full_size_rows_against_the_oracle(orc):
    """Parity at the headline size inside pytest (round-5 verdict 9; so far only bench.py compared config 2 with the oracle):
    2000 sampled rows of the 10^6 x 10^4 @ 1 % product with a dense 10^4 x 128 operand, mixed signs, FMA tolerance
    against sum |a_k b_k|; and the same rows of the matrix-vector product through the stream kernel."""
    import sparse_amd as sp
    from bench import make_csr_device
    from util import assert_within_fma_bound

    M, K, N = 1_000_000, 10_000, 128
    data, idx, ptr = make_csr_device(M, K, 0.01, seed=0)
    data = data - 0.3                                      # mixed signs: cancellation inside the rows
    a = sp.GCXS((data, idx, ptr), shape=(M, K), compressed_axes=(0,))
    g = torch.Generator(device="cuda").manual_seed(5)
    b = torch.rand((K, N), device="cuda", generator=g) - 0.5
    c = a @ b
    c = a @ b                                              # second product: the cached block stream (the steady state)
    rows = torch.from_numpy(np.random.default_rng(0).choice(M, 2000, replace=False)).sort().values.cuda()
    lo, hi = ptr[rows].long(), ptr[rows + 1].long()
    lens = (hi - lo)
    sp_ptr = torch.zeros(len(rows) + 1, dtype=torch.int64, device="cuda")
    sp_ptr[1:] = torch.cumsum(lens, 0)
    take = torch.repeat_interleave(lo - sp_ptr[:-1], lens) + torch.arange(int(sp_ptr[-1]), device="cuda")
    sd, si, spn = data[take].cpu().numpy(), idx[take].cpu().numpy().astype(np.int64), sp_ptr.cpu().numpy()
    bn = b.cpu().numpy()
    want = orc.dot_csr_ndarray((len(rows), N), sd, si, spn, bn)
    assert_within_fma_bound(c[rows].cpu().numpy(), want, sd, si, spn, bn)
    v = b[:, :1].contiguous()
    y = a @ v
    assert_within_fma_bound(y[rows].cpu().numpy(), orc.dot_csr_ndarray((len(rows), 1), sd, si, spn, bn[:, :1].copy()), sd, si, spn,
                            bn[:, :1])
