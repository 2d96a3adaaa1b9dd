"""Round-2 additions: oracle-sampled parity at BASELINE.json's FULL sizes (configs 2, 3, 4), invalidation of the layouts
cached on a container, NumPy's boolean arithmetic, the export / constructor guards, and the sharded products on a real
RCCL process group (world_size 1: the code path of `bench.py --gpus N`, on the one GPU this box has)."""
import os
import sys

import numpy as np
import pytest
import torch

from util import assert_within_fma_bound, random_csr, random_dense

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def sp():
    import sparse_amd

    return sparse_amd


# ---- full-size parity against the oracle on a row sample (cheap on one CPU core) ----------------------------------------
def _sample_rows_csr(data, idx, ptr, rows):
    """Host CSR of the given rows (sorted) of a device CSR."""
    p = ptr.cpu().numpy().astype(np.int64)
    segs = [np.arange(p[r], p[r + 1]) for r in rows]
    sel = torch.from_numpy(np.concatenate(segs)).to(data.device)
    sub_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    sub_ptr[1:] = np.cumsum([len(s) for s in segs])
    return data[sel].cpu().numpy(), idx[sel].cpu().numpy().astype(np.int64), sub_ptr


def test_config2_full_size_against_the_oracle_on_10k_rows(orc, sp):
    """config 2 at full size (10^8 stored elements): 10^4 random rows of the product, through the PRODUCT path (cached
    block stream, executor), against the oracle's restatement of `_dot_csr_ndarray` (_common.py:744-753) — FMA mode within
    1e-6 * sum|a_k b_k| (north_star's tolerance), exact mode bit for bit."""
    from bench import make_csr_device
    from sparse_amd import _settings

    M, K, N = 1_000_000, 10_000, 128
    data, idx, ptr = make_csr_device(M, K, 0.01, seed=2024)
    g = torch.Generator(device="cuda").manual_seed(5)
    b = torch.rand((K, N), generator=g, device="cuda", dtype=torch.float32) - 0.5
    a = sp.GCXS((data, idx, ptr), shape=(M, K), compressed_axes=(0,))
    rows = np.sort(np.random.default_rng(0).choice(M, size=10_000, replace=False))
    hd, hi, hp = _sample_rows_csr(data, idx, ptr, rows)
    hb = b.cpu().numpy()
    want = orc.dot_csr_ndarray((len(rows), N), hd, hi, hp, hb)
    old = _settings.EXACT_MULADD
    try:
        _settings.EXACT_MULADD = False
        got = (a @ b)[torch.from_numpy(rows).cuda()].cpu().numpy()
        assert getattr(a, "_tiled_layouts", None), "config 2 must take the inspector/executor path"
        assert_within_fma_bound(got, want, hd, hi, hp, hb)
        _settings.EXACT_MULADD = True
        got = (a @ b)[torch.from_numpy(rows).cuda()].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    finally:
        _settings.EXACT_MULADD = old


@pytest.mark.parametrize("dt,it", [(np.float64, np.int64), (np.float32, np.int32)])
def test_config3_full_size_against_the_oracle(orc, sp, dt, it):
    """config 3 at full size: COO (512^3 @ 1 %) . dense (512, 512), axes=1 — the WHOLE product against the oracle's
    restatement of `_dot_coo_ndarray` (_common.py:999-1012); exact mode is bit-identical."""
    from sparse_amd import _settings

    n = 512
    c3 = sp.random((n, n, n), density=0.01, random_state=3, dtype=dt, idx_dtype=it)
    g = torch.Generator(device="cuda").manual_seed(1)
    d = (torch.rand((n, n), generator=g, device="cuda", dtype=torch.float64) - 0.5).to(torch.float64 if dt == np.float64 else torch.float32)
    at = c3.reshape((n * n, n))
    want = orc.dot_coo_ndarray(at.coords.cpu().numpy(), at.data.cpu().numpy(), d.cpu().numpy(), (n * n, n))
    old = _settings.EXACT_MULADD
    try:
        _settings.EXACT_MULADD = True
        got = sp.tensordot(c3, d, axes=1).reshape(n * n, n).cpu().numpy()
        assert got.dtype == want.dtype and np.array_equal(got, want)
        _settings.EXACT_MULADD = False
        got = sp.tensordot(c3, d, axes=1).reshape(n * n, n).cpu().numpy()
        ptr = np.zeros(n * n + 1, dtype=np.int64)
        np.cumsum(np.bincount(at.coords[0].cpu().numpy(), minlength=n * n), out=ptr[1:])
        assert_within_fma_bound(got, want, at.data.cpu().numpy(), at.coords[1].cpu().numpy(), ptr, d.cpu().numpy())
    finally:
        _settings.EXACT_MULADD = old


@pytest.mark.parametrize("tdt,tol", [(torch.bfloat16, 2e-6), (torch.float32, 2e-6)])
def test_config4_full_size_sddmm_against_float64_on_a_sample(sp, tdt, tol):
    """config 4 at full size (mask 10^5 x 10^5 @ 0.1 %, K = 256): 50 000 sampled outputs against the float64 evaluation
    of s * <a_i, b_j> on the host (the reference's `s * (a @ b)`, examples/sddmm_example.py:51-52, cannot form the 80 GB
    dense product).  Products of bf16 inputs are exact in fp32; the fp32 accumulation of 256 terms is within
    2e-6 * sum|terms|."""
    M = 100_000
    s = sp.random((M, M), density=0.001, random_state=4, dtype=np.float32, idx_dtype=np.int32)
    g = torch.Generator(device="cuda").manual_seed(2)
    a = (torch.rand((M, 256), generator=g, device="cuda") - 0.5).to(tdt)
    bt = (torch.rand((M, 256), generator=g, device="cuda") - 0.5).to(tdt)
    r = sp.sddmm(s, a, bt=bt)
    assert torch.equal(r.coords, s.coords) or r.nnz <= s.nnz
    pick = np.sort(np.random.default_rng(1).choice(s.nnz, size=50_000, replace=False))
    tp = torch.from_numpy(pick).cuda()
    i, j = s.coords[0][tp].long(), s.coords[1][tp].long()
    ha, hb = a[i].double().cpu().numpy(), bt[j].double().cpu().numpy()
    hs = s.data[tp].double().cpu().numpy()
    want = hs * np.einsum("ik,ik->i", ha, hb)
    terms = np.abs(hs) * np.einsum("ik,ik->i", np.abs(ha), np.abs(hb))
    from sparse_amd import _kernels as Kn

    got = Kn.sddmm_coo(s.coords, s.data, a, bt)[tp].double().cpu().numpy()
    assert np.all(np.abs(got - want) <= tol * terms)


# ---- layouts cached on a container follow its buffers --------------------------------------------
def _big_case(sp, seed, fmt):
    data, idx, ptr = random_csr(70000, 1500, 0.01, seed, np.float32, np.int32)
    b = random_dense(1500, 128, seed + 1, np.float32)
    d = torch.device("cuda")
    a = sp.GCXS(tuple(torch.from_numpy(x).to(d) for x in (data, idx, ptr)), shape=(70000, 1500), compressed_axes=(0,))
    if fmt == "csc":
        a = a.change_compressed_axes((1,))
    elif fmt == "coo":
        a = a.tocoo()
    return a, torch.from_numpy(b).to(d)


@pytest.mark.parametrize("fmt", ["csr", "csc", "coo"])
def test_cached_layouts_follow_in_place_updates(sp, monkeypatch, fmt):
    """product, in-place op, product again: the CSR twin / view and the tiled block stream must be rebuilt, whether the
    container's buffers were replaced (`a *= 2`, `np.multiply(a, 2, out=a)`) or written in place (`a.data.mul_(2)`)."""
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    a, b = _big_case(sp, 100, fmt)
    r1 = a @ b
    r1b = a @ b       # (a COO gets its block stream at the second product)
    assert torch.equal(r1, r1b) and getattr(a, "_tiled_layouts", None)
    a *= 2
    r2 = a @ b
    assert torch.equal(r2, r1 * 2), "stale cached layout after `a *= 2`"
    a @ b
    np.multiply(a, 2, out=a)
    assert torch.equal(a @ b, r1 * 4), "stale cached layout after a ufunc with out=a"
    a @ b
    a.data.mul_(2)    # same buffer, new contents: torch's version counter moves
    assert torch.equal(a @ b, r1 * 8), "stale cached layout after an in-place write to a.data"
    other = sp.GCXS(a) if fmt != "coo" else sp.COO(a)
    other @ b
    a @ b
    a += a
    assert torch.equal(a @ b, r1 * 16), "stale cached layout after `a += a`"


# ---- NumPy's boolean arithmetic ---------------------------------------------------------------
def test_bool_add_multiply_are_logical(sp):
    rng = np.random.default_rng(0)
    x = rng.random((40, 50)) < 0.3
    y = rng.random((40, 50)) < 0.3
    sx, sy = sp.COO.from_numpy(x), sp.COO.from_numpy(y)
    for f in (np.add, np.multiply, np.maximum, np.minimum, np.logical_xor, np.not_equal):
        r = f(sx, sy)
        want = f(x, y)
        assert r.dtype == want.dtype == np.bool_
        assert np.array_equal(r.todense(), want)
        assert set(np.unique(r.data.view(torch.uint8).cpu().numpy()).tolist()) <= {1}, "bool storage must hold 0/1 bytes only"
    # True + True stays a canonical True: a later equality / sum sees it as 1
    s = (sx + sy) + sx
    assert int(s.sum().todense()) == int(((x | y) | x).sum())


def test_bool_duplicates_sum_logically(sp):
    coords = np.array([[0, 0, 1, 1, 1], [3, 3, 2, 2, 4]])
    data = np.array([True, True, True, False, True])
    s = sp.COO(coords, data, shape=(2, 5))
    assert s.dtype == np.bool_ and s.data.dtype == torch.bool
    want = np.zeros((2, 5), dtype=bool)
    np.logical_or.at(want, (coords[0], coords[1]), data)
    assert np.array_equal(s.todense(), want)
    assert set(np.unique(s.data.view(torch.uint8).cpu().numpy()).tolist()) <= {1}


# ---- guards ---------------------------------------------------------------------------------------------------------------
def test_to_scipy_sparse_checks_the_fill_value(sp):
    x = sp.COO.from_numpy(np.eye(4))
    y = x + 1
    with pytest.raises(ValueError, match="fill_value"):
        y.to_scipy_sparse()
    assert y.to_scipy_sparse(accept_fv=1).shape == (4, 4)
    g = sp.GCXS(x) + 1
    with pytest.raises(ValueError, match="fill_value"):
        g.to_scipy_sparse()
    assert x.to_scipy_sparse().nnz == 4


def test_constructor_refuses_out_of_range_coordinates(sp):
    with pytest.raises(IndexError):
        sp.COO(np.array([[0, 5], [1, 1]]), np.array([1.0, 2.0]), shape=(3, 3))
    with pytest.raises(IndexError):
        sp.COO(np.array([[0, -1], [1, 1]]), np.array([1.0, 2.0]), shape=(3, 3))


def test_random_leaves_the_global_rng_alone_and_accepts_generators(sp):
    torch.manual_seed(1234)
    before = torch.random.get_rng_state().clone()
    cuda_before = torch.cuda.get_rng_state().clone()
    sp.random((50, 60), density=0.1)
    assert torch.equal(torch.random.get_rng_state(), before) and torch.equal(torch.cuda.get_rng_state(), cuda_before)
    a = sp.random((50, 60), density=0.1, random_state=np.random.default_rng(7))
    b = sp.random((50, 60), density=0.1, random_state=np.random.default_rng(7))
    assert torch.equal(a.coords, b.coords) and torch.equal(a.data, b.data)
    c = sp.random((50, 60), nnz=17, random_state=3, data_rvs=lambda n: np.arange(1, n + 1, dtype=np.float64))
    assert c.nnz == 17 and sorted(c.data.cpu().tolist()) == list(range(1, 18))


# ---- the multi-GPU code path on a real RCCL group (world_size 1) ------------------------------------------------------------
@pytest.fixture(scope="module")
def rccl_group():
    import torch.distributed as dist

    if dist.is_initialized():
        yield dist.group.WORLD
        return
    from conftest import init_single_rank_group

    init_single_rank_group("nccl")
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_sharded_spmm_and_sddmm_on_rccl_world_size_1(orc, sp, rccl_group):
    """`sharded_spmm` / `sharded_sddmm` with backend "nccl" (= RCCL): partition, shard, all-gather of the
    dense operand through the collective, local product — against the oracle."""
    import torch.distributed as dist
    from sparse_amd import _dist

    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    M, K, N = 3000, 1100, 128
    data, idx, ptr = random_csr(M, K, 0.02, 7, np.float32, np.int32)
    b = random_dense(K, N, 8, np.float32)
    d = torch.device("cuda")
    td, ti, tp = (torch.from_numpy(x).to(d) for x in (data, idx, ptr))
    bounds = _dist.partition_rows_by_nnz(tp, 1)
    sd, si, sptr, r0, r1 = _dist.shard_csr(td, ti, tp, 0, 1, bounds)
    assert (r0, r1) == (0, M)
    a_local = sp.GCXS((sd, si, sptr), shape=(r1 - r0, K), compressed_axes=(0,))
    b_shard = _dist.row_shard(torch.from_numpy(b).to(d), 0, 1)
    got = _dist.sharded_spmm(a_local, b_shard, K)
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, b)
    # the gather itself ran through RCCL: a 1-rank all_gather_into_tensor is the identity
    assert torch.equal(_dist.all_gather_rows(b_shard, K), torch.from_numpy(b).to(d))
    # SDDMM: mask rows and A rows co-sharded, Bt gathered
    s = sp.random((400, 300), density=0.05, random_state=9, dtype=np.float32, idx_dtype=np.int32)
    a2 = torch.rand((400, 64), device=d) - 0.5
    bt = torch.rand((300, 64), device=d) - 0.5
    r = _dist.sharded_sddmm(s, a2, _dist.row_shard(bt, 0, 1), 300)
    i, j = s.coords[0].long(), s.coords[1].long()
    want = s.data.double() * (a2[i].double() * bt[j].double()).sum(1)
    keep = want.float() != 0
    assert r.nnz == int(keep.sum()) and torch.allclose(r.data.double(), want[keep], rtol=1e-5, atol=1e-7)
    # sparse x sparse: B's CSR triplet through the ragged all-gather
    g = sp.random((500, 500), density=0.01, random_state=11, format="gcxs", compressed_axes=(0,))
    c = _dist.sharded_spgemm(g, g)
    ref = g @ g
    assert torch.equal(c.indptr.long(), ref.indptr.long()) and torch.equal(c.indices.long(), ref.indices.long()) and torch.equal(c.data, ref.data)


# ---- dense @ GCXS takes the same inspector/executor path as GCXS @ dense -----------------------
def test_dense_times_gcxs_uses_the_cached_executor(sp):
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(31)
    Kd, N, M = 3000, 70_000, 128
    b = sp.random((Kd, N), density=0.01, random_state=5, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(1,))
    a = torch.from_numpy(rng.random((M, Kd), dtype=np.float32)).cuda()
    r1 = a @ b if False else sp.matmul(a, b)
    view = b.__dict__.get("_t_view")
    assert view is not None and getattr(view, "_tiled_layouts", None), "dense @ GCXS did not build the block stream"
    r2 = sp.matmul(a, b)                      # second product: cached layouts, same bits
    assert torch.equal(r1, r2) and tuple(r1.shape) == (M, N)
    bt = b.T                                   # (N, Kd) compressed by rows
    ref = K.dot_csr_ndarray((N, M), bt.data, bt.indices, bt.indptr, a.t().contiguous()).t()
    assert torch.equal(r1, ref), "executor and row-group kernel differ"
    b.data.mul_(2)                             # in-place edit of b: the view's cached stream must not survive it
    assert torch.equal(sp.matmul(a, b), ref * 2)


def test_deferred_nan_warning_is_raised_later_not_lost(sp):
    """`_settings.NAN_WARNING = "deferred"`: matmul does not wait for its NaN scans; the warning comes from a later call or
    from flush_warnings() - never silently dropped, never duplicated."""
    import warnings

    from sparse_amd import _settings

    rng = np.random.default_rng(3)
    a = sp.random((300, 200), density=0.05, random_state=1, format="gcxs", compressed_axes=(0,))
    b = torch.from_numpy(rng.random((200, 64))).cuda()
    bad = b.clone()
    bad[3, 5] = float("nan")
    old = _settings.NAN_WARNING
    try:
        _settings.NAN_WARNING = "deferred"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sp.matmul(a, bad)
            sp.matmul(a, b)
            torch.cuda.synchronize()
            sp.flush_warnings()
        assert sum("Nan will not be propagated" in str(x.message) for x in w) == 1
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sp.matmul(a, b)
            sp.flush_warnings()
        assert not w
    finally:
        _settings.NAN_WARNING = old
    with pytest.warns(RuntimeWarning, match="Nan will not be propagated"):   # the default: before matmul returns
        sp.matmul(a, bad)


def test_spgemm_prune_uses_the_pack_kernels_zero_count():
    """`GCXS @ GCXS` prunes exact zeros like the reference (`GCXS(..., prune=True)`, _common.py:374-379).  Above
    PRUNE_COUNT_FIRST elements the prune first counts the zeros; for a row-local product the pack kernel has counted them
    already (no pass over the result).  Cancelling products (+1 * 1 + 1 * -1) must still disappear, and the remembered
    count must not outlive a write to the data."""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(4)
    n = 3000
    dense_a = np.zeros((n, n), dtype=np.float32)
    dense_b = np.zeros((n, n), dtype=np.float32)
    rows = rng.integers(0, n, 60000)
    dense_a[rows, rng.integers(0, n, 60000)] = rng.integers(1, 4, 60000).astype(np.float32)
    dense_b[rng.integers(0, n, 60000), rng.integers(0, n, 60000)] = rng.integers(1, 4, 60000).astype(np.float32)
    # make exact cancellations: rows 0..99 of A hold (+1 at column 0, +1 at column 1), rows 0/1 of B are negatives of each other
    dense_a[:100, :] = 0
    dense_a[:100, 0] = 1
    dense_a[:100, 1] = 1
    dense_b[0, :] = 0
    dense_b[1, :] = 0
    dense_b[0, :50] = 2
    dense_b[1, :50] = -2
    a = sp.GCXS.from_numpy(dense_a, compressed_axes=(0,))
    b = sp.GCXS.from_numpy(dense_b, compressed_axes=(0,))
    old = K.PRUNE_COUNT_FIRST
    K.PRUNE_COUNT_FIRST = 1
    try:
        c = a @ b
    finally:
        K.PRUNE_COUNT_FIRST = old
    want = dense_a.astype(np.float64) @ dense_b.astype(np.float64)
    assert np.array_equal(c.todense(), want.astype(np.float32))
    assert c.nnz == np.count_nonzero(want)
    # the remembered count is tied to the tensor's version
    t = torch.zeros(8, device="cuda")
    K.note_zero_bits_count(t, torch.tensor([8], device="cuda"))
    assert K.count_eq_bits(t, 0.0) == 8
    t[0] = 1.0
    assert K.count_eq_bits(t, 0.0) == 7
