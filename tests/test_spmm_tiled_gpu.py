"""Inspector/executor SpMM (csrc/spmm_tiled.hip): parity with the CPU oracle and BIT-identity with the
row-group kernel's FMA mode (same k-ascending fused multiply-adds)."""
import numpy as np
import pytest
import torch

from util import assert_within_fma_bound, random_csr, random_dense

pytestmark = pytest.mark.gpu


def Kn_params():
    from sparse_amd import _kernels as Kn

    return Kn.tiled_params()


def _run(M, K, density, idt, seed=0, **kw):
    from sparse_amd import _kernels as Kn

    data, idx, ptr = random_csr(M, K, density, seed, np.float32, idt, **kw)
    b = random_dense(K, 128, seed + 1, np.float32)
    d = torch.device("cuda")
    td, ti, tp, tb = (torch.from_numpy(x).to(d) for x in (data, idx, ptr, b))
    layout = Kn.csr_tiled_layout(td, ti, tp, M, K)
    got = Kn.dot_csr_ndarray_tiled(layout, (M, 128), K, tb)
    ref = Kn.dot_csr_ndarray((M, 128), td, ti, tp, tb, exact=False)
    torch.cuda.synchronize()
    return (data, idx, ptr, b), got, ref, layout


@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("M,K,density", [(300, 200, 0.05), (1000, 3000, 0.01), (257, 64, 0.5), (64, 1000, 0.2),
                                         (5000, 10000, 0.01), (1, 70, 1.0), (4097, 129, 0.1), (16, 128, 1.0), (333, 160, 0.3),
                                         (333, 161, 0.3), (90, 159, 1.0), (64, 321, 0.5)])
def test_tiled_bit_identical_to_rowgroup_fma(orc, idt, M, K, density):
    (data, idx, ptr, b), got, ref, layout = _run(M, K, density, idt)
    assert torch.equal(got, ref)
    want = orc.dot_csr_ndarray((M, 128), data, idx, ptr, b)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, b)   # the executor itself, 1e-6 * sum|a_k b_k|
    blocks, blk_off, _ = layout
    rg, kb, gpb, epb, slack, _, _ = Kn_params()
    assert bool((blk_off[1:] >= blk_off[:-1]).all()) and blocks.numel() >= (int(blk_off[-1]) + slack) * epb * 2
    ent = blocks.view(-1, 2)[: int(blk_off[-1]) * epb].cpu().numpy()
    real = ent[ent[:, 0] != 0]                                         # padding entries (and the blocks between row groups) are all-zero
    assert len(real) == len(data)
    assert sorted(real[:, 1].view(np.float32).tolist()) == sorted(data.tolist())  # a permutation of A's values


@pytest.mark.parametrize("M,K,density", [(700, 20000, 0.004), (100, 128, 0.3), (513, 256, 0.1), (2000, 7936, 0.01),
                                         (40, 16000, 0.02), (100, 320, 0.3), (700, 9760, 0.01), (300, 9920, 0.02),
                                         (50, 19680, 0.01)])
def test_tiled_phase_chunking_and_tile_edges(orc, M, K, density):
    """More than 61 / 122 tiles (the phase loop runs in chunks of 61), K an exact multiple of the tile (no
    partial tile), one tile only."""
    (data, idx, ptr, b), got, ref, _ = _run(M, K, density, np.int32, seed=41)
    assert torch.equal(got, ref)
    want = orc.dot_csr_ndarray((M, 128), data, idx, ptr, b)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, b)


def test_tiled_edge_rows():
    (_, _, _, _), got, ref, _ = _run(777, 900, 0.03, np.int32, seed=3, empty_rows=(0, 1, 2, 400, 401, 776), long_row=300)
    assert torch.equal(got, ref)
    (_, _, _, _), got, ref, _ = _run(300, 50, 0.0, np.int32)
    assert torch.equal(got, ref) and not got.any()


def _product_case(M=70000, K=1500, density=0.01, N=128, seed=5):
    import sparse_amd as sp

    data, idx, ptr = random_csr(M, K, density, seed, np.float32, np.int32)
    b = random_dense(K, N, seed + 1, np.float32)
    d = torch.device("cuda")
    a = sp.GCXS(tuple(torch.from_numpy(x).to(d) for x in (data, idx, ptr)), shape=(M, K), compressed_axes=(0,))
    return a, torch.from_numpy(b).to(d), (data, idx, ptr, b)


def test_product_path_builds_and_caches_the_block_stream(orc, monkeypatch):
    """`a @ dense` (reference `_dot` csr x ndarray row, _common.py:339-503): the first eligible product builds
    and caches the tiled layout; "never" keeps the row-group kernel; both equal the oracle within fp32 FMA
    tolerance and are bit-identical to each other."""
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "EXACT_MULADD", False)
    a, b, (data, idx, ptr, bh) = _product_case()
    monkeypatch.setattr(_settings, "TILED_SPMM", "never")
    r1 = a @ b
    assert not getattr(a, "_tiled_layouts", None)
    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    r2 = a @ b
    layout = a._tiled_layouts[torch.float32]
    assert layout is not None
    r3 = a @ b
    assert a._tiled_layouts[torch.float32] is layout
    assert torch.equal(r1, r2) and torch.equal(r2, r3)
    want = orc.dot_csr_ndarray((a.shape[0], 128), data, idx, ptr, bh)
    assert_within_fma_bound(r3.cpu().numpy(), want, data, idx, ptr, bh)


@pytest.mark.parametrize("N", [256, 384])
def test_product_path_column_panels(orc, monkeypatch, N):
    from sparse_amd import _settings, _kernels as Kn

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", False)
    a, b, (data, idx, ptr, bh) = _product_case(N=N, seed=8)
    got = a @ b
    assert a._tiled_layouts
    ref = Kn.dot_csr_ndarray((a.shape[0], N), a.data, a.indices, a.indptr, b, exact=False)
    assert torch.equal(got, ref)
    want = orc.dot_csr_ndarray((a.shape[0], N), data, idx, ptr, bh)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, bh)


def test_product_path_csc_operand_builds_the_stream_from_the_csc_arrays(orc, monkeypatch):
    """Default-compressed tall matrix (csc): the CSC-native inspector builds the block stream (late round 4: no CSR twin;
    until then the cached twin fed the CSR inspector)."""
    import sparse_amd as sp
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", False)
    a, b, (data, idx, ptr, bh) = _product_case(seed=11)
    acsc = a.change_compressed_axes((1,))
    got = acsc @ b
    assert acsc._tiled_layouts and getattr(acsc, "_csr_twin", None) is None
    assert torch.equal(got, a @ b)      # the CSR operand's own stream: the same products bit for bit
    want = orc.dot_csr_ndarray((a.shape[0], 128), data, idx, ptr, bh)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, bh)


@pytest.mark.parametrize("M,K,density", [(300, 200, 0.05), (5000, 10000, 0.01), (4097, 129, 0.1), (64, 1000, 0.2)])
def test_tiled_exact_mode_is_bit_identical_to_the_reference_loop(orc, M, K, density):
    """SPAMD_EXACT_MULADD: a rounded product then a rounded add per term in storage order = reference
    `_dot_csr_ndarray` (_common.py:744-753) bit for bit (the oracle is its C restatement, -ffp-contract=off)."""
    from sparse_amd import _kernels as Kn

    data, idx, ptr = random_csr(M, K, density, 31, np.float32, np.int32)
    b = random_dense(K, 128, 32, np.float32)
    d = torch.device("cuda")
    td, ti, tp, tb = (torch.from_numpy(x).to(d) for x in (data, idx, ptr, b))
    layout = Kn.csr_tiled_layout(td, ti, tp, M, K)
    got = Kn.dot_csr_ndarray_tiled(layout, (M, 128), K, tb, exact=True)
    ref = Kn.dot_csr_ndarray((M, 128), td, ti, tp, tb, exact=True)
    want = orc.dot_csr_ndarray((M, 128), data, idx, ptr, b)
    assert torch.equal(got, ref)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_exact_mode_uses_the_tiled_path_and_small_n_keeps_the_rowgroup_kernel(monkeypatch):
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    a, b, _ = _product_case(seed=12)
    r = a @ b
    assert getattr(a, "_tiled_layouts", None)
    monkeypatch.setattr(_settings, "TILED_SPMM", "never")
    assert torch.equal(r, a @ b)
    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", False)
    a2, b2, _ = _product_case(N=4, seed=13)     # (results of at most 4 columns: the row-vector kernel; from 5 on the executor, padded B)
    a2 @ b2
    assert not getattr(a2, "_tiled_layouts", None)


@pytest.mark.parametrize("N", [64, 100, 130, 300])
def test_product_path_pads_ragged_result_widths(orc, monkeypatch, N):
    """N that is not a whole number of column panels: B is zero-padded to the next panel and the result sliced;
    exact mode stays bit-identical to the reference loop."""
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    a, b, (data, idx, ptr, bh) = _product_case(N=N, seed=14)
    got = a @ b
    assert a._tiled_layouts and got.shape == (a.shape[0], N) and got.is_contiguous()
    want = orc.dot_csr_ndarray((a.shape[0], N), data, idx, ptr, bh)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("M,K,density", [(300, 200, 0.05), (5000, 10000, 0.01), (4097, 129, 0.1), (33, 32768, 0.002)])
def test_direct_inspector_equals_the_key_sort_recipe(idt, M, K, density):
    """Both inspectors must produce the same block stream bit for bit (same stable order, same padding)."""
    from sparse_amd import _kernels as Kn

    data, idx, ptr = random_csr(M, K, density, 21, np.float32, idt, empty_rows=(0, 7), long_row=3)
    d = torch.device("cuda")
    td, ti, tp = (torch.from_numpy(x).to(d) for x in (data, idx, ptr))
    l1 = Kn.csr_tiled_layout(td, ti, tp, M, K)
    b1, o1, _ = l1
    b2, o2, _ = Kn.csr_tiled_layout(td, ti, tp, M, K, force_sort=True)
    Kn.TILED_ONE_PASS_INSPECTOR = False
    try:
        b3, o3, _ = Kn.csr_tiled_layout(td, ti, tp, M, K)       # the two-pass builder (count, scan, fill)
    finally:
        Kn.TILED_ONE_PASS_INSPECTOR = True
    assert torch.equal(o2, o3) and torch.equal(b2[: int(o2[-1]) * 16], b3[: int(o2[-1]) * 16])
    _assert_same_lists(l1, (b2, o2), K)


def _assert_same_lists(one_pass, recipe, K):
    """The one-pass inspector starts every row group at a closed-form upper bound (zeroed blocks in between) and gives
    each group its own end: list by list its blocks must be those of the key-sort recipe, bit for bit."""
    from sparse_amd import _kernels as Kn

    assert one_pass.group_ends
    (b1, o1, dt), (b2, o2) = one_pass, recipe
    kb = Kn.tiled_params(dt)[1]
    ntiles = -(-K // kb)
    o1h = o1.cpu().numpy().reshape(-1, ntiles + 1)
    o2h = o2.cpu().numpy()
    b1h, b2h = b1.cpu().numpy().reshape(-1, 16), b2.cpu().numpy().reshape(-1, 16)
    assert o1h.shape[0] * ntiles == o2h.size - 1
    assert np.all(np.diff(o1h, axis=1) >= 0) and np.all(o1h[1:, 0] >= o1h[:-1, -1])     # lists in order, groups do not overlap
    for g in range(o1h.shape[0]):
        assert np.array_equal(np.diff(o1h[g]), np.diff(o2h[g * ntiles:(g + 1) * ntiles + 1])), g
        assert np.array_equal(b1h[o1h[g, 0]:o1h[g, -1]], b2h[o2h[g * ntiles]:o2h[(g + 1) * ntiles]]), g
        nxt = o1h[g + 1, 0] if g + 1 < o1h.shape[0] else o1h[g, -1]
        assert not b1h[o1h[g, -1]:nxt].any(), g                                           # the gap is zeroed
        assert nxt - o1h[g, -1] <= ntiles                                                  # < one block per list


def test_unsorted_rows_fall_back_to_the_key_sort_recipe(orc):
    from sparse_amd import _kernels as Kn

    M, K = 500, 700
    data, idx, ptr = random_csr(M, K, 0.05, 22, np.float32, np.int32)
    idx = idx.copy(); data = data.copy()
    for r in range(M):  # reverse every row: same matrix, descending column order
        idx[ptr[r]:ptr[r + 1]] = idx[ptr[r]:ptr[r + 1]][::-1]
        data[ptr[r]:ptr[r + 1]] = data[ptr[r]:ptr[r + 1]][::-1]
    b = random_dense(K, 128, 23, np.float32)
    d = torch.device("cuda")
    td, ti, tp, tb = (torch.from_numpy(x).to(d) for x in (data, idx, ptr, b))
    layout = Kn.csr_tiled_layout(td, ti, tp, M, K)
    got = Kn.dot_csr_ndarray_tiled(layout, (M, 128), K, tb)
    want = orc.dot_csr_ndarray((M, 128), data, idx, ptr, b)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, b)


def test_product_path_defers_the_sortedness_verdict(orc, monkeypatch):
    """`a @ dense` does not wait for the one-pass inspector's verdict before launching the executor: the layout is built
    with `defer_check`, the verdict is read behind the first product.  Rows with descending column indices (which no
    constructor here produces, but a user-supplied triplet may hold) are caught there: the layout is rebuilt by the
    key-sort recipe and the product repeated — same result as for the sorted matrix, order of accumulation aside."""
    import sparse_amd as sp
    from sparse_amd import _settings, _kernels as Kn

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", False)
    M, K, N = 70000, 1500, 128
    data, idx, ptr = random_csr(M, K, 0.01, 33, np.float32, np.int32)
    b = random_dense(K, N, 34, np.float32)
    d = torch.device("cuda")
    tb = torch.from_numpy(b).to(d)
    a = sp.GCXS(tuple(torch.from_numpy(x).to(d) for x in (data, idx, ptr)), shape=(M, K), compressed_axes=(0,))
    got = a @ tb
    lay = a._tiled_layouts[torch.float32]
    assert lay.group_ends and lay.pending is None          # one-pass layout, verdict consumed by the first product
    ridx, rdata = idx.copy(), data.copy()
    for r in range(M):                                      # reverse every row: same matrix, descending column order
        ridx[ptr[r]:ptr[r + 1]] = idx[ptr[r]:ptr[r + 1]][::-1]
        rdata[ptr[r]:ptr[r + 1]] = data[ptr[r]:ptr[r + 1]][::-1]
    ar = sp.GCXS(tuple(torch.from_numpy(x).to(d) for x in (rdata, ridx, ptr)), shape=(M, K), compressed_axes=(0,))
    lay_r = Kn.csr_tiled_layout(ar.data, ar.indices, ar.indptr, M, K, defer_check=True)
    assert lay_r.pending is not None
    with pytest.raises(Kn.UnsortedColumns):
        Kn.dot_csr_ndarray_tiled(lay_r, (M, N), K, tb)
    got_r = ar @ tb
    lay2 = ar._tiled_layouts[torch.float32]
    assert not lay2.group_ends and lay2.pending is None    # rebuilt by the key-sort recipe
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert_within_fma_bound(got.cpu().numpy(), want, data, idx, ptr, b)
    assert_within_fma_bound(got_r.cpu().numpy(), want, data, idx, ptr, b)
    assert torch.equal(ar @ tb, got_r)


# ---- float64 (the reference's default dtype): one column per lane, 64-column panels, 5-entry blocks ------------
@pytest.mark.parametrize("N", [64, 128, 192])
@pytest.mark.parametrize("M,K,density", [(300, 200, 0.05), (5000, 10000, 0.01), (4097, 129, 0.1), (700, 20000, 0.004),
                                         (1, 70, 1.0)])
def test_tiled_f64_bit_identical_to_rowgroup_and_reference(orc, M, K, density, N):
    from sparse_amd import _kernels as Kn

    data, idx, ptr = random_csr(M, K, density, 51, np.float64, np.int32, empty_rows=(0,) if M > 10 else ())
    b = random_dense(K, N, 52, np.float64)
    d = torch.device("cuda")
    td, ti, tp, tb = (torch.from_numpy(x).to(d) for x in (data, idx, ptr, b))
    layout = Kn.csr_tiled_layout(td, ti, tp, M, K)
    assert layout[2] == torch.float64
    for exact in (False, True):
        got = Kn.dot_csr_ndarray_tiled(layout, (M, N), K, tb, exact=exact)
        ref = Kn.dot_csr_ndarray((M, N), td, ti, tp, tb, exact=exact)
        assert torch.equal(got, ref), exact
    want = orc.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want.view(np.uint64))   # exact mode = reference bits
    b2, o2, _ = Kn.csr_tiled_layout(td, ti, tp, M, K, force_sort=True)
    _assert_same_lists(layout, (b2, o2), K)


def test_product_path_float64_and_mixed_precision(orc, monkeypatch):
    """float64 operands take the float64 block stream; float32 values x float64 B promote like the reference
    (`_dot_dtype`, _common.py:635-636) and build a second, float64 stream of the same matrix."""
    from sparse_amd import _settings
    import sparse_amd as sp

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    data, idx, ptr = random_csr(70000, 1500, 0.01, 61, np.float64, np.int32)
    b = random_dense(1500, 64, 62, np.float64)
    d = torch.device("cuda")
    a = sp.GCXS(tuple(torch.from_numpy(x).to(d) for x in (data, idx, ptr)), shape=(70000, 1500), compressed_axes=(0,))
    r = a @ torch.from_numpy(b).to(d)
    assert torch.float64 in a._tiled_layouts and r.dtype == torch.float64
    assert np.array_equal(r.cpu().numpy(), orc.dot_csr_ndarray((70000, 64), data, idx, ptr, b))
    a32 = sp.GCXS((torch.from_numpy(data.astype(np.float32)).to(d), a.indices, a.indptr), shape=(70000, 1500),
                  compressed_axes=(0,))
    r2 = a32 @ torch.from_numpy(b).to(d)
    assert r2.dtype == torch.float64 and torch.float64 in a32._tiled_layouts
    want = orc.dot_csr_ndarray((70000, 64), data.astype(np.float32).astype(np.float64), idx, ptr, b)
    assert np.array_equal(r2.cpu().numpy(), want)


def test_coo_operand_gets_the_block_stream_at_its_second_product(orc, monkeypatch):
    """`COO @ dense` (reference `_dot_coo_ndarray`, _common.py:979-1014): a canonical 2-D COO is CSR order plus row
    pointers; temporaries (first product) keep the row-group kernel, an array multiplied again gets the tiled path."""
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "TILED_SPMM", "auto")
    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    a, b, (data, idx, ptr, bh) = _product_case(seed=15)
    c = a.tocoo()
    r1 = c @ b
    assert not getattr(c, "_tiled_layouts", None)
    r2 = c @ b
    assert c._tiled_layouts
    want = orc.dot_csr_ndarray((a.shape[0], 128), data, idx, ptr, bh)
    assert torch.equal(r1, r2) and np.array_equal(r2.cpu().numpy(), want)
