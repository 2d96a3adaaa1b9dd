"""The CSC-native inspector (csrc/spmm_tiled.hip `tl_csc_*`, C ABI `spamd_spmm_tiled_inspect_csc`): the executor's block stream
built from the CSC arrays of `csc @ dense` (reference `_dot_csc_ndarray`, _common.py:869-904) without a CSC -> CSR
conversion.  Every product from it must be the row-group kernel's on the CSR arrays bit for bit (an output element's terms are
added k-ascending by both)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    import sparse_amd

    return sparse_amd


def _case(M, Kd, density, dtype, itype, seed, empty_cols=(), dense_cols=()):
    from bench import make_csr_device
    from sparse_amd import _kernels as K

    vt = torch.float32 if dtype == torch.int32 else dtype
    data, idx, ptr = make_csr_device(M, Kd, density, seed=seed, dtype=vt)
    if len(empty_cols) or len(dense_cols):
        rows = K.csr_to_keys(ptr, torch.zeros_like(idx), M, 1)
        keep = ~torch.isin(idx, torch.tensor(list(empty_cols) + list(dense_cols), device=idx.device, dtype=idx.dtype))
        rows, cols, vals = rows[keep], idx[keep].to(torch.int64), data[keep]
        for c in dense_cols:      # every row holds this column
            rows = torch.cat([rows, torch.arange(M, device=rows.device)])
            cols = torch.cat([cols, torch.full((M,), c, device=rows.device, dtype=torch.int64)])
            vals = torch.cat([vals, torch.rand(M, device=rows.device, dtype=vt)])
        order = torch.argsort(rows * Kd + cols)
        rows, idx, data = rows[order], cols[order].to(idx.dtype), vals[order]
        ptr = torch.zeros(M + 1, dtype=ptr.dtype, device=ptr.device)
        ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=M), 0)
    if dtype == torch.int32:
        data = (data * 2000 - 1000).to(torch.int32)
    idx, ptr = idx.to(itype), ptr.to(itype)
    return data, idx, ptr


@pytest.mark.parametrize("M, Kd, density, N", [(3000, 700, 0.02, 128), (70_001, 1500, 0.01, 256), (560 * 7 + 13, 10_000, 0.003, 128),
                                               (1121, 321, 0.3, 128), (559, 161, 0.5, 128), (40_000, 40_960, 0.0005, 128)])
@pytest.mark.parametrize("dtype, itype", [(torch.float32, torch.int32), (torch.float64, torch.int64), (torch.int32, torch.int64)])
def test_csc_inspector_products_equal_the_row_group_kernel(dtype, itype, M, Kd, density, N):
    """shapes around the block (560 rows) and tile (160 columns) sizes, row blocks whose runs exceed one staging window
    (30-50 % density: up to 45 000 elements in a block x tile), the largest K the one-pass builders take"""
    from sparse_amd import _kernels as K

    data, idx, ptr = _case(M, Kd, density, dtype, itype, seed=M % 31)
    cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)          # the CSC arrays
    n = N if dtype != torch.float64 else N // 2
    b = torch.randint(-50, 50, (Kd, n), device="cuda", dtype=torch.int32) if dtype == torch.int32 else \
        torch.rand((Kd, n), device="cuda", dtype=dtype) - 0.5
    lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dtype)
    assert lay is not None
    got = K.dot_csr_ndarray_tiled(lay, (M, n), Kd, b)
    assert torch.equal(got, K.dot_csr_ndarray((M, n), data, idx, ptr, b))
    # the CSR inspector's stream gives the same products (its lists hold the same entries in row order)
    assert torch.equal(got, K.dot_csr_ndarray_tiled(K.csr_tiled_layout(data, idx, ptr, M, Kd, dtype=dtype), (M, n), Kd, b))


def test_csc_inspector_empty_and_full_columns():
    from sparse_amd import _kernels as K

    M, Kd = 9000, 1000
    data, idx, ptr = _case(M, Kd, 0.01, torch.float32, torch.int32, seed=4, empty_cols=(0, 1, 159, 160, 161, 999), dense_cols=(5, 480))
    cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)
    assert int(cp[1]) == 0 and int(cp[6] - cp[5]) == M
    b = torch.rand((Kd, 128), device="cuda") - 0.5
    got = K.dot_csr_ndarray_tiled(K.csc_tiled_layout(cd, ci, cp, M, Kd), (M, 128), Kd, b)
    assert torch.equal(got, K.dot_csr_ndarray((M, 128), data, idx, ptr, b))


def test_rows_out_of_order_inside_a_column_are_reported_and_the_product_recovers(sp):
    """`GCXS((data, indices, indptr))` takes the caller's arrays as they are: the CSC inspector reports rows that do not
    ascend inside a column (its lists are then empty), and `a @ b` falls back to the CSR route with the right result"""
    from sparse_amd import _kernels as K

    M, Kd = 70_000, 1200
    data, idx, ptr = _case(M, Kd, 0.01, torch.float32, torch.int32, seed=9)
    cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)
    a0 = int(cp[3])
    ci2 = ci.clone()
    ci2[a0], ci2[a0 + 1] = ci[a0 + 1].clone(), ci[a0].clone()
    cd2 = cd.clone()
    cd2[a0], cd2[a0 + 1] = cd[a0 + 1].clone(), cd[a0].clone()          # the same matrix, one column's entries swapped
    b = torch.rand((Kd, 128), device="cuda") - 0.5
    lay = K.csc_tiled_layout(cd2, ci2, cp, M, Kd)
    with pytest.raises(K.UnsortedColumns):
        K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, b)
    a = sp.GCXS((cd2, ci2, cp), shape=(M, Kd), compressed_axes=(1,))
    got = a @ b
    assert torch.equal(got, K.dot_csr_ndarray((M, 128), data, idx, ptr, b))


def test_default_construction_never_builds_a_csr_twin_for_the_executor(sp):
    """the reference's default `format="gcxs"` of a tall matrix (float64, int64 indices, compressed by columns)"""
    a = sp.random((80_000, 3000), density=0.01, random_state=2, format="gcxs")
    assert a.compressed_axes == (1,) and a.data.dtype == torch.float64
    b = torch.rand((3000, 64), device="cuda", dtype=torch.float64)
    r1 = a @ b
    assert getattr(a, "_csr_twin", None) is None and a._tiled_layouts
    ref = sp.GCXS(a.tocoo(), compressed_axes=(0,))
    assert torch.equal(r1, ref @ b) and torch.equal(a @ b, r1)
    rt = (b.t().contiguous() @ a.T)            # dense @ sparse: the transposed view shares the buffers and takes the CSR inspector
    assert torch.equal(rt.t(), r1)


@pytest.mark.parametrize("M", [1_400_123, 2_800_123])
def test_csc_inspector_more_row_groups_than_the_lds_histogram_holds(M):
    """up to 77 824 row groups (2.72 x 10^6 rows) two 16-bit counts share a word of the LDS histogram (round 6; one count per
    word up to 38 912 groups before); above that the list sizes come from the workgroups' own runs (`tl_csc_split_kernel`,
    `tl_csc_count_kernel`)"""
    from sparse_amd import _kernels as K

    Kd = 330
    data, idx, ptr = _case(M, Kd, 0.004, torch.float64, torch.int64, seed=8)
    cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)
    b = torch.rand((Kd, 64), device="cuda", dtype=torch.float64) - 0.5
    got = K.dot_csr_ndarray_tiled(K.csc_tiled_layout(cd, ci, cp, M, Kd), (M, 64), Kd, b)
    assert torch.equal(got, K.dot_csr_ndarray((M, 64), data, idx, ptr, b))


def test_csc_inspector_repeated_rows_that_overflow_a_16_bit_count_are_reported(sp):
    """a non-canonical operand: one row stored 70 000 times in one column.  The LDS histogram's 16-bit count of that row group
    wraps; the kernel notices (the counts no longer add up to the elements it walked) and reports the operand like rows out of
    order - `a @ b` then takes the CSR route and the product is the canonical operand's"""
    from sparse_amd import _kernels as K

    M, Kd, rep = 5000, 200, 70_000
    rng = np.random.default_rng(3)
    rows = np.sort(np.concatenate([np.full(rep, 1234), rng.integers(0, M, 3000)]))       # column 7 holds all of them
    indptr = np.zeros(Kd + 1, np.int64)
    indptr[8:] = len(rows)
    data = rng.random(len(rows))
    b = rng.random((Kd, 64))
    dev = torch.device("cuda:0")
    cd, ci, cp = (torch.from_numpy(x).to(dev) for x in (data, rows.astype(np.int64), indptr))
    lay = K.csc_tiled_layout(cd, ci, cp, M, Kd)
    with pytest.raises(K.UnsortedColumns):
        K.dot_csr_ndarray_tiled(lay, (M, 64), Kd, torch.from_numpy(b).to(dev))
    a = sp.GCXS((data, rows.astype(np.int64), indptr), shape=(M, Kd), compressed_axes=(1,), device="cuda:0")
    got = a @ b
    want = np.zeros((M, 64))
    np.add.at(want, rows, data[:, None] * b[7][None, :])
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)


# ---- against the ORACLE itself (round-4 verdict: the tests above compare with the row-group kernel, a second HIP path) ---------

@pytest.mark.parametrize("dtype, itype", [(torch.float32, torch.int32), (torch.float64, torch.int64)])
@pytest.mark.parametrize("M, Kd, density", [(3000, 700, 0.02), (560 * 3 + 1, 4000, 0.004), (1121, 321, 0.3)])
def test_csc_inspector_products_against_the_oracle(orc, dtype, itype, M, Kd, density):
    """the reference's own loop for a CSC operand (`_dot_csc_ndarray`, _common.py:893-902: out[indices[k]] += data[k] * b[j] per
    column j) restated in oracle.c, on the CSC arrays the inspector reads: exact mode bit for bit, FMA mode within
    1e-6 x sum |a||b| of it"""
    from sparse_amd import _kernels as K

    data, idx, ptr = _case(M, Kd, density, dtype, itype, seed=3)
    cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)
    n = 128 if dtype == torch.float32 else 64
    b = torch.rand((Kd, n), device="cuda", dtype=dtype) - 0.5
    lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dtype)
    assert lay is not None
    hd, hi, hp, hb = (t.cpu().numpy() for t in (cd, ci, cp, b))
    want = orc.dot_csc_ndarray((M, Kd), (Kd, n), hd, hi, hp, hb)
    exact = K.dot_csr_ndarray_tiled(lay, (M, n), Kd, b, exact=True).cpu().numpy()
    assert exact.tobytes() == want.tobytes()
    fma = K.dot_csr_ndarray_tiled(lay, (M, n), Kd, b).cpu().numpy()
    bound = orc.dot_csc_ndarray((M, Kd), (Kd, n), np.abs(hd), hi, hp, np.abs(hb))
    assert np.all(np.abs(fma - want) <= 1e-6 * bound + 1e-300)


def test_default_operand_at_config2_size_against_the_oracle(sp, orc):
    """the reference-default construction of BASELINE config 2 (float64 values, int64 indices, compressed by columns: CSC) through
    `a @ b` - the CSC-native inspector + the float64 executor at full size - against the oracle's `_dot_csr_ndarray` loop on
    2500 sampled rows (whole rows, gathered from the CSR arrays the operand was made from; exact mode: bit for bit)"""
    from bench import make_csr_device
    from sparse_amd import _settings

    M, Kd, N = 1_000_000, 10_000, 128
    d, i, p = make_csr_device(M, Kd, 0.01, seed=77, dtype=torch.float64)
    a = sp.GCXS((d, i.to(torch.int64), p.to(torch.int64)), shape=(M, Kd), compressed_axes=(0,)).change_compressed_axes((1,))
    assert a.compressed_axes == (1,) and a.data.dtype == torch.float64 and a.indices.dtype == torch.int64
    b = torch.rand((Kd, N), device="cuda", dtype=torch.float64) - 0.5
    old = _settings.EXACT_MULADD
    try:
        _settings.EXACT_MULADD = True
        r = a @ b
    finally:
        _settings.EXACT_MULADD = old
    assert getattr(a, "_tiled_layouts", None) and a.__dict__.get("_csr_twin") is None, "the CSC-native inspector, no CSR twin"
    pick = np.sort(np.random.default_rng(5).choice(M, size=2500, replace=False))
    hp = p.cpu().numpy()
    seg = np.concatenate([np.arange(hp[r_], hp[r_ + 1]) for r_ in pick])
    sub_ptr = np.zeros(len(pick) + 1, dtype=np.int64)
    sub_ptr[1:] = np.cumsum([hp[r_ + 1] - hp[r_] for r_ in pick])
    want = orc.dot_csr_ndarray((len(pick), N), d.cpu().numpy()[seg], i.cpu().numpy()[seg].astype(np.int64), sub_ptr, b.cpu().numpy())
    got = r[torch.from_numpy(pick).cuda()].cpu().numpy()
    assert got.tobytes() == want.tobytes()
