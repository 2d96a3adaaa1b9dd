"""Round-4 additions: run-to-run determinism of every kernel that uses atomics or look-back ("race
detection": the reference is single-threaded, so ANY run-to-run difference here would be a bug of this backend), and the
hygiene fixes of the round (leading-dimension check of the tiled executor, integer scalars that do not fit the traced
compute type, the merge workspace after an abandoned call)."""
import ctypes as C
import os
import sys

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
    t = t.contiguous()
    return t.view(torch.uint8).cpu().numpy().tobytes()


def _same_sparse(x, y):
    if type(x) is not type(y) or x.shape != y.shape or x.nnz != y.nnz:
        return False
    if hasattr(x, "indptr"):
        return _bits(x.data) == _bits(y.data) and _bits(x.indices) == _bits(y.indices) and _bits(x.indptr) == _bits(y.indptr)
    return _bits(x.data) == _bits(y.data) and _bits(x.linear_loc()) == _bits(y.linear_loc())


# ---- run twice, compare bit for bit -------------------------------------------------------------------------------------

def test_spgemm_is_bit_identical_run_to_run(sp):
    """bucket slots are claimed with LDS atomics (csrc/spgemm_rows.hip): the ORDER of the claims varies from run to run,
    the result must not.  Rows of every size class (short rows, the 24-products-per-thread class, declined rows that go
    through the global form)."""
    from sparse_amd import _dot

    n = 40_000
    g = sp.random((n, n), density=2.5e-3, random_state=5, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    ref = g @ g
    for _ in range(3):
        _dot.drop_derived(g)
        again = g @ g
        assert _same_sparse(ref, again)
    # sparse x sparse with a skewed operand (one very long row: the global form merges in)
    h = sp.random((3000, 3000), density=0.02, random_state=6, dtype=np.float64, format="gcxs", compressed_axes=(0,))
    r1, r2 = h @ h, h @ h
    assert _same_sparse(r1, r2)


def test_fused_merge_is_bit_identical_run_to_run(sp):
    """tiles chain their output offsets by decoupled look-back and take tickets with a same-address atomic
    (csrc/merge.hip): the result must not depend on which tile wins."""
    x = sp.random((1000, 1000, 1000), nnz=700_000, random_state=1)
    y = sp.random((1000, 1000, 1000), nnz=500_000, random_state=2)
    for f in (lambda: x + y, lambda: x * y, lambda: sp.maximum(x, y), lambda: x - y):
        ref = f()
        for _ in range(4):
            assert _same_sparse(ref, f())
    # the two-launch and the count / scan / fill forms give the same bits as the fused one
    from sparse_amd import _umath

    ref = x + y
    for fused, single in ((False, True), (False, False)):
        old = _umath.MERGE_FUSED, _umath.MERGE_SINGLE_PASS
        _umath.MERGE_FUSED, _umath.MERGE_SINGLE_PASS = fused, single
        try:
            assert _same_sparse(ref, x + y)
        finally:
            _umath.MERGE_FUSED, _umath.MERGE_SINGLE_PASS = old


def test_group_reduce_is_bit_identical_run_to_run(sp):
    """runs that cross tiles are closed by a fix-up pass (csrc/group_reduce.hip); floating-point sums must come out in
    one fixed association whatever the scheduling"""
    x = sp.random((300, 400, 500), nnz=2_000_000, random_state=3)
    for ax in (2, 0, (0, 1), (1, 2), None):
        ref = x.sum(axis=ax)
        for _ in range(3):
            got = x.sum(axis=ax)
            assert _same_sparse(ref, got) if hasattr(ref, "nnz") else _bits(torch.as_tensor(np.asarray(ref))) == _bits(torch.as_tensor(np.asarray(got)))
    long_runs = sp.random((4, 3_000_000), nnz=4_000_000, random_state=4)      # runs longer than 64 K elements
    ref = long_runs.sum(axis=1)
    for _ in range(3):
        assert _same_sparse(ref, long_runs.sum(axis=1))


def test_sddmm_panel_order_is_bit_identical_run_to_run(sp):
    """scattered non-temporal stores of the panel-ordered SDDMM (csrc/sddmm_panel.hip): every result slot has ONE writer"""
    from sparse_amd import _kernels as K

    M = 30_000
    s = sp.random((M, M), nnz=3_000_000, random_state=8, dtype=np.float32, idx_dtype=np.int32)
    dev = s.data.device
    gen = torch.Generator(device="cpu").manual_seed(0)
    a = torch.rand((M, 256), generator=gen).to(dev).to(torch.bfloat16)
    bt = torch.rand((M, 256), generator=gen).to(dev).to(torch.bfloat16)
    panels = K.sddmm_panels(s.coords, (M, M), K.sddmm_panel_width(bt))
    runs = [K.sddmm_coo(s.coords, s.data, a, bt, panels=panels) for _ in range(4)]
    assert all(_bits(runs[0]) == _bits(r) for r in runs[1:])
    plain = K.sddmm_coo(s.coords, s.data, a, bt)
    assert _bits(plain) == _bits(runs[0])


def test_tiled_executor_is_bit_identical_run_to_run(sp):
    from sparse_amd import _kernels as K

    M, Kd, N = 140_000, 4000, 128
    g = sp.random((M, Kd), density=0.01, random_state=9, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    b = torch.rand((Kd, N), device=g.data.device, dtype=torch.float32)
    lay = K.csr_tiled_layout(g.data, g.indices, g.indptr, M, Kd)
    ref = K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b)
    for _ in range(3):
        lay2 = K.csr_tiled_layout(g.data, g.indices, g.indptr, M, Kd)
        assert _bits(ref) == _bits(K.dot_csr_ndarray_tiled(lay2, (M, N), Kd, b))


# ---- hygiene ------------------------------------------------------------------------------------------------------------

def test_tiled_executor_rejects_a_leading_dimension_below_the_stored_width(sp):
    """`spamd_spmm_tiled` stores N - panel + last_cols columns per row: an `ldo` below that would overlap rows (round-3
    advice)."""
    from sparse_amd import _ffi, _kernels as K
    from sparse_amd._device import ptr, stream_ptr

    M, Kd = 70_000, 2000
    g = sp.random((M, Kd), density=0.01, random_state=10, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    b = torch.rand((Kd, 128), device=g.data.device, dtype=torch.float32)
    lay = K.csr_tiled_layout(g.data, g.indices, g.indptr, M, Kd)
    out = torch.empty((M, 128), device=b.device, dtype=torch.float32)
    lib = _ffi.lib()
    blocks, blk_off, _ = lay
    flags_full = _ffi.TILED_GROUP_ENDS if lay.group_ends else 0
    EINVAL = -1     # SPAMD_EINVAL (include/sparse_amd.h)
    args = lambda ldo, flags: (_ffi.F32, C.c_int64(M), C.c_int64(Kd), C.c_int64(128), C.c_void_p(ptr(blocks)), C.c_void_p(ptr(blk_off)),
                               C.c_void_p(ptr(b)), C.c_int64(128), C.c_void_p(ptr(out)), C.c_int64(ldo), C.c_uint(flags),
                               C.c_void_p(stream_ptr(b.device)))
    assert lib.spamd_spmm_tiled(*args(126, flags_full)) == EINVAL
    assert lib.spamd_spmm_tiled(*args(30, flags_full | (32 << 16))) == EINVAL        # 32 stored columns need ldo >= 32
    assert lib.spamd_spmm_tiled(*args(32, flags_full | (32 << 16))) == 0
    torch.cuda.synchronize()
    if lay.pending is not None:
        assert int(lay.pending[0]) == 0


def test_integer_scalar_beyond_the_compute_type_is_not_wrapped(sp):
    """`elemwise(lambda v: v < 2**40, int32 array)`: NumPy compares exactly; a device replay with int32(2**40) == 0 would
    not (round-3 advice) - such a graph goes to the host path and the result matches NumPy."""
    d = np.where(np.random.default_rng(0).random((50, 60)) < 0.3, np.random.default_rng(1).integers(-9, 9, (50, 60)), 0).astype(np.int32)
    x = sp.COO.from_numpy(d)
    got = sp.elemwise(lambda v: v < 2 ** 40, x)
    assert np.array_equal(np.asarray(got.todense()), d < 2 ** 40)
    got = sp.elemwise(lambda v: v + 3, x)          # in range: still traced on the device
    assert np.array_equal(np.asarray(got.todense()), d + 3)


def test_merge_workspace_survives_an_abandoned_call(sp):
    """an exception between launch and wait must leave the look-back workspace zeroed for the next merge"""
    from sparse_amd import _umath

    x = sp.random((1000, 1000, 1000), nnz=300_000, random_state=11)
    y = sp.random((1000, 1000, 1000), nnz=300_000, random_state=12)
    ref = x + y
    real = _umath._MergeWorkspace.wait_total

    def boom(self, k):
        raise KeyboardInterrupt

    _umath._MergeWorkspace.wait_total = boom
    try:
        with pytest.raises(KeyboardInterrupt):
            x + y
    finally:
        _umath._MergeWorkspace.wait_total = real
    for _ in range(3):
        assert _same_sparse(ref, x + y)


# ---- int32 values through the inspector / executor (round 4) --------------------------------------------------------------

def test_int32_executor_equals_the_row_group_kernel_and_numpy(sp):
    """`_dot_dtype` makes int32 x int32 -> int32 with wrap-around (reference test_compressed_2d.py:18-47 runs integer
    operands); the executor's int32 variant (v_mul_lo_u32 + v_add_u32 on the float32 layout) must give the row-group
    kernel's bits - and NumPy's, overflow included."""
    from sparse_amd import _kernels as K
    import scipy.sparse as sps

    rng = np.random.default_rng(5)
    M, Kd = 70_000, 3000
    for N, big in ((128, False), (40, True), (256, False), (6, False)):
        x = sps.random(M, Kd, density=0.01, format="csr", random_state=rng, dtype=np.float64)
        x.sort_indices()
        hi = 2 ** 31 - 1 if big else 1000
        data = rng.integers(-hi, hi, size=x.nnz, dtype=np.int64).astype(np.int32)
        b = rng.integers(-hi, hi, size=(Kd, N), dtype=np.int64).astype(np.int32)
        td, ti, tp, tb = (torch.from_numpy(v).to("cuda:0") for v in (data, x.indices.astype(np.int32), x.indptr.astype(np.int32), b))
        want = K.dot_csr_ndarray((M, N), td, ti, tp, tb)
        if N >= 8:
            lay = K.csr_tiled_layout(td, ti, tp, M, Kd, dtype=torch.int32)
            assert lay[2] == torch.int32
            npad = -(-N // 128) * 128
            bp = torch.zeros((Kd, npad), dtype=torch.int32, device="cuda:0")
            bp[:, :N] = tb
            got = K.dot_csr_ndarray_tiled(lay, (M, N), Kd, bp)
            assert got.dtype == torch.int32 and torch.equal(got, want), N
        a = sp.GCXS((td, ti, tp), shape=(M, Kd), compressed_axes=(0,))
        through = a @ tb
        assert through.dtype == torch.int32 and torch.equal(through, want), N
        # NumPy (int64 accumulate, wrapped to int32 = two's complement arithmetic mod 2^32) on a sample of rows
        pick = rng.choice(M, 300, replace=False)
        ref = (sps.csr_matrix((data.astype(np.int64), x.indices, x.indptr), shape=(M, Kd))[pick] @ b.astype(np.int64))
        ref = np.asarray(ref).astype(np.int64).astype(np.int32) if not big else None
        if ref is not None:
            assert np.array_equal(want.cpu().numpy()[pick], ref), N


def test_int32_product_takes_the_executor(sp):
    from sparse_amd import _dot

    a = sp.random((131072, 4000), density=0.01, random_state=2, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    ai = sp.GCXS(((a.data * 100).to(torch.int32), a.indices, a.indptr), shape=a.shape, compressed_axes=(0,))
    b = torch.randint(-50, 50, (4000, 128), device=a.data.device, dtype=torch.int32)
    r1 = ai @ b
    assert torch.int32 in ai.__dict__.get("_tiled_layouts", {})
    from sparse_amd import _kernels as K
    assert torch.equal(r1, K.dot_csr_ndarray((131072, 128), ai.data, ai.indices, ai.indptr, b))


def test_large_coo_operand_takes_the_inspector_at_its_first_product(sp):
    """round-3 verdict, item 3 (iii): from COO_TILED_FIRST_NNZ stored elements a COO operand gets its block stream at the
    FIRST eligible product (smaller ones at the second, as before); the row pointers keep the coordinates' int32 width; the
    result is the row-group kernel's, bit for bit."""
    from sparse_amd import _dot, _kernels as K

    M, Kd, N = 400_000, 6000, 128
    g = sp.random((M, Kd), density=0.01, random_state=12, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    rows = K.csr_to_keys(g.indptr, torch.zeros_like(g.indices), M, 1).to(torch.int32)
    coo = sp.COO(torch.stack([rows, g.indices]), g.data, shape=(M, Kd), has_duplicates=False, sorted=True)
    b = torch.rand((Kd, N), device=g.data.device, dtype=torch.float32)
    assert coo.nnz >= _dot.COO_TILED_FIRST_NNZ
    r = coo @ b
    assert torch.float32 in coo.__dict__.get("_tiled_layouts", {}), "first product of a large COO did not build the block stream"
    assert coo._csr_view[2].dtype == torch.int32
    assert torch.equal(r, K.dot_csr_ndarray((M, N), g.data, g.indices, g.indptr, b))
    old = _dot.COO_TILED_FIRST_NNZ
    try:
        _dot.COO_TILED_FIRST_NNZ = 10 ** 12
        coo2 = sp.COO(torch.stack([rows, g.indices]), g.data, shape=(M, Kd), has_duplicates=False, sorted=True)
        coo2 @ b
        assert not coo2.__dict__.get("_tiled_layouts")          # the round-3 policy: at the second product
        coo2 @ b
        assert torch.float32 in coo2.__dict__.get("_tiled_layouts", {})
    finally:
        _dot.COO_TILED_FIRST_NNZ = old


@pytest.mark.parametrize("dtype, N, nnz_rows, first", [(torch.float32, 512, 60_000, True), (torch.float32, 512, 30_000, False),
                                                        (torch.float64, 128, 120_000, True), (torch.float32, 128, 120_000, False)])
def test_coo_first_product_bound_follows_the_result_width(sp, dtype, N, nnz_rows, first):
    """Late round 4 (`_dot._coo_first_product_tiled`): a COO operand's FIRST product takes the inspector from
    nnz x (bytes of a result row) = 4 x 10^9 on when the result is wider than one 512-byte panel (the row-group kernel's cost
    per element grows with the width); otherwise at its second product, as before.  Bit-identical either way."""
    from sparse_amd import _kernels as K

    M, Kd = nnz_rows, 4000                      # 40 stored elements per row
    g = sp.random((M, Kd), density=0.01, random_state=3, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    data = g.data.to(dtype)
    rows = K.csr_to_keys(g.indptr, torch.zeros_like(g.indices), M, 1).to(torch.int32)
    coo = sp.COO(torch.stack([rows, g.indices]), data, shape=(M, Kd), has_duplicates=False, sorted=True)
    b = torch.rand((Kd, N), device=data.device, dtype=dtype) - 0.5
    want = K.dot_csr_ndarray((M, N), data, g.indices, g.indptr, b)
    r1 = coo @ b
    assert bool(coo.__dict__.get("_tiled_layouts")) == first
    r2 = coo @ b
    assert coo.__dict__.get("_tiled_layouts")
    assert torch.equal(r1, want) and torch.equal(r2, want)


@pytest.mark.parametrize("dtype, M, N, density, takes", [
    (torch.float32, 12_000, 512, 0.01, True), (torch.float32, 9_000, 512, 0.01, False),
    (torch.float64, 21_000, 128, 0.01, True), (torch.float64, 6_000, 512, 0.01, True),
    (torch.float32, 46_000, 128, 0.01, True), (torch.float32, 40_000, 128, 0.01, False),
    (torch.float32, 50_000, 64, 0.01, False), (torch.int32, 12_000, 512, 0.01, True),
    (torch.float32, 70_000, 512, 0.0015, True), (torch.float32, 70_000, 128, 0.0015, False),
    (torch.float64, 70_000, 128, 0.0016, True), (torch.float64, 70_000, 128, 0.001, False)])
def test_executor_bounds_follow_the_result_width(sp, dtype, M, N, density, takes):
    """Round 4 (`_dot._tiled_min_rows`, `_dot._tiled_min_density`): the fewest rows and the lowest density at which
    `a @ dense` takes the inspector/executor shrink with the number of 512-byte column panels of the result (measured
    crossovers, tools/r04/m_crossover.py and density_crossover.py); either way the result is the row-group kernel's
    bit for bit (same k-ascending FMA per output element)."""
    from bench import make_csr_device
    from sparse_amd import _kernels

    Kd = 4000
    data, idx, ptr = make_csr_device(M, Kd, density, seed=M % 97, dtype=torch.float32 if dtype == torch.int32 else dtype)
    if dtype == torch.int32:
        data = (data * 1000).to(torch.int32)
        b = torch.randint(-1000, 1000, (Kd, N), device="cuda", dtype=torch.int32)
    else:
        b = torch.rand((Kd, N), device="cuda", dtype=dtype) - 0.5
    a = sp.GCXS((data, idx, ptr), shape=(M, Kd), compressed_axes=(0,))
    got = a @ b
    assert bool(getattr(a, "_tiled_layouts", None)) == takes
    assert torch.equal(got, _kernels.dot_csr_ndarray((M, N), data, idx, ptr, b))
