"""CPU-side checks: the C ABI loads and exports what include/sparse_amd.h declares, and the
host logic around the kernels (axes normalisation, argument contracts, sharding arithmetic)
matches the reference's behaviour.  No compute calls: there is no GPU here."""
import ctypes
import os

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(hiplib):
    from sparse_amd import _ffi

    declared = _ffi.header_symbols()
    assert declared, "no symbols parsed from include/sparse_amd.h"
    assert set(declared) == set(_ffi.SIGNATURES), "ctypes table out of step with the header"
    for name in declared:
        assert hasattr(hiplib, name), name
    assert hiplib.spamd_target_arch() == b"gfx950"
    assert hiplib.spamd_version() >= 100


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sparse_amd import _ffi

    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.HipBackendError, match="no CPU fallback"):
        _ffi.lib()


def test_no_device_fails_loudly():
    import torch

    from sparse_amd import _device, _ffi, _kernels

    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    with pytest.raises(_ffi.HipBackendError):
        _device.default_device()
    t = torch.zeros(3)
    with pytest.raises(_ffi.HipBackendError, match="no CPU fallback"):
        _kernels.dot_csr_ndarray((2, 1), t, torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.int64),
                                 torch.zeros((1, 1)))


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparse_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


@pytest.mark.parametrize("a_shape,b_shape,axes", [
    ((3, 4), (4, 5), 1), ((3, 4, 5), (4, 5, 6), 2), ((3, 4, 5), (5, 4, 2), ((1, 2), (1, 0))),
    ((3, 4, 5), (5, 7), ((-1,), (0,))), ((6,), (6, 2), ((0,), (0,))), ((2, 3), (4, 5), 0),
])
def test_tensordot_plan_matches_numpy(a_shape, b_shape, axes):
    from sparse_amd._dot import tensordot_plan

    rng = np.random.default_rng(0)
    a, b = rng.random(a_shape), rng.random(b_shape)
    na, sa, nb, sb, olda, oldb = tensordot_plan(a_shape, b_shape, axes)
    got = (a.transpose(na).reshape(sa) @ b.transpose(nb).reshape(sb)).reshape(olda + oldb)
    assert np.allclose(got, np.tensordot(a, b, axes))


def test_tensordot_plan_errors():
    from sparse_amd._dot import tensordot_plan

    with pytest.raises(ValueError, match="shape-mismatch for sum"):
        tensordot_plan((3, 4), (5, 6), 1)
    with pytest.raises(ValueError, match="does not have enough dimensions"):
        tensordot_plan((), (5, 6), 1)
    assert tensordot_plan((), (), 0) is None


def test_argument_contracts():
    from sparse_amd import _utils as U

    class Fake:
        size, dtype = 4, np.dtype("f8")

        def __init__(self, fv):
            self.fill_value = np.float64(fv)

    U.check_zero_fill_value(Fake(0.0), Fake(-0.0))
    with pytest.raises(ValueError, match="argument 1 had a fill value of 0.5"):
        U.check_zero_fill_value(Fake(0.0), Fake(0.5))
    assert U.normalize_axis(-1, 3) == 2 and U.normalize_axis((0, -2), 3) == (0, 1)
    with pytest.raises(ValueError, match="Invalid axis index 3 for ndim=3"):
        U.normalize_axis(3, 3)
    U.check_compressed_axes(3, (0, 2))
    for bad, msg in (((0, 1, 2), "cannot compress all axes"), ((1, 0), "sorted without repeats"),
                     ((0, 5), "axis out of range"), ((0.5,), "integers")):
        with pytest.raises(ValueError, match=msg):
            U.check_compressed_axes(3, bad)
    assert bool(U.equivalent(np.float64(0.0), np.float64(-0.0), loose=True))
    assert not bool(U.equivalent(np.float64(0.0), np.float64(-0.0)))
    assert bool(U.equivalent(np.nan, np.nan, loose=True))
    assert U.can_store(np.int8, 127) and not U.can_store(np.int8, 128)


def test_dot_dtype_rule():
    from sparse_amd._kernels import dot_dtype

    assert dot_dtype(np.float32, np.float32) == np.float32
    assert dot_dtype(np.float32, np.float64) == np.float64
    assert dot_dtype(np.int32, np.int32) == np.int32
    assert dot_dtype(np.int64, np.float32) == np.float64


def test_nnz_balanced_row_partition():
    import torch

    from sparse_amd._dist import partition_rows_by_nnz, row_bounds, shard_csr

    rng = np.random.default_rng(1)
    counts = rng.integers(0, 200, 1000)
    counts[100:300] = 0
    counts[500] = 20000  # one very long row
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]))
    for world in (1, 2, 4, 8):
        b = partition_rows_by_nnz(indptr, world)
        assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:]))
        data = torch.arange(int(indptr[-1]))
        pieces = [shard_csr(data, data, indptr, r, world, b) for r in range(world)]
        assert sum(int(p[0].numel()) for p in pieces) == int(indptr[-1])
        for d, _, ip, r0, r1 in pieces:
            assert int(ip[0]) == 0 and int(ip[-1]) == d.numel() and ip.numel() == r1 - r0 + 1
        if world > 1:
            sizes = [int(p[0].numel()) for p in pieces]
            assert max(sizes) <= int(indptr[-1]) / world + 20000  # balanced up to one row
    assert [row_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_tiled_stream_slack_covers_the_line_touch(hiplib):
    """The tiled SpMM executor prefetches up to 64 lines (64-byte blocks) past the start of a list and its scalar ring
    over-reads up to 3 blocks: the readable slack the inspector appends to the stream must cover both (a shorter
    slack is an out-of-bounds read at the end of the allocation — found by tools/fuzz.py with the caching allocator
    switched off)."""
    import ctypes as C

    vals = [C.c_int() for _ in range(7)]
    for code in (0, 1):   # SPAMD_F32, SPAMD_F64
        assert hiplib.spamd_spmm_tiled_params(code, *[C.byref(v) for v in vals]) == 0
        assert vals[4].value >= 64 + 3


def test_compiler_stays_out_of_the_tiled_kernel_accumulators():
    """The tiled SpMM executor keeps its accumulators in fixed VGPRs across several asm blocks; the compiler's own code
    runs between them.  tools/check_tiled_regs.py cross-compiles the kernel and verifies that no compiler-generated
    instruction touches that register block (a dropped `amdgpu_num_vgpr` request would corrupt results silently)."""
    import shutil
    import subprocess
    import sys

    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_tiled_regs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


class _Shape:
    def __init__(self, *shape):
        self.shape = shape


@pytest.mark.parametrize("call,terms,out", [
    (("ab,bc", _Shape(2, 3), _Shape(3, 4)), ["ab", "bc"], "ac"),
    (("ab,bc->ca", _Shape(2, 3), _Shape(3, 4)), ["ab", "bc"], "ca"),
    (("...ab,...b->...a", _Shape(5, 2, 3), _Shape(3)), ["zab", "b"], "za"),
    (("a...,a...", _Shape(2, 3, 4), _Shape(2, 4)), ["ayz", "az"], "yz"),
    (("aab->b", _Shape(2, 2, 3)), ["aab"], "b"),
    (("ba", _Shape(2, 3)), ["ba"], "ab"),
    ((_Shape(4, 4), [0, 0]), ["AA"], ""),
    ((_Shape(4, 4), [Ellipsis, 1], [Ellipsis]), ["zB"], "z"),
    ((",ab,->ab", _Shape(), _Shape(2, 3), _Shape()), ["", "ab", ""], "ab"),
])
def test_einsum_subscript_normalisation(call, terms, out):
    """`parse_einsum` (sparse_amd/_einsum.py): explicit labels for every ellipsis (right-aligned), the classical implicit
    output (ellipsis labels, then the labels that occur once, sorted), the sublist calling form — the grammar of
    `numpy.einsum` that the reference's parser accepts (_common.py:1163-1323)."""
    from sparse_amd._einsum import parse_einsum

    got_terms, got_out, arrays = parse_einsum(call)
    assert got_terms == terms and got_out == out and len(arrays) == len(terms)


@pytest.mark.parametrize("bad,exc", [(("a+b->c", _Shape(2), _Shape(2)), ValueError), (("i->&", _Shape(2)), ValueError),
                                     (("i->ij", _Shape(2)), ValueError), (("ij->jij", _Shape(2, 2)), ValueError),
                                     (("a..,a...", _Shape(2), _Shape(2)), ValueError), ((".i...", _Shape(2, 2)), ValueError),
                                     (("a,a->->", _Shape(2), _Shape(2)), ValueError), ((), ValueError),
                                     (("ab", _Shape(2)), ValueError), (("a", _Shape(2, 2)), ValueError),
                                     ((0, _Shape(2), _Shape(2)), TypeError), (([0, 0], _Shape(2), _Shape(2)), TypeError)])
def test_einsum_subscript_errors(bad, exc):
    from sparse_amd._einsum import parse_einsum

    with pytest.raises(exc):
        parse_einsum(bad)


def test_basic_index_normalisation():
    """`_indexing._normalise`: ellipsis expansion, trailing full slices, and the IndexError / NotImplementedError
    contracts, independent of any device work."""
    from sparse_amd._indexing import _normalise

    full = slice(None)
    assert _normalise(2, 3) == (2, full, full)
    assert _normalise((Ellipsis, 1), 3) == (full, full, 1)
    assert _normalise((0, Ellipsis, None, 1), 4) == (0, full, full, None, 1)
    assert _normalise((None, slice(1, 3)), 2) == (None, slice(1, 3), full)
    assert _normalise((), 0) == ()
    for bad in ((0, 0, 0), (Ellipsis, Ellipsis), ("a",), (1.5,)):
        with pytest.raises(IndexError):
            _normalise(bad, 2)
    # one 1-D integer (or boolean) array index is on the path; several of them, or N-D ones, are not
    got = _normalise(([0, 1], Ellipsis), 2)
    assert isinstance(got[0], np.ndarray) and got[0].dtype == np.int64 and got[1] == full
    assert _normalise((np.array([True, False]),), 1)[0].dtype == bool
    for fancy in (([0, 1], [1, 0]), (np.zeros((2, 2), dtype=np.int64),)):
        with pytest.raises(NotImplementedError):
            _normalise(fancy, 2)
    with pytest.raises(IndexError):
        _normalise((np.array([0.5, 1.0]),), 2)


def test_sddmm_dispatch_models(hiplib):
    """The host-side choices of `sparse_amd.sddmm` are pure arithmetic on shapes (no GPU needed): which lane-group
    widths have a row-cached kernel (asked of the library), the panel width, and the traffic models that pick the
    element order and the tile dispatch (measured at config 4's shapes)."""
    import torch
    from types import SimpleNamespace
    from sparse_amd import _kernels as K

    assert K.sddmm_has_panels(torch.bfloat16, 256) and K.sddmm_has_panels(torch.float32, 256) and K.sddmm_has_panels(torch.float64, 512)
    assert not K.sddmm_has_panels(torch.bfloat16, 200) and not K.sddmm_has_panels(torch.float32, 7)
    assert not K.sddmm_has_panels(torch.int32, 256)
    a = torch.empty((100_000, 256), dtype=torch.bfloat16, device="meta")
    bt = torch.empty((100_000, 256), dtype=torch.bfloat16, device="meta")
    w = K.sddmm_panel_width(bt)
    assert w == 6250          # 16 panels (two per XCD) of 3.05 MiB; (3 << 20) // 512 = 6144 with shared panels
    assert K.sddmm_panels_pay(10_000_000, a, bt, w) and not K.sddmm_panels_pay(3_000_000, a, bt, w)
    assert not K.sddmm_panels_pay(10_000_000, a, bt, 0) and not K.sddmm_panels_pay(1000, a, bt, w)
    small = torch.empty((4096, 256), dtype=torch.bfloat16, device="meta")
    assert K.sddmm_panel_width(small) == 0
    # tiles: a block-sparse plan pays, a half-filled clustered one (its left-over samples would leave the panel order) does not
    plan = lambda ntiles, dense, rest: SimpleNamespace(tiles=torch.empty(ntiles, device="meta"), n_dense_samples=dense,
                                                       rest=torch.empty(rest, device="meta"), nnz=dense + rest)
    assert K.sddmm_tiles_pay(plan(9765, 9_999_360, 0), a, bt, w)
    assert not K.sddmm_tiles_pay(plan(13671, 7_001_614, 2_996_267), a, bt, w)
    assert not K.sddmm_tiles_pay(plan(0, 0, 10_000_000), a, bt, w)
    assert not K.sddmm_tiles_pay(plan(1000, 300_000, 0), a, bt, w)      # 300 samples per tile: below what a tile product costs


def test_one_pass_inspector_group_offsets_never_overlap():
    """The tiled-SpMM inspector starts row group g at ceil((e0 + g * tiles * (EPB - 1)) / EPB) blocks (csrc/spmm_tiled.hip
    `tl_group_first_block`): for any split of the elements into lists, a group's lists (each padded to whole blocks) end
    before the next group starts, the gap is less than one block per list, and the total stays inside the allocation
    ceil(nnz / EPB) + lists."""
    rng = np.random.default_rng(0)
    for epb in (8, 5):
        for _ in range(200):
            tiles = int(rng.integers(1, 70))
            groups = int(rng.integers(1, 40))
            counts = rng.integers(0, 40, size=(groups, tiles)) * (rng.random((groups, tiles)) < 0.8)
            e0 = np.concatenate([[0], np.cumsum(counts.sum(axis=1))])
            first = lambda e, g: (int(e) + g * tiles * (epb - 1) + epb - 1) // epb
            for g in range(groups):
                used = int(np.sum(-(-counts[g] // epb)))
                assert first(e0[g], g) + used <= first(e0[g + 1], g + 1)
                assert first(e0[g + 1], g + 1) - (first(e0[g], g) + used) <= tiles
            assert first(e0[-1], groups) <= -(-int(e0[-1]) // epb) + groups * tiles


def test_elemwise_tracer_records_numpys_own_dtypes_and_refuses_what_is_not_exact():
    """`_trace.build`: a plain callable over exactly-rounded operations becomes a graph whose every node carries the dtype
    NumPy itself gives that step; anything else (transcendental functions, keywords, data-dependent control flow, other
    dtypes) is refused, so that the caller evaluates it on the host as the reference does."""
    import numpy as np

    from sparse_amd import _trace

    f32, f64, i32, i64 = (np.dtype(x) for x in ("f4", "f8", "i4", "i8"))

    def dt(func, *spec):
        root = _trace.build(func, list(spec))
        return None if root is None else root.dtype

    A = lambda d: ("array", d)   # noqa: E731
    S = lambda v: ("scalar", v)  # noqa: E731
    assert dt(lambda a, b, c: a * b + c, A(f64), A(f64), A(f64)) == f64
    assert dt(lambda a, b: a * b + 3, A(f32), A(f32)) == f32                      # Python scalars are weak (NEP 50)
    assert dt(lambda a, b: a * b + np.float64(3), A(f32), A(f32)) == f64
    assert dt(lambda a, b: a / b, A(i32), A(i64)) == f64
    assert dt(lambda a, b: (a > b) & (a != 0), A(f64), A(f32)) == np.dtype(bool)
    assert dt(lambda a, b, c, d: np.maximum(a, b) - c * d, A(f64), A(f64), A(f64), S(2.0)) == f64
    assert dt(lambda a, b: np.where(a > 0, a, b), A(f32), A(f64)) == f64
    assert dt(lambda a: abs(a) ** 2 - (-a), A(i32)) == i32
    assert dt(lambda a: a.astype(np.float32) * 2, A(i64)) == f32
    assert dt(lambda a: np.sin(a) ** 2, A(f64)) is None                          # not bit-identical on the device
    assert dt(lambda a: a ** 3, A(f64)) is None
    assert dt(lambda a: a // 2, A(i64)) == i64 and dt(lambda a, b: a % b, A(f32), A(f64)) == f64      # (late round 6: NumPy's divmod rules in the kernel)
    assert dt(lambda a: a << 2, A(i32)) == i32 and dt(lambda a, b: np.copysign(a, b), A(i32), A(f32)) == f64
    assert dt(lambda a, b: np.hypot(a, b), A(f64), A(f64)) is None and dt(lambda a: np.gcd(a, 6), A(i64)) is None
    assert dt(lambda a: np.clip(a, 0, 1), A(f64)) is None
    assert dt(lambda a: a if a > 0 else -a, A(f64)) is None                       # data-dependent control flow
    assert dt(lambda a: a + 1, A(np.dtype("f2"))) is None and dt(lambda a: a + 1j, A(f64)) is None
    assert dt(lambda a: 5.0, A(f64)) is None                                      # a constant is not a graph


def test_committed_bench_rows_move_the_bytes_they_claim():
    """Round-3 verdict, item 1: a `paths` row whose kernels moved fewer HBM-side bytes (rocprofv3 PMC passes,
    profiles/paths_pmc.json) than 0.9 x its algorithmic bytes claims work the timed region does not do.  Checked on the
    committed line of the latest evidence run; rows whose operands fit the Infinity Cache are exempt (their fabric counters
    undercount by design)."""
    import glob
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = sorted(glob.glob(os.path.join(root, "profiles", "r0*_bench_line.json")))
    assert lines
    line = json.load(open(lines[-1]))
    if "paths" not in line:
        # round 5: the driver-parsed line is compact; the full rows of the same run are published beside it
        line["paths"] = json.load(open(lines[-1].replace("_bench_line.json", "_paths.json")))
        assert line["roofline"]["paths_accounting_errors"] == []
    pmc = json.load(open(os.path.join(root, "profiles", "paths_pmc.json")))["rows"]
    checked = 0
    for rid, row in line["paths"].items():
        if not isinstance(row, dict) or rid not in pmc or pmc[rid].get("cache_resident"):
            continue
        assert pmc[rid]["pmc_bytes"] >= 0.9 * row["algorithmic_bytes"], (rid, pmc[rid]["pmc_bytes"], row["algorithmic_bytes"])
        checked += 1
    assert checked >= 5
    assert not line["paths"].get("_accounting_errors")


def test_executor_policy_bounds_follow_the_result_width():
    """`_dot._tiled_min_rows / _tiled_min_density / _coo_first_product_tiled` (round 4: measured crossovers, see their
    docstrings): wider results and 8-byte values lower the bounds, results narrower than a panel keep rounds 1-3's, and the
    functions are monotone in the width."""
    from sparse_amd import _dot

    assert _dot._tiled_min_rows(128) == 65536 and _dot._tiled_min_rows(508) == 65536        # narrower than a 512-byte panel
    assert _dot._tiled_min_rows(512) == 45056
    assert _dot._tiled_min_rows(1024) == 20480 and _dot._tiled_min_rows(2048) == 10240 and _dot._tiled_min_rows(4096) == 5120
    assert _dot._tiled_min_rows(1 << 20) == 4096                                               # never below 4096 rows
    widths = [512 * k for k in range(1, 40)]
    rows = [_dot._tiled_min_rows(w) for w in widths]
    assert all(a >= b for a, b in zip(rows, rows[1:]))
    small_b, big_b = 1 << 20, 32 << 20
    assert _dot._tiled_min_density(128, small_b) == 12 and _dot._tiled_min_density(128, big_b) == 6
    assert [_dot._tiled_min_density(w, small_b) for w in (512, 1024, 2048, 4096)] == [9, 6, 5, 5]
    assert _dot._tiled_min_density(512, big_b) == 5
    # a COO operand's first product: one panel from 2 x 10^7 stored elements, wider from nnz x row bytes = 4 x 10^9
    assert not _dot._coo_first_product_tiled(19_999_999, 512) and _dot._coo_first_product_tiled(20_000_000, 512)
    assert not _dot._coo_first_product_tiled(1_900_000, 2048) and _dot._coo_first_product_tiled(2_000_000, 2048)
    assert _dot._coo_first_product_tiled(1_000_000, 4096) and not _dot._coo_first_product_tiled(7_000_000, 256)


def test_csc_inspector_workspace_covers_its_arrays(hiplib):
    """`spamd_spmm_tiled_inspect_csc_ws` (int32 words): split[(row blocks + 1) x K] + four count partials + the relative
    offsets + two 8-byte arrays over the groups, whatever the alignment padding"""
    rg, kb, gpb = 35, 160, 16
    for M, K in ((1, 1), (559, 161), (560 * 3, 1000), (1_000_000, 10_000), (70_001, 40_960)):
        groups = -(-(-(-M // rg)) // gpb) * gpb
        ntiles = -(-K // kb)
        need = (groups // gpb + 1) * K + 4 * groups * ntiles + groups * (ntiles + 1) + 1 + 2 * (2 * groups + 1)
        assert hiplib.spamd_spmm_tiled_inspect_csc_ws(M, K) >= need


def test_broadcast_axis_groups_of_the_one_pass_expansion():
    """`_broadcast._broadcast_groups`: the target's axes as (broadcast)(own)(broadcast)(own)(broadcast) group sizes, or None when
    the operand's own axes form more than two groups (csrc/broadcast.hip takes the former, the sort-based construction the rest)."""
    from sparse_amd._broadcast import _broadcast_groups as g

    assert g((30, 1, 40), (30, 25, 40)) == (1, 30, 25, 40, 1)          # the reference benchmark's left operand
    assert g((1, 35, 40), (20, 35, 40)) == (1, 1, 20, 35 * 40, 1)      # ... and its right one: a leading group
    assert g((30, 40, 1), (30, 40, 7)) == (1, 1, 1, 30 * 40, 7)        # trailing
    assert g((1, 9, 1, 11, 1), (4, 9, 5, 11, 3)) == (4, 9, 5, 11, 3)   # all five groups
    assert g((6, 1, 1, 7), (6, 3, 2, 7)) == (1, 6, 6, 7, 1)            # adjacent broadcast axes are one group
    assert g((1, 1), (5, 6)) == (1, 1, 1, 1, 30)                        # nothing of its own: one trailing group
    assert g((5, 1, 6, 1, 7), (5, 2, 6, 3, 7)) is None                 # three groups of own axes
    assert g((2, 1, 3), (2, 1, 3)) == (1, 1, 1, 6, 1)                  # (axes of size 1 in both shapes belong to their neighbours)
    for xs, shape in (((30, 1, 40), (30, 25, 40)), ((1, 9, 1, 11, 1), (4, 9, 5, 11, 3))):
        b0, k1, b1, k2, b2 = g(xs, shape)
        assert b0 * k1 * b1 * k2 * b2 == int(np.prod(shape)) and k1 * k2 == int(np.prod(xs))


def test_slab_merge_plan_of_leading_axis_reductions():
    """`_kernels.lead_last_plan`: ranges of a power-of-two number of kept cells holding ~2048 elements each; the sort keeps
    problems with more than 2048 runs and key spaces far larger than the element count."""
    from sparse_amd._kernels import lead_last_plan as plan

    assert plan(10 ** 6, 1000, 10 ** 6) == (2048, 489)          # config 1, sum(axis=0)
    assert plan(2 * 10 ** 6, 1000, 10 ** 6) == (1024, 977)      # ... of the sum of two such arrays
    assert plan(7200, 2000, 4) == (1, 4)                        # four cells for 7200 elements: the kernel will give up, not the plan
    assert plan(10 ** 6, 4096, 10 ** 6) is None                 # more runs than a workgroup has threads for
    assert plan(1000, 100, 10 ** 12) is None                    # 10^3 elements in a 10^12-cell key space
    assert plan(0, 10, 10) is None and plan(2 ** 31, 10, 10 ** 6) is None
    assert plan(10 ** 7, 1000, 10 ** 6) is None                 # beyond ~4 x 10^6 elements the radix sort is the faster order
    assert plan(4 * 10 ** 6, 1000, 10 ** 6) is None             # ... and so is a boundary table beyond the L2 (1000 runs x 1955 ranges)
    for n, S, P in ((10 ** 5, 37, 10 ** 7), (4 * 10 ** 6, 100, 10 ** 6), (123456, 1, 999)):
        p = plan(n, S, P)
        if p is not None:
            cells, ranges = p
            assert cells & (cells - 1) == 0 and 1 <= cells <= 2048 and (ranges - 1) * cells < P <= ranges * cells


def test_gcxs_index_width_follows_the_number_of_stored_elements():
    """int32 indices with int64 pointers: one width for both - the indices' while int32 pointers can hold the number of stored
    elements, int64 beyond (round 6: 2.25 x 10^9 elements had their pointers narrowed to int32; tools/r06/big_nnz_check.py runs
    the products at that size on the GPU)"""
    import torch

    from sparse_amd._gcxs import unified_index_dtype

    assert unified_index_dtype(torch.int32, 10) == torch.int32
    assert unified_index_dtype(torch.int32, 2 ** 31 - 1) == torch.int32
    assert unified_index_dtype(torch.int32, 2 ** 31) == torch.int64
    assert unified_index_dtype(torch.int64, 5) == torch.int64


def test_hub_row_piece_plan():
    """`_dot.hot_piece_plan`: the pieces of the hot rows cover their elements exactly once and in order, no piece is longer
    than the piece size (32 .. 4096, a multiple of 32) or crosses a row, and no combine loop is longer than the fan-in where
    two levels are used."""
    import numpy as np

    from sparse_amd import _dot as D

    rng = np.random.default_rng(0)
    for case in range(200):
        H = int(rng.integers(1, 40))
        lens = rng.integers(4096, int(rng.choice([5000, 40_000, 2_000_000])), size=H)
        gaps = rng.integers(0, 1000, size=H)
        p0 = np.cumsum(gaps + np.concatenate(([0], lens[:-1])))
        p1 = p0 + lens
        vptr, vfirst, g1 = D.hot_piece_plan(p0, p1)
        sizes = np.diff(vptr)
        piece = int(sizes.max())
        assert vptr[0] == 0 and vptr[-1] == lens.sum() and (sizes > 0).all()
        assert 32 <= piece <= 4096 and piece % 32 == 0
        ends = np.cumsum(lens)
        assert np.isin(ends, vptr).all()                             # no piece crosses a hot row's end
        pieces_per_row = np.diff(np.searchsorted(vptr, np.concatenate(([0], ends))))
        assert (pieces_per_row == -(-lens // piece)).all()
        if g1 is None:
            assert pieces_per_row.max() <= D.HOT_COMBINE_FAN and (np.diff(vfirst) == pieces_per_row).all()
        else:
            assert pieces_per_row.max() > D.HOT_COMBINE_FAN
            assert g1[0] == 0 and g1[-1] == len(vptr) - 1 and (np.diff(g1) > 0).all() and np.diff(g1).max() <= D.HOT_COMBINE_FAN
            assert np.isin(np.searchsorted(vptr, ends), g1).all()    # a group never spans two rows
            assert (np.diff(vfirst) == -(-pieces_per_row // D.HOT_COMBINE_FAN)).all() and vfirst[-1] == len(g1) - 1
    v, f, g = D.hot_piece_plan([0], [4096])
    assert v.tolist() == list(range(0, 4097, 32)) and f.tolist() == [0, 128] and g is None
