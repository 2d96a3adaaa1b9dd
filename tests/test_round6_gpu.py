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


def _run_bench_under_launcher(extra):
    """`bench.py` the way the driver launches it for N > 1 (torch.distributed.run, one rank per GPU, RCCL), at N = 1: the
    launcher path - RANK / LOCAL_RANK / WORLD_SIZE from the environment, init_process_group("nccl"), barriers, the max over
    ranks - had never run under the driver (round-5 verdict 3c).  Returns the parsed last stdout line."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_runs_under_the_distributed_launcher():
    line = _run_bench_under_launcher(["--steps", "2", "--warmup", "1", "--no-cpu", "--no-paths"])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["unit"] == "GFLOP/s" and line["value"] > 0
    assert line["config"]["world_size"] == 1 and "roofline" in line and 0 < line["roofline"]["frac"] < 1


def test_spgemm_workload_runs_under_the_distributed_launcher():
    """configs[4] at a tenth of its side (10^5 x 10^5 at 10^-3: the same 100 stored elements per row and 10^4 products per
    row), in two pieces per rank"""
    line = _run_bench_under_launcher(["--workload", "spgemm", "--spgemm-n", "100000", "--spgemm-density", "0.001", "--steps", "2",
                                      "--warmup", "1", "--chunk-rows", "50000", "--no-cpu"])
    assert line["metric"].startswith("GCXS x GCXS SpGEMM") and line["n_gpus"] == 1 and line["value"] > 0
    cfg = line["config"]
    assert cfg["pieces_rank0"] == 2 and sum(cfg["products_per_rank"]) > 9e8 and line["roofline"]["kernel_ms"] > 0


def test_results_beyond_2_to_31_stored_elements(orc):
    """Two adjacent row blocks of config 5 on one GPU through `a @ b`: 2.5 x 10^9 products and ~2.49 x 10^9 stored elements in
    ONE result (`_dot_csr_csr` uses intp throughout, _common.py:669-671; no test had crossed 2^31): pointers int64 and
    monotone, 200 sampled rows bit-equal to the oracle (columns after the canonical sort; values are the same left-to-right
    sums), a second run identical on the sampled rows, and the first block's rows equal to the block computed alone."""
    import sparse_amd as sp

    n = 1_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip("needs ~100 GB of free HBM")
    g = sp.random((n, n), density=1e-4, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    rows = n // 4
    p1 = int(g.indptr[rows])
    a = sp.GCXS((g.data[:p1].contiguous(), g.indices[:p1].contiguous(), g.indptr[:rows + 1].contiguous()), shape=(rows, n),
                compressed_axes=(0,))
    c = a @ g
    assert c.nnz > 2 ** 31 and c.indptr.dtype == torch.int64 and int(c.indptr[-1]) == c.nnz
    assert bool((c.indptr[1:] >= c.indptr[:-1]).all()) and int(c.indptr[0]) == 0
    rng = np.random.default_rng(3)
    pick = np.sort(np.concatenate([rng.choice(rows, 196, replace=False), [0, 1, rows - 2, rows - 1]]))
    hA = [t.cpu().numpy() for t in (a.data, a.indices, a.indptr)]
    hB = [t.cpu().numpy() for t in (g.data, g.indices, g.indptr)]
    segs = [np.arange(hA[2][r], hA[2][r + 1]) for r in pick]
    sub_ptr = np.zeros(len(pick) + 1, dtype=hA[2].dtype)
    sub_ptr[1:] = np.cumsum([len(s) for s in segs])
    sel = np.concatenate(segs)
    wd, wi, wp = orc.dot_csr_csr((len(pick), n), hA[0][sel], hB[0], hA[1][sel], hB[1], sub_ptr, hB[2])
    cp = c.indptr.cpu().numpy()
    got = []
    for j, r in enumerate(pick):
        lo, hi = int(cp[r]), int(cp[r + 1])
        gi, gd = c.indices[lo:hi].cpu().numpy(), c.data[lo:hi].cpu().numpy()
        wl, wh = int(wp[j]), int(wp[j + 1])
        o = np.argsort(wi[wl:wh], kind="stable")
        keep = wd[wl:wh][o].view(np.uint32) != 0
        assert np.array_equal(gi, wi[wl:wh][o][keep]), r
        assert np.array_equal(gd, wd[wl:wh][o][keep]), r
        got.append((gi, gd))
    assert hi > 2 ** 31                      # the last sampled row lies beyond the 32-bit range of positions
    del c
    torch.cuda.empty_cache()
    c2 = a @ g
    cp2 = c2.indptr.cpu().numpy()
    assert np.array_equal(cp, cp2)
    for (gi, gd), r in zip(got, pick):
        lo, hi = int(cp2[r]), int(cp2[r + 1])
        assert np.array_equal(c2.indices[lo:hi].cpu().numpy(), gi) and np.array_equal(c2.data[lo:hi].cpu().numpy(), gd)


def test_all_gather_csr_rebases_pointers_beyond_2_to_31():
    """`all_gather_csr`'s pointer rebasing with a total beyond 2^31 stored elements (one rank, RCCL at world size 1 is enough
    to run the packing / slicing / rebasing code on the device): 2.2 x 10^9 one-byte... no - int8 values are not a CSR value
    type; float32 values and int32 columns, 17.6 GB of them."""
    import torch.distributed as dist

    from sparse_amd import _dist

    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip("needs ~60 GB of free HBM")
    started = False
    if not dist.is_initialized():
        from conftest import init_single_rank_group

        init_single_rank_group("nccl")
        started = True
    try:
        rows, per = 2_200_000, 1000
        nnz = rows * per                               # 2.2e9 > 2^31
        data = torch.empty(nnz, dtype=torch.float32, device="cuda")
        data[::1000003] = 1.5
        idx = torch.empty(nnz, dtype=torch.int32, device="cuda")
        idx[-5:] = torch.arange(5, dtype=torch.int32, device="cuda")
        ptr = torch.arange(0, nnz + 1, per, dtype=torch.int64, device="cuda")
        d, i, ip = _dist.all_gather_csr(data, idx, ptr)
        assert ip.dtype == torch.int64 and int(ip[-1]) == nnz and int(ip[rows // 2]) == (rows // 2) * per
        assert bool((ip[1:] - ip[:-1] == per).all())
        assert torch.equal(i[-5:], idx[-5:]) and float(d[2 * 1000003]) == 1.5 and d.numel() == nnz
    finally:
        if started:
            dist.destroy_process_group()


@pytest.mark.parametrize("dshape", [(50,), (40, 1), (30, 1, 1), (1, 40, 50), (40, 50), (1,), (30, 40, 50)])
def test_dense_operand_that_broadcasts_into_the_sparse_shape(dshape):
    """`x * d[None, :]` - a row / column scaling: the dense operand is never copied out at the sparse array's shape (round 6:
    it was - 8 GB for a vector of 1000 against a 1000^3 array); results as NumPy's on the dense arrays, either operand order,
    and the reference's error when func(fill, dense) is not constant (reference _umath.py:505-555)"""
    import sparse_amd as sp

    rng = np.random.default_rng(len(dshape) * 7 + dshape[0])
    xd = np.where(rng.random((30, 40, 50)) < 0.2, rng.random((30, 40, 50)) - 0.5, 0.0)
    d = rng.random(dshape) + 0.5
    x = sp.COO.from_numpy(xd)
    for f in (np.multiply, np.divide):
        got = f(x, d)
        assert isinstance(got, sp.COO) and got.shape == xd.shape and np.array_equal(got.todense(), f(xd, d))
    got = np.multiply(d, x)
    assert np.array_equal(got.todense(), d * xd)
    g = sp.GCXS.from_numpy(xd.reshape(30, 2000))
    if len(dshape) == 1 and dshape[0] == 50:
        dv = rng.random(2000) + 0.5
        assert np.array_equal((g * dv).todense(), xd.reshape(30, 2000) * dv)
    if dshape == (1,):        # one value: func(fill, dense) IS constant - a sparse result with that fill value
        got = x + d
        assert isinstance(got, sp.COO) and got.fill_value == d[0] and np.array_equal(got.todense(), xd + d)
    elif dshape != (30, 40, 50):
        with pytest.raises(ValueError, match="dense array"):
            x + d


def test_scaling_a_large_array_by_a_vector_does_not_touch_its_full_shape():
    """10^6 stored elements of a 4000^3 COO times a vector of 4000: the full shape has 6.4 x 10^10 cells (512 GB of float64)"""
    import sparse_amd as sp

    x = sp.random((4000, 4000, 4000), density=1.6e-5, random_state=3)
    v = np.random.default_rng(1).random(4000) + 1.0
    y = x * v
    c = x.coords.cpu().numpy()
    assert y.nnz == x.nnz and np.array_equal(y.data.cpu().numpy(), x.data.cpu().numpy() * v[c[2]])
    y1 = x * v[:, None]
    assert np.array_equal(y1.data.cpu().numpy(), x.data.cpu().numpy() * v[c[1]])


@pytest.mark.parametrize("dtype, idt", [(np.float32, np.int32), (np.float64, np.int64), (np.int64, np.int64)])
@pytest.mark.parametrize("n_row, n, per_row", [(700, 900, 120), (2500, 2500, 60), (300, 7500, 170), (5000, 1500, 25)])
def test_dense_ish_sparse_products_take_the_accumulator_kernel_and_keep_their_bits(dtype, idt, n_row, n, per_row, orc):
    """A @ B whose result has about one product per cell or more (round 6): once the row products are known the
    dense-accumulator kernel takes the product (a wave per row; the bucket kernel's rows overflowed their buckets and were
    redone by the global form: 3000 x 3000 with 300 per row 28.6 -> 1.1 ms).  Same indices and the same bits as the other
    kernels give, and as the oracle's `_dot_csr_csr` (reference _common.py:639-717) on sampled rows."""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    kw = dict(dtype=np.float64 if np.dtype(dtype).kind == "i" else dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
    a = sp.random((n_row, n), density=per_row / n, random_state=11, **kw)
    b = sp.random((n, n), density=per_row / n, random_state=12, **kw)
    if np.dtype(dtype).kind == "i":
        a = sp.GCXS(((a.data * 200 - 100).to(torch.int64), a.indices, a.indptr), shape=a.shape, compressed_axes=(0,))
        b = sp.GCXS(((b.data * 200 - 100).to(torch.int64), b.indices, b.indptr), shape=b.shape, compressed_axes=(0,))
    c = a @ b
    took = K.SPGEMM_STATS.get("kernel")
    assert took == "small", took
    K.SPGEMM_SMALL = False
    try:
        ref = a @ b
        assert K.SPGEMM_STATS.get("kernel") != "small"
    finally:
        K.SPGEMM_SMALL = True
    assert c.nnz == ref.nnz and torch.equal(c.indptr.long(), ref.indptr.long()) and torch.equal(c.indices.long(), ref.indices.long())
    assert torch.equal(c.data, ref.data)
    pick = np.sort(np.random.default_rng(2).choice(n_row, size=40, replace=False))
    hA = [t.cpu().numpy() for t in (a.data, a.indices, a.indptr)]
    hB = [t.cpu().numpy() for t in (b.data, b.indices, b.indptr)]
    segs = [np.arange(hA[2][r], hA[2][r + 1]) for r in pick]
    sub_ptr = np.zeros(len(pick) + 1, dtype=hA[2].dtype)
    sub_ptr[1:] = np.cumsum([len(s) for s in segs])
    sel = np.concatenate(segs)
    wd, wi, wp = orc.dot_csr_csr((len(pick), n), hA[0][sel], hB[0], hA[1][sel], hB[1], sub_ptr, hB[2])
    cp = c.indptr.cpu().numpy()
    for j, r in enumerate(pick):
        lo, hi = int(cp[r]), int(cp[r + 1])
        o = np.argsort(wi[wp[j]:wp[j + 1]], kind="stable")
        keep = wd[wp[j]:wp[j + 1]][o] != 0 if np.dtype(dtype).kind == "i" else wd[wp[j]:wp[j + 1]][o].view(np.uint64 if wd.itemsize == 8 else np.uint32) != 0
        assert np.array_equal(c.indices[lo:hi].cpu().numpy(), wi[wp[j]:wp[j + 1]][o][keep])
        assert np.array_equal(c.data[lo:hi].cpu().numpy(), wd[wp[j]:wp[j + 1]][o][keep])


def test_bucket_kernel_spreads_few_columns_over_its_passes():
    """float64 products of ~10^4 per row over 10^4 columns: the 8-pass class of the bucket kernel (csrc/spgemm_rows.hip) gave
    pass 0 the first 4096 columns - 41 % of a row's products for a pass with room for 14 % - and declined 97 % of the rows
    (round 6: columns spread over all buckets whatever their number)"""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    g = sp.random((2000, 10_000), density=0.01, random_state=4, format="gcxs", compressed_axes=(0,))
    h = sp.random((10_000, 10_000), density=0.0095, random_state=5, format="gcxs", compressed_axes=(0,))
    K.SPGEMM_SMALL_SECOND = False
    try:
        c = g @ h
        assert K.SPGEMM_STATS.get("kernel") == "buckets" and K.SPGEMM_STATS.get("heavy_or_declined", 0) < 20, dict(K.SPGEMM_STATS)
    finally:
        K.SPGEMM_SMALL_SECOND = True
    c2 = g @ h
    assert c.nnz == c2.nnz and torch.equal(c.indices.long(), c2.indices.long()) and torch.equal(c.data, c2.data)


def test_group_reduce_is_memory_safe_on_keys_that_are_no_keys():
    """`x.prod(axis=0)` of a nearly dense array: the slab merge (csrc/lead_rotate.hip) declines its ranges (more than 64 elements
    per cell), leaves its output buffers unwritten and says so in a device word the caller reads LATER - the grouped reduce
    runs on whatever those buffers hold.  Recycled memory full of 0xff bytes (the merge's own boundary table) made every key
    -1 = the kernels' "nothing before" mark: no run at all, and the boundary fix-up stored at run -1 - a device fault,
    round 6, tools/fuzz_dense.py.  The first element is a run head whatever its group."""
    from sparse_amd import _reduce as R

    dev = torch.device("cuda:0")
    n = 1_094_340
    data = torch.randint(-9, 10, (n,), device=dev, dtype=torch.int64)
    for keys in (torch.full((n,), -1, device=dev, dtype=torch.int64), torch.full((n,), -141, device=dev, dtype=torch.int64),
                 torch.randint(-2 ** 62, 2 ** 62, (n,), device=dev, dtype=torch.int64)):
        for op in ("multiply", "add"):
            out = R.group_reduce(keys, 141, data, op, key_bound=1_155_072, sync=False)
            torch.cuda.synchronize()
            assert 1 <= int(out[3][0]) <= n
    # and the reduction that met it: 141 x 64 x 64 x 2, about 5 % zeros, over the leading axis
    import sparse_amd as sp

    d = np.random.default_rng(5).integers(-9, 10, (141, 64, 64, 2)).astype(np.int64)
    x = sp.COO.from_numpy(d)
    junk = [torch.full((1 << 20,), -1, device=dev, dtype=torch.int64) for _ in range(6)]      # blocks of 0xff for the allocator to recycle
    del junk
    for name in ("prod", "sum", "max"):
        assert np.array_equal(getattr(x, name)(axis=0).todense(), getattr(d, name)(axis=0))


@pytest.mark.parametrize("shape", [(1, 1), (3, 70), (64, 64), (65, 129), (128, 100_003), (1000, 1), (1, 5000), (257, 513)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int32, torch.int16, torch.uint8, torch.bfloat16])
def test_transposed_copy(shape, dtype):
    """csrc/transpose.hip (the dense operand of `dense @ sparse`): the same bits as torch's `x.t().contiguous()`"""
    from sparse_amd import _kernels as K

    x = (torch.rand(shape, device="cuda", dtype=torch.float64) * 200 - 100).to(dtype)
    got = K.transposed_copy(x)
    want = x.t().contiguous()
    assert got.shape == want.shape and got.dtype == want.dtype and got.is_contiguous()
    assert torch.equal(got.float() if dtype == torch.bfloat16 else got, want.float() if dtype == torch.bfloat16 else want)
    # a strided operand keeps torch's copy
    y = x[:, ::2] if shape[1] > 1 else x
    assert torch.equal(K.transposed_copy(y), y.t().contiguous())


def test_csr_times_dense_with_more_than_2_to_31_result_elements():
    """17 x 10^6 rows x 128 columns = 2.2 x 10^9 result elements (8.7 GB): the executor (what `a @ b` takes here) and the
    row-group kernel against a float64 evaluation of rows at the head, around element 2^31 and at the very end"""
    import sparse_amd as sp
    from bench import make_csr_device
    from sparse_amd import _kernels as K

    M, Kd, N = 17_000_000, 1000, 128
    d, i, p = make_csr_device(M, Kd, 0.003, 9)
    b = torch.rand((Kd, N), device="cuda") - 0.5
    a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
    rows = np.concatenate([np.arange(0, 3), np.arange(16_777_214, 16_777_219), np.arange(M - 3, M)])
    pc = p[torch.from_numpy(np.concatenate([rows, rows + 1])).cuda()].cpu().numpy().reshape(2, -1)

    def check(c):
        for k, r in enumerate(rows):
            lo, hi = int(pc[0][k]), int(pc[1][k])
            want = (d[lo:hi].double()[:, None] * b[i[lo:hi].long()].double()).sum(0)
            assert torch.allclose(c[int(r)].double(), want, rtol=1e-5, atol=1e-6), int(r)

    c = a @ b
    assert getattr(a, "_tiled_layouts", None) and c.shape == (M, N)
    check(c)
    del c
    check(K.dot_csr_ndarray((M, N), d, i, p, b, keep_order=True))


@pytest.mark.parametrize("n", [1, 5, 63, 511, 2049, 8192 * 4 + 3, 3_000_017, 40_000_003])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_everything_reduced_without_the_keys(n, dtype):
    """axis=None: one group, found by `spamd_reduce_all` from the values alone (csrc/group_reduce.hip).  Exact for integers,
    max / min; sums to re-association; the same results as the grouped path over the keys; the workspace's ticket word is
    left ready (two calls in a row); a value array that starts off a 16-byte boundary takes the element-wise loads."""
    import sparse_amd as sp
    from sparse_amd import _reduce as R

    rng = np.random.default_rng(n % 1000)
    size = max(4 * n, 8)
    lin = np.sort(rng.choice(size, size=n, replace=False)) if n < 5_000_000 else np.arange(n, dtype=np.int64) * 4 + 1
    vals = (rng.integers(-9, 10, size=n) if np.dtype(dtype).kind == "i" else rng.random(n) - 0.3).astype(dtype)
    vals[vals == 0] = 1
    shape = (2, size // 2)
    x = sp.COO(np.stack(np.unravel_index(lin, shape)), vals, shape=shape)
    dense_sum = vals.sum(dtype=np.float64 if np.dtype(dtype).kind == "f" else np.int64)
    for rep in range(2):
        got = x.sum()
        if np.dtype(dtype).kind == "i":
            assert int(got) == int(dense_sum)
        else:
            assert abs(float(got) - float(dense_sum)) <= (1e-4 if dtype == np.float32 else 1e-10) * max(1.0, np.abs(vals).sum())
    assert x.max() == max(vals.max(), 0) and x.min() == min(vals.min(), 0)
    assert bool(x.any()) and not bool(x.all())
    kd = x.sum(keepdims=True)
    assert kd.shape == (1, 1)
    R.REDUCE_ALL_DIRECT = False
    try:
        old = (x.max(), x.min(), x.sum())
    finally:
        R.REDUCE_ALL_DIRECT = True
    assert old[0] == x.max() and old[1] == x.min()
    if np.dtype(dtype).kind == "i":
        assert old[2] == x.sum()
    # the C ABI on a pointer 1 element past an aligned one
    d = torch.from_numpy(vals).cuda()
    if n > 1:
        g, v, c, ng = R.reduce_all(d[1:], "add")
        assert ng.tolist() == [1, 0] and int(c[0]) == n - 1 and int(g[0]) == 0
        want = vals[1:].sum(dtype=np.float64 if np.dtype(dtype).kind == "f" else np.int64)
        if np.dtype(dtype).kind == "i":
            assert int(v[0]) == int(np.asarray(want).astype(dtype))
        else:
            assert abs(float(v[0]) - float(want)) <= (1e-4 if dtype == np.float32 else 1e-10) * max(1.0, np.abs(vals).sum())


def test_everything_reduced_nan_rules_and_products():
    """np.maximum propagates a NaN, np.fmax skips it, wherever it sits among the pieces; a product over all elements."""
    import sparse_amd as sp

    n = 2_100_000
    vals = np.full(n, 0.5)
    vals[n - 7] = np.nan
    x = sp.COO(np.arange(n)[None, :] * 2, vals, shape=(2 * n,))
    assert np.isnan(x.max()) and np.isnan(x.min())
    assert sp.nanmax(x) == 0.5 and sp.nanmin(x) == 0.0
    y = sp.COO(np.arange(40)[None, :], np.full(40, 2.0), shape=(40,), fill_value=1.0)
    assert y.prod() == 2.0 ** 40


@pytest.mark.parametrize("ca", [(0,), (1,)])
def test_gcxs_reduced_over_every_axis_needs_no_coordinates(ca):
    """A GCXS with every axis reduced goes to `spamd_reduce_all` with its stored values alone (no conversion to COO); axis given
    as None, as a tuple of all axes (negative too), with keepdims; 3-D; an operand dtype the kernels do not cover still
    takes the host route."""
    import sparse_amd as sp

    x = sp.random((300, 200), density=0.05, random_state=5, format="gcxs", compressed_axes=ca)
    d = x.todense()
    for ax in (None, (0, 1), (-1, 0)):
        assert abs(float(x.sum(axis=ax)) - d.sum()) < 1e-9
        assert float(x.max(axis=ax)) == d.max() and float(x.min(axis=ax)) == d.min()
    kd = x.sum(axis=None, keepdims=True)
    assert kd.shape == (1, 1) and abs(float(kd.todense()[0, 0]) - d.sum()) < 1e-9
    assert abs(float(x.sum(axis=(0,)).todense().sum()) - d.sum()) < 1e-9          # (one axis: the grouped path as before)
    y = sp.random((20, 30, 40), density=0.05, random_state=6, format="gcxs")
    assert abs(float(y.sum()) - y.todense().sum()) < 1e-9
    assert abs(float(y.sum(axis=(0, 1, 2))) - y.todense().sum()) < 1e-9
    with pytest.raises(Exception):
        x.sum(axis=(0, 2))
    c = sp.GCXS.from_numpy((d * (1 + 2j)).astype(np.complex128)) if hasattr(sp.GCXS, "from_numpy") else None
    if c is not None:
        assert abs(complex(c.sum()) - (d * (1 + 2j)).sum()) < 1e-9


@pytest.mark.parametrize("fmt", ["coo", "gcxs"])
@pytest.mark.parametrize("axis", [None, 0, 1, (0, 1)])
def test_variance_of_long_groups_and_nan_skipping_sums(fmt, axis):
    """var / std sum their groups with the grouped reduce (a run of any length; with every axis reduced: spamd_reduce_all) and
    the NaN-skipping sums replace NaNs on the value array: against NumPy on the dense twin, with and without NaNs, ddof."""
    import sparse_amd as sp

    rng = np.random.default_rng(3)
    d = rng.random((3, 200_000)) * (rng.random((3, 200_000)) < 0.3)
    x = sp.asarray(d, format=fmt) if fmt == "gcxs" else sp.COO.from_numpy(d)
    for dd in (0, 1):
        got = x.var(axis=axis, ddof=dd)
        np.testing.assert_allclose(np.asarray(got.todense() if hasattr(got, "todense") else got), d.var(axis=axis, ddof=dd), rtol=1e-10, atol=1e-14)
    got = x.std(axis=axis)
    np.testing.assert_allclose(np.asarray(got.todense() if hasattr(got, "todense") else got), d.std(axis=axis), rtol=1e-10, atol=1e-14)
    f32 = x.astype(np.float32).var(axis=axis)
    np.testing.assert_allclose(np.asarray(f32.todense() if hasattr(f32, "todense") else f32), d.astype(np.float32).var(axis=axis), rtol=2e-4)
    for planted in (False, True):
        e = d.copy()
        if planted:
            e[0, 5] = e[2, 77] = e[1, 199_999] = np.nan
        y = sp.asarray(e, format=fmt) if fmt == "gcxs" else sp.COO.from_numpy(e)
        for name in ("nansum", "nanprod", "nanmean", "nanmax", "nanmin"):
            got = getattr(sp, name)(y, axis=axis)
            want = getattr(np, name)(e, axis=axis)
            np.testing.assert_allclose(np.asarray(got.todense() if hasattr(got, "todense") else got), want, rtol=1e-10, atol=1e-14)
    z = sp.COO.from_numpy(np.where(d > 0.5, d, 2.0), fill_value=2.0)        # a fill value that is not zero
    got = z.var(axis=axis)
    np.testing.assert_allclose(np.asarray(got.todense() if hasattr(got, "todense") else got), np.where(d > 0.5, d, 2.0).var(axis=axis),
                               rtol=1e-10)


@pytest.mark.parametrize("ca", [(0,), (1,)])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_gcxs_slices_along_its_compressed_axis_without_a_coo(ca, idt):
    """`x[a:b]`, `x[i]` (CSR) / `x[:, a:b]`, `x[:, j]` (CSC): a pointer range and the elements between its ends; the result keeps
    the compressed axis (reference _compressed/indexing.py:14-176), equals NumPy's slice of the dense twin and works as an
    operand (product, sum) - also for empty ranges, empty rows at the ends, negative bounds."""
    import sparse_amd as sp
    from sparse_amd import _gcxs as G

    rng = np.random.default_rng(11)
    d = rng.random((230, 170)) * (rng.random((230, 170)) < 0.1)
    d[:7] = 0
    d[100:103] = 0
    d[:, 160:] = 0
    x = sp.asarray(d, format="gcxs", compressed_axes=ca, idx_dtype=idt) if False else sp.GCXS(sp.COO.from_numpy(d), compressed_axes=ca, idx_dtype=idt)
    n = d.shape[ca[0]]
    pre = (slice(None),) * ca[0]
    for sl in (slice(0, 5), slice(3, 120), slice(100, 103), slice(-40, None), slice(50, 50), slice(60, 20), slice(None), slice(n - 1, n + 5)):
        key = pre + (sl,)
        assert G._compressed_axis_slice(x, key) is not None
        got = x[key]
        assert isinstance(got, sp.GCXS) and got.compressed_axes == ca and got.shape == d[key].shape
        assert got.indices.dtype == x.indices.dtype and got.indptr.dtype == x.indptr.dtype
        assert np.array_equal(got.todense(), d[key])
        if got.shape[0] and got.shape[1]:
            b = rng.random((got.shape[1], 3))
            np.testing.assert_allclose(got @ b, d[key] @ b, rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(got.sum(axis=0).todense(), d[key].sum(axis=0), rtol=1e-12, atol=1e-14)
    for i in (0, 8, 101, n - 1, -1, -n):
        key = pre + (i,)
        got = x[key]
        assert got.shape == d[key].shape and np.array_equal(got.todense(), d[key])
    with pytest.raises(IndexError):
        x[pre + (n,)]
    # other forms still take the general route
    for key in ((slice(None, None, 2),), (slice(3, 9), slice(2, 5)), (np.array([1, 5]),), (None,)):
        assert G._compressed_axis_slice(x, key) is None
    assert np.array_equal(x[::2].todense(), d[::2]) and np.array_equal(x[3:9, 2:5].todense(), d[3:9, 2:5])


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64, np.int32])
def test_a_hot_coordinate_among_unique_ones_is_summed_by_the_grouped_reduce(dtype, monkeypatch):
    """`COO(coords, data)` with 3 x 10^5 duplicates of one coordinate (and of a second one, at the end) among 2 x 10^5 unique
    ones: the run-per-thread sum would walk the long runs alone (220 ms per 10^6 duplicates); the grouped reduce takes them.
    Exact for integers, 1e-12 / 1e-5 for floats; without a long run the left-to-right sums are kept bit for bit."""
    import sparse_amd as sp
    from sparse_amd import _reduce as R

    rng = np.random.default_rng(5)
    hot, uniq = 300_000, 200_000
    keys = np.concatenate([np.full(hot, 7), rng.choice(np.arange(8, 10_000_000), size=uniq, replace=False), np.full(hot, 10_000_001)])
    vals = (rng.integers(-5, 6, size=keys.size) if np.dtype(dtype).kind == "i" else rng.random(keys.size)).astype(dtype)
    perm = rng.permutation(keys.size)
    keys, vals = keys[perm], vals[perm]
    shape = (5000, 4000)
    called = []
    real = R.segment_reduce
    monkeypatch.setattr(R, "segment_reduce", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    c = sp.COO(np.stack(np.unravel_index(keys, shape)), vals, shape=shape)
    assert not called and c.nnz == uniq + 2
    want = np.zeros(shape, dtype=np.float64 if np.dtype(dtype).kind == "f" else dtype)
    np.add.at(want.reshape(-1), keys, vals)
    got = c.todense()
    if np.dtype(dtype).kind == "i":
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=1e-12 if dtype == np.float64 else 1e-4)
    # short runs only: the run-per-thread kernel, left to right
    k2 = np.repeat(rng.choice(10_000_000, size=50_000, replace=False), 3)
    v2 = rng.random(k2.size).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(-5, 6, size=k2.size).astype(dtype)
    c2 = sp.COO(np.stack(np.unravel_index(k2, shape)), v2, shape=shape)
    assert called
    order = np.argsort(k2, kind="stable")
    seq = (v2[order].reshape(-1, 3)[:, 0] + v2[order].reshape(-1, 3)[:, 1]) + v2[order].reshape(-1, 3)[:, 2]
    assert np.array_equal(c2.data.cpu().numpy(), seq)          # (sums that are zero stay stored: prune=False, as the reference's)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64])
@pytest.mark.parametrize("form", ["csr", "csc", "coo"])
def test_hub_rows_are_multiplied_in_pieces(dtype, form):
    """A matrix with two hub rows (15 000 and 5000 stored elements beside 3 per row) is multiplied as (matrix without them) +
    (their pieces as rows of their own) + `spamd_hot_rows_combine`: results as SciPy's on the host for 1 .. 128 columns,
    integers exact; the split is found once and kept with the operand; an operand without hub rows is left alone."""
    import scipy.sparse as ss

    import sparse_amd as sp
    from sparse_amd import _dot as D

    rng = np.random.default_rng(8)
    M, Kd = 150_000, 20_000
    r = np.concatenate([rng.integers(0, M, size=3 * M), np.full(15_000, 4242), np.full(5000, M - 1)])
    c = np.concatenate([rng.integers(0, Kd, size=3 * M), rng.choice(Kd, size=15_000, replace=False), rng.choice(Kd, size=5000, replace=False)])
    v = (rng.integers(-4, 5, size=r.size) if np.dtype(dtype).kind == "i" else rng.random(r.size) - 0.5).astype(dtype)
    host = ss.coo_matrix((v, (r, c)), shape=(M, Kd)).tocsr()
    host.sum_duplicates()
    a = sp.GCXS.from_scipy_sparse(host) if form == "csr" else (
        sp.GCXS.from_scipy_sparse(host.tocsc()) if form == "csc" else sp.COO.from_scipy_sparse(host))
    for n in (1, 4, 16, 128):
        b = (rng.integers(-3, 4, size=(Kd, n)) if np.dtype(dtype).kind == "i" else rng.random((Kd, n)) - 0.5).astype(dtype)
        got = np.asarray(a @ b)
        want = host @ b
        if np.dtype(dtype).kind == "i":
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-4 if dtype == np.float32 else 1e-11, atol=2e-4 if dtype == np.float32 else 1e-11)
    split = a.__dict__.get("_hot_split")
    if form != "csc" or split is not None:       # (a CSC operand may go through its own inspector, which needs no CSR twin)
        assert split is not None and sorted(split[3].tolist()) == [4242, M - 1]
        assert split[0].nnz + split[1].nnz == a.nnz and split[0].shape == (M, Kd)
        assert split[4] is not None          # (15 000 elements in pieces of 32: two levels of the combine)
    plain = sp.GCXS.from_scipy_sparse(host[:4000])
    plain @ np.ones((Kd, 4), dtype=dtype)
    assert plain.__dict__.get("_hot_split", None) is None
    D.HOT_ROW_SPLIT = False
    try:
        a2 = sp.GCXS.from_scipy_sparse(host)
        b = (rng.integers(-3, 4, size=(Kd, 8)) if np.dtype(dtype).kind == "i" else rng.random((Kd, 8)) - 0.5).astype(dtype)
        ref = np.asarray(a2 @ b)
    finally:
        D.HOT_ROW_SPLIT = True
    np.testing.assert_allclose(np.asarray(a @ b), ref, rtol=2e-4 if dtype == np.float32 else 1e-11, atol=2e-4 if dtype == np.float32 else 1e-11)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64])
def test_sparse_product_with_an_output_element_of_many_products(dtype):
    """a @ a.T with a hub row: the diagonal element of that row is the sum of 8000 products - a run the one-thread-per-element
    sum would walk alone; the grouped reduce takes it (`_kernels._spgemm_keys`).  Against SciPy; integers exact; the same
    result (to re-association) as with the probe switched off."""
    import scipy.sparse as ss

    import sparse_amd as sp
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(4)
    n = 20_000
    r = np.concatenate([rng.integers(0, n, size=5 * n), np.full(8000, 321)])
    c = np.concatenate([rng.integers(0, n, size=5 * n), rng.choice(n, size=8000, replace=False)])
    v = (rng.integers(-3, 4, size=r.size) if np.dtype(dtype).kind == "i" else rng.random(r.size) - 0.5).astype(dtype)
    h = ss.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    h.sum_duplicates()
    a = sp.GCXS.from_scipy_sparse(h)
    want = (h @ h.T).toarray()
    got = (a @ a.T).todense()
    old = K.SPGEMM_RUN_LONG
    K.SPGEMM_RUN_LONG = False
    try:
        ref = (a @ a.T).todense()
    finally:
        K.SPGEMM_RUN_LONG = old
    if np.dtype(dtype).kind == "i":
        assert np.array_equal(got, want) and np.array_equal(ref, want)
    else:
        tol = dict(rtol=1e-11, atol=1e-11) if dtype == np.float64 else dict(rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(got, want, **tol)
        np.testing.assert_allclose(got, ref, **tol)


@pytest.mark.parametrize("axis", [0, 1, -1])
def test_gcxs_matrices_joined_along_their_compressed_axis(axis):
    """`concatenate` of matrices compressed along the joined axis appends the arrays and shifts the pointers
    (`_batched._concatenate_compressed`): the dense result of NumPy's concatenate, a GCXS compressed along that axis, the same
    arrays as the route through COO; mixed value and index types, an operand without rows, one without stored elements."""
    import sparse_amd as sp
    from sparse_amd import _batched as B

    rng = np.random.default_rng(2)
    ax = axis % 2
    shapes = [(40, 30), (0, 30), (25, 30), (7, 30)] if ax == 0 else [(30, 40), (30, 0), (30, 25), (30, 7)]
    ds = [rng.random(s) * (rng.random(s) < 0.2) for s in shapes]
    ds[3][:] = 0
    ds[2] = ds[2].astype(np.float32)
    xs = [sp.GCXS(sp.COO.from_numpy(d), compressed_axes=(ax,), idx_dtype=(np.int64 if k == 2 else np.int32)) for k, d in enumerate(ds)]
    assert B._concatenate_compressed(xs, axis, None) is not None
    got = sp.concatenate(xs, axis=axis)
    want = np.concatenate(ds, axis=axis)
    assert isinstance(got, sp.GCXS) and got.compressed_axes == (ax,) and got.shape == want.shape and got.dtype == want.dtype
    assert got.indices.dtype == torch.int64 and got.indptr.dtype == torch.int64
    assert np.array_equal(got.todense(), want)
    slow = sp.concatenate([x.tocoo() for x in xs], axis=axis).asformat("gcxs", compressed_axes=(ax,))
    assert np.array_equal(slow.data.cpu().numpy(), got.data.cpu().numpy())
    assert np.array_equal(slow.indices.cpu().numpy(), got.indices.cpu().numpy())
    assert np.array_equal(slow.indptr.cpu().numpy(), got.indptr.cpu().numpy())
    b = rng.random((want.shape[1], 3))
    np.testing.assert_allclose(got @ b, want @ b, rtol=1e-12, atol=1e-14)
    # the other axis, a requested layout, a different fill value: the general route
    assert B._concatenate_compressed(xs[:1] + xs[2:3], 1 - ax, None) is None
    assert B._concatenate_compressed(xs, axis, (1 - ax,)) is None
    with pytest.raises(ValueError):
        sp.concatenate([xs[0], sp.GCXS(sp.COO.from_numpy(ds[2] + 1.0, fill_value=1.0), compressed_axes=(ax,))], axis=axis)


@pytest.mark.parametrize("ca", [(0,), (1,)])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_gcxs_slices_along_its_uncompressed_axis_without_a_coo(ca, idt):
    """`x[:, a:b]` (CSR) / `x[a:b]` (CSC): the elements inside the range compacted, the pointers read from the scan
    (`_gcxs._uncompressed_axis_slice`); NumPy's slice of the dense twin, the layout and index widths kept, usable as an operand."""
    import sparse_amd as sp
    from sparse_amd import _gcxs as G

    rng = np.random.default_rng(12)
    d = rng.random((210, 190)) * (rng.random((210, 190)) < 0.1)
    d[:, :9] = 0
    d[50:60] = 0
    x = sp.GCXS(sp.COO.from_numpy(d), compressed_axes=ca, idx_dtype=idt)
    un = 1 - ca[0]
    n = d.shape[un]
    pre = (slice(None),) * un
    for sl in (slice(0, 9), slice(5, 100), slice(-30, None), slice(70, 70), slice(90, 20), slice(1, n), slice(n - 1, n + 9), slice(None)):
        key = pre + (sl,)
        assert G._uncompressed_axis_slice(x, key) is not None or ca == (1,) and G._compressed_axis_slice(x, key) is not None
        got = x[key]
        assert isinstance(got, sp.GCXS) and got.compressed_axes == ca and got.shape == d[key].shape
        assert got.indices.dtype == x.indices.dtype and got.indptr.dtype == x.indptr.dtype
        assert np.array_equal(got.todense(), d[key])
        if got.shape[0] and got.shape[1]:
            b = rng.random((got.shape[1], 3))
            np.testing.assert_allclose(got @ b, d[key] @ b, rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(got.sum(axis=1).todense(), d[key].sum(axis=1), rtol=1e-12, atol=1e-14)
    assert G._uncompressed_axis_slice(x, pre + (slice(0, 50, 2),)) is None and G._uncompressed_axis_slice(x, pre + (3,)) is None
    e = sp.GCXS(sp.COO.from_numpy(np.zeros((20, 30))), compressed_axes=ca)
    assert e[pre + (slice(2, 9),)].nnz == 0


@pytest.mark.parametrize("fmt", ["coo", "gcxs"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int32])
def test_round_and_clip_stay_on_the_device(fmt, dtype):
    """`x.round(d)` = NumPy's own multiply / rint / divide recipe and `x.clip(lo, hi)` = minimum(maximum(x, lo), hi) as device
    ufuncs: the values, dtypes and fill values NumPy gives on the dense twin (bit for bit), no host evaluation."""
    import sparse_amd as sp

    rng = np.random.default_rng(6)
    d = ((rng.random((300, 200)) * 200 - 100) * (rng.random((300, 200)) < 0.2)).astype(dtype)
    if np.dtype(dtype).kind == "f":
        d[3, 4] = np.nan
    x = sp.asarray(d, format=fmt) if fmt == "gcxs" else sp.COO.from_numpy(d)
    counts = lambda: {k: v for k, v in sp.fallback_stats().items() if k != "recent"}
    before = counts()
    if np.dtype(dtype).kind == "f":
        for dec in (0, 1, 2, 5, -1, -2):
            got = x.round(dec)
            want = np.round(d, dec)
            assert got.dtype == want.dtype and np.array_equal(got.todense(), want, equal_nan=True), dec
    for lo, hi in ((-10, 20), (0.5, 30.25), (None, 5), (-3.5, None), (2, 1000), (-1000, -2)):
        got = x.clip(lo, hi)
        want = np.clip(d, lo, hi)
        assert got.dtype == want.dtype, (lo, hi, got.dtype, want.dtype)
        assert np.array_equal(got.todense(), want, equal_nan=True), (lo, hi)
        assert got.nnz <= d.size
    assert counts() == before
    with pytest.raises(ValueError):
        x.clip()


_LATE_OPS = ["floor_divide", "remainder", "fmod", "copysign", "hypot", "arctan2", "left_shift", "right_shift"]


@pytest.mark.parametrize("name", _LATE_OPS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64, np.int32])
def test_late_binary_ufuncs_on_the_device(name, dtype):
    """floor_divide / remainder / fmod (NumPy's npy_divmod rules for floats, Python's floor rules for integers), copysign, the
    shifts (bit for bit) and hypot / arctan2 (to 4 ulp) between two sparse arrays, with a scalar on either side, through the
    operators (`//`, `%`, `<<`, `>>`) and inside a traced lambda: NumPy's values, dtypes and fill values on the dense twins,
    zeros and negative operands included; no host evaluation."""
    import warnings

    import sparse_amd as sp

    f = getattr(np, name)
    kind = np.dtype(dtype).kind
    if name in ("left_shift", "right_shift") and kind == "f":
        pytest.skip("integer ufunc")
    rng = np.random.default_rng(7)
    shape = (120, 90)
    def draw():
        v = rng.integers(-40, 41, size=shape) if kind == "i" else np.round(rng.random(shape) * 80 - 40, 2)
        return (v * (rng.random(shape) < 0.3)).astype(dtype)
    da, db = draw(), draw()
    if name in ("left_shift", "right_shift"):
        db = np.abs(db) % 70            # (counts beyond the width included)
        db[0, :4] = [-1, 31, 32, 64]
    a, b = sp.COO.from_numpy(da), sp.COO.from_numpy(db)
    exact = name not in ("hypot", "arctan2")

    def same(got, want):
        gd = got.todense() if hasattr(got, "todense") else np.asarray(got)
        assert gd.dtype == want.dtype, (gd.dtype, want.dtype)
        if exact:
            assert np.array_equal(gd, want, equal_nan=True)
            if want.dtype.kind == "f":
                ok = ~np.isnan(want)          # (the sign of a NaN is the platform's: x86 fmod(x, 0) sets it, the device does not)
                assert np.array_equal(np.signbit(gd)[ok], np.signbit(want)[ok])
        else:
            np.testing.assert_allclose(gd, want, rtol=1e-6 if want.dtype == np.float32 else 1e-15, atol=0)
        fv = np.asarray(got.fill_value)
        assert fv.dtype == want.dtype

    counts = lambda: {k: v for k, v in sp.fallback_stats().items() if k != "recent"}
    before = counts()
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        same(f(a, b), f(da, db))
        sc = 3 if kind == "i" or name.endswith("shift") else 2.5
        same(f(a, sc), f(da, sc))
        same(f(sc, b), f(sc, db))
        if name == "floor_divide":
            same(a // b, da // db)
            same(a // sc, da // sc)
            same(sp.elemwise(lambda p, q: p // q + p, a, b), da // db + da)
        if name == "remainder":
            same(a % b, da % db)
            same(a % sc, da % sc)
            same(np.mod(a, -sc), np.mod(da, -sc))
        if name == "left_shift":
            same(a << 2, da << 2)
        if name == "right_shift":
            same(a >> 2, da >> 2)
        g = sp.GCXS(a), sp.GCXS(b)
        same(f(g[0], g[1]), f(da, db))
    assert counts() == before


def test_invert_imag_and_float_power_stay_on_the_device():
    import sparse_amd as sp

    rng = np.random.default_rng(9)
    d = (rng.integers(-50, 50, size=(80, 70)) * (rng.random((80, 70)) < 0.3)).astype(np.int32)
    counts = lambda: {k: v for k, v in sp.fallback_stats().items() if k != "recent"}
    before = counts()
    for arr in (d, d.astype(np.int64), d != 0):
        x = sp.COO.from_numpy(arr)
        got = ~x
        want = ~arr
        assert got.dtype == want.dtype and np.array_equal(got.todense(), want) and got.fill_value == want.dtype.type(~arr.dtype.type(0))
        assert np.array_equal(np.invert(sp.GCXS(x)).todense(), want)
    f = sp.COO.from_numpy(d.astype(np.float64) / 7)
    im = f.imag
    assert im.nnz == 0 and im.dtype == np.float64 and im.shape == f.shape and not im.todense().any()
    assert np.array_equal(f.real.todense(), d.astype(np.float64) / 7)
    fp = np.float_power(sp.COO.from_numpy(np.abs(d).astype(np.float32)), 1.5)
    want = np.float_power(np.abs(d).astype(np.float32), 1.5)
    assert fp.dtype == want.dtype
    np.testing.assert_allclose(fp.todense(), want, rtol=1e-14)
    assert counts() == before


@pytest.mark.parametrize("axis", [1, -1, 2])
@pytest.mark.parametrize("fmt", ["coo", "gcxs"])
def test_joining_along_an_inner_axis_merges_sorted_keys(axis, fmt):
    """`concatenate` along an axis that is not the first: the operands' keys in the result's shape are merged, not sorted
    (`_batched.CONCAT_MERGE`); NumPy's result on the dense twins, the same stored arrays as the sorting route; three operands of
    mixed value types, one without stored elements, one of extent 0 along the axis; `stack` along the last axis."""
    import sparse_amd as sp
    from sparse_amd import _batched as B

    rng = np.random.default_rng(21)
    base = (6, 9, 11)
    def make(ext, dtype, dens=0.3):
        sh = list(base)
        sh[axis] = ext
        return (rng.random(sh) * (rng.random(sh) < dens)).astype(dtype)
    ds = [make(5, np.float64), make(0, np.float64), make(7, np.float32), make(3, np.float64, 0.0), make(4, np.float64)]
    xs = [sp.COO.from_numpy(d) if fmt == "coo" else sp.asarray(d, format="gcxs") for d in ds]
    got = sp.concatenate(xs, axis=axis)
    want = np.concatenate(ds, axis=axis)
    assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got.todense(), want)
    B.CONCAT_MERGE = False
    try:
        ref = sp.concatenate(xs, axis=axis)
    finally:
        B.CONCAT_MERGE = True
    gc, rc = (got.tocoo(), ref.tocoo()) if fmt == "gcxs" else (got, ref)
    assert np.array_equal(gc.coords.cpu().numpy(), rc.coords.cpu().numpy()) and np.array_equal(gc.data.cpu().numpy(), rc.data.cpu().numpy())
    assert gc.fill_value == rc.fill_value and np.asarray(gc.fill_value).dtype == np.asarray(rc.fill_value).dtype
    b = [sp.COO.from_numpy(d != 0) for d in (ds[0], ds[4][:, :, :] if axis != 0 else ds[4])]
    bb = sp.concatenate(b, axis=axis)
    assert bb.dtype == bool and np.array_equal(bb.todense(), np.concatenate([ds[0] != 0, ds[4] != 0], axis=axis))
    st = sp.stack([xs[0], xs[0] * 2], axis=-1)
    assert np.array_equal(st.todense(), np.stack([ds[0], ds[0] * 2], axis=-1))


def test_einsum_with_a_sparse_operand_contracted_away():
    """"ij,ij->" and "ijk,jk->i" between sparse operands: the operand that keeps no axis is not turned into a one-row matrix over
    the product of the contracted extents (10^9 row pointers at 10^5 x 10^4); NumPy's einsum on the dense twins."""
    import sparse_amd as sp

    rng = np.random.default_rng(13)
    a = rng.random((30, 40, 50)) * (rng.random((30, 40, 50)) < 0.1)
    b = rng.random((40, 50)) * (rng.random((40, 50)) < 0.3)
    c = rng.random((30, 40, 50)) * (rng.random((30, 40, 50)) < 0.1)
    A, B, C = sp.COO.from_numpy(a), sp.COO.from_numpy(b), sp.COO.from_numpy(c)
    for sub, ops, dense in (("ijk,ijk->", (A, C), (a, c)), ("ijk,jk->i", (A, B), (a, b)), ("jk,ijk->i", (B, A), (b, a)),
                            ("ijk,jk->", (A, B), (a, b)), ("ijk,kj->i", (A, sp.COO.from_numpy(b.T.copy())), (a, b.T))):
        got = sp.einsum(sub, *ops)
        want = np.einsum(sub, *dense)
        np.testing.assert_allclose(np.asarray(got.todense()), want, rtol=1e-12, atol=1e-14)
    big = sp.random((100_000, 10_000), density=1e-4, random_state=3)
    other = sp.random((100_000, 10_000), density=1e-4, random_state=4)
    tot = sp.einsum("ij,ij->", big, other)
    assert abs(float(tot.todense()) - float((big * other).sum())) < 1e-9


def test_tensordot_with_a_sparse_operand_contracted_away_over_a_wide_extent():
    """`tensordot(x, y, axes=2)` and friends between sparse operands when one keeps no axis and the contracted extents multiply
    to 2^22 or more: the aligned multiply + sum (einsum's general route), not a one-row matrix with that many row pointers."""
    import sparse_amd as sp

    x = sp.random((3000, 2000), density=2e-3, random_state=1)
    y = sp.random((3000, 2000), density=2e-3, random_state=2)
    want = float((x.todense() * y.todense()).sum())
    for got in (sp.tensordot(x, y, axes=2), sp.tensordot(x, y, axes=([0, 1], [0, 1])), sp.tensordot(x, y.T, axes=([0, 1], [1, 0])),
                sp.tensordot(x.asformat("gcxs"), y, axes=2)):
        assert got.shape == () and abs(float(got.todense()) - want) < 1e-9
    z = sp.random((4, 3000, 2000), density=1e-3, random_state=3)
    zw = np.tensordot(z.todense(), y.todense(), axes=([1, 2], [0, 1]))
    r = sp.tensordot(z, y, axes=([1, 2], [0, 1]))
    assert r.shape == (4,) and np.allclose(r.todense(), zw, rtol=1e-12)
    r2 = sp.tensordot(y, z, axes=([0, 1], [1, 2]))
    assert r2.shape == (4,) and np.allclose(r2.todense(), zw, rtol=1e-12)
    g = sp.tensordot(z.asformat("gcxs"), y.asformat("gcxs"), axes=([1, 2], [0, 1]))
    assert isinstance(g, sp.GCXS) and np.allclose(g.todense(), zw, rtol=1e-12)
