"""A9 parity: the SDDMM kernel vs a NumPy float64 evaluation of the reference's formulation
`s * (a @ b)` (examples/sddmm_example.py:51-52) at the mask's coordinates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1, 7, 64, 192, 200, 256, 384, 1000])
@pytest.mark.parametrize("dt", ["bf16", "f32", "f64"])
def test_sddmm_vs_dense_formulation(dt, K):
    import sparse_amd as sp

    rng = np.random.default_rng(K)
    M, N, nnz = 300, 250, 4001
    lin = np.sort(rng.choice(M * N, nnz, replace=False))
    coords = np.stack([lin // N, lin % N])
    sval = rng.random(nnz) - 0.5
    a, b = rng.random((M, K)) - 0.5, rng.random((K, N)) - 0.5
    tdt = {"bf16": torch.bfloat16, "f32": torch.float32, "f64": torch.float64}[dt]
    at, bt = torch.from_numpy(a).cuda().to(tdt), torch.from_numpy(b.T.copy()).cuda().to(tdt)
    s = sp.COO(coords, sval.astype(np.float64 if dt == "f64" else np.float32), shape=(M, N))
    r = sp.sddmm(s, at, bt=bt)
    a64, b64 = at.double().cpu().numpy(), bt.double().cpu().numpy()  # the values the kernel saw
    want = s.data.double().cpu().numpy() * np.einsum("ik,ik->i", a64[coords[0]], b64[coords[1]])
    got = r.todense()[coords[0], coords[1]]
    absum = np.abs(s.data.double().cpu().numpy()) * np.einsum("ik,ik->i", np.abs(a64[coords[0]]), np.abs(b64[coords[1]]))
    tol = 1e-14 if dt == "f64" else 2e-6  # fp32 accumulation, relative to sum |terms|
    assert np.all(np.abs(got - want) <= tol * absum + 1e-300)
    assert r.nnz == np.count_nonzero(got)


def test_sddmm_empty_and_errors():
    import sparse_amd as sp

    s = sp.COO(np.zeros((2, 0), dtype=np.int64), np.zeros(0, np.float32), shape=(5, 6))
    r = sp.sddmm(s, torch.zeros((5, 8), device="cuda"), bt=torch.zeros((6, 8), device="cuda"))
    assert r.nnz == 0 and r.shape == (5, 6)
    with pytest.raises(ValueError, match="shape-mismatch"):
        sp.sddmm(s, torch.zeros((5, 8), device="cuda"), bt=torch.zeros((6, 9), device="cuda"))


# ---- dense-tile (matrix-core) form: csrc/sddmm_mfma.hip -----------------------------------------------------------------
def _clustered_mask(rng, M, N, n_blocks, block_density, sprinkle):
    """32 x 32 blocks filled at `block_density` plus a uniform sprinkle: (sorted unique linear indices)"""
    tr, tc = -(-M // 32), -(-N // 32)
    blocks = rng.choice(tr * tc, n_blocks, replace=False)
    lin = []
    for b in blocks:
        r0, c0 = (b // tc) * 32, (b % tc) * 32
        rr, cc = np.meshgrid(np.arange(r0, min(r0 + 32, M)), np.arange(c0, min(c0 + 32, N)), indexing="ij")
        keep = rng.random(rr.shape) < block_density
        lin.append((rr[keep] * N + cc[keep]).ravel())
    lin.append(rng.choice(M * N, sprinkle, replace=False))
    return np.unique(np.concatenate(lin))


@pytest.mark.parametrize("shape_k", [((2048, 2048), 256), ((1000, 777), 48), ((70, 3000), 16), ((513, 515), 400)])
@pytest.mark.parametrize("idx", ["int32", "int64"])
def test_sddmm_dense_tiles_on_the_matrix_cores(shape_k, idx, monkeypatch):
    """Populated tiles run as bf16 MFMA tile products, the rest through the sampled kernel; both against the fp64
    evaluation of the reference formulation `s * (a @ b)` within 4e-6 * sum |terms| (fp32 accumulation of exact bf16
    products; the matrix core adds 16 products at a time), and against each other within the same bound."""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    (M, N), Kd = shape_k
    rng = np.random.default_rng(M + Kd)
    lin = _clustered_mask(rng, M, N, n_blocks=40, block_density=0.6, sprinkle=3000)
    coords = np.stack([lin // N, lin % N]).astype(idx)
    nnz = lin.size
    sval = (rng.random(nnz) - 0.5).astype(np.float32)
    at = torch.from_numpy(rng.random((M, Kd)) - 0.5).cuda().to(torch.bfloat16)
    bt = torch.from_numpy(rng.random((N, Kd)) - 0.5).cuda().to(torch.bfloat16)
    s = sp.COO(coords, sval, shape=(M, N))
    plan = K.sddmm_plan(s.coords, s.shape)
    # the plan partitions the samples: every sample is in exactly one dense tile run or in `rest`
    seg = plan.seg_start.cpu().numpy()
    perm = plan.perm.cpu().numpy()
    covered = np.concatenate([perm[seg[t]:seg[t + 1]] for t in plan.tiles.cpu().numpy()] + [plan.rest.cpu().numpy()])
    assert np.array_equal(np.sort(covered), np.arange(nnz))
    assert plan.n_dense_samples > 0.5 * nnz and plan.rest.numel() > 0      # both paths are exercised
    for t in plan.tiles.cpu().numpy():
        assert seg[t + 1] - seg[t] >= plan.threshold
    got = K.sddmm_coo_mfma(plan, s.coords, s.shape, s.data, at, bt)
    assert got is not None, "the hybrid path declined a clustered mask"
    sampled = K.sddmm_coo(s.coords, s.data, at, bt)
    a64, b64 = at.double().cpu().numpy(), bt.double().cpu().numpy()
    want = sval.astype(np.float64) * np.einsum("ik,ik->i", a64[coords[0]], b64[coords[1]])
    absum = np.abs(sval.astype(np.float64)) * np.einsum("ik,ik->i", np.abs(a64[coords[0]]), np.abs(b64[coords[1]]))
    g = got.double().cpu().numpy()
    assert np.all(np.abs(g - want) <= 4e-6 * absum + 1e-300)
    assert np.all(np.abs(g - sampled.double().cpu().numpy()) <= 4e-6 * absum + 1e-300)
    # the samples left to the sampled kernel are bit-identical to the all-sampled result
    rest = plan.rest.cpu().numpy()
    assert np.array_equal(got.cpu().numpy()[rest], sampled.cpu().numpy()[rest])
    # product entry point (told that the tiles pay: at these sizes its traffic model says they do not): same values,
    # plan cached on the mask
    monkeypatch.setattr(K, "sddmm_tiles_pay", lambda plan, a, bt, width: True)
    r = sp.sddmm(s, at, bt=bt)
    assert getattr(s, "_sddmm_plan", None) is not None
    assert np.array_equal(r.todense()[coords[0], coords[1]], np.where(g == 0, 0, got.cpu().numpy()))


def test_sddmm_uniform_mask_stays_on_the_sampled_kernel():
    """BASELINE config 4's regime (about one sample per tile): no tile reaches the threshold, the dispatcher hands
    everything to the sampled kernel and the result is bit-identical to it."""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(5)
    M = N = 4096
    lin = np.sort(rng.choice(M * N, 16000, replace=False))
    coords = np.stack([lin // N, lin % N])
    s = sp.COO(coords, rng.random(lin.size).astype(np.float32), shape=(M, N))
    at = torch.rand((M, 64), device="cuda").to(torch.bfloat16)
    bt = torch.rand((N, 64), device="cuda").to(torch.bfloat16)
    plan = K.sddmm_plan(s.coords, s.shape)
    assert plan.tiles.numel() == 0 and plan.rest.numel() == lin.size
    assert K.sddmm_coo_mfma(plan, s.coords, s.shape, s.data, at, bt) is None
    r = sp.sddmm(s, at, bt=bt)
    assert torch.equal(r.data, K.sddmm_coo(s.coords, s.data, at, bt))


def _mask(rng, M, N, nnz, idx=np.int32):
    lin = np.sort(rng.choice(M * N, nnz, replace=False))
    return np.stack([lin // N, lin % N]).astype(idx), lin


@pytest.mark.parametrize("dt,Kd", [("bf16", 128), ("bf16", 256), ("bf16", 512), ("bf16", 2048), ("f32", 64), ("f32", 256), ("f32", 512),
                                   ("f64", 64), ("f64", 256), ("f64", 512),   # 16-, 32- and 64-lane groups, 1 / 2 / 4 vectors per lane
                                   ("bf16", 384), ("f32", 192), ("f64", 96), ("bf16", 768), ("f32", 768), ("f64", 192)])   # 3 vectors per lane (round 6)
@pytest.mark.parametrize("idx", [np.int32, np.int64])
def test_sddmm_column_panel_order_is_bit_identical(dt, Kd, idx):
    """spamd_sddmm_panels walks the mask one panel of Bt rows at a time; every stored element is computed by the same
    lanes in the same order as in the mask's own order, so the two results are equal bit for bit - for ragged sizes
    (nnz not a multiple of the step), several panel widths and chunk lengths, and for a subset of the elements."""
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(Kd)
    M, N, nnz = 700, 5000, 60_013
    coords_h, _ = _mask(rng, M, N, nnz, idx)
    tdt = {"bf16": torch.bfloat16, "f32": torch.float32, "f64": torch.float64}[dt]
    at = (torch.rand((M, Kd), device="cuda", dtype=torch.float64) - 0.5).to(tdt)
    bt = (torch.rand((N, Kd), device="cuda", dtype=torch.float64) - 0.5).to(tdt)
    coords = torch.from_numpy(coords_h).cuda()
    sval = (torch.rand(nnz, device="cuda", dtype=torch.float64) - 0.5).to(torch.float64 if dt == "f64" else torch.float32)
    ref = K.sddmm_coo(coords, sval, at, bt)
    # (the row-cached kernel itself against the float64 evaluation, for every group width)
    a64, b64 = at.double().cpu().numpy(), bt.double().cpu().numpy()
    ch = coords_h.astype(np.int64)
    want = sval.double().cpu().numpy() * np.einsum("ik,ik->i", a64[ch[0]], b64[ch[1]])
    absum = np.abs(sval.double().cpu().numpy()) * np.einsum("ik,ik->i", np.abs(a64[ch[0]]), np.abs(b64[ch[1]]))
    assert np.all(np.abs(ref.double().cpu().numpy() - want) <= (1e-14 if dt == "f64" else 2e-6) * absum + 1e-300)
    for width, xcd in ((64, False), (64, True), (300, True), (1000, False), (4999, False), (5000, True)):
        plan = K.sddmm_panels(coords, (M, N), width, xcd=xcd)
        # the order really is panel-major (XCD-major: panel p of XCD p % 8 is that XCD's (p / 8)-th), row-major inside a
        # panel, and `pos` is a permutation
        pc = plan.cols.cpu().numpy().astype(np.int64) // width
        npanels = (N - 1) // width + 1
        if xcd and npanels >= 8:
            per = -(-npanels // 8)
            assert plan.xstate is not None
            first = plan.xstate.cpu().numpy()
            assert first[0] == 0 and first[8] == nnz and plan.xmax == int(np.max(np.diff(first)))
            pc = (pc % 8) * per + pc // 8
            for x in range(8):
                assert np.all(pc[first[x]:first[x + 1]] // per == x)
        else:
            assert plan.xstate is None
        assert np.all(np.diff(pc) >= 0)
        pos = plan.pos.cpu().numpy()
        assert np.array_equal(np.sort(pos), np.arange(nnz))
        inside = np.diff(pc) == 0
        assert np.all(np.diff(pos)[inside] > 0)
        for chunk in (0, 16, 48, 1000):
            plan.chunk = chunk
            assert torch.equal(K.sddmm_coo(coords, sval, at, bt, panels=plan), ref), (width, chunk)
    # a subset of the elements, written into a caller-provided result
    subset = torch.from_numpy(np.sort(rng.choice(nnz, 20_001, replace=False))).cuda()
    plan = K.sddmm_panels(coords, (M, N), 300, subset=subset)
    out = torch.full((nnz,), -7.0, dtype=ref.dtype, device="cuda")
    K._sddmm_panels_into(plan, sval, sval, at, bt, out)
    keep = torch.zeros(nnz, dtype=torch.bool, device="cuda")
    keep[subset] = True
    assert torch.equal(out[keep], ref[keep]) and bool((out[~keep] == -7.0).all())


def test_sddmm_product_path_uses_panels_and_follows_the_mask(monkeypatch):
    """`sparse_amd.sddmm` builds the panel plan once per mask (COO, or the kept COO view of a GCXS mask), re-gathers the
    mask values when they change in place, drops the plan when the coordinates' container is replaced, and its result
    matches the float64 evaluation of the reference's formulation."""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    monkeypatch.setattr(K, "sddmm_panels_pay", lambda n, a, bt, width: bool(width))   # (the traffic model would keep so small a mask in its own order)
    monkeypatch.setattr(K, "SDDMM_PANEL_BYTES", 64 * 256 * 2)   # 64 Bt rows per panel
    rng = np.random.default_rng(11)
    M, N, Kd, nnz = 500, 3000, 256, 40_000
    coords_h, _ = _mask(rng, M, N, nnz)
    sval = (rng.random(nnz) - 0.5).astype(np.float32)
    at = (torch.rand((M, Kd), device="cuda") - 0.5).to(torch.bfloat16)
    bt = (torch.rand((N, Kd), device="cuda") - 0.5).to(torch.bfloat16)
    a64, b64 = at.double().cpu().numpy(), bt.double().cpu().numpy()
    dots = np.einsum("ik,ik->i", a64[coords_h[0]], b64[coords_h[1]])
    absd = np.einsum("ik,ik->i", np.abs(a64[coords_h[0]]), np.abs(b64[coords_h[1]]))

    def check(r, values):
        got = r.todense()[coords_h[0], coords_h[1]]
        assert np.all(np.abs(got - values.astype(np.float64) * dots) <= 2e-6 * np.abs(values) * absd + 1e-300)

    s = sp.COO(coords_h, sval, shape=(M, N))
    check(sp.sddmm(s, at, bt=bt), sval)
    plans = s._sddmm_plan
    key = ("panels", "all", K.sddmm_panel_width(bt))
    assert key in plans and plans[key].count == nnz
    first = plans[key]
    check(sp.sddmm(s, at, bt=bt), sval)
    assert s._sddmm_plan[key] is first            # built once
    s.data.mul_(2.0)                               # in-place change of the mask values: new result, no stale values
    check(sp.sddmm(s, at, bt=bt), 2 * sval)
    g = sp.GCXS.from_numpy(s.todense(), compressed_axes=(0,)) if False else s.asformat("gcxs", compressed_axes=(0,))
    r = sp.sddmm(g, at, bt=bt)
    assert isinstance(r, sp.GCXS)
    check(r, 2 * sval)
    view = g._coo_view
    sp.sddmm(g, at, bt=bt)
    assert g._coo_view is view and key in view._sddmm_plan


def test_sddmm_mfma_rest_in_panel_order(monkeypatch):
    """Per-tile dispatch with the left-over samples taken in column-panel order: equal, element for element, to the
    dispatch with the left-over samples in the mask's own order."""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(3)
    M = N = 2048
    Kd = 128
    tiles = rng.choice((M // 32) * (N // 32), 300, replace=False)
    pos = np.argsort(rng.random((300, 1024)), axis=1)[:, :600]
    r = (tiles // (N // 32))[:, None] * 32 + pos // 32
    c = (tiles % (N // 32))[:, None] * 32 + pos % 32
    lin = np.unique(np.concatenate([(r.astype(np.int64) * N + c).ravel(), rng.choice(M * N, 50_000, replace=False)]))
    coords_h = np.stack([lin // N, lin % N]).astype(np.int32)
    s = sp.COO(coords_h, (rng.random(lin.size) - 0.5).astype(np.float32), shape=(M, N))
    at = (torch.rand((M, Kd), device="cuda") - 0.5).to(torch.bfloat16)
    bt = (torch.rand((N, Kd), device="cuda") - 0.5).to(torch.bfloat16)
    plan = K.sddmm_plan(s.coords, s.shape)
    assert plan.tiles.numel() > 0 and plan.rest.numel() > 1000
    plain = K.sddmm_coo_mfma(plan, s.coords, s.shape, s.data, at, bt, force=True)
    restp = K.sddmm_panels(s.coords, s.shape, 100, subset=plan.rest)
    paneled = K.sddmm_coo_mfma(plan, s.coords, s.shape, s.data, at, bt, force=True, rest_panels=restp)
    assert torch.equal(plain, paneled)
    # and through the product entry point with the thresholds lowered so that both plans are used
    monkeypatch.setattr(K, "sddmm_panels_pay", lambda n, a, bt, width: bool(width))
    monkeypatch.setattr(K, "sddmm_tiles_pay", lambda plan, a, bt, width: True)
    monkeypatch.setattr(K, "SDDMM_PANEL_BYTES", 100 * Kd * 2)
    out = sp.sddmm(s, at, bt=bt)
    assert ("panels", "rest", K.sddmm_panel_width(bt)) in s._sddmm_plan
    want = np.where(plain.cpu().numpy() == 0, 0, plain.cpu().numpy())
    assert np.array_equal(out.todense()[coords_h[0], coords_h[1]], want)


def test_sddmm_panel_order_traffic_model():
    """The choice between the two element orders follows the traffic model: config 4's 10^7 samples take the panel
    order, 3 * 10^6 samples of the same shapes (what the dense-tile dispatch leaves over on a clustered mask) and any
    operand that fits the L2 keep the mask's own order."""
    from sparse_amd import _kernels as K

    a = torch.empty((100_000, 256), dtype=torch.bfloat16, device="cuda")
    bt = torch.empty((100_000, 256), dtype=torch.bfloat16, device="cuda")
    w = K.sddmm_panel_width(bt)
    assert w == 6250          # 16 panels (two per XCD) of 3.05 MiB; (3 << 20) // 512 = 6144 with shared panels
    assert K.sddmm_panels_pay(10_000_000, a, bt, w)
    assert not K.sddmm_panels_pay(3_000_000, a, bt, w)
    small = torch.empty((4096, 256), dtype=torch.bfloat16, device="cuda")
    assert K.sddmm_panel_width(small) == 0 and not K.sddmm_panels_pay(10_000_000, a, small, 0)
    odd = torch.empty((100_000, 200), dtype=torch.bfloat16, device="cuda")   # no row-cached kernel for K = 200
    assert K.sddmm_panel_width(odd) == 0


def test_sddmm_tile_dispatch_traffic_model():
    """Dense tiles + sampled rest vs everything sampled, by the traffic model: a block-sparse mask (full tiles) takes
    the matrix cores, a uniform mask does not."""
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(0)
    M = N = 16384
    a = torch.empty((M, 256), dtype=torch.bfloat16, device="cuda")
    bt = torch.empty((N, 256), dtype=torch.bfloat16, device="cuda")
    tiles = rng.choice((M // 32) * (N // 32), 2000, replace=False)
    full = np.arange(1024)
    lin = np.sort((((tiles // (N // 32))[:, None] * 32 + full // 32).astype(np.int64) * N + (tiles % (N // 32))[:, None] * 32 + full % 32).ravel())
    coords = torch.from_numpy(np.stack([lin // N, lin % N]).astype(np.int32)).cuda()
    plan = K.sddmm_plan(coords, (M, N))
    assert plan.tiles.numel() == 2000 and plan.rest.numel() == 0
    assert K.sddmm_tiles_pay(plan, a, bt, K.sddmm_panel_width(bt))
    lin = np.sort(rng.choice(M * N, 2_000_000, replace=False))
    coords = torch.from_numpy(np.stack([lin // N, lin % N]).astype(np.int32)).cuda()
    plan = K.sddmm_plan(coords, (M, N))
    assert plan.tiles.numel() == 0 and not K.sddmm_tiles_pay(plan, a, bt, K.sddmm_panel_width(bt))


@pytest.mark.parametrize("dt, Kd", [("f32", 100), ("bf16", 200), ("f64", 48), ("f32", 33)])
def test_sddmm_inner_dimensions_without_a_row_cached_kernel_are_padded(dt, Kd):
    """From 200 000 samples on, an inner dimension whose rows have no row-cached kernel (384-byte rows: K = 96 or 100 in float32)
    is zero-padded to the next one that has (round 6: the generic gather, without a panel order, took 2x as long); the sums
    against the float64 evaluation, the padding switched off for comparison"""
    import sparse_amd as sp
    from sparse_amd import _kernels as K

    rng = np.random.default_rng(Kd)
    M, N, nnz = 3000, 5000, 250_000
    tdt = {"bf16": torch.bfloat16, "f32": torch.float32, "f64": torch.float64}[dt]
    s = sp.random((M, N), nnz=nnz, random_state=5, dtype=np.float64 if dt == "f64" else np.float32)
    at = (torch.rand((M, Kd), device="cuda", dtype=torch.float64) - 0.5).to(tdt)
    bt = (torch.rand((N, Kd), device="cuda", dtype=torch.float64) - 0.5).to(tdt)
    pa, pb = K.sddmm_pad_inner(at, bt, nnz)
    assert (pa.shape[1] > Kd) == (Kd != 33)        # (132-byte rows would become 256-byte ones: more than 1.5x, left alone)
    r = sp.sddmm(s, at, bt=bt)
    c = s.coords.cpu().numpy()
    a64, b64 = at.double().cpu().numpy(), bt.double().cpu().numpy()
    sv = s.data.double().cpu().numpy()
    want = sv * np.einsum("ik,ik->i", a64[c[0]], b64[c[1]])
    absum = np.abs(sv) * np.einsum("ik,ik->i", np.abs(a64[c[0]]), np.abs(b64[c[1]]))
    got = r.todense()[c[0], c[1]]
    assert np.all(np.abs(got - want) <= (1e-14 if dt == "f64" else 2e-6) * absum + 1e-300)
