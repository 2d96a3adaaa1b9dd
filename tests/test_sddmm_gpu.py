"""A9 parity: the SDDMM kernel vs a NumPy float64 evaluation of the reference's formulation
`s * (a @ b)` (examples/sddmm_example.py:51-52) at the mask's coordinates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1, 7, 64, 200, 256, 1000])
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
def test_sddmm_dense_tiles_on_the_matrix_cores(shape_k, idx):
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
    # product entry point: same values, plan cached on the mask
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
