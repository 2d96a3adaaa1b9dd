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
