"""Parity of the gpr-indexed LDS-tile SpMM kernel (fp32, N = 128, FMA mode) — debug build (phase A /
windows / tiles with memory accumulators) and asm build — against the CPU oracle."""
import numpy as np
import pytest
import torch

from util import random_csr, random_dense

pytestmark = pytest.mark.gpu


def _check(orc, monkeypatch, mode, M, K, density, idt, seed=0, **kw):
    from sparse_amd import _kernels as Kn
    import scipy.sparse as sps

    monkeypatch.setenv("SPAMD_SPMM_VARIANT", f"GIDX={mode}")
    data, idx, ptr = random_csr(M, K, density, seed, np.float32, idt, **kw)
    b = random_dense(K, 128, seed + 1, np.float32)
    d = torch.device("cuda")
    out = Kn.dot_csr_ndarray((M, 128), *(torch.from_numpy(x).to(d) for x in (data, idx, ptr, b)), exact=False)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = orc.dot_csr_ndarray((M, 128), data, idx, ptr, b)
    absum = sps.csr_matrix((np.abs(data).astype(np.float64), idx, ptr), shape=(M, K)) @ np.abs(b).astype(np.float64)
    assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= 1e-6 * absum + 1e-30)


@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("M,K,density", [(300, 200, 0.05), (1000, 3000, 0.01), (257, 64, 0.5), (64, 1000, 0.2),
                                         (5000, 10000, 0.01), (1, 70, 1.0)])
def test_gidx_matches_oracle(orc, monkeypatch, mode, idt, M, K, density):
    _check(orc, monkeypatch, mode, M, K, density, idt)


@pytest.mark.parametrize("mode", [2, 1])
def test_gidx_edge_rows(orc, monkeypatch, mode):
    _check(orc, monkeypatch, mode, 777, 900, 0.03, np.int32, seed=3, empty_rows=(0, 1, 2, 400, 401, 776), long_row=300)
    _check(orc, monkeypatch, mode, 300, 50, 0.0, np.int32)
