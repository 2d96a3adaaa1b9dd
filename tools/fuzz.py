"""Randomised cross-checks of kernels that exist in two forms (run on the GPU box):
   tiled vs row-group SpMM (bit-identical, both arithmetic modes, fp32/fp64), single-pass vs two-pass merge,
   row-local vs global SpGEMM, grouped reduce vs a float64 scatter-add, SDDMM in column-panel order vs the mask's own
   order (and vs float64).   python tools/fuzz.py [seconds]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K, _umath as U
from bench import make_csr_device, make_powerlaw_csr_device

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ONLY_SPMM = len(sys.argv) > 2 and sys.argv[2] == "spmm"   # with PYTORCH_NO_CUDA_MEMORY_CACHING=1: catches reads past a buffer
rng = np.random.default_rng(int(time.time()))
t_end = time.time() + budget
n = {"spmm": 0, "merge": 0, "spgemm": 0, "reduce": 0, "sddmm": 0}
while time.time() < t_end:
    # ---- SpMM
    M = int(rng.choice([1, 31, 513, 5000, 70001, 200000]))
    Kd = int(rng.choice([1, 127, 128, 129, 1000, 7936, 16000, 33000]))
    dens = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3]))
    if M * Kd * dens > 3e7:
        dens = 3e7 / (M * Kd)
    dt = torch.float32 if rng.random() < 0.5 else torch.float64
    panel = 128 if dt == torch.float32 else 64
    N = panel * int(rng.integers(1, 4))
    it = torch.int32 if rng.random() < 0.7 else torch.int64
    if rng.random() < 0.4 and M >= 513 and Kd >= 127 and dens > 0:
        # round 5: Zipf row lengths (the balanced, row-mapped layout; full rows, 30 % empty rows) at any size
        K.TILED_BALANCE_MIN_NNZ = 0
        data, idx, ptr = make_powerlaw_csr_device(M, Kd, max(int(M * Kd * dens), 1), seed=int(rng.integers(1 << 30)), dtype=dt, idx_dtype=it,
                                                  alpha=float(rng.choice([0.7, 1.0, 1.4])))
        n["zipf"] = n.get("zipf", 0) + 1
    else:
        data, idx, ptr = make_csr_device(M, Kd, dens, seed=int(rng.integers(1 << 30)), dtype=dt, idx_dtype=it)
    b = torch.randn((Kd, N), device="cuda", dtype=dt)
    layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
    n["balanced"] = n.get("balanced", 0) + (layout.rowmap is not None)
    for exact in (False, True):
        got = K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, exact=exact)
        ref = K.dot_csr_ndarray((M, N), data, idx, ptr, b, exact=exact)
        assert torch.equal(got, ref), ("spmm", M, Kd, dens, dt, N, exact)
    n["spmm"] += 1
    # ---- the dispatcher's other kernels against the k-ascending row-group kernel: B resident in LDS for a short contracted
    #      axis (bit-identical, both modes, any value type) and the row-vector kernel for results of 1..4 columns (tree order:
    #      within rounding for floats, identical for integers)
    M2 = int(rng.choice([8192, 9001, 40000, 150000]))
    K2 = int(rng.choice([1, 2, 17, 64, 300, 575, 576]))
    dens2 = float(rng.choice([0.0, 0.004, 0.03, 0.2, 0.9]))
    dt2 = [torch.float32, torch.float64, torch.int32, torch.int64][int(rng.integers(4))]
    vec = 16 // torch.empty(0, dtype=dt2).element_size()
    N2 = vec * int(rng.integers(32 // vec, 80))
    d2, i2, p2 = make_csr_device(M2, K2, dens2, seed=int(rng.integers(1 << 30)), dtype=torch.float32,
                                 idx_dtype=torch.int32 if rng.random() < 0.7 else torch.int64)
    d2 = d2.to(dt2) if dt2.is_floating_point else ((d2 - 0.5) * 40).to(dt2)
    b2 = torch.randn((K2, N2), device="cuda", dtype=torch.float64)
    b2 = b2.to(dt2) if dt2.is_floating_point else (b2 * 9).to(dt2)
    for exact in ((False, True) if dt2.is_floating_point else (False,)):
        got = K.dot_csr_ndarray((M2, N2), d2, i2, p2, b2, exact=exact)
        ref = K.dot_csr_ndarray((M2, N2), d2, i2, p2, b2, exact=exact, keep_order=True)
        assert torch.equal(got, ref), ("ldsb", M2, K2, dens2, dt2, N2, exact)
    nv = int(rng.integers(1, 5))
    bv = b2[:, :nv].contiguous()
    got = K.dot_csr_ndarray((M2, nv), d2, i2, p2, bv)
    ref = K.dot_csr_ndarray((M2, nv), d2, i2, p2, bv, keep_order=True)
    if dt2.is_floating_point:
        bound = K.dot_csr_ndarray((M2, nv), d2.abs(), i2, p2, bv.abs(), keep_order=True)
        tol = 1e-6 if dt2 == torch.float32 else 1e-14
        assert bool(((got - ref).abs() <= tol * bound + 1e-300).all()), ("rowvec", M2, K2, dens2, dt2, nv)
    else:
        assert torch.equal(got, ref), ("rowvec", M2, K2, dens2, dt2, nv)
    n["narrow"] = n.get("narrow", 0) + 1
    del d2, i2, p2, b2, bv
    if ONLY_SPMM:
        del data, idx, ptr, b, layout, got, ref
        continue
    # ---- merge (elementwise add / multiply / maximum) single-pass vs two-pass
    shape = (int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 300)))
    size = shape[0] * shape[1] * shape[2]
    nnz = int(min(size, rng.integers(0, 3_000_000)))
    x = sp.random(shape, nnz=nnz, random_state=int(rng.integers(1 << 30)))
    y = sp.random(shape, nnz=int(min(size, rng.integers(0, 3_000_000))), random_state=int(rng.integers(1 << 30)))
    for f in (lambda a, c: a + c, lambda a, c: a * c, lambda a, c: np.maximum(a, c)):
        U.MERGE_SINGLE_PASS = True
        r1 = f(x, y)
        U.MERGE_SINGLE_PASS = False
        r2 = f(x, y)
        U.MERGE_SINGLE_PASS = True
        assert torch.equal(r1.linear_loc(), r2.linear_loc()) and torch.equal(r1.data, r2.data), ("merge", shape)
    n["merge"] += 1
    # ---- grouped reduce vs scatter-add in float64
    ax = int(rng.integers(0, 3))
    s = x.sum(axis=ax).todense()
    want = torch.zeros(shape, dtype=torch.float64, device="cuda")
    if x.nnz:
        want[tuple(x.coords.long())] = x.data.double()
    assert np.allclose(s, want.sum(dim=ax).cpu().numpy(), rtol=1e-11, atol=1e-13), ("reduce", shape, ax)
    n["reduce"] += 1
    # ---- reductions over leading axes: the slab merge (csrc/lead_rotate.hip) against the key sort, bit for bit
    lead = (0,) if rng.integers(0, 2) else (0, 1)
    K.LEAD_LAST = False
    w1, w2 = x.max(axis=lead), x.sum(axis=lead)
    K.LEAD_LAST = True
    g1, g2 = x.max(axis=lead), x.sum(axis=lead)
    for g, w in ((g1, w1), (g2, w2)):
        assert torch.equal(g.linear_loc(), w.linear_loc()) and torch.equal(g.data.view(torch.int64) if g.data.dtype == torch.float64 else g.data, w.data.view(torch.int64) if w.data.dtype == torch.float64 else w.data), ("lead", shape, lead)
    n["lead"] = n.get("lead", 0) + 1
    # ---- SpGEMM
    ng = int(rng.choice([50, 2000, 20000]))
    dg = float(rng.choice([0.0005, 0.005, 0.02]))
    g1 = sp.random((ng, ng), density=dg, random_state=int(rng.integers(1 << 30)), format="gcxs", compressed_axes=(0,))
    g2 = sp.random((ng, ng), density=dg, random_state=int(rng.integers(1 << 30)), format="gcxs", compressed_axes=(0,))
    K.SPGEMM_ROW_LOCAL = True
    c1 = g1 @ g2
    K.SPGEMM_ROW_LOCAL = False
    c2 = g1 @ g2
    K.SPGEMM_ROW_LOCAL = True
    assert torch.equal(c1.indptr.long(), c2.indptr.long()) and torch.equal(c1.indices.long(), c2.indices.long()) \
        and torch.equal(c1.data, c2.data), ("spgemm", ng, dg)
    n["spgemm"] += 1
    # ---- SDDMM: column-panel order (also of a subset) vs the mask's own order, bit for bit; both vs float64
    Ms, Ns = int(rng.integers(1, 3000)), int(rng.integers(1, 6000))
    Ks = int(rng.choice([16, 64, 96, 128, 192, 200, 256, 384, 512, 768]))       # (96 / 192 / 384 / 768: three vectors per lane, round 6)
    sdt = [torch.bfloat16, torch.float32, torch.float64][int(rng.integers(0, 3))]
    ns = int(min(Ms * Ns, rng.integers(0, 400_000)))
    m = sp.random((Ms, Ns), nnz=ns, random_state=int(rng.integers(1 << 30)), dtype=np.float64 if sdt == torch.float64 else np.float32,
                  idx_dtype=np.int32 if rng.random() < 0.5 else np.int64)
    at = (torch.rand((Ms, Ks), device="cuda", dtype=torch.float64) - 0.5).to(sdt)
    btt = (torch.rand((Ns, Ks), device="cuda", dtype=torch.float64) - 0.5).to(sdt)
    ref = K.sddmm_coo(m.coords, m.data, at, btt)
    if ns:
        cl = m.coords.long()
        want = m.data.double() * (at.double()[cl[0]] * btt.double()[cl[1]]).sum(dim=1)
        bound = m.data.double().abs() * (at.double()[cl[0]].abs() * btt.double()[cl[1]].abs()).sum(dim=1)
        assert bool(((ref.double() - want).abs() <= (1e-14 if sdt == torch.float64 else 2e-6) * bound + 1e-300).all()), ("sddmm", Ms, Ns, Ks, sdt)
    if ns and K.sddmm_has_panels(sdt, Ks):
        width = int(rng.integers(1, Ns + 1))
        pl = K.sddmm_panels(m.coords, m.shape, width)
        pl.chunk = int(rng.choice([0, 16, 100]))
        assert torch.equal(K.sddmm_coo(m.coords, m.data, at, btt, panels=pl), ref), ("sddmm panels", Ms, Ns, Ks, sdt, width)
        sub = torch.nonzero(torch.rand(ns, device="cuda") < 0.3).reshape(-1)
        if sub.numel():
            pl = K.sddmm_panels(m.coords, m.shape, width, subset=sub)
            out = torch.zeros_like(ref)
            K._sddmm_panels_into(pl, m.data, m.data.to(ref.dtype), at, btt, out)
            assert torch.equal(out[sub], ref[sub]), ("sddmm subset", Ms, Ns, Ks, sdt, width)
    n["sddmm"] += 1
torch.cuda.synchronize()
print("fuzz ok:", n)
