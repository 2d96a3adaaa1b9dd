"""`einsum` against `numpy.einsum` on the dense arrays (the reference tests the same way,
sparse/numba_backend/tests/test_einsum.py), plus the argument contracts and result formats it pins."""
import numpy as np
import pytest
import torch

PATTERNS = [
    # single operand: transposes, traces, diagonals, partial sums
    "ab->ba", "abc->cab", "aa", "aa->a", "abab->ba", "aab->b", "abc->", "abc->b", "...a->...", "a...b->b...a",
    # two operands through the tensordot route
    "ab,bc", "ab,cb", "ba,bc", "ab,bc->ca", "abcd,cd", "abcd,cdef->feba", "abc,cba", "ab,ab", "abc,bd->acd",
    # two operands that need the aligned-multiply route (batch labels, repeated labels, dropped free labels)
    "ab,ab->ab", "ab,ab->a", "ab,bc->", "ab,bc->b", "aab,bc->ac", "ab,bcc->ac", "aab,ccb->ac", "ab,b->ab",
    "...ab,...ab", "...ab,...b->...a", "a...,a...", "abcd,ad",
    # scalars and many operands
    "a,->a", ",ab,->ab", ",,->", "a,b,ab->ab", "a,ab,abc->abc", "ab,cd,ef->acdf", "ab,cd,de->be", "ab,bcd,cd->abd",
    "eb,cb,fb->cef", "dd,fb,be,cdb->cef", "bca,cdb,dbf,afc->", "ab,ab,cd,cd->ac", "ea,fb,gc,hd,abcd->efgh",
    "bb,ff,be->e", "afd,ba,cc,dc->bf",
]


def _operands(subscripts, density, seed):
    import sparse_amd as sp

    rng = np.random.default_rng(seed)
    terms = subscripts.split("->")[0].split(",")
    dense = []
    for t in terms:
        nd = len(t.replace("...", "xy"))     # an ellipsis stands for two axes here ("a...,a..." -> 3-D operands)
        d = np.zeros((4,) * nd)
        m = rng.random(d.shape) < density
        d[m] = rng.random(int(m.sum())) - 0.3
        dense.append(d)
    # a 0-d operand stores its value as one element (as `sparse.random(())` does): `from_numpy` of a 0-d array would
    # make the value the FILL value, which einsum rejects
    return dense, [sp.COO.from_numpy(d) if d.ndim else sp.COO(np.zeros((0, 1), dtype=np.int64), d.reshape(1), shape=())
                   for d in dense]


@pytest.mark.gpu
@pytest.mark.parametrize("density", [0.15, 1.0])
@pytest.mark.parametrize("subscripts", PATTERNS)
def test_einsum_matches_numpy(subscripts, density):
    import sparse_amd as sp

    dense, sparse = _operands(subscripts, density, seed=len(subscripts) * 7 + int(density * 10))
    want = np.einsum(subscripts, *dense)
    got = sp.einsum(subscripts, *sparse)
    assert isinstance(got, sp.SparseArray) and got.shape == want.shape
    assert np.allclose(got.todense(), want, rtol=1e-12, atol=1e-14), subscripts
    assert np.array_equal(np.einsum(subscripts, *sparse).todense(), got.todense())   # __array_function__ route


@pytest.mark.gpu
@pytest.mark.parametrize("args", [[[0, 0]], [[0, Ellipsis]], [[Ellipsis, 1], [Ellipsis]], [[0, 1], [0]], [[0, 1], [1, 0]]])
def test_einsum_sublist_form(args):
    import sparse_amd as sp

    d = np.random.default_rng(3).random((4, 4)) * (np.random.default_rng(4).random((4, 4)) < 0.6)
    assert np.allclose(sp.einsum(sp.COO.from_numpy(d), *args).todense(), np.einsum(d, *args))


@pytest.mark.gpu
def test_einsum_contracts():
    import sparse_amd as sp

    x, y = sp.random((2,), density=0.5, random_state=1), sp.random((2,), density=0.5, random_state=2)
    with pytest.raises(ValueError):
        sp.einsum()
    for bad in ("a+b->c", "i->&", "i->ij", "ij->jij", "a..,a...", ".i...", "a,a->->"):
        with pytest.raises(ValueError):
            sp.einsum(bad, x, y)
    for bad in (0, [0, 0]):
        with pytest.raises(TypeError):
            sp.einsum(bad, x, y)
    with pytest.raises(ValueError):   # non-zero fill value
        sp.einsum("cba", sp.random((2, 2, 2), density=0.5, random_state=3, fill_value=2))
    z = sp.random((2, 3, 4), density=0.5, random_state=4)
    with pytest.raises(ValueError):   # repeated label over different extents
        sp.einsum("aab", z)
    with pytest.raises(ValueError):   # inconsistent extents across operands
        sp.einsum("abc,acb", z, sp.random((2, 3, 4), density=0.5, random_state=5))


@pytest.mark.gpu
@pytest.mark.parametrize("formats,expected", [(("coo",), "coo"), (("gcxs",), "gcxs"), (("coo", "coo"), "coo"),
                                              (("coo", "dense"), "coo"), (("dense", "coo"), "coo"),
                                              (("gcxs", "dense"), "gcxs"), (("dense", "gcxs"), "gcxs"),
                                              (("gcxs", "gcxs"), "gcxs"), (("coo", "gcxs"), "coo"),
                                              (("dense", "coo", "gcxs"), "coo")])
def test_einsum_result_format_and_dtype(formats, expected):
    import sparse_amd as sp

    rng = np.random.default_rng(8)
    dense = [rng.standard_normal((2, 2, 2)) * (rng.random((2, 2, 2)) < 0.6) for _ in formats]
    ops = [d if f == "dense" else sp.COO.from_numpy(d).asformat(f) for d, f in zip(dense, formats)]
    eq = {1: "abc->bc", 2: "abc,cda->abd", 3: "abc,cad,dea->abe"}[len(ops)]
    out = sp.einsum(eq, *ops)
    assert out.format == expected and np.allclose(out.todense(), np.einsum(eq, *dense), rtol=1e-12, atol=1e-14)
    if len(ops) == 2:
        r = sp.einsum("abc,cda->abd", *ops, dtype=np.float32)
        assert r.dtype == np.float32 if all(f != "dense" for f in formats) else True


@pytest.mark.gpu
def test_einsum_spmm_pattern_uses_the_matmul_kernels():
    """'ij,jk->ik' with a sparse A and a dense B is the headline product: same numbers as `A @ B`."""
    import sparse_amd as sp

    a = sp.random((3000, 500), density=0.01, random_state=6, dtype=np.float32, format="gcxs", compressed_axes=(0,))
    b = torch.rand((500, 64), device="cuda", dtype=torch.float32)
    r = sp.einsum("ij,jk->ik", a, b)
    assert r.format == "gcxs" and np.array_equal(r.todense(), (a @ b).cpu().numpy())
    r2 = sp.einsum("ij,jk->ki", a, b)
    assert np.array_equal(r2.todense(), (a @ b).cpu().numpy().T)


@pytest.mark.gpu
def test_einsum_against_reference_fixture():
    """tests/golden/einsum.npz: what the real reference's `einsum` returned for the same operands
    (oracle/gen_golden.py `gen_einsum`) — values, stored-element counts and result formats."""
    import os

    import sparse_amd as sp

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "einsum.npz"))
    for k in range(int(g["n_einsum"])):
        pat = str(g[f"e{k}_pat"])
        ops = [sp.COO.from_numpy(g[f"e{k}_op{j}"]) for j in range(int(g[f"e{k}_n"]))]
        r = sp.einsum(pat, *ops)
        want = g[f"e{k}_dense"]
        assert r.shape == want.shape and np.allclose(r.todense(), want, rtol=1e-13, atol=1e-15), pat
        assert r.nnz <= max(int(g[f"e{k}_nnz"]), 1), pat   # never more stored elements than the reference keeps
    a, b = g["fa"], g["fb"]
    for entry in g["formats"]:
        fa_, fb_, want_fmt = str(entry).split(",")
        oa = a if fa_ == "dense" else sp.COO.from_numpy(a).asformat(fa_)
        ob = b if fb_ == "dense" else sp.COO.from_numpy(b).asformat(fb_)
        r = sp.einsum("abc,cda->abd", oa, ob)
        assert r.format == want_fmt and np.allclose(r.todense(), g[f"f_{fa_}_{fb_}"], rtol=1e-13, atol=1e-15), entry
