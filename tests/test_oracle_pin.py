"""Pin the CPU oracle (oracle/) against fixtures produced by the REAL reference
(oracle/gen_golden.py ran pydata/sparse's numba_backend source under the no-op numba stub).
Bit-exact: the C restatement keeps the reference's loop order and separate mul/add."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dot():
    return np.load(os.path.join(GOLD, "dot.npz"))


@pytest.fixture(scope="module")
def conv():
    return np.load(os.path.join(GOLD, "convert.npz"))


def test_csr_csc_dense_kernels_bit_exact(orc, dot):
    n = int(dot["n_gcxs_dense"])
    assert n == 50
    for k in range(n):
        p = f"gd{k}_"
        data, idx, ptr, b, want = (dot[p + s] for s in ("data", "indices", "indptr", "b", "out"))
        M, K = (int(v) for v in dot[p + "shape"])
        if tuple(dot[p + "ca"]) == (0,):
            got = orc.dot_csr_ndarray(want.shape, data, idx, ptr, b)
        else:
            got = orc.dot_csc_ndarray((M, K), b.shape, data, idx, ptr, b)
        assert got.dtype == want.dtype, k
        assert np.array_equal(got, want), k


def test_edge_rows_and_nan(orc, dot):
    got = orc.dot_csr_ndarray(dot["edge_out"].shape, dot["edge_data"], dot["edge_indices"], dot["edge_indptr"], dot["edge_b"])
    assert np.array_equal(got, dot["edge_out"], equal_nan=True)
    assert np.isnan(got[20]).all() and not got[3].any() and not got[4].any()


def test_coo_dense_kernel(orc, dot):
    got = orc.dot_coo_ndarray(dot["coo_coords"], dot["coo_data"], dot["coo_b"], dot["coo_out"].shape)
    assert np.array_equal(got, dot["coo_out"])


def test_small_values_kat(orc, dot):
    """reference tests/test_dot.py:289-300"""
    a = dot["small_a"]
    r, c = np.nonzero(a)
    got = orc.dot_coo_ndarray(np.stack([r, c]), a[r, c], dot["small_b"], dot["small_out"].shape)
    assert np.array_equal(got, dot["small_out"])


@pytest.mark.parametrize("tag", ["csr", "csc"])
def test_spgemm_kernel_raw_order(orc, dot, tag):
    """`_dot_csr_csr` incl. its unsorted (reverse discovery) row order; csc@csc = (B^T A^T)^T."""
    ad, ai, ap = (dot[f"gg_{tag}_a_{s}"] for s in ("data", "indices", "indptr"))
    bd, bi, bp = (dot[f"gg_{tag}_b_{s}"] for s in ("data", "indices", "indptr"))
    if tag == "csr":
        data, indices, indptr = orc.dot_csr_csr((40, 45), ad, bd, ai, bi, ap, bp)
    else:
        data, indices, indptr = orc.dot_csr_csr((45, 40), bd, ad, bi, ai, bp, ap)
    keep = data != 0  # the GCXS ctor prunes explicit zeros (prune=True)
    assert np.array_equal(indptr[-1:], [len(data)])
    assert np.array_equal(indices[keep], dot[f"gg_{tag}_raw_indices"])
    assert np.array_equal(data[keep], dot[f"gg_{tag}_raw_data"])


def test_match_arrays(orc):
    rng = np.random.default_rng(0)
    a = np.sort(rng.integers(0, 50, 200))
    b = np.sort(rng.integers(0, 50, 150))
    ai, bi = orc.match_arrays(a, b)
    want = [(i, j) for i in range(len(a)) for j in range(len(b)) if a[i] == b[j]]
    assert list(zip(ai.tolist(), bi.tolist())) == want


def test_canonicalize_and_gcxs_conversion(orc, conv):
    shape = tuple(conv["shape"])
    c, d = orc.coo_canonicalize(conv["raw_coords"], conv["raw_data"], shape)
    assert np.array_equal(c, conv["can_coords"]) and np.array_equal(d, conv["can_data"])
    c, d = orc.coo_canonicalize(conv["raw_coords"], conv["raw_data"], shape, prune=True)
    assert np.array_equal(c, conv["pruned_coords"]) and np.array_equal(d, conv["pruned_data"])
    for i in range(6):
        ca = tuple(conv[f"g{i}_ca"])
        data, indices, indptr, _ = orc.coo_to_gcxs(conv["can_coords"], conv["can_data"], shape, ca)
        assert np.array_equal(data, conv[f"g{i}_data"])
        assert np.array_equal(indices, conv[f"g{i}_indices"])
        assert np.array_equal(indptr, conv[f"g{i}_indptr"])
    c, d = orc.coo_canonicalize(np.array([[0, 1, 2]]), np.array([0.0, -0.0, 1.0]), (4,), prune=True)
    assert np.array_equal(c, conv["negzero_coords"]) and np.signbit(d[0]) and len(d) == 2


def test_elemwise_union_merge_restatement(orc):
    g = np.load(os.path.join(GOLD, "elemwise.npz"))
    shape = tuple(g["shape"])
    kx, ky = orc.linear_loc(g["x_coords"], shape), orc.linear_loc(g["y_coords"], shape)
    for name, f in {"add": np.add, "subtract": np.subtract, "multiply": np.multiply, "maximum": np.maximum,
                    "minimum": np.minimum, "greater": np.greater, "not_equal": np.not_equal}.items():
        keys, vals, fill, _ = orc.elemwise_zero_fill(f, kx, g["x_data"], ky, g["y_data"])
        want_k = orc.linear_loc(g[f"{name}_coords"], shape)
        assert np.array_equal(keys, want_k), name
        assert np.array_equal(vals, g[f"{name}_data"]), name


def test_grouped_reduce_restatement(orc):
    g = np.load(os.path.join(GOLD, "reduce.npz"))
    shape = tuple(g["shape"])
    coords, data = g["x_coords"], g["x_data"]
    # sum over axis 2: groups are (i, j) pairs in C order == runs of the leading key
    vals, heads, counts = orc.grouped_reduce(data, coords[0] * shape[1] + coords[1], np.add)
    dense = np.zeros(shape[0] * shape[1])
    dense[(coords[0] * shape[1] + coords[1])[heads]] = vals
    k = next(i for i in range(int(g["n_reduce"])) if str(g[f"r{i}_name"]) == "sum" and
             np.array_equal(g[f"r{i}_axis"], 2) and not g[f"r{i}_keepdims"])
    assert np.array_equal(dense.reshape(shape[:2]), g[f"r{k}_dense"])
