"""HIP path vs fixtures produced by the REAL reference (tests/golden/*.npz, made by
oracle/gen_golden.py).  Index structure is compared bit-exactly; floating-point values
bit-exactly wherever the HIP path keeps the reference's operation order (exact mode), else
within the stated tolerance."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP_RTOL = {np.dtype("float32"): 2e-6, np.dtype("float64"): 1e-14}


def _npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def assert_coo(x, coords, data, exact=True, rtol=None):
    """`assert_eq` of the reference (_utils.py:11-49) against stored golden arrays: canonical
    coords compared exactly; data exactly or to tolerance."""
    import sparse_amd

    assert isinstance(x, sparse_amd.COO)
    c, d = _npy(x.coords), _npy(x.data)
    assert c.shape == coords.shape, (c.shape, coords.shape)
    assert np.array_equal(c, coords)
    assert d.dtype == data.dtype, (d.dtype, data.dtype)
    if exact or d.dtype.kind not in "fc":
        assert np.array_equal(d, data, equal_nan=True)
    else:
        assert np.allclose(d, data, rtol=rtol or 1e-6, atol=0, equal_nan=True)


@pytest.fixture(scope="module")
def sp():
    import sparse_amd

    return sparse_amd


@pytest.fixture(scope="module")
def dot():
    return np.load(os.path.join(GOLD, "dot.npz"))


@pytest.fixture(scope="module")
def conv():
    return np.load(os.path.join(GOLD, "convert.npz"))


# ---------------------------------------------------------------- A6 / T1-T3: conversions ------
def test_coo_canonical_form(sp, conv):
    shape = tuple(conv["shape"])
    x = sp.COO(conv["raw_coords"], conv["raw_data"], shape=shape)
    assert_coo(x, conv["can_coords"], conv["can_data"])
    xp = sp.COO(conv["raw_coords"], conv["raw_data"], shape=shape, prune=True)
    assert_coo(xp, conv["pruned_coords"], conv["pruned_data"])
    assert np.array_equal(x.todense(), conv["dense"])
    z = sp.COO(np.array([[0, 1, 2]]), np.array([0.0, -0.0, 1.0]), shape=(4,), prune=True)
    assert_coo(z, conv["negzero_coords"], conv["negzero_data"])
    assert np.signbit(_npy(z.data)[0])  # -0.0 is not the fill value (bit-wise prune)
    y = sp.COO.from_numpy(conv["dense"])
    assert_coo(y, conv["pruned_coords"], conv["pruned_data"])


def test_coo_gcxs_roundtrips(sp, conv):
    shape = tuple(conv["shape"])
    x = sp.COO(conv["can_coords"], conv["can_data"], shape=shape, has_duplicates=False, sorted=True)
    for i in range(6):
        ca = tuple(int(c) for c in conv[f"g{i}_ca"])
        g = sp.GCXS(x, compressed_axes=ca)
        assert g.compressed_axes == ca
        assert np.array_equal(_npy(g.data), conv[f"g{i}_data"])
        assert np.array_equal(_npy(g.indices), conv[f"g{i}_indices"])
        assert np.array_equal(_npy(g.indptr), conv[f"g{i}_indptr"])
        assert_coo(g.tocoo(), conv["can_coords"], conv["can_data"])
        assert np.array_equal(g.todense(), conv["dense"])
    g = sp.GCXS(x, compressed_axes=(0,))
    h = g.change_compressed_axes((2, 3))
    for k in ("data", "indices", "indptr"):
        assert np.array_equal(_npy(getattr(h, k)), conv[f"cca_{k}"])
    t = g.transpose((2, 0, 3, 1))
    assert t.shape == tuple(conv["tr_shape"]) and t.compressed_axes == tuple(conv["tr_ca"])
    for k in ("data", "indices", "indptr"):
        assert np.array_equal(_npy(getattr(t, k)), conv[f"tr_{k}"])
    r = g.reshape((35, 24))
    assert r.compressed_axes == tuple(conv["rs_ca"])
    for k in ("data", "indices", "indptr"):
        assert np.array_equal(_npy(getattr(r, k)), conv[f"rs_{k}"])
    assert_coo(x.transpose((3, 1, 0, 2)), conv["ct_coords"], conv["ct_data"])
    assert_coo(x.reshape((35, 24)), conv["cr_coords"], conv["cr_data"])


# ---------------------------------------------------------------- A1/A2: GCXS @ dense ----------
@pytest.mark.parametrize("exact", [True, False])
def test_gcxs_dense_products(sp, dot, exact, monkeypatch):
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "EXACT_MULADD", exact)
    for k in range(int(dot["n_gcxs_dense"])):
        p = f"gd{k}_"
        ca = tuple(int(c) for c in dot[p + "ca"])
        a = sp.GCXS((dot[p + "data"], dot[p + "indices"], dot[p + "indptr"]), shape=tuple(dot[p + "shape"]),
                    compressed_axes=ca)
        want = dot[p + "out"]
        got = sp.tensordot(a, dot[p + "b"], axes=1)
        assert isinstance(got, np.ndarray) and got.dtype == want.dtype and got.shape == want.shape, k
        if exact or want.dtype.kind != "f":
            assert np.array_equal(got, want), k
        else:
            assert np.allclose(got, want, rtol=FP_RTOL[want.dtype], atol=FP_RTOL[want.dtype]), k


def test_edge_rows_nan_warning(sp, dot, monkeypatch):
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    a = sp.GCXS((dot["edge_data"], dot["edge_indices"], dot["edge_indptr"]), shape=(40, 300), compressed_axes=(0,))
    with pytest.warns(RuntimeWarning, match="Nan will not be propagated"):  # reference tests/test_dot.py:173-196
        got = a @ dot["edge_b"]
    assert np.array_equal(got, dot["edge_out"], equal_nan=True)


# ---------------------------------------------------------------- A3: COO with dense -------------
def test_coo_dense_products(sp, dot, monkeypatch):
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    x = sp.COO(dot["coo_coords"], dot["coo_data"], shape=(50, 30))
    assert np.array_equal(sp.tensordot(x, dot["coo_b"], axes=1), dot["coo_out"])
    assert np.array_equal(sp.tensordot(dot["coo_a2"], x, axes=1), dot["coo_out2"])
    assert_coo(sp.tensordot(x, dot["coo_b"], axes=1, return_type=sp.COO), dot["coo_sp_coords"], dot["coo_sp_data"])
    assert_coo(sp.tensordot(dot["coo_a2"], x, axes=1, return_type=sp.COO), dot["coo_sp2_coords"], dot["coo_sp2_data"])
    # literal KAT: tiny values must not be lost (reference tests/test_dot.py:289-300)
    a = sp.COO.from_numpy(dot["small_a"])
    assert np.array_equal(sp.dot(a, dot["small_b"]), dot["small_out"])


@pytest.mark.parametrize("tag,ca", [("csr", (0,)), ("csc", (1,))])
def test_sparse_returning_gcxs_dense(sp, dot, tag, ca):
    a = sp.GCXS((dot[f"sp_{tag}_data"], dot[f"sp_{tag}_indices"], dot[f"sp_{tag}_indptr"]), shape=(30, 25),
                compressed_axes=ca)
    r = sp.tensordot(a, dot[f"sp_{tag}_b"], axes=1, return_type=sp.GCXS)
    assert isinstance(r, sp.GCXS) and r.compressed_axes == tuple(dot[f"sp_{tag}_out_ca"])
    assert r.nnz == int(dot[f"sp_{tag}_out_nnz"])
    assert np.array_equal(r.todense(), dot[f"sp_{tag}_out_dense"])


# ---------------------------------------------------------------- A4/A5: sparse @ sparse ---------
@pytest.mark.parametrize("tag,ca", [("csr", (0,)), ("csc", (1,))])
def test_spgemm_gcxs(sp, dot, tag, ca):
    a = sp.GCXS(tuple(dot[f"gg_{tag}_a_{s}"] for s in ("data", "indices", "indptr")), shape=(40, 35), compressed_axes=ca)
    b = sp.GCXS(tuple(dot[f"gg_{tag}_b_{s}"] for s in ("data", "indices", "indptr")), shape=(35, 45), compressed_axes=ca)
    r = a @ b
    assert isinstance(r, sp.GCXS) and r.compressed_axes == tuple(dot[f"gg_{tag}_out_ca"])
    # rows of the reference come out unsorted (Appendix C.2): compare after canonical sort
    assert_coo(r.tocoo(), dot[f"gg_{tag}_out_coords"], dot[f"gg_{tag}_out_data"])
    assert r.nnz == len(dot[f"gg_{tag}_raw_data"])


def test_spgemm_coo_and_integers(sp, dot):
    x = sp.COO(dot["cc_x_coords"], dot["cc_x_data"], shape=(30, 20))
    y = sp.COO(dot["cc_y_coords"], dot["cc_y_data"], shape=(20, 25))
    assert_coo(x @ y, dot["cc_out_coords"], dot["cc_out_data"])
    xi = sp.GCXS((dot["ci_data"], dot["ci_indices"], dot["ci_indptr"]), shape=(25, 25), compressed_axes=(0,))
    assert_coo((xi @ xi).tocoo(), dot["ci_out_coords"], dot["ci_out_data"])


def test_nd_tensordot(sp, dot, monkeypatch):
    from sparse_amd import _settings

    monkeypatch.setattr(_settings, "EXACT_MULADD", True)
    c3 = sp.COO(dot["t3_coords"], dot["t3_data"], shape=(12, 10, 8))
    assert np.array_equal(sp.tensordot(c3, dot["t3_d"], axes=1), dot["t3_out"])
    assert np.array_equal(sp.tensordot(c3, dot["t3_dd"], axes=((0, 1), (1, 0))), dot["t3_out2"])
    g3 = sp.GCXS(c3, compressed_axes=(1,))
    assert np.array_equal(sp.tensordot(g3, dot["t3_d"], axes=((2,), (0,))), dot["t3g_out"])


# ---------------------------------------------------------------- A7: elementwise ----------------
def test_elementwise_binary_unary(sp):
    g = np.load(os.path.join(GOLD, "elemwise.npz"))
    shape = tuple(g["shape"])
    x = sp.COO(g["x_coords"], g["x_data"], shape=shape)
    y = sp.COO(g["y_coords"], g["y_data"], shape=shape)
    for name in ("add", "subtract", "multiply", "maximum", "minimum", "greater", "less", "not_equal",
                 "greater_equal", "equal", "divide"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = getattr(np, name)(x, y)
        assert_coo(r, g[f"{name}_coords"], g[f"{name}_data"])
        want_fill = g[f"{name}_fill"]
        assert np.array_equal(np.asarray(r.fill_value), want_fill, equal_nan=True), name
    exact_unary = {"negative": np.negative, "abs": np.abs, "mul_scalar": lambda v: v * 3.0,
                   "add_scalar": lambda v: v + 1.0, "astype_f32": lambda v: v.astype(np.float32),
                   "sqrt_abs": lambda v: np.sqrt(np.abs(v))}
    for name, f in exact_unary.items():
        r = f(x)
        assert_coo(r, g[f"u_{name}_coords"], g[f"u_{name}_data"])
        assert np.array_equal(np.asarray(r.fill_value), g[f"u_{name}_fill"])
    for name, f in {"exp": np.exp, "sin": np.sin}.items():  # libm vs device math: 2 ulp
        r = f(x)
        assert_coo(r, g[f"u_{name}_coords"], g[f"u_{name}_data"], exact=False, rtol=5e-16)
    xb = sp.COO(g["x_coords"], np.where(np.arange(len(g["x_data"])) % 3 == 0, 0.0, g["x_data"]), shape=shape)
    assert_coo(xb.astype(bool), g["u_astype_bool_coords"], g["u_astype_bool_data"])
    xi = sp.COO(g["xi_coords"], g["xi_data"], shape=shape)
    yi = sp.COO(g["yi_coords"], g["yi_data"], shape=shape)
    for name in ("add", "multiply", "bitwise_and"):
        assert_coo(getattr(np, name)(xi, yi), g[f"i_{name}_coords"], g[f"i_{name}_data"])
    gx, gy = sp.GCXS(x, compressed_axes=(1,)), sp.GCXS(y, compressed_axes=(1,))
    r = gx + gy
    assert isinstance(r, sp.GCXS) and r.compressed_axes == tuple(g["gadd_ca"])
    for k in ("data", "indices", "indptr"):
        assert np.array_equal(_npy(getattr(r, k)), g[f"gadd_{k}"])


def test_sddmm_matches_reference_formulation(sp):
    g = np.load(os.path.join(GOLD, "elemwise.npz"))
    s = sp.COO(g["sd_s_coords"], g["sd_s_data"], shape=(20, 30))
    # (1) the reference's own spelling: s * (a @ b) through the sparse (x) dense elementwise path
    r = s * (g["sd_a"] @ g["sd_b"])
    assert_coo(r, g["sd_out_coords"], g["sd_out_data"])
    # (2) the dedicated kernel (no dense intermediate): same structure, fp64 dot in another order
    r2 = sp.sddmm(s, g["sd_a"], g["sd_b"])
    assert_coo(r2, g["sd_out_coords"], g["sd_out_data"], exact=False, rtol=1e-14)


# ---------------------------------------------------------------- A8: reductions -------------------
def test_reductions(sp):
    g = np.load(os.path.join(GOLD, "reduce.npz"))
    shape = tuple(g["shape"])
    x = sp.COO(g["x_coords"], g["x_data"], shape=shape)
    done = 0
    for k in range(int(g["n_reduce"])):
        name = str(g[f"r{k}_name"])
        axis = g[f"r{k}_axis"]
        axis = None if axis.ndim == 0 and int(axis) == -99 else (int(axis) if axis.ndim == 0 else tuple(int(a) for a in axis))
        r = getattr(x, name)(axis=axis, keepdims=bool(g[f"r{k}_keepdims"]))
        want = g[f"r{k}_dense"]
        got = r.todense()
        assert got.shape == want.shape and got.dtype == want.dtype, (name, axis)
        if name in ("sum", "mean", "prod", "var", "std") and want.dtype.kind == "f":
            assert np.allclose(got, want, rtol=1e-12, atol=1e-15), (name, axis)
        else:
            assert np.array_equal(got, want), (name, axis)
        if r.ndim and name not in ("var", "std"):
            assert r.nnz == int(g[f"r{k}_nnz"]), (name, axis)
        assert np.allclose(np.asarray(r.fill_value, dtype=np.float64), np.asarray(g[f"r{k}_fill"], dtype=np.float64)), (name, axis)
        done += 1
    assert done == int(g["n_reduce"])
    r = (x + 1).sum(axis=1)  # non-zero fill value: Appendix D5
    assert np.allclose(r.todense(), g["fv_dense"], rtol=1e-13) and float(r.fill_value) == float(g["fv_fill"])
    gx = sp.GCXS(x, compressed_axes=(1,))
    for j, axis in enumerate((0, (0, 2), None)):
        assert np.allclose(gx.sum(axis=axis).todense(), g[f"g{j}_dense"], rtol=1e-13)
    xi = sp.COO(g["x_coords"].astype(np.uint8), (g["x_data"] * 50).astype(np.int32), shape=shape)
    assert np.array_equal(xi.sum(axis=(0, 2)).todense(), g["isum"])
    assert np.array_equal(xi.max(axis=1).todense(), g["imax"])
