"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE (pydata/sparse numba_backend,
imported in place from /root/reference under the no-op numba stub, see ref_loader.py).

    python oracle/gen_golden.py            # rewrites every fixture (deterministic seeds)
    python oracle/gen_golden.py array_api  # only tests/golden/array_api.npz

TEST INFRASTRUCTURE.  Runs only in the build container (the GPU box has no /root/reference);
the fixtures it writes are committed, travel to the GPU box, and pin both the CPU oracle
(tests/test_oracle_pin.py) and the HIP kernels (tests/test_*_gpu.py).  The reference ships no
golden vectors of its own, so these are produced with fixed seeds
(`sparse.random(..., random_state=<int>)`) plus the literal cases its tests do hold
(`test_small_values`, tests/test_dot.py:289-300).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def _rand_gcxs(sp, shape, density, seed, dtype, idt, ca):
    x = sp.random(shape, density=density, format="gcxs", compressed_axes=ca, random_state=seed, idx_dtype=idt)
    if np.dtype(dtype).kind == "f":
        x = (x - 0.25 * (x != 0)).astype(dtype)  # mixed signs, keeps sparsity pattern
        x = sp.GCXS(x, compressed_axes=ca, idx_dtype=idt) if not isinstance(x, sp.GCXS) else x
    else:
        x = (x * 100).astype(dtype)
    return x


def gen_dot(sp):
    """A1/A2/A3/A4/A5: every `_dot` kernel, dense and sparse return types."""
    rng = np.random.default_rng(7)
    cases = {}
    k = 0
    # --- GCXS(csr|csc) @ ndarray, dense out -------------------------------------------------
    for dtype, idt in ((np.float32, np.int32), (np.float32, np.int64), (np.float64, np.int64), (np.int64, np.int64),
                       (np.int32, np.int32)):
        for N in (1, 7, 64, 128, 130):
            for ca in ((0,), (1,)):
                M, K = 60, 45
                a = _rand_gcxs(sp, (M, K), 0.15, 100 + k, dtype, idt, ca)
                if np.dtype(dtype).kind == "f":
                    b = (rng.random((K, N)) - 0.5).astype(dtype)
                    b[:, 0] = 0  # an all-zero dense column
                else:
                    b = rng.integers(-9, 9, size=(K, N)).astype(dtype)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    c = sp.tensordot(a, b, axes=1)
                assert isinstance(c, np.ndarray)
                pre = f"gd{k}_"
                cases[pre + "data"], cases[pre + "indices"], cases[pre + "indptr"] = a.data, a.indices, a.indptr
                cases[pre + "ca"] = np.array(ca)
                cases[pre + "shape"] = np.array((M, K))
                cases[pre + "b"] = b
                cases[pre + "out"] = c
                k += 1
    cases["n_gcxs_dense"] = np.array(k)

    # --- rows: empty rows + one dense row + NaN in A (matmul warns, NaN*0 skipped structurally) ---
    a = sp.random((40, 300), density=0.02, format="gcxs", compressed_axes=(0,), random_state=5).astype(np.float32)
    d = a.todense()
    d[3, :] = 0
    d[4, :] = 0
    d[17, :] = np.linspace(-1, 1, 300, dtype=np.float32)
    d[17, 0] = 1
    d[20, 5] = np.nan
    a = sp.GCXS.from_numpy(d, compressed_axes=(0,))
    b = (rng.random((300, 128)) - 0.5).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = a @ b
    cases.update(edge_data=a.data, edge_indices=a.indices, edge_indptr=a.indptr, edge_b=b, edge_out=c)

    # --- COO @ ndarray / ndarray @ COO, dense + sparse returns --------------------------------
    x = sp.random((50, 30), density=0.2, format="coo", random_state=11).astype(np.float64)
    b = rng.random((30, 9))
    a2 = rng.random((8, 50))
    cases.update(coo_coords=x.coords, coo_data=x.data, coo_b=b, coo_out=sp.tensordot(x, b, axes=1),
                 coo_a2=a2, coo_out2=sp.tensordot(a2, x, axes=1))
    r = sp.tensordot(x, b, axes=1, return_type=sp.COO)
    cases.update(coo_sp_coords=r.coords, coo_sp_data=r.data)
    r = sp.tensordot(a2, x, axes=1, return_type=sp.COO)
    cases.update(coo_sp2_coords=r.coords, coo_sp2_data=r.data)

    # --- sparse-returning GCXS @ dense (A1s, A2) ---------------------------------------------
    for tag, ca in (("csr", (0,)), ("csc", (1,))):
        a = _rand_gcxs(sp, (30, 25), 0.2, 21, np.float64, np.int64, ca)
        b = rng.random((25, 6))
        b[:, 2] = 0
        b[rng.random((25, 6)) < 0.5] = 0
        r = sp.tensordot(a, b, axes=1, return_type=sp.GCXS)
        cases.update({f"sp_{tag}_data": a.data, f"sp_{tag}_indices": a.indices, f"sp_{tag}_indptr": a.indptr,
                      f"sp_{tag}_b": b, f"sp_{tag}_out_dense": r.todense(), f"sp_{tag}_out_nnz": np.array(r.nnz),
                      f"sp_{tag}_out_ca": np.array(r.compressed_axes)})

    # --- SpGEMM: GCXS @ GCXS (csr and csc), COO @ COO -----------------------------------------
    for tag, ca in (("csr", (0,)), ("csc", (1,))):
        a = _rand_gcxs(sp, (40, 35), 0.12, 31, np.float64, np.int64, ca)
        b = _rand_gcxs(sp, (35, 45), 0.12, 32, np.float64, np.int64, ca)
        r = a @ b
        assert isinstance(r, sp.GCXS)
        rc = r.tocoo()
        cases.update({f"gg_{tag}_a_data": a.data, f"gg_{tag}_a_indices": a.indices, f"gg_{tag}_a_indptr": a.indptr,
                      f"gg_{tag}_b_data": b.data, f"gg_{tag}_b_indices": b.indices, f"gg_{tag}_b_indptr": b.indptr,
                      f"gg_{tag}_out_coords": rc.coords, f"gg_{tag}_out_data": rc.data,
                      f"gg_{tag}_out_ca": np.array(r.compressed_axes),
                      f"gg_{tag}_raw_indices": r.indices, f"gg_{tag}_raw_indptr": r.indptr, f"gg_{tag}_raw_data": r.data})
    x = sp.random((30, 20), density=0.2, random_state=41)
    y = sp.random((20, 25), density=0.2, random_state=42)
    r = x @ y
    cases.update(cc_x_coords=x.coords, cc_x_data=x.data, cc_y_coords=y.coords, cc_y_data=y.data,
                 cc_out_coords=r.coords, cc_out_data=r.data)
    # integer SpGEMM must be exact
    xi = (sp.random((25, 25), density=0.3, random_state=43, format="gcxs", compressed_axes=(0,)) * 10).astype(np.int64)
    r = (xi @ xi).tocoo()
    cases.update(ci_data=xi.data, ci_indices=xi.indices, ci_indptr=xi.indptr, ci_out_coords=r.coords, ci_out_data=r.data)

    # --- literal KAT of the reference: test_small_values (tests/test_dot.py:289-300) ----------
    a = sp.COO.from_numpy(np.array([[3.6e-100, 0.0, 0.0], [0.0, 4.5e-225, 0.0]]))
    bb = np.array([[1e-120, 0.0], [0.0, 2.0], [1.0, 1.0]])
    cases.update(small_a=a.todense(), small_b=bb, small_out=sp.dot(a, bb))

    # --- N-D tensordot: 3-D COO x dense, axes=1 (BASELINE config 3, reduced) and GCXS 3-D -----
    c3 = sp.random((12, 10, 8), density=0.1, random_state=51)
    dmat = rng.random((8, 6))
    cases.update(t3_coords=c3.coords, t3_data=c3.data, t3_d=dmat, t3_out=sp.tensordot(c3, dmat, axes=1))
    g3 = sp.GCXS(c3, compressed_axes=(1,))
    cases.update(t3g_out=sp.tensordot(g3, dmat, axes=((2,), (0,))))
    dd = rng.random((10, 12, 5))
    cases.update(t3_dd=dd, t3_out2=sp.tensordot(c3, dd, axes=((0, 1), (1, 0))))
    _save("dot", **cases)


def gen_convert(sp):
    """A6 / T1-T3: canonicalisation and format conversion (integer side, bit-exact)."""
    rng = np.random.default_rng(3)
    cases = {}
    shape = (7, 5, 6, 4)
    nnz = 150
    coords = np.stack([rng.integers(0, s, nnz) for s in shape])
    data = rng.integers(-3, 4, nnz).astype(np.float64)  # duplicates + explicit zeros + cancellation
    x = sp.COO(coords, data, shape=shape)  # sort + sum duplicates (no prune)
    xp = sp.COO(coords, data, shape=shape, prune=True)
    cases.update(raw_coords=coords, raw_data=data, shape=np.array(shape), can_coords=x.coords, can_data=x.data,
                 pruned_coords=xp.coords, pruned_data=xp.data)
    for i, ca in enumerate([(0,), (1,), (3,), (0, 1), (1, 3), (0, 2, 3)]):
        g = sp.GCXS(x, compressed_axes=ca)
        cases.update({f"g{i}_ca": np.array(ca), f"g{i}_data": g.data, f"g{i}_indices": g.indices, f"g{i}_indptr": g.indptr})
        back = g.tocoo()
        assert np.array_equal(back.coords, x.coords)
    g = sp.GCXS(x, compressed_axes=(0,))
    h = g.change_compressed_axes((2, 3))
    cases.update(cca_data=h.data, cca_indices=h.indices, cca_indptr=h.indptr)
    t = g.transpose((2, 0, 3, 1))
    cases.update(tr_ca=np.array(t.compressed_axes), tr_data=t.data, tr_indices=t.indices, tr_indptr=t.indptr,
                 tr_shape=np.array(t.shape))
    r = g.reshape((35, 24))
    cases.update(rs_ca=np.array(r.compressed_axes), rs_data=r.data, rs_indices=r.indices, rs_indptr=r.indptr)
    ct = x.transpose((3, 1, 0, 2))
    cr = x.reshape((35, 24))
    cases.update(ct_coords=ct.coords, ct_data=ct.data, cr_coords=cr.coords, cr_data=cr.data)
    # -0.0 survives pruning (Appendix C.1)
    z = sp.COO(np.array([[0, 1, 2]]), np.array([0.0, -0.0, 1.0]), shape=(4,), prune=True)
    cases.update(negzero_coords=z.coords, negzero_data=z.data)
    cases.update(dense=x.todense())
    _save("convert", **cases)


def gen_elemwise(sp):
    """A7: binary/unary elementwise on same-shape zero-fill COO operands (+ one broadcast and one
    non-zero-fill case for the general path)."""
    cases = {}
    shape = (9, 8, 7)
    x = sp.random(shape, density=0.2, random_state=61)
    y = sp.random(shape, density=0.2, random_state=62)
    x = sp.COO(x.coords, x.data - 0.5, shape=shape)
    y = sp.COO(y.coords, y.data - 0.5, shape=shape)
    # force exact cancellation at a few coincident positions
    both = sp.COO(x.coords[:, :5], -x.data[:5], shape=shape)
    y = (y + both)
    y = sp.COO(y.coords, y.data, shape=shape)
    cases.update(shape=np.array(shape), x_coords=x.coords, x_data=x.data, y_coords=y.coords, y_data=y.data)
    ops = {"add": np.add, "subtract": np.subtract, "multiply": np.multiply, "maximum": np.maximum,
           "minimum": np.minimum, "greater": np.greater, "less": np.less, "not_equal": np.not_equal,
           "greater_equal": np.greater_equal, "equal": np.equal, "divide": np.divide}
    for name, f in ops.items():
        with np.errstate(all="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = f(x, y)
        cases.update({f"{name}_coords": r.coords, f"{name}_data": r.data, f"{name}_fill": np.asarray(r.fill_value)})
    for name, f in {"negative": np.negative, "abs": np.abs, "exp": np.exp, "sin": np.sin, "sqrt_abs": None,
                    "astype_f32": None, "astype_bool": None, "mul_scalar": None, "add_scalar": None}.items():
        with np.errstate(all="ignore"):
            if name == "sqrt_abs":
                r = np.sqrt(np.abs(x))
            elif name == "astype_f32":
                r = x.astype(np.float32)
            elif name == "astype_bool":
                r = sp.COO(x.coords, np.where(np.arange(x.nnz) % 3 == 0, 0.0, x.data), shape=shape).astype(bool)
            elif name == "mul_scalar":
                r = x * 3.0
            elif name == "add_scalar":
                r = x + 1.0
            else:
                r = f(x)
        cases.update({f"u_{name}_coords": r.coords, f"u_{name}_data": r.data, f"u_{name}_fill": np.asarray(r.fill_value)})
    # integer operands
    xi = sp.COO(x.coords, (x.data * 20).astype(np.int64), shape=shape)
    yi = sp.COO(y.coords, (y.data * 20).astype(np.int64), shape=shape)
    for name, f in {"add": np.add, "multiply": np.multiply, "bitwise_and": np.bitwise_and}.items():
        r = f(xi, yi)
        cases.update({f"i_{name}_coords": r.coords, f"i_{name}_data": r.data})
    cases.update(xi_data=xi.data, yi_data=yi.data, xi_coords=xi.coords, yi_coords=yi.coords)
    # broadcasting (general path): (9,8,7) * (8,1)
    z = sp.random((8, 1), density=0.6, random_state=63)
    r = x * z
    cases.update(bz_coords=z.coords, bz_data=z.data, bmul_coords=r.coords, bmul_data=r.data)
    # GCXS in -> GCXS out with same compressed axes
    gx, gy = sp.GCXS(x, compressed_axes=(1,)), sp.GCXS(y, compressed_axes=(1,))
    r = gx + gy
    cases.update(gadd_ca=np.array(r.compressed_axes), gadd_data=r.data, gadd_indices=r.indices, gadd_indptr=r.indptr)
    # sparse (*) dense gather path (SDDMM formulation, A9): s * (a @ b)
    s = sp.random((20, 30), density=0.1, random_state=64)
    rng = np.random.default_rng(65)
    a, b = rng.random((20, 6)), rng.random((6, 30))
    r = s * (a @ b)
    cases.update(sd_s_coords=s.coords, sd_s_data=s.data, sd_a=a, sd_b=b, sd_out_coords=r.coords, sd_out_data=r.data)
    _save("elemwise", **cases)


def gen_reduce(sp):
    """A8: reductions over axis sets, with and without non-zero fill values."""
    cases = {}
    shape = (6, 7, 5)
    x = sp.random(shape, density=0.3, random_state=71)
    x = sp.COO(x.coords, x.data - 0.4, shape=shape)
    cases.update(shape=np.array(shape), x_coords=x.coords, x_data=x.data)
    k = 0
    for name in ("sum", "prod", "max", "min", "mean", "var", "std", "any", "all"):
        for axis in (None, 0, 1, 2, (0, 1), (1, 2), (0, 2), (0, 1, 2)):
            for keepdims in (False, True):
                with np.errstate(all="ignore"), warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    r = getattr(x, name)(axis=axis, keepdims=keepdims)
                d = r.todense() if hasattr(r, "todense") else np.asarray(r)
                cases[f"r{k}_name"] = np.array(name)
                cases[f"r{k}_axis"] = np.array(-99 if axis is None else axis)
                cases[f"r{k}_keepdims"] = np.array(keepdims)
                cases[f"r{k}_dense"] = d
                cases[f"r{k}_nnz"] = np.array(getattr(r, "nnz", -1))
                cases[f"r{k}_fill"] = np.asarray(getattr(r, "fill_value", 0))
                k += 1
    cases["n_reduce"] = np.array(k)
    # non-zero fill value: (x + 1).sum(axis=1) has fill_value = n_cols (Appendix D5)
    r = (x + 1).sum(axis=1)
    cases.update(fv_dense=r.todense(), fv_fill=np.asarray(r.fill_value), fv_nnz=np.array(r.nnz))
    # GCXS reductions
    g = sp.GCXS(x, compressed_axes=(1,))
    for j, axis in enumerate((0, (0, 2), None)):
        r = g.sum(axis=axis)
        cases[f"g{j}_dense"] = r.todense()
    # integer + narrow idx dtype (reference test_reduce_narrow_idx_dtype, tests/test_coo.py:783-799)
    xi = sp.COO(x.coords.astype(np.uint8), (x.data * 50).astype(np.int32), shape=shape)
    cases.update(isum=xi.sum(axis=(0, 2)).todense(), imax=xi.max(axis=1).todense())
    _save("reduce", **cases)


def gen_nd(sp):
    """N2: N-D matmul broadcasting (`_matmul_recurser`), stack / concatenate, x[i]."""
    cases = {}
    rng = np.random.default_rng(81)
    shapes = [((2, 3, 4, 5), (2, 3, 5, 6)), ((2, 1, 3, 4), (1, 5, 4, 6)), ((3, 4), (2, 4, 5)), ((2, 3, 4), (4, 5)),
              ((1, 1, 4), (3, 4, 2)), ((3, 2, 4, 5), (1, 5, 3))]
    for k, (sa, sb) in enumerate(shapes):
        a = sp.random(sa, density=0.4, random_state=90 + k)
        b = sp.random(sb, density=0.4, random_state=190 + k)
        bd = rng.random(sb)
        r_ss = sp.matmul(a, b)
        r_sd = sp.matmul(a, bd)
        r_ds = sp.matmul(bd.swapaxes(-1, -2) if False else rng.random(sa), b)
        cases.update({f"m{k}_a_coords": a.coords, f"m{k}_a_data": a.data, f"m{k}_a_shape": np.array(sa),
                      f"m{k}_b_coords": b.coords, f"m{k}_b_data": b.data, f"m{k}_b_shape": np.array(sb),
                      f"m{k}_bd": bd, f"m{k}_ss": r_ss.todense() if hasattr(r_ss, "todense") else r_ss,
                      f"m{k}_ss_sparse": np.array(hasattr(r_ss, "todense")),
                      f"m{k}_sd": r_sd.todense() if hasattr(r_sd, "todense") else r_sd})
        ga, gb = sp.GCXS(a), sp.GCXS(b)
        r_gg = sp.matmul(ga, gb)
        cases[f"m{k}_gg"] = r_gg.todense()
        cases[f"m{k}_gg_fmt"] = np.array(r_gg.format)
    cases["n_matmul"] = np.array(len(shapes))
    xs = [sp.random((4, 5, 3), density=0.3, random_state=300 + i) for i in range(3)]
    for i, x in enumerate(xs):
        cases.update({f"s{i}_coords": x.coords, f"s{i}_data": x.data})
    for ax in (0, 1, 2, 3, -1):
        r = sp.stack(xs, axis=ax)
        cases.update({f"stack{ax}_coords": r.coords, f"stack{ax}_data": r.data, f"stack{ax}_shape": np.array(r.shape)})
    for ax in (0, 1, 2):
        r = sp.concatenate(xs, axis=ax)
        cases.update({f"cat{ax}_coords": r.coords, f"cat{ax}_data": r.data, f"cat{ax}_shape": np.array(r.shape)})
    r = sp.concatenate([sp.GCXS(x, compressed_axes=(0,)) for x in xs], axis=1)
    cases.update(gcat_fmt=np.array(r.format), gcat_dense=r.todense())
    for i in (0, 2, -1):
        r = xs[0][i]
        cases.update({f"take{i}_coords": r.coords, f"take{i}_data": r.data})
    _save("nd", **cases)


def gen_matrix(sp):
    """Breadth: the parameter grids of the reference's own tests — tensordot over operand formats x
    return_type (tests/test_dot.py:15-80), binary elementwise over ndim 1-4 and {COO,GCXS}
    (tests/test_elemwise.py:143-203), reductions over dtypes (tests/test_coo.py:43-200)."""
    cases = {}
    rng = np.random.default_rng(91)
    # ---- tensordot grid -------------------------------------------------------------------------
    grid = [((3, 4, 5), (4, 3), ((1, 0), (0, 1))), ((3, 4), (4, 5), 1), ((4, 5), (5, 6), ((1,), (0,))),
            ((3, 4, 5), (5, 4, 2), ((1, 2), (1, 0))), ((2, 3, 4), (4, 3, 2), 0 if False else ((2, 1), (0, 1)))]
    k = 0
    for sa, sb, axes in grid:
        a = sp.random(sa, density=0.5, random_state=400 + k)
        b = sp.random(sb, density=0.5, random_state=500 + k)
        ops = {"coo": lambda x: x, "gcxs": lambda x: sp.GCXS(x), "dense": lambda x: x.todense()}
        for fa in ("coo", "gcxs", "dense"):
            for fb in ("coo", "gcxs", "dense"):
                if fa == "dense" and fb == "dense":
                    continue
                for rt_name, rt in (("none", None), ("coo", sp.COO), ("gcxs", sp.GCXS), ("ndarray", np.ndarray)):
                    r = sp.tensordot(ops[fa](a), ops[fb](b), axes=axes, return_type=rt)
                    kind = "ndarray" if isinstance(r, np.ndarray) else r.format
                    d = r if isinstance(r, np.ndarray) else r.todense()
                    cases[f"td{k}_{fa}_{fb}_{rt_name}_kind"] = np.array(kind)
                    cases[f"td{k}_{fa}_{fb}_{rt_name}_nnz"] = np.array(-1 if isinstance(r, np.ndarray) else r.nnz)
                    if f"td{k}_dense" not in cases:
                        cases[f"td{k}_dense"] = d
                    assert np.allclose(d, cases[f"td{k}_dense"])
        cases.update({f"td{k}_a_coords": a.coords, f"td{k}_a_data": a.data, f"td{k}_a_shape": np.array(sa),
                      f"td{k}_b_coords": b.coords, f"td{k}_b_data": b.data, f"td{k}_b_shape": np.array(sb),
                      f"td{k}_axes": np.array(axes, dtype=object) if False else np.array(str(axes))})
        k += 1
    cases["n_td"] = np.array(k)
    # ---- binary elementwise over ndim 1..4 --------------------------------------------------------
    k = 0
    for shape in ((17,), (6, 7), (4, 5, 6), (3, 4, 2, 5)):
        x = sp.random(shape, density=0.4, random_state=600 + k)
        y = sp.random(shape, density=0.4, random_state=700 + k)
        x = sp.COO(x.coords, x.data - 0.5, shape=shape)
        for name in ("add", "subtract", "multiply", "maximum", "greater", "less_equal", "not_equal"):
            with np.errstate(all="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r = getattr(np, name)(x, y)
            cases[f"ew{k}_{name}_coords"], cases[f"ew{k}_{name}_data"] = r.coords, r.data
            cases[f"ew{k}_{name}_fill"] = np.asarray(r.fill_value)
            if len(shape) > 1:
                rg = getattr(np, name)(sp.GCXS(x), sp.GCXS(y))
                cases[f"ew{k}_{name}_gfmt"] = np.array(rg.format)
                rm = getattr(np, name)(sp.GCXS(x), y)
                cases[f"ew{k}_{name}_mfmt"] = np.array(rm.format)
        cases.update({f"ew{k}_x_coords": x.coords, f"ew{k}_x_data": x.data, f"ew{k}_y_coords": y.coords,
                      f"ew{k}_y_data": y.data, f"ew{k}_shape": np.array(shape)})
        k += 1
    cases["n_ew"] = np.array(k)
    # ---- reductions over dtypes -------------------------------------------------------------------------
    k = 0
    base = sp.random((5, 6, 4), density=0.35, random_state=800)
    for dt in (np.float64, np.float32, np.int64, np.int32):
        x = sp.COO(base.coords, ((base.data - 0.4) * (1 if np.dtype(dt).kind == "f" else 20)).astype(dt), shape=base.shape)
        for name in ("sum", "prod", "max", "min", "mean", "any", "all"):
            for axis in (None, 0, (0, 2), 1):
                with np.errstate(all="ignore"), warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    r = getattr(x, name)(axis=axis)
                cases[f"rd{k}_dense"] = r.todense()
                cases[f"rd{k}_meta"] = np.array(f"{np.dtype(dt).name}|{name}|{axis}")
                k += 1
    cases.update(rd_coords=base.coords, rd_data=base.data, n_rd=np.array(k))
    _save("matrix", **cases)


def gen_select(sp):
    """N3: `where`, NaN-skipping reductions, creation functions (reference _coo/common.py:334-733,
    _common.py:1561-1857)."""
    cases = {}
    shape = (5, 6, 7)
    rng = np.random.default_rng(91)

    def arr(seed, density, nan=0.0):
        r = np.random.default_rng(seed)
        d = np.zeros(shape)
        m = r.random(shape) < density
        d[m] = r.random(int(m.sum())) - 0.4
        if nan:
            d[r.random(shape) < nan] = np.nan
        return d

    c, x, y = arr(1, 0.4), arr(2, 0.5), arr(3, 0.3)
    xn = arr(4, 0.5, nan=0.2)
    xn[1, 2, :] = np.nan
    cases.update(shape=np.array(shape), c=c, x=x, y=y, xn=xn)
    sc, sx, sy, sxn = (sp.COO.from_numpy(a) for a in (c > 0, x, y, xn))
    k = 0
    for label, r in (("where(c,x,y)", sp.where(sc, sx, sy)), ("where(c,2.5,y)", sp.where(sc, 2.5, sy)),
                     ("where(c,x,-1.0)", sp.where(sc, sx, -1.0)), ("where(isnan(xn),0,xn)", sp.where(np.isnan(sxn), 0, sxn)),
                     ("where(xn,x,y)", sp.where(sxn, sx, sy))):
        cases[f"w{k}_label"] = np.array(label)
        cases[f"w{k}_dense"] = r.todense()
        cases[f"w{k}_nnz"] = np.array(r.nnz)
        cases[f"w{k}_fill"] = np.asarray(r.fill_value)
        k += 1
    cases["n_where"] = np.array(k)
    for j, a in enumerate(sp.where(sy)):
        cases[f"w1arg_{j}"] = np.asarray(a)
    k = 0
    for name in ("nansum", "nanprod", "nanmax", "nanmin", "nanmean"):
        for axis in (None, 0, 2, (0, 1)):
            with np.errstate(all="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r = getattr(sp, name)(sxn, axis=axis)
            cases[f"n{k}_name"] = np.array(name)
            cases[f"n{k}_axis"] = np.array(-99 if axis is None else axis)
            cases[f"n{k}_dense"] = r.todense() if hasattr(r, "todense") else np.asarray(r)
            cases[f"n{k}_nnz"] = np.array(getattr(r, "nnz", -1))
            cases[f"n{k}_fill"] = np.asarray(getattr(r, "fill_value", 0))
            k += 1
    cases["n_nan"] = np.array(k)
    for j, (n, m, kk) in enumerate(((4, None, 0), (3, 5, 1), (5, 3, -2), (3, 3, 7))):
        e = sp.eye(n, m, k=kk, dtype=np.float32)
        cases[f"eye{j}_args"] = np.array([n, -1 if m is None else m, kk])
        cases[f"eye{j}_coords"] = e.coords
        cases[f"eye{j}_data"] = e.data
    f = sp.full((2, 3), 7, dtype=np.int32)
    cases.update(full_fill=np.asarray(f.fill_value), full_nnz=np.array(f.nnz), full_dense=f.todense(),
                 ones_plus=(sx + sp.ones(shape)).todense())
    _save("select", **cases)


def gen_einsum(sp):
    """N4: einsum through the real reference (`_common.py:1400-1476`): values, result format and dtype."""
    cases = {}
    pats = ["ab,bc->ac", "ab,cb", "abc,cd->abd", "aab,bc->ac", "ab,ab->a", "a,ab,abc->abc", "abc->ca", "aa->a", "ab,bc,cd->ad",
            "...ab,...b->...a", "bca,cdb,dbf,afc->", "ab,b->ab"]
    rng = np.random.default_rng(77)
    for k, pat in enumerate(pats):
        terms = pat.split("->")[0].split(",")
        ops = []
        for j, t in enumerate(terms):
            nd = len(t.replace("...", "x"))
            d = rng.random((4,) * nd) - 0.3
            d[rng.random(d.shape) < 0.6] = 0.0
            cases[f"e{k}_op{j}"] = d
            ops.append(sp.COO.from_numpy(d))
        r = sp.einsum(pat, *ops)
        cases[f"e{k}_pat"] = np.array(pat)
        cases[f"e{k}_n"] = np.array(len(ops))
        cases[f"e{k}_dense"] = r.todense()
        cases[f"e{k}_nnz"] = np.array(r.nnz)
    cases["n_einsum"] = np.array(len(pats))
    # result formats (reference tests/test_einsum.py `format_test_cases`)
    a = rng.random((2, 2, 2)); a[rng.random(a.shape) < 0.4] = 0
    b = rng.random((2, 2, 2)); b[rng.random(b.shape) < 0.4] = 0
    cases.update(fa=a, fb=b)
    fm = []
    for fa_, fb_ in (("coo", "coo"), ("gcxs", "gcxs"), ("coo", "gcxs"), ("coo", "dense"), ("dense", "gcxs")):
        oa = a if fa_ == "dense" else sp.COO.from_numpy(a).asformat(fa_)
        ob = b if fb_ == "dense" else sp.COO.from_numpy(b).asformat(fb_)
        r = sp.einsum("abc,cda->abd", oa, ob)
        fm.append(f"{fa_},{fb_},{type(r).__name__.lower()}")
        cases[f"f_{fa_}_{fb_}"] = r.todense()
    cases["formats"] = np.array(fm)
    _save("einsum", **cases)


def gen_general(sp, module="general_cases", out_name="general"):
    """N2/N3: arbitrary callables, N operands, keywords, dense operands, non-zero-fill var/std, broadcast N-D matmul -
    the cases of tests/general_cases.py evaluated by the reference (`_Elemwise`, _umath.py:392-751; `_matmul_recurser`,
    _common.py:278-293; `var/std`, _sparse_array.py:704-876).  With `module="array_api_cases"`: the array-namespace
    functions of `_coo/common.py` / `_common.py` (tests/array_api_cases.py -> array_api.npz)."""
    import importlib

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    gc = importlib.import_module(module)

    inp = gc.inputs()
    out = {f"in_{k}": v for k, v in inp.items()}
    for k, (name, fn) in enumerate(gc.CASES):
        r = gc.evaluate(sp, fn, inp)
        out[f"c{k}_name"] = np.array(name)
        out[f"c{k}_kind"] = np.array(r["kind"])
        if r["kind"] == "error":
            out[f"c{k}_error"] = np.array(r["error"])
            print(f"    {name}: raises {r['error']}")
            continue
        out[f"c{k}_dense"] = r["dense"]
        if r["kind"] == "sparse":
            out[f"c{k}_nnz"], out[f"c{k}_fill"], out[f"c{k}_cls"] = np.array(r["nnz"]), r["fill"], np.array(r["cls"])
    out["n_cases"] = np.array(len(gc.CASES))
    if out_name == "array_api":   # the public names of the backend (`__all__`, numba_backend/__init__.py:179-350)
        import sparse.numba_backend as nb

        out["reference_all"] = np.array(sorted(nb.__all__))
    _save(out_name, **out)


def main():
    sp = ref_loader.load()
    print("reference:", sp.__file__)
    if sys.argv[1:] == ["array_api"]:      # only the array-namespace cases
        gen_general(sp, "array_api_cases", "array_api")
        return
    gen_dot(sp)
    gen_convert(sp)
    gen_elemwise(sp)
    gen_reduce(sp)
    gen_nd(sp)
    gen_matrix(sp)
    gen_select(sp)
    gen_einsum(sp)
    gen_general(sp)
    gen_general(sp, "array_api_cases", "array_api")


if __name__ == "__main__":
    main()
