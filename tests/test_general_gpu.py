"""N2 / N3: the general elementwise algorithm (any callable, any number of operands, keywords,
dense operands, non-zero fill values), var/std with a fill value and broadcast N-D matmul, against what the REAL reference
returned for the same inputs (tests/golden/general.npz, written by oracle/gen_golden.py `gen_general` from the cases in
tests/general_cases.py)."""
import os

import numpy as np
import pytest

import general_cases as gc

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "general.npz"))


@pytest.mark.parametrize("k", range(len(gc.CASES)), ids=[c[0] for c in gc.CASES])
def test_case_matches_the_reference(k):
    import sparse_amd as sp

    name, fn = gc.CASES[k]
    assert str(G[f"c{k}_name"]) == name, "tests/golden/general.npz is stale: run oracle/gen_golden.py"
    inp = {key[3:]: G[key] for key in G.files if key.startswith("in_")}
    got = gc.evaluate(sp, fn, inp)
    kind = str(G[f"c{k}_kind"])
    assert got["kind"] == kind, (name, got)
    if kind == "error":
        assert got["error"] == str(G[f"c{k}_error"]), name
        return
    want = G[f"c{k}_dense"]
    assert got["dense"].shape == want.shape and got["dense"].dtype == want.dtype, name
    if "matmul" in name or "var" in name or "std" in name:      # sums: same terms, another order
        assert np.allclose(got["dense"], want, rtol=1e-12, atol=1e-14, equal_nan=True), name
    else:                                                          # elementwise: the same function on the same values
        assert np.array_equal(got["dense"], want, equal_nan=True), name
    if kind == "sparse":
        assert got["cls"] == str(G[f"c{k}_cls"]), name
        wide = np.complex128 if np.iscomplexobj(G[f"c{k}_fill"]) else np.float64
        assert np.allclose(np.asarray(got["fill"], dtype=wide), np.asarray(G[f"c{k}_fill"], dtype=wide),
                           rtol=1e-12, atol=0, equal_nan=True), name
        assert np.asarray(got["fill"]).dtype == G[f"c{k}_fill"].dtype, name
        if "var" not in name and "std" not in name:   # (a variance that cancels to 0 exactly in one order need not in another)
            assert got["nnz"] == int(G[f"c{k}_nnz"]), name


def test_to_device_round_trip():
    import torch

    import sparse_amd as sp

    x = sp.random((30, 40), density=0.1, random_state=3)
    assert x.to_device(x.device) is x and x.to_device("cuda") is x
    g = sp.GCXS(x)
    assert g.to_device(torch.device("cuda", x.device.index or 0)) is g
    with pytest.raises(ValueError):
        x.to_device("cpu")
    with pytest.raises(ValueError):
        x.to_device(x.device, stream=1)
    if torch.cuda.device_count() > 1:
        y = x.to_device("cuda:1")
        assert y.device.index == 1 and np.array_equal(y.todense(), x.todense())


@pytest.mark.parametrize("shapes", [((5, 6, 7), (6, 1)), ((5, 6, 7), (7,)), ((4, 1, 7), (4, 6, 7)), ((8, 9), (8, 1)), ((3, 4, 5, 6), (4, 1, 6))])
def test_broadcast_multiply_matches_reduced_coordinates(shapes):
    """x * y with one operand broadcasting into the other: the reduced-coordinate gather (no replicas) must equal NumPy
    on the dense arrays - values, nnz and fill - and must step aside (materialisation) when func(0, y) is not the fill
    value everywhere (an inf in y: 0 * inf = NaN is a stored element)."""
    import sparse_amd as sp
    from sparse_amd import _umath

    sa, sb = shapes
    rng = np.random.default_rng(sum(sa) + len(sb))
    # the operand with the full shape has mixed signs, the broadcasting one is non-negative: 0 * (negative) = -0.0 is a
    # STORED element bit-wise (reference `equivalent`, _utils.py:448-452), so with negative values in the broadcasting
    # operand its own positions matter and the path must (and does) step aside
    full_first = int(np.prod(sa)) >= int(np.prod(sb))
    da = np.where(rng.random(sa) < 0.4, rng.random(sa) - (0.5 if full_first else 0.0), 0.0)
    db = np.where(rng.random(sb) < 0.6, rng.random(sb) - (0.0 if full_first else 0.5), 0.0)
    taken = []
    orig = _umath._broadcast_matched

    def spy(*a, **k):
        r = orig(*a, **k)
        taken.append(r is not None)
        return r

    _umath._broadcast_matched = spy
    try:
        for x, y in ((da, db), (db, da)):
            r = sp.COO.from_numpy(x) * sp.COO.from_numpy(y)
            want = x * y
            assert np.array_equal(r.todense(), want) and r.fill_value == 0
            assert r.nnz == np.count_nonzero(want.view(np.uint64)), "bit-wise pruning: -0.0 products are stored"
        assert taken == [True, True], "the reduced-coordinate path was not taken"
        del taken[:]
        r = sp.COO.from_numpy(da) + sp.COO.from_numpy(db)          # add: positions where x stores nothing are not fill
        assert np.array_equal(r.todense(), da + db) and taken == [False]
        small, big = (db, da) if full_first else (da, db)
        bad = small.copy()
        bad.flat[0] = np.inf                                        # 0 * inf = NaN is a stored element: materialise
        r = sp.COO.from_numpy(big) * sp.COO.from_numpy(bad)
        with np.errstate(invalid="ignore"):
            want = big * bad
        assert np.array_equal(r.todense(), want, equal_nan=True) and taken[-1] is False
    finally:
        _umath._broadcast_matched = orig
