"""Randomised comparison of the product API with NumPy on the dense arrays (small shapes, odd sizes, empty
dimensions): conversions, transposes, reshapes, elementwise with broadcasting, reductions, tensordot, matmul,
SDDMM.  Run on the GPU box:  python tools/fuzz_dense.py [seconds] [seed]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
t_end = time.time() + budget
it = 0
import os
BIG = bool(os.environ.get("FUZZ_BIG"))
TRACE = len(sys.argv) > 3      # any third argument: print every iteration's case and synchronise after every section (to find a device fault)


def mark(what):
    if TRACE:
        torch.cuda.synchronize()
        print("  ok", what, flush=True)



def rand_dense(shape, dtype, density):
    d = np.zeros(shape, dtype=dtype)
    m = rng.random(shape) < density
    if np.dtype(dtype).kind == "f":
        d[m] = (rng.random(int(m.sum())) - 0.5).astype(dtype)
    else:
        d[m] = rng.integers(-9, 10, int(m.sum())).astype(dtype)
    return d


def close(a, b, dtype, atol=None):
    a = a.todense() if hasattr(a, "todense") else np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a)
    if np.dtype(dtype).kind == "f":
        return a.shape == np.shape(b) and np.allclose(a, b, rtol=2e-5 if dtype == np.float32 else 1e-11, atol=atol or (5e-5 if dtype == np.float32 else 1e-12), equal_nan=True)
    return a.shape == np.shape(b) and np.array_equal(a, b)


while time.time() < t_end:
    it += 1
    nd = int(rng.integers(1, 5))
    shape = tuple(int(rng.choice([0, 1, 2, 3, 5, 17, 64, 65])) if rng.random() < 0.85 else int(rng.integers(1, 200)) for _ in range(nd))
    if BIG:       # (FUZZ_BIG=1: larger arrays - several tiles per kernel, runs across tiles, declined fast paths)
        shape = tuple(int(rng.choice([1, 2, 3, 17, 64, 141, 257, 1000, 2049])) for _ in range(nd))
        if not 2e5 < np.prod(shape) <= 2e7:
            continue
    if np.prod(shape) > (2e7 if BIG else 2e6):
        continue
    dtype = rng.choice([np.float64, np.float32, np.int64, np.int32])
    dens = float(rng.choice([0.0, 0.02, 0.3, 1.0]))
    d = rand_dense(shape, dtype, dens)
    x = sp.COO.from_numpy(d)
    ctx = (it, shape, np.dtype(dtype).name, dens)
    if TRACE:
        print(ctx, flush=True)
    assert close(x, d, dtype), ("roundtrip", ctx)
    # conversions
    if nd >= 1:
        k = int(rng.integers(1, nd + 1)) if nd > 1 else 1
        ca = tuple(sorted(rng.choice(nd, size=min(k, nd), replace=False).tolist()))
        if len(ca) < nd or nd == 1:
            try:
                g = sp.GCXS.from_coo(x, compressed_axes=ca) if nd > 1 else sp.GCXS.from_coo(x)
            except ValueError:
                g = None
            if g is not None:
                assert close(g, d, dtype), ("gcxs", ca, ctx)
                assert close(g.tocoo(), d, dtype), ("gcxs->coo", ca, ctx)
                if nd == 2:
                    g2 = g.change_compressed_axes((1 - ca[0],))
                    assert close(g2, d, dtype), ("swap", ctx)
                    assert close(g.T, d.T, dtype), ("gcxs.T", ctx)
    mark("conversions")
    perm = tuple(rng.permutation(nd).tolist())
    assert close(x.transpose(perm), d.transpose(perm), dtype), ("transpose", perm, ctx)
    if d.size:
        news = (d.size,) if rng.random() < 0.5 else ((shape[0], d.size // shape[0]) if shape[0] else (d.size,))
        assert close(x.reshape(news), d.reshape(news), dtype), ("reshape", news, ctx)
    mark("transpose/reshape")
    # elementwise with broadcasting
    bshape = tuple(s if rng.random() < 0.7 else 1 for s in shape)[int(rng.integers(0, nd)):]
    e = rand_dense(bshape, dtype, float(rng.choice([0.05, 0.5])))
    y = sp.COO.from_numpy(e)
    for name, f in (("add", np.add), ("mul", np.multiply), ("max", np.maximum), ("sub", np.subtract)):
        assert close(f(x, y), f(d, e), dtype), (name, bshape, ctx)
    assert close(x * 3, d * 3, dtype) and close(-x, -d, dtype) and close(abs(x), abs(d), dtype), ("scalar", ctx)
    mark("elementwise")
    # reductions
    for name in ("sum", "max", "min", "prod"):
        axes = [None] + [int(rng.integers(0, nd))] + ([tuple(sorted(rng.choice(nd, 2, replace=False).tolist()))] if nd >= 2 else [])
        for ax in axes:
            if name in ("max", "min") and (d.size == 0 or (ax is not None and any(shape[a] == 0 for a in (ax if isinstance(ax, tuple) else (ax,))))):
                continue
            want = getattr(d, name)(axis=ax)
            if TRACE and d.size > 500_000:
                np.save("gpurun_out/fuzz_case.npy", d)
                print("   reduce", name, ax, flush=True)
                if name == "prod" and not getattr(sys, "_ffi_traced", False):      # every C-ABI call by name, synchronised: the last one printed faulted
                    from sparse_amd import _ffi
                    sys._ffi_traced = True
                    real = _ffi.call

                    def traced(fname, *a):
                        print("      call", fname, [v for v in a if isinstance(v, int) and abs(v) < (1 << 40)][:8], flush=True)
                        r = real(fname, *a)
                        torch.cuda.synchronize()
                        return r
                    _ffi.call = traced
                    import sparse_amd._kernels as _K, sparse_amd._reduce as _R
                    for mod in (_K, _R):
                        if hasattr(mod, "_ffi"):
                            mod._ffi.call = traced
            got = getattr(x, name)(axis=ax)
            if TRACE and d.size > 500_000:
                torch.cuda.synchronize()
            # fp32 running sums drift like eps * n * |partial sum| (the reference's reduceat accumulates in fp32 too)
            assert close(got, want, dtype, atol=(1e-7 * d.size + 5e-5) if dtype == np.float32 and name in ('sum', 'prod') else None), (name, ax, ctx)
    mark("reductions")
    # tensordot / matmul with a dense operand
    if nd >= 1 and np.dtype(dtype).kind == "f":
        kdim = shape[-1]
        n = int(rng.choice([1, 3, 64, 130]))
        bd = (rng.random((kdim, n)) - 0.5).astype(dtype)
        bt = torch.from_numpy(bd).cuda()
        assert close(sp.tensordot(x, bt, axes=1), np.tensordot(d, bd, axes=1), dtype), ("tensordot", n, ctx)
        mark(("tensordot", n))
        if nd == 2:
            assert close(x @ bt, d @ bd, dtype), ("matmul", n, ctx)
            g = sp.GCXS.from_coo(x, compressed_axes=(int(rng.integers(0, 2)),))
            mark("matmul")
            assert close(g @ bt, d @ bd, dtype), ("gcxs matmul", n, ctx)
            mark("gcxs matmul")
            x2 = sp.COO.from_numpy(rand_dense((shape[1], int(rng.choice([1, 7, 70]))), dtype, 0.2))
            assert close(x @ x2, d @ x2.todense(), dtype), ("spgemm", ctx)
            mark("spgemm")
            ad = (rng.random((shape[0], 16)) - 0.5).astype(np.float32)
            bdn = (rng.random((shape[1], 16)) - 0.5).astype(np.float32)
            if dtype == np.float32:
                r = sp.sddmm(x, torch.from_numpy(ad).cuda(), bt=torch.from_numpy(bdn).cuda())
                assert close(r, d * (ad @ bdn.T), dtype), ("sddmm", ctx)
torch.cuda.synchronize()
print("fuzz_dense ok:", it, "iterations")
