"""Indexing, concatenate / stack, triangles, diagonal, nonzero (N2: _coo/indexing.py:12-133, _compressed/indexing.py:14-176,
_coo/common.py:132-249) at 10^7 stored elements: ms per call.  Looking for calls that cost a multiple of their neighbours."""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from bench import dev_time

NNZ = 10_000_000
for shape in [(100_000, 10_000), (1000, 1000, 1000)]:
    size = int(np.prod(shape))
    g = torch.Generator(device="cuda").manual_seed(1)
    lin = torch.unique(torch.randint(0, size, (NNZ,), device="cuda", generator=g))
    vals = torch.rand(lin.numel(), device="cuda", dtype=torch.float64) + 0.1
    c = sp.COO._from_sorted_keys(lin, vals, shape, 0.0, torch.int64)
    c.coords
    gx = c.asformat("gcxs")
    fancy = np.arange(0, shape[0], 7)
    for label, x in (("coo", c), ("gcxs", gx)):
        ops = {
            "x[5]": lambda: x[5],
            "x[:, 5]": lambda: x[:, 5],
            "x[10:500]": lambda: x[10:500],
            "x[:, 10:500]": lambda: x[:, 10:500],
            "x[::2]": lambda: x[::2],
            "x[:, ::2]": lambda: x[:, ::2],
            "x[::-1]": lambda: x[::-1],
            "x[fancy]": lambda: x[fancy],
            "x[None]": lambda: x[None],
            "x[..., -1]": lambda: x[..., -1],
            "concat0": lambda: sp.concatenate([x, x], axis=0),
            "concat-1": lambda: sp.concatenate([x, x], axis=-1),
            "stack0": lambda: sp.stack([x, x], axis=0),
            "stack-1": lambda: sp.stack([x, x], axis=-1),
            "nonzero": lambda: x.nonzero(),
        }
        if x.ndim == 2:
            ops.update({"tril": lambda: sp.tril(x), "triu": lambda: sp.triu(x, 3), "diagonal": lambda: sp.diagonal(x)})
        row = [str(shape), label]
        for name, f in ops.items():
            try:
                f()
                f()
                row.append(f"{name} {dev_time(f, 3):.2f}")
            except Exception as e:
                row.append(f"{name} {type(e).__name__}:{str(e)[:30]}")
        print(" | ".join(row), flush=True)
    del c, gx, lin, vals
