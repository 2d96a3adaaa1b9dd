"""Conversions and layout changes (N1: core.py:1294-1371, 1034-1111, 725-807; compressed.py:25-77, 388-460; convert.py) at
10^7 stored elements, by shape: ms per call.  Looking for calls that cost a multiple of their neighbours."""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from bench import dev_time

NNZ = 10_000_000
SHAPES = [(100_000, 10_000), (10_000, 100_000), (1_000_000, 1000), (1000, 1_000_000), (1000, 1000, 1000), (100, 100, 100, 1000),
          (10_000_000, 100), (31_623, 31_623)]
for shape in SHAPES:
    size = int(np.prod(shape))
    g = torch.Generator(device="cuda").manual_seed(1)
    lin = torch.unique(torch.randint(0, size, (NNZ,), device="cuda", generator=g))
    vals = torch.rand(lin.numel(), device="cuda", dtype=torch.float64) + 0.1
    c = sp.COO._from_sorted_keys(lin, vals, shape, 0.0, torch.int64)
    c.coords          # (materialised once, as a COO built by the user has them)
    ops = {
        "tocoo->gcxs": lambda: c.asformat("gcxs"),
        "T": lambda: c.T.linear_loc(),
        "transpose(rev)": lambda: c.transpose(tuple(reversed(range(c.ndim)))).linear_loc(),
        "reshape(-1)": lambda: c.reshape((size,)).linear_loc(),
        "reshape(2d)": lambda: c.reshape((shape[0], size // shape[0])).linear_loc(),
        "astype(f32)": lambda: c.astype(np.float32),
        "copy": lambda: c.copy(),
        "neg": lambda: -c,
    }
    gx = c.asformat("gcxs")
    ops.update({
        "gcxs->coo": lambda: gx.tocoo().linear_loc(),
        "gcxs.T": lambda: gx.T,
        "change_ca": lambda: gx.change_compressed_axes((gx.ndim - 1,)),
        "gcxs.reshape(2d)": lambda: gx.reshape((shape[0], size // shape[0])),
        "gcxs.astype": lambda: gx.astype(np.float32),
    })
    if size <= 2_000_000_000:
        ops["todense"] = lambda: c.todense_device()
        ops["gcxs.todense"] = lambda: gx.todense_device()
    row = [str(shape)]
    for name, f in ops.items():
        try:
            f()
            f()
            row.append(f"{name} {dev_time(f, 5):.2f}")
        except Exception as e:
            row.append(f"{name} {type(e).__name__}:{str(e)[:40]}")
    print(" | ".join(row), flush=True)
    del c, gx, lin, vals
