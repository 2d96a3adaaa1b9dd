"""Random operands with hub rows through the two-part product (`_dot._hot_row_split`, csrc/hot_rows.hip) against the same
product with the split switched off: integers exact, floats to re-association (1e-11 / 2e-4 of sum |terms|).  Hub rows of
4096 .. K elements (one to many, adjacent ones, the first and the last row, a full row), CSR / CSC / COO operands, four value
types, int32 / int64 indices, 1 .. 200 columns, repeated products on the memoised split, `dense @ sparse`.
    python tools/fuzz_hub.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from sparse_amd import _dot as D

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
cases = fails = splits = 0
t_end = time.time() + budget
while time.time() < t_end:
    M = int(rng.choice([1, 2, 50, 1000, 40_000, 300_000]))
    Kd = int(rng.choice([4096, 5000, 20_000, 70_000, 300_000]))
    per = float(rng.choice([0.0, 0.5, 3, 20]))
    nb = int(min(M * per, 3_000_000))
    base = torch.randint(0, M * Kd, (nb,), device="cuda", generator=g)
    nh = int(rng.choice([0, 1, 1, 2, 5, 40]))
    hub_rows = rng.choice(M, size=min(nh, M), replace=False)
    if len(hub_rows) and rng.random() < 0.3:
        hub_rows[0] = 0
    if len(hub_rows) > 1 and rng.random() < 0.3:
        hub_rows[1] = M - 1
    if len(hub_rows) > 2 and rng.random() < 0.3 and hub_rows[0] + 1 < M:
        hub_rows[2] = hub_rows[0] + 1
    extra = []
    for r in set(int(x) for x in hub_rows):
        ln = Kd if rng.random() < 0.2 else int(rng.integers(min(4000, Kd), min(Kd, 200_000) + 1))
        extra.append(torch.randperm(Kd, device="cuda", generator=g)[:ln] + r * Kd)
    lin = torch.unique(torch.cat([base] + extra)) if (nb or extra) else torch.empty(0, dtype=torch.int64, device="cuda")
    if lin.numel() == 0:
        continue
    dt = [np.float32, np.float64, np.int32, np.int64][int(rng.integers(0, 4))]
    tdt = {np.float32: torch.float32, np.float64: torch.float64, np.int32: torch.int32, np.int64: torch.int64}[dt]
    vals = (torch.randint(-4, 5, (lin.numel(),), device="cuda", generator=g) if np.dtype(dt).kind == "i"
            else torch.rand(lin.numel(), device="cuda", generator=g, dtype=torch.float64) - 0.5).to(tdt)
    idt = torch.int32 if rng.random() < 0.5 else torch.int64
    c = sp.COO._from_sorted_keys(lin, vals, (M, Kd), np.dtype(dt).type(0), idt)
    form = str(rng.choice(["csr", "csc", "coo", "rhs"]))
    n = int(rng.choice([1, 2, 4, 5, 8, 12, 13, 16, 64, 128, 200]))
    if form == "rhs":          # dense @ sparse: the hub rows of the operand become hub columns of c.T
        a = sp.GCXS(c, compressed_axes=(0,)).T
        b = (torch.randint(-3, 4, (n, Kd), device="cuda", generator=g) if np.dtype(dt).kind == "i"
             else torch.rand(n, Kd, device="cuda", generator=g, dtype=torch.float64) - 0.5).to(tdt)
        f = lambda: b @ a
    else:
        a = c if form == "coo" else sp.GCXS(c, compressed_axes=(0,) if form == "csr" else (1,))
        b = (torch.randint(-3, 4, (Kd, n), device="cuda", generator=g) if np.dtype(dt).kind == "i"
             else torch.rand(Kd, n, device="cuda", generator=g, dtype=torch.float64) - 0.5).to(tdt)
        f = lambda: a @ b
    try:
        got = [f(), f()]
        target = a if form != "rhs" else a.__dict__.get("_t_view", a)
        splits += 1 if getattr(target, "__dict__", {}).get("_hot_split") is not None else 0
        D.HOT_ROW_SPLIT = False
        for o in (a, target):
            o.__dict__.pop("_hot_split", None)
        ref = f()
        D.HOT_ROW_SPLIT = True
        ok = True
        for r in got:
            r, q = torch.as_tensor(r), torch.as_tensor(ref)
            if np.dtype(dt).kind == "i":
                ok = ok and torch.equal(r, q)
            else:
                scale = float(q.abs().max()) + 1.0
                ok = ok and r.shape == q.shape and float((r - q).abs().max()) <= (1e-11 if dt == np.float64 else 2e-4) * scale * 50
    except Exception as e:        # noqa: BLE001 - a fuzzer reports and goes on
        D.HOT_ROW_SPLIT = True
        ok = False
        print("EXCEPTION", type(e).__name__, str(e)[:200], flush=True)
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", M, Kd, per, nh, np.dtype(dt).name, idt, form, n, flush=True)
        if fails > 5:
            break
    del c, a, b, lin, vals
print(f"fuzz_hub: {cases} cases ({splits} split), {fails} mismatches (seed {seed})")
sys.exit(1 if fails else 0)
