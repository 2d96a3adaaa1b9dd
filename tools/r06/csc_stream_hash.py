"""The CSC inspector's outputs as hashes (one line per case): run once per library (SPAMD_LIB=...) and diff the lines - a
change of the inspector's kernels must leave the block stream and blk_off the same bytes.  `--time`: ms per layout at
config 2's size (f32/int32, f64/int64), 10 launches, HIP events."""
import hashlib
import sys

sys.path.insert(0, "/root/repo")
import torch

from bench import make_csr_device, dev_time
from sparse_amd import _kernels as K

CASES = [(3000, 700, 0.02), (70_001, 1500, 0.01), (560 * 7 + 13, 10_000, 0.003), (1121, 321, 0.3), (559, 161, 0.5),
         (40_000, 40_960, 0.0005), (1_400_000, 500, 0.002), (1_000_000, 10_000, 0.001)]


def digest(lay, groups, ntiles):
    blocks, blk_off = lay[0], lay[1]
    end = int(blk_off.view(groups, ntiles + 1)[-1, -1])
    h = hashlib.sha256()
    h.update(blk_off.cpu().numpy().tobytes())
    h.update(blocks[: end * 16].cpu().numpy().tobytes())
    return h.hexdigest()[:16], end


for M, Kd, dens in CASES:
    for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
        d, i, p = make_csr_device(M, Kd, dens, seed=M % 97, dtype=dt)
        cd, ci, cp = K.csx_swap_2d(d, i.to(it), p.to(it), M, Kd)
        rg, kb, gpb, epb, slack, direct_max, _ = K.tiled_params(dt)
        ntiles = -(-Kd // kb)
        groups = -(-(-(-M // rg)) // gpb) * gpb
        out = []
        for rep in range(2):
            lay = K.csc_tiled_layout(cd, ci.to(it), cp.to(it), M, Kd, dtype=dt)
            out.append(digest(lay, groups, ntiles))
        print(M, Kd, dens, str(dt)[6:], str(it)[6:], out[0][0], out[0][1], "repeat-same" if out[0] == out[1] else "REPEAT-DIFFERS", flush=True)
        del lay, d, i, p, cd, ci, cp

if "--time" in sys.argv:
    M, Kd = 1_000_000, 10_000
    d, i, p = make_csr_device(M, Kd, 0.01, 1234)
    for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
        cd, ci, cp = K.csx_swap_2d(d.to(dt), i.to(it), p.to(it), M, Kd)
        ci, cp = ci.to(it), cp.to(it)
        for _ in range(3):
            lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
        ms = dev_time(lambda: K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt), 10)
        print("time", str(dt)[6:], str(it)[6:], f"{ms:.3f} ms", flush=True)
