"""group_reduce on keys that are NOT sorted group keys (what it sees when keys_lead_last declined its ranges and left its
output buffers unwritten): must not fault whatever the bits"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sparse_amd import _reduce as R
dev = torch.device("cuda:0")
n = 1094340
g = torch.Generator(device=dev).manual_seed(1)
kinds = {
    "random 63-bit": lambda: torch.randint(0, 2 ** 62, (n,), device=dev, generator=g, dtype=torch.int64),
    "random signed": lambda: torch.randint(-2 ** 62, 2 ** 62, (n,), device=dev, generator=g, dtype=torch.int64),
    "float bits": lambda: torch.rand(n, device=dev, generator=g, dtype=torch.float64).view(torch.int64),
    "small unsorted": lambda: torch.randint(0, 1155072, (n,), device=dev, generator=g, dtype=torch.int64),
    "descending": lambda: torch.arange(n, 0, -1, device=dev, dtype=torch.int64),
    "all ones bits": lambda: torch.full((n,), -1, device=dev, dtype=torch.int64),
    "int32 pairs": lambda: torch.randint(-2 ** 31, 2 ** 31, (2 * n,), device=dev, generator=g, dtype=torch.int32).view(torch.int64),
}
for name, mk in kinds.items():
    for op in ("multiply", "add", "maximum"):
        keys = mk()
        data = torch.randint(-9, 10, (n,), device=dev, generator=g, dtype=torch.int64)
        print(name, op, end=" ", flush=True)
        out = R.group_reduce(keys, 141, data, op, key_bound=1155072, sync=False)
        torch.cuda.synchronize()
        print("ok groups", int(out[3][0]), flush=True)
