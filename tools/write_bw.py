import torch
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tf = t(lambda: x.fill_(1.0)); print(f"fill 1 GiB: {tf:.3f} ms = {x.numel()*4/tf/1e9:.2f} TB/s")
tz = t(lambda: x.zero_()); print(f"zero_ 1 GiB: {tz:.3f} ms = {x.numel()*4/tz/1e9:.2f} TB/s")
tc = t(lambda: y.copy_(x)); print(f"copy 1 GiB: {tc:.3f} ms = {2*x.numel()*4/tc/1e9:.2f} TB/s (read+write)")
ts = t(lambda: x.sum()); print(f"sum (read) 1 GiB: {ts:.3f} ms = {x.numel()*4/ts/1e9:.2f} TB/s")
