"""Achievable HBM read / copy bandwidth on this box (torch kernels, 2 GiB buffers): the ceiling the sweep kernels
are measured against in DESIGN.md."""
import torch, time
n = 1 << 29
x = torch.ones(n, dtype=torch.float32, device="cuda:0")
y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
ts = t(lambda: x.sum())
tc = t(lambda: y.copy_(x))
tm = t(lambda: torch.mul(x, 2.0, out=y))
print(f"read (sum)  {n*4/ts/1e12:.2f} TB/s   copy {2*n*4/tc/1e12:.2f} TB/s (r+w)   scale {2*n*4/tm/1e12:.2f} TB/s (r+w)")
for m in (1 << 26, 1 << 27):
    xs = x[:m]
    ts = t(lambda: xs.sum(), 30)
    print(f"read {m*4/2**20:.0f} MiB: {m*4/ts/1e12:.2f} TB/s")
