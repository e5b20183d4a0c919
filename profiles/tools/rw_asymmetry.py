"""Measure write-only, read-only and copy bandwidth on the GPU box (torch kernels, HIP events): how asymmetric is HBM here?"""
import torch

dev = torch.device("cuda:0")
n = 1 << 28  # 1 GiB of fp32
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


t = timed(lambda: x.fill_(1.0))
print(f"fill  (write only): {4 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: y.copy_(x))
print(f"copy  (read+write): {8 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: x.sum())
print(f"sum   (read only) : {4 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: torch.add(x, 1.0, out=y))
print(f"add   (read+write): {8 * n / t / 1e12:.2f} TB/s")
