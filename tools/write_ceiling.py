#!/usr/bin/env python3
"""Write-only and copy bandwidth of the device with plain torch kernels: context for a kernel whose bytes are 96 % writes."""
import torch
dev = torch.device("cuda:0")
n = 8 * (1 << 30) // 4
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)
t = timeit(lambda: a.fill_(1.5))
print(f"fill 8 GiB: {t:.3f} ms -> {n * 4 / t / 1e6:.0f} GB/s written")
t = timeit(lambda: a.zero_())
print(f"zero 8 GiB (memset): {t:.3f} ms -> {n * 4 / t / 1e6:.0f} GB/s written")
t = timeit(lambda: b.copy_(a))
print(f"copy 8 GiB: {t:.3f} ms -> {2 * n * 4 / t / 1e6:.0f} GB/s read+written")
t = timeit(lambda: a.sum())
print(f"sum 8 GiB: {t:.3f} ms -> {n * 4 / t / 1e6:.0f} GB/s read")
