#!/usr/bin/env python3
"""Device-to-device copy, fill and read-reduce bandwidth on GPU 0 (GB/s), as a yardstick for the bandwidth kernels."""
import time
import torch
n = 1 << 28                                       # 1 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, bytes_moved, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return bytes_moved * reps / (time.perf_counter() - t0) / 1e9
print("fill  %.0f GB/s" % t(lambda: a.fill_(1.0), 4 * n))
print("copy  %.0f GB/s (read + write)" % t(lambda: b.copy_(a), 8 * n))
print("sum   %.0f GB/s" % t(lambda: a.sum(), 4 * n))
