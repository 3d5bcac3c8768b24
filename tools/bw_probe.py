#!/usr/bin/env python3
"""HBM read / write / copy bandwidth of plain torch kernels on this device (context for the GEMM epilogue numbers)."""
import torch
n = 1 << 30
x = torch.empty(n // 2, dtype=torch.float16, device="cuda")
y = torch.empty_like(x)
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3
print(f"fill  (write 1 GiB): {n / t(lambda: x.fill_(1.0)) / 1e12:.2f} TB/s")
print(f"copy  (read+write 2 GiB): {2 * n / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s")
print(f"sum   (read 1 GiB): {n / t(lambda: x.sum()) / 1e12:.2f} TB/s")
