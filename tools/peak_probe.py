#!/usr/bin/env python3
"""Measured peaks of THIS box, to put next to the nominal ones ("measure both peaks on the box and report
against measured and nominal"): dense bf16 GEMM through the vendor library (torch.matmul -> hipBLASLt) and HBM copy / fill.
Not part of the product path."""
import torch


def timeit(f, it=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device='cuda', dtype=torch.bfloat16)
    b = torch.randn(n, n, device='cuda', dtype=torch.bfloat16)
    ms = timeit(lambda: torch.matmul(a, b))
    print('bf16 GEMM %5d^3 (hipBLASLt): %8.3f ms  %7.1f TFLOP/s' % (n, ms, 2.0 * n ** 3 / ms / 1e9))
x = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
y = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
ms = timeit(lambda: y.copy_(x))
print('HBM copy 1 GiB: %.3f ms  %.2f TB/s (read + write)' % (ms, 2.0 * (1 << 30) / ms / 1e9))
ms = timeit(lambda: y.zero_())
print('HBM fill 1 GiB: %.3f ms  %.2f TB/s' % (ms, (1 << 30) / ms / 1e9))
