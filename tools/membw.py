#!/usr/bin/env python3
"""HBM streaming rates as torch sees them (fill = pure write, sum = pure read, copy = read + write), at the tensor
sizes the generator moves between convolutions.  Usage: python tools/membw.py"""
import torch

def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (67, 268, 1072):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device='cuda')
    y = torch.empty_like(x)
    print('%5d MB  fill %6.2f TB/s   sum %6.2f TB/s   copy %6.2f TB/s (r+w bytes)' % (
        mb, n * 4 / t(lambda: y.fill_(1.0)) / 1e12, n * 4 / t(lambda: x.sum()) / 1e12,
        2 * n * 4 / t(lambda: y.copy_(x)) / 1e12))
