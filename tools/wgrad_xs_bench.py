#!/usr/bin/env python3
"""wgrad of a 256 -> 256 3x3 layer at 64 x 64, N = 32: the prepared-operand route (xs_transpose + wgrad_bf16x3) against
ap_conv2d_wgrad_xs (both operands read as split copies).  Run under rocprofv3 --kernel-trace --stats for per-kernel times:
    APAMD_PRECISION=bf16x3|bf16 python tools/wgrad_xs_bench.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops

dev = torch.device('cuda:0')
n, c, m, h = 32, 256, 256, 64
prec = ops.DEFAULT_PRECISION
x = torch.randn(n, c, h, h, device=dev)
f = ops.Feat(x)
ops.presplit(f, prec)
g = ops.Feat(torch.randn(n, m, h, h, device=dev))
ops.presplit(g, prec)
for name, kw in (('prepared', {}), ('split copies', {'g_xs': g.xs})):
    for _ in range(3):
        ops.wgrad(3, 1, 1, ops.PAD_REFLECT, g, [f], (m, c, 3, 3), precision=prec, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.wgrad(3, 1, 1, ops.PAD_REFLECT, g, [f], (m, c, 3, 3), precision=prec, **kw)
    torch.cuda.synchronize()
    print('%-14s %.1f us per operator' % (name, (time.perf_counter() - t0) / 20 * 1e6))
