#!/usr/bin/env python3
"""Times ap_act_bwd's fold / add forms on the residual-stream shape of the train step (32 x 256 x 64 x 64)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animateportrait_amd import ops      # noqa: E402

dev = torch.device('cuda:0')
n, c, h, w = 32, 256, 64, 64
g1p = torch.randn(n, c, h + 2, w + 2, device=dev)
g1 = torch.randn(n, c, h, w, device=dev)
g2 = torch.randn(n, c, h, w, device=dev)
junk = torch.empty(512 * 1024 * 1024 // 4, device=dev)     # flushes the memory-side cache between repetitions


def run(name, fn, nbytes):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(5):
        junk.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print('%-28s %7.1f us  %.2f TB/s (cold)' % (name, t, nbytes / t / 1e6))


plane = n * c * h * w * 4
run('fold(pad 1) + g2', lambda: ops.fold_add(g1p, 1, g2), plane * 3 + (g1p.numel() * 4 - plane))
run('fold(pad 1)', lambda: ops.fold_add(g1p, 1, None), plane * 2 + (g1p.numel() * 4 - plane))
run('g1 + g2 (16-byte lanes)', lambda: ops.fold_add(g1, 0, g2), plane * 3)
