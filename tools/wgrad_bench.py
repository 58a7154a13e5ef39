#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient kernel on the generator's shapes (B=32 = one G step at bs=16).
Usage: python tools/wgrad_bench.py [iters] [name filter]   (env: APAMD_ABLATE, APAMD_WGRAD_BLOCKS)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops

LAYERS = [
    # name, cin segs, cout, k, stride, pad, mode, H (input)
    ('res 256->256 k3 @64', [256], 256, 3, 1, 1, ops.PAD_REFLECT, 64),
    ('merge 768->256 k3 @64', [256, 256, 256], 256, 3, 1, 1, ops.PAD_ZERO, 64),
    ('down 64->128 k3s2 @256', [64], 128, 3, 2, 1, ops.PAD_ZERO, 256),
    ('stem 3->64 k7 @256', [3], 64, 7, 1, 3, ops.PAD_REFLECT, 256),
    ('final 64->1 k7 @256', [64], 1, 7, 1, 3, ops.PAD_REFLECT, 256),
    ('D 256->512 k4 @32', [256], 512, 4, 1, 1, ops.PAD_ZERO, 32),
    ('D 1->64 k4s2 @256', [1], 64, 4, 2, 1, ops.PAD_ZERO, 256),
    ('D 2->64 k4s2 @256', [2], 64, 4, 2, 1, ops.PAD_ZERO, 256),
    ('D 64->128 k4s2 @128', [64], 128, 4, 2, 1, ops.PAD_ZERO, 128),
    ('stem 3->32 k7 @256', [3], 32, 7, 1, 3, ops.PAD_REFLECT, 256),
    ('land 1->8 k3 @256 (x2 batch)', [1], 8, 3, 1, 1, ops.PAD_ZERO, 256),
    ('land 8->16 k3s2 @256', [8], 16, 3, 2, 1, ops.PAD_ZERO, 256),
    ('land 16->16 k3s2 @128', [16], 16, 3, 2, 1, ops.PAD_ZERO, 128),
    ('down 128->256 k3s2 @128', [128], 256, 3, 2, 1, ops.PAD_ZERO, 128),
    ('down 64->64 k3s2 @256', [64], 64, 3, 2, 1, ops.PAD_ZERO, 256),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device('cuda:0')
    n = 32
    for name, segs, cout, k, stride, pad, mode, h in LAYERS:
        if only and only not in name:
            continue
        ho = (h + 2 * pad - k) // stride + 1
        g = ops.Feat(torch.randn(n, cout, ho, ho, device=dev))
        srcs = []
        for c in segs:
            x = torch.randn(n, c, h, h, device=dev)
            srcs.append(ops.Feat(x, torch.zeros(n * c, device=dev), torch.ones(n * c, device=dev), ops.ACT_RELU))
        shape = (cout, sum(segs), k, k)
        for _ in range(2):
            ops.wgrad(k, stride, pad, mode, g, srcs, shape)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.wgrad(k, stride, pad, mode, g, srcs, shape)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flops = 2.0 * n * ho * ho * cout * sum(segs) * k * k
        print('%-26s %8.1f us  %6.1f TFLOP/s' % (name, ms * 1e3, flops / (ms * 1e-3) / 1e12), flush=True)


if __name__ == '__main__':
    main()
