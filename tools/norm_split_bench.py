#!/usr/bin/env python3
"""Micro-benchmark of the streaming passes between two convolutions (ap_norm_apply_split / ap_instnorm_apply) on the
generator's shapes.  Prints time and the algorithmic HBM rate (bytes that must move / time).
Usage: python tools/norm_split_bench.py [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device('cuda:0')
    for n, c, h in ((16, 256, 64), (16, 128, 128), (16, 64, 256)):
        x = torch.randn(n, c, h, h, device=dev)
        res = torch.randn(n, c, h, h, device=dev)
        m = torch.zeros(n * c, device=dev)
        r = torch.ones(n * c, device=dev)
        mb = x.numel() * 4 / 1e6

        def feat():
            return ops.Feat(x, m, r, ops.ACT_RELU)

        cases = [
            ('split only        (r 1, w 1)', 2, lambda: ops._norm_apply_split(feat(), None, False, True)),
            ('apply only        (r 1, w 1)', 2, lambda: ops._norm_apply_split(feat(), None, True, False)),
            ('apply+res         (r 2, w 1)', 3, lambda: ops._norm_apply_split(feat(), ops.Feat(res), True, False)),
            ('apply+res+split   (r 2, w 2)', 4, lambda: ops._norm_apply_split(feat(), ops.Feat(res), True, True)),
            ('torch copy        (r 1, w 1)', 2, lambda: res.copy_(x)),
        ]
        for name, mult, fn in cases:
            t = timeit(fn, iters)
            print('%dx%dx%dx%d  %-30s %7.1f us  %5.2f TB/s' % (n, c, h, h, name, t * 1e6, mult * mb * 1e6 / t / 1e12), flush=True)


if __name__ == '__main__':
    main()
