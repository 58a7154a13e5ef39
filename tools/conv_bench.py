#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernel on the generator's shapes (B=16, ngf=64).
Usage: python tools/conv_bench.py [iters]   (env: APAMD_CONV_CI / APAMD_CONV_COTILE / APAMD_CONV_LDS_TARGET)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops
from animateportrait_amd.networks import ConvLayer

LAYERS = [
    # name, segs, cout, k, stride, pad, mode, transposed, H
    ('res 256->256 k3 @64', [256], 256, 3, 1, 1, ops.PAD_REFLECT, False, 64),
    ('res2 288->256 k3 @64', [256, 16, 16], 256, 3, 1, 1, ops.PAD_REFLECT, False, 64),
    ('merge 768->256 k3 @64', [256, 256, 256], 256, 3, 1, 1, ops.PAD_ZERO, False, 64),
    ('down 64->128 k3s2 @256', [64], 128, 3, 2, 1, ops.PAD_ZERO, False, 256),
    ('down 128->256 k3s2 @128', [128], 256, 3, 2, 1, ops.PAD_ZERO, False, 128),
    ('stem 3->64 k7 @256', [3], 64, 7, 1, 3, ops.PAD_REFLECT, False, 256),
    ('stem 3->32 k7 @256', [3], 32, 7, 1, 3, ops.PAD_REFLECT, False, 256),
    ('up 256->128 @64', [256], 128, 3, 2, 1, ops.PAD_ZERO, True, 64),
    ('up 128->64 @128', [128], 64, 3, 2, 1, ops.PAD_ZERO, True, 128),
    ('final 64->1 k7 @256', [64], 1, 7, 1, 3, ops.PAD_REFLECT, False, 256),
    ('D 256->512 k4 @32', [256], 512, 4, 1, 1, ops.PAD_ZERO, False, 32),
    ('D 128->256 k4s2 @64', [128], 256, 4, 2, 1, ops.PAD_ZERO, False, 64),
    ('res 256->256 k3 @64 (again, clocks warm)', [256], 256, 3, 1, 1, ops.PAD_REFLECT, False, 64),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device('cuda:0')
    n = 16
    for name, segs, cout, k, stride, pad, mode, tr, h in LAYERS:
        if only and only not in name:
            continue
        layer = ConvLayer(segs, cout, k, stride, pad, mode, tr, 1 if tr else 0).to(dev)
        torch.nn.init.normal_(layer.weight, 0, 0.02)
        srcs = []
        for c in segs:
            x = torch.randn(n, c, h, h, device=dev)
            m = torch.zeros(n * c, device=dev)
            r = torch.ones(n * c, device=dev)
            srcs.append(ops.Feat(x, m, r, ops.ACT_RELU))
        kw = dict(act=ops.ACT_NONE) if os.environ.get('APAMD_BENCH_NOSTATS') else dict(norm_act=ops.ACT_RELU)
        for _ in range(2):
            y = layer.run(srcs, **kw)
        prof = ops.LaunchProfiler()
        ops.PROFILER = prof
        for _ in range(iters):
            y = layer.run(srcs, **kw)
        ops.PROFILER = None
        agg = prof.summary()
        for kn, v in agg.items():
            print('%-26s %-30s %8.1f us  %6.1f TFLOP/s' % (name, kn, v['ms'] * 1e3 / iters, v['flops'] / (v['ms'] * 1e-3) / 1e12),
                  flush=True)


if __name__ == '__main__':
    main()
