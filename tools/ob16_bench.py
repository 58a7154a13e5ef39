#!/usr/bin/env python3
"""A/B of the trunk 3x3 layer (256 -> 256 @ 64 x 64, 2B = 32, plain bf16 arithmetic) with fp32 and with bf16 raw output.
    python tools/ob16_bench.py [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops
from animateportrait_amd.networks import ConvLayer

ops.DEFAULT_PRECISION = ops.PRECISION_BF16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device('cuda:0')
n = 32
for cin in ([256], [256, 16, 16]):
    layer = ConvLayer(cin, 256, 3, 1, 1, ops.PAD_REFLECT).to(dev)
    torch.nn.init.normal_(layer.weight, 0, 0.02)
    srcs = [ops.Feat(torch.randn(n, c, 64, 64, device=dev), torch.zeros(n * c, device=dev), torch.ones(n * c, device=dev), ops.ACT_RELU) for c in cin]
    for ob in (False, True, False, True):
        for _ in range(3):
            y = layer.run(srcs, norm_act=ops.ACT_RELU, out_bf16=ob)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            y = layer.run(srcs, norm_act=ops.ACT_RELU, out_bf16=ob)
        e1.record()
        torch.cuda.synchronize()
        print('cin %-14s out_bf16=%-5s %7.1f us per layer call (dtype %s)' % (cin, ob, e0.elapsed_time(e1) / iters * 1e3, y.data.dtype), flush=True)
