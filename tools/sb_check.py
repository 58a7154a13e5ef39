#!/usr/bin/env python3
"""conv_bf16x3_sb (one LDS stage per workgroup, two workgroups per CU; APAMD_CONV_SB=1) against conv_bf16x3 on the generator's
3x3 stride-1 layers: bitwise comparison of the outputs and InstanceNorm statistics, then per-launch times for a few start skews.
Usage: python tools/sb_check.py [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops
from animateportrait_amd.networks import ConvLayer

LAYERS = [
    ('res 256->256 @64 B16', [256], 256, ops.PAD_REFLECT, 64, 64, 16),
    ('res2 288->256 @64 B16', [256, 16, 16], 256, ops.PAD_REFLECT, 64, 64, 16),
    ('merge 768->256 @64 B16', [256, 256, 256], 256, ops.PAD_ZERO, 64, 64, 16),
    ('ragged 272->200 40x52 B5', [256, 16], 200, ops.PAD_REFLECT, 40, 52, 5),
]


def run(layer, srcs, sb, skew=0):
    os.environ['APAMD_CONV_SB'] = str(sb)
    os.environ['APAMD_CONV_SB_SKEW'] = str(skew)
    return layer.run(srcs, norm_act=ops.ACT_RELU)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for name, segs, cout, mode, h, w, n in LAYERS:
        layer = ConvLayer(segs, cout, 3, 1, 1, mode, False, 0).to(dev)
        torch.nn.init.normal_(layer.weight, 0, 0.02)
        srcs = []
        for c in segs:
            x = torch.randn(n, c, h, w, device=dev)
            srcs.append(ops.Feat(x, torch.zeros(n * c, device=dev), torch.ones(n * c, device=dev), ops.ACT_RELU))
        a = run(layer, srcs, 0)
        b = run(layer, srcs, 1)
        a2 = run(layer, srcs, 0)
        print('%-28s bitwise identical: output %s, mean %s (max diff %.3g), rstd %s (%.3g); conv_bf16x3 twice: %s' % (
            name, torch.equal(a.data, b.data), torch.equal(a.mean, b.mean), float((a.mean - b.mean).abs().max()),
            torch.equal(a.rstd, b.rstd), float((a.rstd - b.rstd).abs().max()),
            torch.equal(a.data, a2.data) and torch.equal(a.mean, a2.mean)), flush=True)
        for sb, skew in ((0, 0), (1, 0), (1, 4), (1, 8), (1, 16), (0, 0), (1, 0), (1, 8)):
            for _ in range(3):
                run(layer, srcs, sb, skew)
            prof = ops.LaunchProfiler()
            ops.PROFILER = prof
            for _ in range(iters):
                run(layer, srcs, sb, skew)
            ops.PROFILER = None
            for kn, v in prof.summary().items():
                if True:
                    print('    sb=%d skew=%-4d %-34s %8.1f us' % (sb, skew, kn[:34], v['ms'] * 1e3 / iters), flush=True)
    os.environ['APAMD_CONV_SB'] = '0'


if __name__ == '__main__':
    main()
