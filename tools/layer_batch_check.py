#!/usr/bin/env python3
"""Batch independence of ONE layer's backward (conv -> InstanceNorm -> ReLU) at full-width shapes: the gradients of the batch against
the sum / concatenation of the single-sample gradients, in the chosen arithmetic.   python tools/layer_batch_check.py [precision] [n]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops
from animateportrait_amd.autograd import Tape, conv_forward
from animateportrait_amd.networks import ConvLayer

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ops.DEFAULT_PRECISION = {'bf16': ops.PRECISION_BF16, 'bf16x3': ops.PRECISION_BF16X3, 'fp32': ops.PRECISION_FP32}[precision]
dev = torch.device('cuda:0')
CASES = [
    ('up 128->64 @128', [128], 64, 3, 2, 1, 'zero', True, 128),
    ('up 256->128 @64', [256], 128, 3, 2, 1, 'zero', True, 64),
    ('res 256->256 @64', [256], 256, 3, 1, 1, 'reflect', False, 64),
    ('res2 288->256 @64', [256, 16, 16], 256, 3, 1, 1, 'reflect', False, 64),
    ('down 64->128 s2 @256', [64], 128, 3, 2, 1, 'zero', False, 256),
    ('down 128->256 s2 @128', [128], 256, 3, 2, 1, 'zero', False, 128),
    ('merge 768->256 @64', [256, 256, 256], 256, 3, 1, 1, 'zero', False, 64),
    ('stem 3->64 k7 @256', [3], 64, 7, 1, 3, 'reflect', False, 256),
    ('D 64->128 k4s2 @128', [64], 128, 4, 2, 1, 'zero', False, 128),
    ('D 256->512 k4 @32', [256], 512, 4, 1, 1, 'zero', False, 32),
]
only = os.environ.get('ONLY')
for name, segs, cout, k, stride, pad, mode, tr, H in CASES:
    if only and only not in name:
        continue
    g = torch.Generator(device=dev).manual_seed(7)
    cin = sum(segs)
    xs = [torch.randn(N, c, H, H, device=dev, generator=g) * 1.5 + 0.3 for c in segs]
    layer = ConvLayer(segs, cout, k, stride, pad, ops.PAD_REFLECT if mode == 'reflect' else ops.PAD_ZERO, tr, 1 if tr else 0).to(dev)
    torch.nn.init.normal_(layer.weight, 0.0, 0.05)

    def run(sl, up=None):
        tape = Tape()
        feats = [tape.track(ops.Feat(x[sl].contiguous())) for x in xs]
        out = conv_forward(tape, layer, feats, norm_act=ops.ACT_RELU)
        if up is None:
            up = torch.randn(out.data.shape, device=dev, generator=g)
        tape.add(out, up[sl].contiguous() if up.shape[0] != out.data.shape[0] else up, 0)
        tape.backward()
        gxs = []
        for f in feats:
            g1, p1, g2 = ops._split_contribs(tape.take(f))
            gxs.append((ops.fold_add(g1, p1, g2) if (p1 or g2 is not None) else g1).double())
        return tape.param_grads[layer.weight].double().clone(), gxs, up, ops.materialize(out).data.double() if False else None

    gw, gx, up, _ = run(slice(0, N))
    sw, sx = None, [[] for _ in segs]
    for i in range(N):
        w1, x1, _, _ = run(slice(i, i + 1), up)
        sw = w1 if sw is None else sw + w1
        for j, t in enumerate(x1):
            sx[j].append(t)
    ew = float((gw - sw).norm() / sw.norm())
    ex = [float((a - torch.cat(b, 0)).norm() / torch.cat(b, 0).norm()) for a, b in zip(gx, sx)]
    print('%-24s n=%2d %s  wgrad vs sum of singles %.2e   dgrad %s' % (name, N, precision, ew, ' '.join('%.2e' % e for e in ex)), flush=True)
    del xs, up, gw, gx, sw, sx
    torch.cuda.empty_cache()
