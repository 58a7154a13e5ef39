#!/usr/bin/env python3
"""Batch independence of the generator's decoder (model3: two up-convolutions + the final 7x7 + tanh) backward."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops
from animateportrait_amd.autograd import Tape, conv_forward as cf
from animateportrait_amd.networks import ConvLayer

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # 0: whole decoder; 1: from the second up-convolution; 2: final layer only
ops.DEFAULT_PRECISION = {'bf16': ops.PRECISION_BF16, 'bf16x3': ops.PRECISION_BF16X3, 'fp32': ops.PRECISION_FP32}[precision]
dev = torch.device('cuda:0')
torch.manual_seed(3)
L = [ConvLayer([256], 128, 3, 2, 1, ops.PAD_ZERO, True, 1), ConvLayer([128], 64, 3, 2, 1, ops.PAD_ZERO, True, 1),
     ConvLayer([64], 1, 7, 1, 3, ops.PAD_REFLECT)]
for l in L:
    l.to(dev)
    torch.nn.init.normal_(l.weight, 0.0, 0.05)
cin, H = [(256, 64), (128, 128), (64, 256)][first]
x = torch.randn(N, cin, H, H, device=dev)
up = torch.randn(N, 1, 256, 256, device=dev)


def run(sl):
    tape = Tape()
    f = tape.track(ops.Feat(x[sl].contiguous()))
    y = f
    if first == 2:      # the final layer reads a virtual feature in the network: emulate with IN statistics of x itself
        y = f
    for i in range(first, 2):
        y = cf(tape, L[i], y, norm_act=ops.ACT_RELU)
    out = cf(tape, L[2], y, act=ops.ACT_TANH)
    tape.add(out, up[sl].contiguous(), 0)
    tape.backward()
    g1, p1, g2 = ops._split_contribs(tape.take(f))
    gx = (ops.fold_add(g1, p1, g2) if (p1 or g2 is not None) else g1).double()
    return [tape.param_grads[l.weight].double().clone() for l in L[first:]], gx


gw, gx = run(slice(0, N))
sw, sx = None, []
for i in range(N):
    w1, x1 = run(slice(i, i + 1))
    sw = w1 if sw is None else [a + b for a, b in zip(sw, w1)]
    sx.append(x1)
print('decoder from layer %d, n=%d %s: wgrad %s  dgrad %.2e' % (first, N, precision, ' '.join('%.2e' % float((a - b).norm() / b.norm()) for a, b in zip(gw, sw)),
                                                              float((gx - torch.cat(sx, 0)).norm() / torch.cat(sx, 0).norm())))
