#!/usr/bin/env python3
"""Batch (in)dependence of the backward pass of ONE network: gradients of a linear functional of the output at batch B against the
sum of the B single-sample backward passes.  Usage: python tools/batch_net_check.py [D|G] [B] [width] [precision]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops, networks as N
from animateportrait_amd.synthetic import make_generator_inputs, generator_args

which = sys.argv[1] if len(sys.argv) > 1 else 'D'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
width = int(sys.argv[3]) if len(sys.argv) > 3 else 64
precision = sys.argv[4] if len(sys.argv) > 4 else 'bf16x3'
ops.DEFAULT_PRECISION = {'bf16': ops.PRECISION_BF16, 'bf16x3': ops.PRECISION_BF16X3, 'fp32': ops.PRECISION_FP32}[precision]
dev = torch.device('cuda:0')
torch.manual_seed(5)
if which == 'D':
    net = N.define_D(2, width, 'basic', 3, 'instance', 'normal', 0.02, [0])
    x = torch.randn(B, 2, 256, 256, device=dev)
    ins = [x]
else:
    net = N.define_G(3, 1, width, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [0], div=3, disp=3)
    ins = [a.to(dev).contiguous() for a in generator_args(make_generator_inputs(B, seed=5))]
params = [p for p in net.parameters()]
names = [k for k, _ in net.named_parameters()]


def run(sl):
    xs = [a[sl].clone().requires_grad_(a.dtype.is_floating_point and i == 0) for i, a in enumerate(ins)]
    y = net(*xs)
    w = torch.randn(y.shape[1:], device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    loss = (y * w).sum()
    gs = torch.autograd.grad(loss, [xs[0]] + params, allow_unused=True)
    return [None if g is None else g.double() for g in gs]


big = run(slice(0, B))
acc = None
for i in range(B):
    g = run(slice(i, i + 1))
    if acc is None:
        acc = [None if t is None else (t.clone() if j else [t]) for j, t in enumerate(g)]
    else:
        for j, t in enumerate(g):
            if t is None:
                continue
            if j == 0:
                acc[0].append(t)
            else:
                acc[j] += t
acc[0] = torch.cat(acc[0], 0)
rows = []
for j, (a, b) in enumerate(zip(big, acc)):
    if a is None or float(b.abs().max()) == 0:
        continue
    rows.append((float((a - b).norm() / b.norm()), 'input' if j == 0 else names[j - 1], tuple(a.shape)))
if os.environ.get('ALLROWS'):
    for r in rows:
        print('   %-40s %.2e' % (r[1], r[0]))
rows.sort(reverse=True)
print('%s B=%d width %d %s: worst %s' % (which, B, width, precision, ', '.join('%s %.1e' % (r[1], r[0]) for r in rows[:8])))
