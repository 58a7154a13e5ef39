#!/usr/bin/env python3
"""Forward batch independence of the generator with and without a tape: output at batch B vs the single-sample outputs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops, networks as N
from animateportrait_amd.synthetic import make_generator_inputs, generator_args
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16x3'
ops.DEFAULT_PRECISION = {'bf16': ops.PRECISION_BF16, 'bf16x3': ops.PRECISION_BF16X3, 'fp32': ops.PRECISION_FP32}[precision]
dev = torch.device('cuda:0')
torch.manual_seed(5)
net = N.define_G(3, 1, 64, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [0], div=3, disp=3)
ins = [a.to(dev).contiguous() for a in generator_args(make_generator_inputs(B, seed=5))]
for grad in (False, True):
    with torch.set_grad_enabled(grad):
        xs = [a.clone().requires_grad_(grad and i == 0) for i, a in enumerate(ins)]
        y = net(*xs).detach()
        ys = torch.cat([net(*[a[i:i + 1].clone().requires_grad_(grad and j == 0) for j, a in enumerate(ins)]).detach() for i in range(B)], 0)
    print('B=%d %s grad=%s: output L-inf %.3e  rel L2 %.3e' % (B, precision, grad, float((y - ys).abs().max()), float((y - ys).norm() / ys.norm())))
