#!/usr/bin/env python3
"""Weight gradients of the 7x7 edge layers at the train step's shapes (2B = 32, 256 x 256), plain bf16: the matrix-pipe route
(csrc/wgrad_k7.h) against the fp32 / vector-ALU kernels it replaces.   python tools/k7_bench.py [iters] [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
H = W = 256
for name, final_form, mw, cn in (('stem 3->64', 0, 64, 3), ('stem 3->32', 0, 32, 3), ('final 64->1', 1, 64, 1)):
    if final_form:
        x = torch.randn(n, mw, H, W, device=dev)
        src = ops.Feat(x, torch.zeros(n * mw, device=dev), torch.ones(n * mw, device=dev), ops.ACT_RELU)
        g = ops.Feat(torch.randn(n, 1, H, W, device=dev))
        shape = (1, mw, 7, 7)
    else:
        src = ops.Feat(torch.randn(n, cn, H, W, device=dev))
        g = ops.Feat(torch.randn(n, mw, H, W, device=dev))
        shape = (mw, cn, 7, 7)
    res = {}
    for route in (True, False, True, False):
        ops.K7_WGRAD = route
        for _ in range(3):
            dw = ops.wgrad(7, 1, 3, ops.PAD_REFLECT, g, [src], shape, precision=ops.PRECISION_BF16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dw = ops.wgrad(7, 1, 3, ops.PAD_REFLECT, g, [src], shape, precision=ops.PRECISION_BF16)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(route, dw)
        print('%-12s %-28s %8.1f us per operator' % (name, 'wgrad_k7 (bf16 matrix pipe)' if route else 'fp32 / vector-ALU kernels',
                                                     e0.elapsed_time(e1) / iters * 1e3), flush=True)
    a, b = res[True], res[False]
    print('%-12s relative L-inf between the routes: %.2e' % (name, float((a - b).abs().max() / b.abs().max())), flush=True)

# ---- the last layer's data gradient (padded coordinates): dgrad_k7.h against the fp32 matrix kernel
from animateportrait_amd import autograd
from animateportrait_amd.networks import ConvLayer
ops.DEFAULT_PRECISION = ops.PRECISION_BF16
layer = ConvLayer([64], 1, 7, 1, 3, ops.PAD_REFLECT).to(dev)
layer.spec.precision = ops.PRECISION_BF16
torch.nn.init.normal_(layer.weight, 0, 0.02)
spec, fold = autograd._dgrad_spec(layer, 64)
packed = layer.packed_dgrad(0, spec, layer.weight.detach())
g = ops.Feat(torch.randn(n, 1, H, W, device=dev))
outs = {}
for route in (True, False, True, False):
    fn = (lambda: ops.final_dgrad_k7(g, layer.weight)) if route else (lambda: ops.conv2d(spec, [g], packed, None).data)
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    outs[route] = y
    print('final dgrad  %-28s %8.1f us per operator' % ('dgrad_k7 (bf16 matrix pipe)' if route else 'fp32 matrix kernel', e0.elapsed_time(e1) / iters * 1e3), flush=True)
print('final dgrad  relative L-inf between the routes: %.2e' % float((outs[True] - outs[False]).abs().max() / outs[False].abs().max()))
