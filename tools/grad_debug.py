import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from animateportrait_amd import ops
from animateportrait_amd.autograd import Tape, conv_forward
from animateportrait_amd.networks import ConvLayer
dev = torch.device('cuda:0')
torch.manual_seed(0)
g = torch.Generator().manual_seed(3)

def rel(a, b):
    return float((a.cpu() - b).abs().max() / b.abs().max())

for (H, cin, cmid) in ((32, 16, 8), (128, 16, 8), (256, 16, 8)):
    # chain: x(plain) -> deconv(cin->cmid)+IN+ReLU -> reflect7x7(cmid->1)+tanh
    n = 2
    x = (torch.randn(n, cin, H // 2, H // 2, generator=g)).requires_grad_(True)
    w1 = (torch.randn(cin, cmid, 3, 3, generator=g) * 0.1).requires_grad_(True)
    w2 = (torch.randn(1, cmid, 7, 7, generator=g) * 0.05).requires_grad_(True)
    b2 = torch.zeros(1, requires_grad=True)
    y1 = F.conv_transpose2d(x, w1, None, stride=2, padding=1, output_padding=1); y1.retain_grad()
    a1 = F.relu(F.instance_norm(y1)); a1.retain_grad()
    out = torch.tanh(F.conv2d(F.pad(a1, (3,) * 4, mode='reflect'), w2, b2))
    up = torch.randn(out.shape, generator=g)
    (out * up).sum().backward()
    L1 = ConvLayer([cin], cmid, 3, 2, 1, ops.PAD_ZERO, True, 1).to(dev)
    L2 = ConvLayer([cmid], 1, 7, 1, 3, ops.PAD_REFLECT).to(dev)
    with torch.no_grad():
        L1.weight.copy_(w1); L2.weight.copy_(w2); L2.bias.zero_()
    tape = Tape()
    xin = tape.track(ops.Feat(x.detach().to(dev)))
    f1 = conv_forward(tape, L1, xin, norm_act=ops.ACT_RELU)
    f2 = conv_forward(tape, L2, f1, act=ops.ACT_TANH)
    print('H=%d fwd rel %.2e' % (H, rel(f2.data, out.detach())))
    tape.add(f2, up.to(dev), 0)
    # run the last layer's backward only, then inspect the gradient that reaches f1
    tape.steps[-1]()
    contribs = list(tape.grads[id(f1)])
    g1, p1, g2 = ops._split_contribs(contribs)
    ga = ops.fold_add(g1, p1, g2)
    print('   grad wrt a1 (dgrad 7x7 + fold): %.2e' % rel(ga, a1.grad))
    dy1 = ops.instnorm_bwd(contribs, f1)
    print('   grad wrt y1 (IN+ReLU bwd):      %.2e' % rel(dy1, y1.grad))
    tape.steps[0]()
    print('   dW2 %.2e  dW1 %.2e  dx %.2e' % (rel(tape.param_grads[L2.weight], w2.grad), rel(tape.param_grads[L1.weight], w1.grad),
          rel(tape.grads[id(xin)][0][0], x.grad)))
