import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from animateportrait_amd import networks as N
from animateportrait_amd.synthetic import make_generator_inputs, generator_args
from oracle import generator as og
dev = torch.device('cuda:0')
z = np.load('tests/golden/gen_ngf8.npz')
d = make_generator_inputs(2, seed=1234)
sd = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234)
G = N.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [0], div=3, disp=3)
G.load_state_dict(sd, strict=True)
y = G(*[a.to(dev) for a in generator_args(d)])
up = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
(y * up.to(dev)).sum().backward()
grads = dict(G.named_parameters())
# oracle grads on CPU for every key
for v in sd.values(): v.requires_grad_(True)
yo = og.generator_forward(sd, *generator_args(d), div=3, disp=3)
(yo * up).sum().backward()
sd64 = {k: v.detach().double().requires_grad_(True) for k, v in sd.items()}
y64 = og.generator_forward(sd64, *[a.double() for a in generator_args(d)], div=3, disp=3)
(y64 * up.double()).sum().backward()
print('key                                 gpu-vs-cpu32   gpu-vs-fp64   cpu32-vs-fp64')
for k in sd:
    if k.endswith('weight'):
        t = sd64[k].grad
        print('%-34s %.2e   %.2e   %.2e' % (k, float((grads[k].grad.cpu() - sd[k].grad).abs().max() / t.abs().max()),
              float((grads[k].grad.cpu().double() - t).abs().max() / t.abs().max()),
              float((sd[k].grad.double() - t).abs().max() / t.abs().max())))
        continue
        g = grads[k].grad.cpu(); r = sd[k].grad
        print('%-34s rel-linf %.2e  (golden: %s)' % (k, float((g - r).abs().max() / r.abs().max()),
              ('%.2e' % float((g - torch.from_numpy(z['grad_' + k])).abs().max() / r.abs().max())) if 'grad_' + k in z.files else '-'))
