"""Oracle (test infrastructure): 70x70 PatchGAN discriminator.

Functional restatement of ``NLayerDiscriminator(input_nc, ndf, n_layers=3,
norm=instance)`` (Module2/models/networks.py:2602-2647) over a dict with the
reference ``state_dict`` keys (``model.0/2/5/8/11``).
"""
import torch
import torch.nn.functional as F

from .generator import inorm


def patchgan_forward(sd, x):
    """networks.py:2620-2645: 4x4 s2 (bias)+LReLU; 2x(4x4 s2 +IN+LReLU);
    4x4 s1 +IN+LReLU; 4x4 s1 (bias) -> (B,1,30,30) for 256x256 input."""
    x = F.leaky_relu(F.conv2d(x, sd['model.0.weight'], sd['model.0.bias'], stride=2, padding=1), 0.2)
    x = F.leaky_relu(inorm(F.conv2d(x, sd['model.2.weight'], sd['model.2.bias'], stride=2, padding=1)), 0.2)
    x = F.leaky_relu(inorm(F.conv2d(x, sd['model.5.weight'], sd['model.5.bias'], stride=2, padding=1)), 0.2)
    x = F.leaky_relu(inorm(F.conv2d(x, sd['model.8.weight'], sd['model.8.bias'], stride=1, padding=1)), 0.2)
    return F.conv2d(x, sd['model.11.weight'], sd['model.11.bias'], stride=1, padding=1)


def patchgan_param_shapes(input_nc, ndf=64):
    """Ordered (key, shape) list == reference state_dict order."""
    out = []
    chans = [(ndf, input_nc), (ndf * 2, ndf), (ndf * 4, ndf * 2), (ndf * 8, ndf * 4), (1, ndf * 8)]
    for idx, (co, ci) in zip((0, 2, 5, 8, 11), chans):
        out.append(('model.%d.weight' % idx, (co, ci, 4, 4)))
        out.append(('model.%d.bias' % idx, (co,)))
    return out
