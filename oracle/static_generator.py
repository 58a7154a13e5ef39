"""Oracle (test infrastructure): the static drawing generator and the streaming-inference composition.

Functional restatement of ``ResnetStyle2Generator`` (Module2/models/networks.py:573-637, selected as
``resnet_style2_9blocks`` at networks.py:155-156 with ``model0_res=0``, ``extra_channel=3``) over a plain dict of
tensors with the reference's ``state_dict`` key names, and of ``GeomCGTIFWTestModel.forward``
(Module2/models/geomcgt_ifw_test_model.py:276-302, 'drawing' branch) with the outputs of the frozen auxiliary nets
(MODNet matte, netF flow) as inputs.  Only tests, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline may use it.
"""
import torch
import torch.nn.functional as F

from .generator import conv_reflect, conv_zero, deconv, inorm, resnet_block


def static_param_shapes(input_nc=3, output_nc=1, ngf=64, n_blocks=9, extra_channel=3):
    """Ordered (key, shape) list == reference state_dict order.  nn.Sequential indices: model0 = [pad, conv(1), IN,
    ReLU, conv(4), IN, ReLU, conv(7), IN, ReLU]; model = [conv(0), IN, ReLU, 9 x ResnetBlock (3..11), deconv(12), IN,
    ReLU, deconv(15), IN, ReLU, pad, conv(19), tanh]."""
    out = []

    def conv(key, co, ci, k):
        out.append((key + '.weight', (co, ci, k, k)))
        out.append((key + '.bias', (co,)))

    conv('model0.1', ngf, input_nc, 7)
    conv('model0.4', ngf * 2, ngf, 3)
    conv('model0.7', ngf * 4, ngf * 2, 3)
    dim = ngf * 4
    conv('model.0', dim, dim + extra_channel, 3)
    for i in range(n_blocks):
        conv('model.%d.conv_block.1' % (3 + i), dim, dim, 3)
        conv('model.%d.conv_block.5' % (3 + i), dim, dim, 3)
    j = 3 + n_blocks
    out.append(('model.%d.weight' % j, (dim, dim // 2, 3, 3)))            # ConvTranspose2d: (in, out, k, k)
    out.append(('model.%d.bias' % j, (dim // 2,)))
    out.append(('model.%d.weight' % (j + 3), (dim // 2, dim // 4, 3, 3)))
    out.append(('model.%d.bias' % (j + 3), (dim // 4,)))
    conv('model.%d' % (j + 7), output_nc, ngf, 7)
    return out


def static_forward(sd, input1, input2, n_blocks=9):
    """forward(input1, input2), networks.py:632-636: f1 = model0(input1); model(cat[f1, input2])."""
    x = F.relu(inorm(conv_reflect(sd, 'model0.1', input1, 3)))
    x = F.relu(inorm(conv_zero(sd, 'model0.4', x, stride=2)))
    x = F.relu(inorm(conv_zero(sd, 'model0.7', x, stride=2)))
    x = torch.cat([x, input2], 1)
    x = F.relu(inorm(conv_zero(sd, 'model.0', x, stride=1)))
    for i in range(n_blocks):
        x = resnet_block(sd, 'model.%d' % (3 + i), x)
    j = 3 + n_blocks
    x = F.relu(inorm(deconv(sd, 'model.%d' % j, x)))
    x = F.relu(inorm(deconv(sd, 'model.%d' % (j + 3), x)))
    return torch.tanh(conv_reflect(sd, 'model.%d' % (j + 7), x, 3))


def style_code(n, size, device=None, dtype=torch.float32):
    """style_B of the 'drawing' branch (geomcgt_ifw_test_model.py:280): channels (0, 1, 0), constant planes."""
    s = torch.tensor([0., 1., 0.], dtype=dtype, device=device).view(1, 3, 1, 1)
    return s.repeat(n, 1, size, size)


def static_drawing(sd_static, real_A):
    """geomcgt_ifw_test_model.py:282-285: 256 -> 512 bilinear, static generator with the style planes at 128^2,
    512 -> 256 bilinear (align_corners=False both ways)."""
    n = real_A.shape[0]
    a512 = F.interpolate(real_A, size=(512, 512), mode='bilinear', align_corners=False)
    y512 = static_forward(sd_static, a512, style_code(n, 128, real_A.device, real_A.dtype))
    return F.interpolate(y512, size=(256, 256), mode='bilinear', align_corners=False)


def streaming_forward(generator_fn, real_A, matte, fakeB_static, land1, land2, warp_motion, iw_flow, if_mask):
    """geomcgt_ifw_test_model.py:276-300 after the frozen nets: ``matte`` (MODNet) and ``iw_flow`` / ``if_mask``
    (netF) are inputs; ``generator_fn`` is the hot-path generator G(input, land1, land2, motion, flow, ifmask).
    Returns (fake_B, fake_B_fore, mask1, masked real_A)."""
    mask = (matte > 0.5).float()
    real_A = ((real_A / 2 + 0.5) * mask + 1 - mask) * 2 - 1
    fake_fore = generator_fn(real_A, land1, land2, warp_motion, iw_flow, if_mask)
    mask1 = F.grid_sample(mask, warp_motion, align_corners=True)
    fake_B = ((fake_fore / 2 + 0.5) * mask1 + (fakeB_static / 2 + 0.5) * (1 - mask1)) * 2 - 1
    return fake_B, fake_fore, mask1, real_A
