"""Oracle (test infrastructure): the landmark-conditioned tri-branch generator.

Functional restatement of ``ResnetConditionTriGenerator32_full_ifw``
(Module2/models/networks.py:1190-1340) over a plain ``dict`` of tensors that
uses the reference's ``state_dict`` key names.  No
``nn.Module``; every layer is one line of ``torch.nn.functional``.
"""
import torch
import torch.nn.functional as F

from .warp import double_feature_warping

EPS = 1e-5  # nn.InstanceNorm2d default, networks.py:33-34


def inorm(x):
    """InstanceNorm2d(affine=False, track_running_stats=False): biased var, eps 1e-5."""
    return F.instance_norm(x, eps=EPS)


def _wb(sd, key):
    return sd[key + '.weight'], sd.get(key + '.bias')


def conv_reflect(sd, key, x, pad):
    w, b = _wb(sd, key)
    return F.conv2d(F.pad(x, (pad,) * 4, mode='reflect'), w, b)


def conv_zero(sd, key, x, stride=1, pad=1):
    w, b = _wb(sd, key)
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def deconv(sd, key, x):
    """ConvTranspose2d(k=3, s=2, p=1, output_padding=1), networks.py:1271-1274."""
    w, b = _wb(sd, key)
    return F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)


def stem7(sd, key, x):
    """ReflectionPad2d(3) + Conv 7x7 + IN + ReLU (networks.py:1218-1221 etc.)."""
    return F.relu(inorm(conv_reflect(sd, key, x, 3)))


def down3(sd, key, x):
    """Conv 3x3 s2 p1 + IN + ReLU (networks.py:1222-1227 etc.)."""
    return F.relu(inorm(conv_zero(sd, key, x, stride=2)))


def landmark_trans(sd, land):
    """model_landmark_trans, networks.py:1280-1282: 1->8 (s1) ->16 (s2) ->16 (s2), last IN has no ReLU."""
    x = F.relu(inorm(conv_zero(sd, 'model_landmark_trans.0', land, 1)))
    x = F.relu(inorm(conv_zero(sd, 'model_landmark_trans.3', x, 2)))
    return inorm(conv_zero(sd, 'model_landmark_trans.6', x, 2))


def resnet_block(sd, prefix, x):
    """ResnetBlock, networks.py:2303-2361 (reflect pad, no dropout)."""
    y = F.relu(inorm(conv_reflect(sd, prefix + '.conv_block.1', x, 1)))
    y = inorm(conv_reflect(sd, prefix + '.conv_block.5', y, 1))
    return x + y


def resnet_block2(sd, prefix, x):
    """ResnetBlock2, networks.py:2363-2421: main branch reflect-padded, shortcut zero-padded."""
    y = F.relu(inorm(conv_reflect(sd, prefix + '.conv_block.1', x, 1)))
    y = inorm(conv_reflect(sd, prefix + '.conv_block.5', y, 1))
    s = inorm(conv_zero(sd, prefix + '.shortcut.0', x, 1))
    return s + y


def is_block2(i, div, disp):
    """networks.py:1260,1334."""
    return (i + disp) % div == 0


def trunk_input(sd, inp, motion, flow, ifmask, formula=False):
    """networks.py:1317-1330: three encoder branches, warps, merge conv."""
    x1 = stem7(sd, 'model_tri00.1', inp)
    x1 = double_feature_warping(x1, motion, flow, ifmask, 0, formula)
    x1 = down3(sd, 'model_tri01.0', x1)
    x1 = down3(sd, 'model_tri02.0', x1)
    x2 = stem7(sd, 'model_tri10.1', inp)
    x2 = down3(sd, 'model_tri11.0', x2)
    x2 = double_feature_warping(x2, motion, flow, ifmask, 1, formula)
    x2 = down3(sd, 'model_tri12.0', x2)
    x3 = stem7(sd, 'model_tri20.1', inp)
    x3 = down3(sd, 'model_tri21.0', x3)
    x3 = down3(sd, 'model_tri22.0', x3)
    x3 = double_feature_warping(x3, motion, flow, ifmask, 2, formula)
    return conv_zero(sd, 'model_tri_merge', torch.cat([x1, x2, x3], 1), 1)


def generator_forward(sd, inp, land1, land2, motion, flow, ifmask,
                      n_blocks=9, div=3, disp=1, formula=False):
    """networks.py:1315-1340.

    inp (B,3,256,256); land1/land2 (B,1,256,256); motion (B,256,256,2);
    flow (B,2,256,256); ifmask (B,1,256,256) -> (B,output_nc,256,256)."""
    x = trunk_input(sd, inp, motion, flow, ifmask, formula)
    l1 = landmark_trans(sd, land1)
    l2 = landmark_trans(sd, land2)
    for i in range(n_blocks):
        if is_block2(i, div, disp):
            x = resnet_block2(sd, 'model2.%d' % i, torch.cat([x, l1, l2], 1))
        else:
            x = resnet_block(sd, 'model2.%d' % i, x)
    x = F.relu(inorm(deconv(sd, 'model3.0', x)))
    x = F.relu(inorm(deconv(sd, 'model3.3', x)))
    return torch.tanh(conv_reflect(sd, 'model3.7', x, 3))


# ----------------------------------------------------------- parameters ----
def generator_param_shapes(input_nc=3, output_nc=1, ngf=64, n_blocks=9, div=3, disp=1):
    """Ordered (key, shape) list == reference state_dict order.

    Order follows nn.Module registration order in networks.py:1251,1284-1295:
    model_tri_merge first (assigned directly in __init__), then the Sequentials."""
    h = ngf // 2
    out = []

    def conv(key, co, ci, k):
        out.append((key + '.weight', (co, ci, k, k)))
        out.append((key + '.bias', (co,)))

    def deconv_(key, ci, co, k):
        out.append((key + '.weight', (ci, co, k, k)))
        out.append((key + '.bias', (co,)))

    conv('model_tri_merge', ngf * 4, ngf * 12, 3)
    conv('model_tri00.1', h, input_nc, 7)
    conv('model_tri01.0', ngf * 2, ngf, 3)
    conv('model_tri02.0', ngf * 4, ngf * 2, 3)
    conv('model_tri10.1', ngf, input_nc, 7)
    conv('model_tri11.0', ngf, ngf, 3)
    conv('model_tri12.0', ngf * 4, ngf * 2, 3)
    conv('model_tri20.1', ngf, input_nc, 7)
    conv('model_tri21.0', ngf * 2, ngf, 3)
    conv('model_tri22.0', ngf * 2, ngf * 2, 3)
    dim = ngf * 4
    for i in range(n_blocks):
        p = 'model2.%d' % i
        if is_block2(i, div, disp):
            conv(p + '.conv_block.1', dim, dim + 32, 3)
            conv(p + '.conv_block.5', dim, dim, 3)
            conv(p + '.shortcut.0', dim, dim + 32, 3)
        else:
            conv(p + '.conv_block.1', dim, dim, 3)
            conv(p + '.conv_block.5', dim, dim, 3)
    deconv_('model3.0', ngf * 4, ngf * 2, 3)
    deconv_('model3.3', ngf * 2, ngf, 3)
    conv('model3.7', output_nc, ngf, 7)
    conv('model_landmark_trans.0', 8, 1, 3)
    conv('model_landmark_trans.3', 16, 8, 3)
    conv('model_landmark_trans.6', 16, 16, 3)
    return out


def init_params(shapes, seed, gain=0.02):
    """init_weights('normal', 0.02): conv weights N(0, gain), biases 0
    (networks.py:71-102).  Draw order == shapes order, one normal_() per weight,
    on a dedicated CPU generator so the GPU box regenerates identical values."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in shapes:
        if key.endswith('.weight'):
            sd[key] = torch.empty(shape, dtype=torch.float32).normal_(0.0, gain, generator=g)
        else:
            sd[key] = torch.zeros(shape, dtype=torch.float32)
    return sd
