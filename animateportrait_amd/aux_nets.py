"""The three frozen third-party networks the Module2 models call, as stock PyTorch-ROCm modules whose ``state_dict`` keys,
shapes and registration order equal the reference classes', so the published checkpoints load strictly:

* ``MobileFaceNet([112, 112], 136)``  68-point landmark regressor of the geometry loss
  (Module2/models/mobilefacenet.py:104-159; built and loaded at geomgm_ifw_fore_model.py:362-366, called in get_lm :410);
* ``Sphere20a``                         identity features inside ``networks.FaceLoss``
  (Module2/models/facenet.py:200-282; loaded by FaceLoss.load_sphere_model, networks.py:3044-3055: ``fc6*`` dropped);
* ``MODNet``                            portrait matting (Module2/models/modnet.py:204-236 with backbones/mobilenetv2.py,
  backbones/wrapper.py; loaded through nn.DataParallel, i.e. ``module.``-prefixed keys, geomgm_ifw_fore_model.py:369-373,
  geomcgt_ifw_test_model.py:218-223).

SURVEY.md section 2 row 12 keeps these nets off the hand-written kernels (depthwise / 1x1 / BatchNorm stacks, ~10 GMAC per
sample of the step's 627): they run on MIOpen through torch, frozen, in eval mode.  Pinned to the reference classes by
tests/golden/make_auxnets_golden.py (key lists + outputs for seeded weights).  ``attach_aux_networks`` is the loader the
models call (the reference hard-codes the three paths under ``checkpoints/``).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class _PReLU(nn.PReLU):
    """nn.PReLU (same ``weight`` key) evaluated as max(x, 0) + w * min(x, 0): the same values, but its backward is two plain
    elementwise kernels -- aten's prelu_backward also forms the (unused: the net is frozen) slope gradient and ran 155 us per
    layer on these small maps, 5 of the 8 ms of a MobileFaceNet forward + backward (tools/aux_bench.py)."""

    def forward(self, x):
        w = self.weight.view(1, -1, *([1] * (x.dim() - 2)))
        return torch.clamp_min(x, 0) + w * torch.clamp_max(x, 0)


# ----------------------------------------------------------------------------------------------- MobileFaceNet
class _ConvBN(nn.Module):
    """conv (no bias) -> BatchNorm2d [-> PReLU]; attribute names are the checkpoint's (conv / bn / prelu)."""

    def __init__(self, cin, cout, k=1, s=1, p=0, groups=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, p, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        if act:
            self.prelu = _PReLU(cout)
        self._act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return self.prelu(x) if self._act else x


class _DepthWise(nn.Module):
    """1x1 expansion to ``mid`` -> 3x3 depthwise (stride) -> linear 1x1 projection, optional identity shortcut."""

    def __init__(self, cin, cout, mid, stride, residual):
        super().__init__()
        self.conv = _ConvBN(cin, mid)
        self.conv_dw = _ConvBN(mid, mid, 3, stride, 1, groups=mid)
        self.project = _ConvBN(mid, cout, act=False)
        self.residual = residual

    def forward(self, x):
        y = self.project(self.conv_dw(self.conv(x)))
        return x + y if self.residual else y


class _ResidualStack(nn.Module):
    def __init__(self, c, n, mid):
        super().__init__()
        self.model = nn.Sequential(*[_DepthWise(c, c, mid, 1, True) for _ in range(n)])

    def forward(self, x):
        return self.model(x)


class _GDC(nn.Module):
    """global depthwise 7x7 -> flatten -> Linear (no bias) -> BatchNorm1d"""

    def __init__(self, emb):
        super().__init__()
        self.conv_6_dw = _ConvBN(512, 512, 7, 1, 0, groups=512, act=False)
        self.linear = nn.Linear(512, emb, bias=False)
        self.bn = nn.BatchNorm1d(emb)

    def forward(self, x):
        return self.bn(self.linear(self.conv_6_dw(x).flatten(1)))


class MobileFaceNet(nn.Module):
    """forward(x (B, 3, 112, 112) in [0, 1]) -> (embedding (B, emb), conv features (B, 512, 7, 7))."""

    def __init__(self, input_size=(112, 112), embedding_size=136):
        super().__init__()
        if input_size[0] != 112:
            raise ValueError('MobileFaceNet: 112 x 112 inputs only (the 7 x 7 global depthwise layer)')
        self.conv1 = _ConvBN(3, 64, 3, 2, 1)
        self.conv2_dw = _ConvBN(64, 64, 3, 1, 1, groups=64)
        self.conv_23 = _DepthWise(64, 64, 128, 2, False)
        self.conv_3 = _ResidualStack(64, 4, 128)
        self.conv_34 = _DepthWise(64, 128, 256, 2, False)
        self.conv_4 = _ResidualStack(128, 6, 256)
        self.conv_45 = _DepthWise(128, 128, 512, 2, False)
        self.conv_5 = _ResidualStack(128, 2, 256)
        self.conv_6_sep = _ConvBN(128, 512)
        self.output_layer = _GDC(embedding_size)

    def forward(self, x):
        for name in ('conv1', 'conv2_dw', 'conv_23', 'conv_3', 'conv_34', 'conv_4', 'conv_45', 'conv_5', 'conv_6_sep'):
            x = getattr(self, name)(x)
        return self.output_layer(x), x


# ----------------------------------------------------------------------------------------------- Sphere20a
class Sphere20a(nn.Module):
    """forward(x (B, 3, 112, 96) in [-1, 1]) -> [four stage maps (/2 /4 /8 /16), the 512-d fc5 vector] -- the feature list
    FaceLoss.compute_loss sums L1 distances over (networks.py:2926-2940).  Stage s: a stride-2 conv + PReLU, then ``units``
    residual pairs of conv + PReLU; layer names conv{s}_{i} / relu{s}_{i} as in the checkpoint."""
    STAGES = ((64, 1), (128, 2), (256, 4), (512, 1))

    def __init__(self):
        super().__init__()
        cin = 3
        for s, (c, units) in enumerate(self.STAGES, 1):
            for i in range(1, 2 * units + 2):
                setattr(self, 'conv%d_%d' % (s, i), nn.Conv2d(cin if i == 1 else c, c, 3, 2 if i == 1 else 1, 1))
                setattr(self, 'relu%d_%d' % (s, i), _PReLU(c))
            cin = c
        self.fc5 = nn.Linear(512 * 7 * 6, 512)

    def forward(self, x):
        feats = []
        for s, (c, units) in enumerate(self.STAGES, 1):
            layer = lambda i, t: getattr(self, 'relu%d_%d' % (s, i))(getattr(self, 'conv%d_%d' % (s, i))(t))   # noqa: E731
            x = layer(1, x)
            for u in range(units):
                x = x + layer(2 * u + 3, layer(2 * u + 2, x))
            feats.append(x)
        feats.append(self.fc5(x.flatten(1)))
        return feats


# ----------------------------------------------------------------------------------------------- MODNet
def _cbr6(cin, cout, k, s, groups=1):
    return [nn.Conv2d(cin, cout, k, s, k // 2, groups=groups, bias=False), nn.BatchNorm2d(cout), nn.ReLU6(inplace=True)]


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, t):
        super().__init__()
        hid = round(cin * t)
        self.shortcut = stride == 1 and cin == cout
        layers = [] if t == 1 else _cbr6(cin, hid, 1, 1)
        layers += _cbr6(hid, hid, 3, stride, groups=hid)
        layers += [nn.Conv2d(hid, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.shortcut else self.conv(x)


class _MobileNetV2(nn.Module):
    """features.0 .. features.18 of MobileNetV2 (alpha 1, expansion 6), no classifier."""
    SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))

    def __init__(self, in_channels):
        super().__init__()
        feats = [nn.Sequential(*_cbr6(in_channels, 32, 3, 2))]
        cin = 32
        for t, c, n, s in self.SETTING:
            for i in range(n):
                feats.append(_InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(nn.Sequential(*_cbr6(cin, 1280, 1, 1)))
        self.features = nn.Sequential(*feats)


class _Backbone(nn.Module):
    enc_channels = (16, 24, 32, 96, 1280)
    TAPS = (1, 3, 6, 13, 18)          # last feature index of the /2 /4 /8 /16 /32 stages

    def __init__(self, in_channels):
        super().__init__()
        self.model = _MobileNetV2(in_channels)

    def forward(self, x):
        out = []
        for i, f in enumerate(self.model.features):
            x = f(x)
            if i in self.TAPS:
                out.append(x)
        return out


class _IBNorm(nn.Module):
    """BatchNorm on the first half of the channels, InstanceNorm (no affine) on the rest"""

    def __init__(self, c):
        super().__init__()
        self.nb = c // 2
        self.bnorm = nn.BatchNorm2d(self.nb, affine=True)
        self.inorm = nn.InstanceNorm2d(c - self.nb, affine=False)

    def forward(self, x):
        return torch.cat((self.bnorm(x[:, :self.nb].contiguous()), self.inorm(x[:, self.nb:].contiguous())), 1)


class _ConvIBNRelu(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, ibn=True, relu=True):
        super().__init__()
        layers = [nn.Conv2d(cin, cout, k, stride=stride, padding=padding)]
        if ibn:
            layers.append(_IBNorm(cout))
        if relu:
            layers.append(nn.ReLU(inplace=True))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x)


class _SE(nn.Module):
    def __init__(self, c, reduction):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(c, c // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(c // reduction, c, bias=False), nn.Sigmoid())

    def forward(self, x):
        return x * self.fc(x.mean((2, 3))).view(x.shape[0], -1, 1, 1)


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


class _LRBranch(nn.Module):
    def __init__(self, backbone):
        super().__init__()
        e = backbone.enc_channels
        self.backbone = backbone
        self.se_block = _SE(e[4], 4)
        self.conv_lr16x = _ConvIBNRelu(e[4], e[3], 5, 1, 2)
        self.conv_lr8x = _ConvIBNRelu(e[3], e[2], 5, 1, 2)
        self.conv_lr = _ConvIBNRelu(e[2], 1, 3, 2, 1, ibn=False, relu=False)

    def forward(self, img, inference):
        enc = self.backbone(img)
        lr8x = self.conv_lr8x(_up2(self.conv_lr16x(_up2(self.se_block(enc[4])))))
        sem = None if inference else torch.sigmoid(self.conv_lr(lr8x))
        return sem, lr8x, enc[0], enc[1]


class _HRBranch(nn.Module):
    def __init__(self, h, e):
        super().__init__()
        self.tohr_enc2x = _ConvIBNRelu(e[0], h, 1)
        self.conv_enc2x = _ConvIBNRelu(h + 3, h, 3, 2, 1)
        self.tohr_enc4x = _ConvIBNRelu(e[1], h, 1)
        self.conv_enc4x = _ConvIBNRelu(2 * h, 2 * h, 3, 1, 1)
        self.conv_hr4x = nn.Sequential(_ConvIBNRelu(3 * h + 3, 2 * h, 3, 1, 1), _ConvIBNRelu(2 * h, 2 * h, 3, 1, 1),
                                       _ConvIBNRelu(2 * h, h, 3, 1, 1))
        self.conv_hr2x = nn.Sequential(_ConvIBNRelu(2 * h, 2 * h, 3, 1, 1), _ConvIBNRelu(2 * h, h, 3, 1, 1),
                                       _ConvIBNRelu(h, h, 3, 1, 1), _ConvIBNRelu(h, h, 3, 1, 1))
        self.conv_hr = nn.Sequential(_ConvIBNRelu(h + 3, h, 3, 1, 1), _ConvIBNRelu(h, 1, 1, ibn=False, relu=False))

    def forward(self, img, enc2x, enc4x, lr8x, inference):
        down = lambda f: F.interpolate(img, scale_factor=f, mode='bilinear', align_corners=False)    # noqa: E731
        enc2x = self.tohr_enc2x(enc2x)
        hr4x = self.conv_enc2x(torch.cat((down(1 / 2), enc2x), 1))
        hr4x = self.conv_enc4x(torch.cat((hr4x, self.tohr_enc4x(enc4x)), 1))
        hr4x = self.conv_hr4x(torch.cat((hr4x, _up2(lr8x), down(1 / 4)), 1))
        hr2x = self.conv_hr2x(torch.cat((_up2(hr4x), enc2x), 1))
        detail = None if inference else torch.sigmoid(self.conv_hr(torch.cat((_up2(hr2x), img), 1)))
        return detail, hr2x


class _FusionBranch(nn.Module):
    def __init__(self, h, e):
        super().__init__()
        self.conv_lr4x = _ConvIBNRelu(e[2], h, 5, 1, 2)
        self.conv_f2x = _ConvIBNRelu(2 * h, h, 3, 1, 1)
        self.conv_f = nn.Sequential(_ConvIBNRelu(h + 3, h // 2, 3, 1, 1), _ConvIBNRelu(h // 2, 1, 1, ibn=False, relu=False))

    def forward(self, img, lr8x, hr2x):
        lr2x = _up2(self.conv_lr4x(_up2(lr8x)))
        f2x = self.conv_f2x(torch.cat((lr2x, hr2x), 1))
        return torch.sigmoid(self.conv_f(torch.cat((_up2(f2x), img), 1)))


class MODNet(nn.Module):
    """forward(img (B, 3, H, W) in [-1, 1], inference) -> (semantic | None, detail | None, matte (B, 1, H, W) in [0, 1]).
    The backbone module is registered under ``backbone`` AND ``lr_branch.backbone`` (one object), as in the reference: its
    tensors appear under both prefixes in the checkpoint."""

    def __init__(self, in_channels=3, hr_channels=32, backbone_pretrained=False):
        super().__init__()
        if backbone_pretrained:
            raise ValueError('MODNet: the ImageNet / human-seg backbone checkpoint is not part of this package; '
                             'load the full matting checkpoint instead')
        self.backbone = _Backbone(in_channels)
        self.lr_branch = _LRBranch(self.backbone)
        self.hr_branch = _HRBranch(hr_channels, self.backbone.enc_channels)
        self.f_branch = _FusionBranch(hr_channels, self.backbone.enc_channels)
        self.fast_layout = False       # set by prepare_frozen: NHWC activations (MIOpen's NHWC solvers: 11.8 -> 8.8 ms at B=16)

    def forward(self, img, inference=True):
        if self.fast_layout and img.is_cuda:
            img = img.contiguous(memory_format=torch.channels_last)
            # ... and MIOpen may TIME its algorithms for this net's shapes once instead of taking its immediate-mode pick
            # (8.8 -> 7.1 ms at B = 16, tools/aux_find_bench.py; MobileFaceNet / Sphere20a do not move).  Scoped to this call.
            bk = torch.backends.cudnn
            with bk.flags(enabled=bk.enabled, benchmark=not os.environ.get('APAMD_NO_MIOPEN_FIND'),
                          deterministic=bk.deterministic, allow_tf32=bk.allow_tf32):
                return self._forward(img, inference)
        return self._forward(img, inference)

    def _forward(self, img, inference):
        sem, lr8x, enc2x, enc4x = self.lr_branch(img, inference)
        detail, hr2x = self.hr_branch(img, enc2x, enc4x, lr8x, inference)
        return sem, detail, self.f_branch(img, lr8x, hr2x).contiguous()


# ----------------------------------------------------------------------------------------------- loaders
MOBILEFACENET_CKPT = 'mobilefacenet_model_best.pth.tar'              # geomgm_ifw_fore_model.py:363
MODNET_CKPT = 'modnet_photographic_portrait_matting.ckpt'            # :371, geomcgt_ifw_test_model.py:222


def fold_batchnorm(net):
    """Fold every eval-mode ``Conv2d -> BatchNorm2d`` pair that sits side by side in a module into the convolution (in place;
    the BatchNorm becomes an Identity).  Only for a net that is already loaded and frozen: its state_dict keys change."""
    from torch.nn.utils.fusion import fuse_conv_bn_eval

    def walk(m):
        names = list(m._modules.keys())
        for a, b in zip(names, names[1:]):
            ma, mb = m._modules[a], m._modules[b]
            if isinstance(ma, nn.Conv2d) and isinstance(mb, nn.BatchNorm2d):
                m._modules[a] = fuse_conv_bn_eval(ma.eval(), mb.eval())
                m._modules[b] = nn.Identity()
        for c in m._modules.values():
            if c is not None:
                walk(c)
    walk(net)
    return net


def _frozen(net, device, fold=True):
    """eval mode, no parameter gradients, BatchNorm folded into the convolutions (the net never trains), MODNet in NHWC."""
    net = net.to(device).eval()
    if fold:
        fold_batchnorm(net)
    if isinstance(net, MODNet) and torch.device(device).type == 'cuda':
        net = net.to(memory_format=torch.channels_last)
        net.fast_layout = True
    for p in net.parameters():
        p.requires_grad_(False)
    return net


def load_mobilefacenet(path, device, fold=True):
    """geomgm_ifw_fore_model.py:362-366: checkpoint dict with the weights under 'state_dict'."""
    net = MobileFaceNet((112, 112), 136)
    ck = torch.load(path, map_location='cpu')
    net.load_state_dict(ck['state_dict'], strict=True)
    return _frozen(net, device, fold)


def load_modnet(path, device, fold=True):
    """:369-373: the checkpoint was saved from nn.DataParallel(MODNet) -- every key carries a ``module.`` prefix."""
    net = MODNet(backbone_pretrained=False)
    sd = torch.load(path, map_location='cpu')
    net.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}, strict=True)
    return _frozen(net, device, fold)


def load_sphere20a(path, device, fold=True):
    """networks.py:3044-3053: the classifier head ``fc6*`` of the published checkpoint is dropped, the rest loads strictly."""
    net = Sphere20a()
    sd = torch.load(path, map_location='cpu')
    net.load_state_dict({k: v for k, v in sd.items() if not k.startswith('fc6')}, strict=True)
    return _frozen(net, device, fold)


class GraphedFrozen(nn.Module):
    """A frozen network whose forward + backward (w.r.t. its input) replay as hipGraphs, one pair per input shape
    (torch.cuda.make_graphed_callables).  MobileFaceNet is ~600 small kernels per forward + backward at the train step's shapes:
    6.2 ms eager, 4.5 ms as a graph; Sphere20a 4.2 -> 3.2 ms (tools/aux_bench.py).  ``pick`` selects the outputs the caller uses
    (every graphed output needs a gradient in backward).  Calls whose input needs no gradient, CPU tensors and autocast / no_grad
    contexts run eagerly.  A graph's outputs and saved activations are STATIC buffers: one call per shape may be in flight
    between forward and backward -- a second grad-requiring call of the same shape before the first one's backward (which would
    overwrite them) is detected and runs eagerly.  The graphs are captured after a device synchronisation: capture warms up on a side stream, and no
    kernel of this package may run beside another stream's."""

    def __init__(self, net, pick=None):
        super().__init__()
        self.net = net
        self.pick = pick
        self._graphs = {}
        self._in_flight = set()       # shapes whose graph ran forward and has not yet run backward
        self._warned = False

    def new_step(self):
        """Step boundary (the model calls it from set_input): a grad-enabled forward whose backward never reached the input -- an
        exception in between, a loss term that was dropped, an evaluation pass under grad -- would otherwise keep its shape marked
        for ever, and every later call of that shape would silently run eagerly.  Nothing of the previous step's graph
        can still be needed once the next batch is set."""
        self._in_flight.clear()

    def _eager(self, x):
        out = self.net(x)
        return out if self.pick is None else self.pick(out)

    def forward(self, x):
        if (not (x.is_cuda and x.requires_grad and torch.is_grad_enabled()) or torch.is_autocast_enabled() or
                os.environ.get('APAMD_NO_AUX_GRAPHS')):
            return self._eager(x)
        key = (tuple(x.shape), x.dtype)
        if key in self._in_flight:
            if not self._warned:
                self._warned = True
                print('[aux] GraphedFrozen(%s): a second grad-requiring call of shape %s before the first one\'s backward runs eagerly '
                      '(new_step() at the step boundary clears forwards whose backward never ran)' % (type(self.net).__name__, key[0]))
            return self._eager(x)
        g = self._graphs.get(key)
        if g is None:
            outer = self

            class _Fn(nn.Module):
                def __init__(self):
                    super().__init__()
                    self.net = outer.net

                def forward(self, t):
                    return outer._eager(t)
            torch.cuda.synchronize()
            sample = torch.zeros_like(x).requires_grad_(True)
            g = self._graphs[key] = torch.cuda.make_graphed_callables(_Fn(), (sample,))
            torch.cuda.synchronize()
        xin = x.contiguous()
        if not xin.requires_grad:
            return self._eager(x)
        self._in_flight.add(key)
        xin = _ReleaseOnBackward.apply(xin, self._in_flight, key)     # its backward runs after the graph's: the buffers are free again
        return g(xin)


class _ReleaseOnBackward(torch.autograd.Function):
    """Identity in front of a graphed callable; when the gradient comes back through it the graph's backward has run."""

    @staticmethod
    def forward(ctx, x, in_flight, key):
        ctx.in_flight, ctx.key = in_flight, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.in_flight.discard(ctx.key)
        return g, None, None


def attach_aux_networks(model, checkpoints_dir='checkpoints', verbose=True):
    """What the reference models do in ``__init__`` with hard-coded paths: load whichever of the frozen nets has its checkpoint
    under ``checkpoints_dir`` (and ``opt.face_recog_model`` for Sphere20a) into the model's ``aux`` slots that are still empty.
    A missing file leaves the slot empty -- the model then reads that net's output from the batch / skips its loss term with a
    notice, as before.  Returns the names that were attached."""
    from . import networks
    aux = getattr(model, 'aux', None)
    if aux is None:
        return []
    opt, dev, done = model.opt, model.device, []
    say = print if verbose else (lambda *a: None)
    p = os.path.join(checkpoints_dir, MODNET_CKPT)
    if 'modnet' in aux and aux['modnet'] is None and os.path.exists(p):
        aux['modnet'] = load_modnet(p, dev)
        done.append('modnet')
    if getattr(model, 'isTrain', False):
        p = os.path.join(checkpoints_dir, MOBILEFACENET_CKPT)
        if 'landmarks' in aux and aux['landmarks'] is None and os.path.exists(p):
            aux['landmarks'] = GraphedFrozen(load_mobilefacenet(p, dev), pick=lambda o: o[0])
            done.append('landmarks')
        p = getattr(opt, 'face_recog_model', None)
        if 'faceloss' in aux and aux['faceloss'] is None and getattr(opt, 'identity_loss', 0) and p and os.path.exists(p):
            if 'senet' in p:
                raise NotImplementedError('--face_recog_model: only the Sphere20a checkpoint is supported (FaceLoss, networks.py:2862-2871)')
            aux['faceloss'] = networks.FaceLoss(GraphedFrozen(load_sphere20a(p, dev), pick=tuple))
            done.append('faceloss')
    for n in done:
        say('[aux] attached frozen network: %s' % n)
    return done
