"""FlowUnet_v2 (the intrinsic-flow regressor ``netF``) on the HIP convolution kernels.

``FlowUnetV2Hip(net)`` takes a loaded ``flow_unet.FlowUnetV2`` (the key-for-key mirror of
Module2/intrinsic_flow_models/networks.py:647-744, pinned to the reference class) and runs its forward pass -- inference
only: the network is frozen in both models (geomgm_ifw_fore_model.py:57-68) -- through libapamd.so:

  * every ``Conv2d -> BatchNorm2d`` pair (eval mode: running statistics) is folded into one convolution at construction,
    ``W' = W * gamma / sqrt(var + eps)``, ``b' = beta - mean * gamma / sqrt(var + eps)`` -- also across the
    ``PixelShuffle(2)`` of the decoder, where output channel ``4c + k`` takes the parameters of shuffled channel ``c``
    (:693-698); InstanceNorm configurations have no running statistics to fold and are refused;
  * ``activation -> conv`` (``ResidualBlock`` :26-60, the down / up-sampling Sequentials, the prediction heads) is the
    consumer-side activation of the conv loader (``Feat.act``), ``torch.cat((x, conv_a(a)))`` is a two-segment source;
  * 1x1 ``channel_mapping`` convolutions run as one-tap layers of the same kernels, stride-2 3x3 as in the generator;
  * ``x + conv(...)`` is one pass of ``ap_norm_apply_split`` (residual port), which also writes the split-bf16 copy the next
    wide convolution stages; ``PixelShuffle`` is ``ap_pixel_shuffle2``; the two final x``start_scale`` bilinear
    up-samplings are ``ap_resize_bilinear`` (``F.upsample(..., align_corners=False)``, :741-742).
Returns what the reference returns: ``(flow_out, vis_out, flow_pyr, feat_out)``.
"""
import torch
import torch.nn as nn

from . import ops
from .networks import ConvLayer
from .ops import Feat, ACT_NONE, ACT_RELU, PAD_ZERO


def _fold(conv, bn, shuffle=1):
    """(weight, bias) of ``bn(shuffle(conv(x)))`` as one convolution; ``shuffle`` = r^2 of a PixelShuffle in between."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
    if bn is None:
        return w, b
    if not isinstance(bn, nn.BatchNorm2d):
        raise NotImplementedError('FlowUnetV2Hip folds BatchNorm2d (norm="batch"); got %s' % type(bn).__name__)
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    t = bn.bias.detach().float() - bn.running_mean.detach().float() * s
    s, t = s.repeat_interleave(shuffle), t.repeat_interleave(shuffle)
    return w * s.view(-1, 1, 1, 1), b * s + t


def _layer(conv, bn, segs=None, shuffle=1):
    k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    layer = ConvLayer(segs or [conv.in_channels], conv.out_channels, k, st, pd, PAD_ZERO)
    w, b = _fold(conv, bn, shuffle)
    with torch.no_grad():
        layer.weight.copy_(w)
        layer.bias.copy_(b)
    for p in layer.parameters():
        p.requires_grad_(False)
    return layer


class _Act:
    """A materialised pre-activation feature and, lazily, its ``activation(f)`` view for the conv loaders: ONE Feat per
    feature, so the split-bf16 copy of relu(f) is made once however many layers read it (an encoder feature feeds the next
    block, the down-sampling layer and -- as ``hiddens`` -- the decoder's channel_mapping)."""
    __slots__ = ('f', '_relu')

    def __init__(self, f, relu_xs=None):
        self.f = f
        self._relu = None
        if relu_xs is not None:
            self._relu = Feat(f.data, act=ACT_RELU)
            self._relu.xs = relu_xs                     # (the loader of a split-bf16 layer takes the copy as it is)
            self._relu.xs_heads_only = ops.DEFAULT_PRECISION == ops.PRECISION_BF16

    @property
    def data(self):
        return self.f.data

    def relu(self):
        if self._relu is None:
            self._relu = Feat(self.f.data, act=ACT_RELU)
        return self._relu


def _add(y, x):
    """x + y for plain features; the same pass writes the split-bf16 copy of relu(x + y) that the next ``activation -> conv``
    stages (ap_norm_apply_split_ex, flag bit 1)."""
    n, c, h, w = y.data.shape
    if c % 8:
        return _Act(Feat(y.data + x.data))                             # (never the case for nf % 8 == 0)
    want_xs = ops.wants_split(c)
    out, xs = ops._norm_apply_split(Feat(y.data), Feat(x.data), want_y=True, want_xs=want_xs, xs_relu=True)
    return _Act(Feat(out), xs)


class _ResBlock(nn.Module):
    def __init__(self, blk):
        super().__init__()
        self.has_a = hasattr(blk, 'conv_a')
        if self.has_a:
            self.conv_a = _layer(blk.conv_a[0], blk.conv_a[1])
            dim = blk.conv[0].out_channels
            self.conv = _layer(blk.conv[0], blk.conv[1], segs=[dim, dim])
        else:
            self.conv = _layer(blk.conv[0], blk.conv[1])

    def run(self, x, a=None):
        """x, a: _Act; returns _Act"""
        if self.has_a:
            ya = _Act(self.conv_a.run([a.relu()]))
            y = self.conv.run([x.relu(), ya.relu()])
        else:
            y = self.conv.run([x.relu()])
        return _add(y, x)


class FlowUnetV2Hip(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.start_scale, self.num_scales, self.n_res = net.start_scale, net.num_scales, net.n_residual_blocks
        mods = list(net.pre_conv)
        self.pre0 = _layer(mods[0][0], mods[0][1])                      # channel_mapping(input_nc, nf)
        self.pre_res, self.pre_down = nn.ModuleList(), nn.ModuleList()
        i = 1
        while i < len(mods):                                            # [ResidualBlock, activation, Conv s2, norm] per level
            self.pre_res.append(_ResBlock(mods[i]))
            self.pre_down.append(_layer(mods[i + 2], mods[i + 3]))
            i += 4
        self.enc_res, self.enc_down = nn.ModuleList(), nn.ModuleList()
        self.dec_up, self.dec_res = nn.ModuleList(), nn.ModuleList()
        self.pred_flow = nn.ModuleList()
        for l in range(self.num_scales):
            self.enc_res.append(nn.ModuleList([_ResBlock(getattr(net, 'enc_%d_res_%d' % (l, k))) for k in range(self.n_res)]))
            dn = getattr(net, 'enc_%d_downsample' % l)
            self.enc_down.append(_layer(dn[1], dn[2]))
            up = getattr(net, 'dec_%d_upsample' % l)
            self.dec_up.append(_layer(up[1], up[3], shuffle=4))
            self.dec_res.append(nn.ModuleList([_ResBlock(getattr(net, 'dec_%d_res_%d' % (l, k))) for k in range(self.n_res)]))
            self.pred_flow.append(_layer(getattr(net, 'pred_flow_%d' % l)[1], None))
        self.pred_vis = _layer(net.pred_vis[1], None)
        # flow_network_warp (geomgm_ifw_fore_model.py:69-84) uses flow_out and vis_out only: with ``heads_only`` the
        # lower pyramid heads are skipped and the two full-resolution heads (c -> 2, c -> 3) run as ONE 5-output layer
        # (a 2- or 3-output layer fills 1/16 of a 32-cout MFMA tile either way)
        self.heads_only = False
        self.use_graph = False             # replay the forward pass as a hipGraph (set by the streaming callers)
        self._graphs = {}
        f0, vs = net.pred_flow_0[1], net.pred_vis[1]
        self.pred_both = ConvLayer([f0.in_channels], 5, 3, 1, 1, PAD_ZERO)
        with torch.no_grad():
            self.pred_both.weight.copy_(torch.cat([f0.weight.detach().float(), vs.weight.detach().float()], 0))
            self.pred_both.bias.copy_(torch.cat([f0.bias.detach().float(), vs.bias.detach().float()], 0))
        for q in self.pred_both.parameters():
            q.requires_grad_(False)

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('FlowUnetV2Hip runs on the MI355X only (the stock-PyTorch mirror is flow_unet.FlowUnetV2)')
        if self.use_graph and not torch.cuda.is_current_stream_capturing():
            return self._replay(x)
        return self._forward(x)

    def _replay(self, x):
        """The ~150 launches of a forward pass as ONE hipGraph launch per input shape.  The network is frozen and its layers
        are small at the deep scales, so a clip spends more host time issuing these launches than the GPU spends running them
        (profiles/r03z_stream_kernel_stats.md: 6.5 ms of host work per batch of 16 against ~3 ms of kernels).  Capture goes
        through torch.cuda.graph (its private pool owns the intermediate buffers); the C-ABI launches land on the capturing
        stream because every call passes torch's current stream."""
        key = (tuple(x.shape), x.device.index)
        ent = self._graphs.get(key)
        if ent is None:
            static_in = torch.empty_like(x, dtype=torch.float32).contiguous()
            static_in.copy_(x)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up: kernel attributes, packed weights, allocator
                for _ in range(2):
                    self._forward(static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self._forward(static_in)
            ent = self._graphs[key] = (graph, static_in, outs)
        graph, static_in, outs = ent
        static_in.copy_(x)
        graph.replay()
        # the graph's output buffers are overwritten by the next replay: hand out copies (a few MB)
        return tuple(o.clone() if torch.is_tensor(o) else ([t.clone() for t in o] if o is not None else None) for o in outs)

    def _forward(self, x):
        f = _Act(self.pre0.run([Feat(x.float().contiguous())]))
        for res, down in zip(self.pre_res, self.pre_down):
            f = _Act(down.run([res.run(f).relu()]))
        hiddens, flow_pyr = [], []
        for l in range(self.num_scales):
            for blk in self.enc_res[l]:
                f = blk.run(f)
                hiddens.append(f)
            f = _Act(self.enc_down[l].run([f.relu()]))
        for l in range(self.num_scales - 1, -1, -1):
            f = _Act(Feat(ops.pixel_shuffle2(self.dec_up[l].run([f.relu()]).data)))
            for k in range(self.n_res - 1, -1, -1):
                f = self.dec_res[l][k].run(f, hiddens.pop())
            if not self.heads_only:
                flow_pyr.insert(0, self.pred_flow[l].run([f.relu()]).data)
        s = self.start_scale

        def up(t):
            return ops.resize_bilinear(t, (t.shape[2] * s, t.shape[3] * s))
        if self.heads_only:
            both = up(self.pred_both.run([f.relu()]).data)               # bilinear up-sampling is per channel
            return both[:, 0:2].contiguous(), both[:, 2:5].contiguous(), None, f.data
        vis = self.pred_vis.run([f.relu()]).data
        return up(flow_pyr[0]), up(vis), flow_pyr, f.data
