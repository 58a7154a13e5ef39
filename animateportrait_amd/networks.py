"""Host-side mirror of the reference's network registry for the hot path.

Same entry points, argument meaning and error behaviour as Module2/models/networks.py:
``define_G`` (:123-201), ``define_D`` (:204-247), ``init_weights`` (:71-102), ``get_scheduler``
(:42-68), ``GANLoss`` (:407-473); same ``state_dict`` keys as
``ResnetConditionTriGenerator32_full_ifw`` (:1190-1340) and ``NLayerDiscriminator`` (:2602-2647)
so reference checkpoints (``<epoch>_net_G_A.pth`` ...) load with ``strict=True``.

The modules own their parameters as ordinary ``nn.Parameter``s; ``forward`` never touches a
torch operator for the math -- every layer is a call into libapamd.so (HIP, gfx950).
"""
import functools

import torch
import torch.nn as nn
from torch.optim import lr_scheduler

from . import ops
from .autograd import (conv_forward, materialize_forward, warp_forward, batch_split_forward,
                       generator_apply, discriminator_apply)
from .ops import (Feat, ConvSpec, ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT,
                  W_OIHW, W_IOHW)

GENERATOR_NAMES = ('resnet_9blocks_rcatland32_full_ifw', 'resnet_style2_9blocks')
# names the reference registry knows (networks.py:152-199) but that are outside this build's hot path
_REFERENCE_ONLY_G = (
    'resnet_9blocks', 'resnet_9blocks_rcatland', 'resnet_9blocks_rcatland3',
    'resnet_9blocks_rcatland32', 'resnet_10blocks_rcatland32', 'resnet_9blocks_rcatland4',
    'resnet_9blocks_rcatland32_fw', 'resnet_9blocks_rcatland32_fw2', 'resnet_9blocks_rcatland32_ifw',
    'resnet_9blocks_rcatland32_ifw_single2', 'resnet_9blocks_rcatland32_full_ifw_colorcoded',
    'resnet_9blocks_rcatland32_full_ifw2', 'resnet_9blocks_rcatland32_full_ifw_single',
    'resnet_9blocks_rcatland32_full_ifw_single2', 'resnet_9blocks_rcatland32_full_ifw_single3', 'regressor',
    'combiner', 'resnet_9blocks_rcatland2', 'resnet_6blocks', 'unet_128', 'unet_256')


class ConvLayer(nn.Module):
    """Parameters of one nn.Conv2d / nn.ConvTranspose2d plus the kernel-side spec.
    state_dict: ``weight`` (OIHW, or IOHW when transposed) and ``bias`` -- as the reference layer."""

    def __init__(self, cin_segments, cout, k, stride=1, pad=0, pad_mode=PAD_ZERO, transposed=False,
                 output_padding=0):
        super().__init__()
        cin = sum(cin_segments)
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.zeros(cout, dtype=torch.float32))
        self.spec = ConvSpec(cin_segments, cout, k, stride, pad, pad_mode, transposed, output_padding,
                             W_IOHW if transposed else W_OIHW)
        self._slots = {}                   # operator variant -> ops.PackedSlot (forward, row / space-to-depth forms, data gradients)
        self._rows_spec = None
        self._s2d_spec = None

    def _packed(self, slot_id, spec, view):
        slot = ops.packed_slot(spec, self.weight, view, self._slots.get(slot_id))
        self._slots[slot_id] = slot
        return slot.buf

    def packed_dgrad(self, seg, spec, view):
        """Packed weights of the data-gradient operator for input segment ``seg``; ``view``: the operand as a (strided)
        view of this layer's parameter (a channel slice, transposed taps): no contiguous copy is made for the batched packer."""
        return self._packed(('dgrad', seg), spec, ops.WeightView(view))

    def packed(self):
        return self._packed('fwd', self.spec, ops.WeightView(self.weight.detach()))

    def run(self, srcs, norm_act=None, act=ACT_NONE, out_octet=False, out_bf16=False):
        """norm_act=None: plain conv + bias + act.  norm_act=ACT_*: conv followed by InstanceNorm
        (bias skipped -- it cancels exactly under the mean subtraction) and that activation,
        both deferred to the consumer.  out_octet: see ops.conv2d (the output is read by the warp kernel only)."""
        if not isinstance(srcs, (list, tuple)):
            srcs = [srcs]
        spec, packed = self.spec, None
        if norm_act is None and ops.conv_d0_ok(spec, srcs, act):
            # the PatchGAN's first layer in plain-bf16 arithmetic: an output stream on the bf16 matrix pipe (csrc/conv_d0.h)
            return ops.conv_d0(srcs[0], self.weight, self.bias.detach(), act)
        if ops.stem_rows_eligible(spec) and not srcs[0].is_split_only:
            # 7x7 stem on <= 4 channels: 1x7 split-bf16 convolution over the row expansion of the input
            spec, packed = self.rows_spec(), self.packed_rows()
            srcs = [ops.presplit_rows(srcs[0], self.spec.k, self.spec.pad, self.spec.pad_mode)]
        elif (srcs[0].s2d is not None or not srcs[0].is_split_only) and ops.s2d_eligible(spec, *srcs[0].data.shape[2:]):
            # 4x4 stride-2 PatchGAN layer / 3x3 stride-2 encoder layer: 2x2 split-bf16 convolution over the space-to-depth
            # copy of the input -- made here, or already written by the producer (the warp kernel, ops.warp_concat s2d=True)
            spec, packed = self.s2d_spec(), self.packed_s2d()
            if srcs[0].s2d is None:
                srcs[0].s2d = ops.presplit_s2d(srcs[0])      # (kept on the source: the weight gradient re-tiles its operand from it)
            srcs = [srcs[0].s2d]
        if packed is None:
            packed = self.packed()
        if norm_act is None:
            return ops.conv2d(spec, srcs, packed, self.bias.detach(), act=act, out_octet=out_octet)
        return ops.conv2d(spec, srcs, packed, None, want_stats=True, out_act=norm_act, out_octet=out_octet,
                          out_bf16=out_bf16 and spec is self.spec)

    def stages_split(self, shape):
        """Whether this layer reads a source of ``shape`` (N, C, H, W) as its split-bf16 copy (then the producer may
        leave out the fp32 tensor in inference: ops.materialize keep_fp32=False)."""
        n, _, h, w = shape
        if ops.stem_rows_eligible(self.spec) or ops.s2d_eligible(self.spec, h, w):
            return False
        return ops.takes_split(self.spec, n, h, w)

    def fused_norm_ok(self, srcs):
        """Inference only: can this layer produce act(IN(conv)) [+ residual] in one launch (ops.conv2d_norm)?"""
        if not isinstance(srcs, (list, tuple)):
            srcs = [srcs]
        return (not ops.stem_rows_eligible(self.spec) and not ops.s2d_eligible(self.spec, *srcs[0].data.shape[2:])
                and ops.fused_norm_ok(self.spec, srcs))

    def run_norm(self, srcs, act=ACT_NONE, residual=None, want_oct=False, want_xs=True):
        if not isinstance(srcs, (list, tuple)):
            srcs = [srcs]
        return ops.conv2d_norm(self.spec, srcs, self.packed(), act=act, residual=residual, want_oct=want_oct, want_xs=want_xs)

    def s2d_spec(self):
        if self._s2d_spec is None:
            self._s2d_spec = ops.s2d_spec(self.spec)
        return self._s2d_spec

    def packed_s2d(self):
        return self._packed('s2d', self.s2d_spec(), ops.WeightView(self.weight.detach(), s2d_c=self.spec.cin_segments[0]))

    def rows_spec(self):
        if self._rows_spec is None:
            self._rows_spec = ops.stem_rows_spec(self.spec)
        return self._rows_spec

    def packed_rows(self):
        return self._packed('rows', self.rows_spec(), ops.WeightView(self.weight.detach(), rows_c=self.spec.cin_segments[0]))


def _seq(**children):
    """Container whose children are registered under the numeric names of the reference's nn.Sequential."""
    return nn.ModuleDict({k.lstrip('_'): v for k, v in children.items()})


class ResnetBlock(nn.Module):
    """networks.py:2303-2361 (reflect padding, no dropout): x + IN(conv(ReLU(IN(conv(x)))))."""

    def __init__(self, dim):
        super().__init__()
        self.conv_block = _seq(_1=ConvLayer([dim], dim, 3, 1, 1, PAD_REFLECT),
                               _5=ConvLayer([dim], dim, 3, 1, 1, PAD_REFLECT))

    def run(self, x, tape=None, consumer=None):
        """consumer: the layer that reads the block's output when it is not another block of the trunk (the first
        up-convolution) -- the output is kept as fp32 unless that layer stages split copies."""
        c1, c5 = self.conv_block['1'], self.conv_block['5']
        if tape is None and not x.virtual and (x.oct is not None or not x.is_split_only) and c1.fused_norm_ok(x):
            # inference: each convolution normalises its own output in the epilogue (ap_conv2d_fwd_norm) -- no raw fp32 output,
            # no norm_split pass; the block's result lives as the split copy (next convolution) + channel-octet fp32 (next residual)
            y = c1.run_norm(x, act=ACT_RELU)
            return c5.run_norm(y, act=ACT_NONE, residual=x, want_oct=True)
        # (raw16: c1's raw output is read by c5's split pass, its own InstanceNorm backward and c5's weight gradient; c5's by the
        # residual pass and its InstanceNorm backward -- all of which read bf16: plain-bf16 training stores them as bf16)
        # inference: the raw outputs leave in the channel-octet layout where their one reader is a split-only norm pass (ops.trunk_octet_ok)
        dim = c1.spec.cout
        keep = consumer is not None and not consumer.stages_split(x.data.shape)
        y = conv_forward(tape, c1, x, norm_act=ACT_RELU, raw16=True, out_octet=tape is None and ops.trunk_octet_ok(dim))
        y = conv_forward(tape, c5, y, norm_act=ACT_NONE, raw16=True,
                         out_octet=tape is None and ops.trunk_octet_ok(dim, residual=x, keep_fp32=keep))
        return materialize_forward(tape, y, residual=x, consumer=consumer)


class ResnetBlock2(nn.Module):
    """networks.py:2363-2421: main branch reflect-padded, shortcut conv zero-padded, both on cat[x,l1,l2]."""

    def __init__(self, segs, dim_out):
        super().__init__()
        self.conv_block = _seq(_1=ConvLayer(segs, dim_out, 3, 1, 1, PAD_REFLECT),
                               _5=ConvLayer([dim_out], dim_out, 3, 1, 1, PAD_REFLECT))
        self.shortcut = _seq(_0=ConvLayer(segs, dim_out, 3, 1, 1, PAD_ZERO))

    def run(self, srcs, tape=None, consumer=None):
        c1, c5, sc = self.conv_block['1'], self.conv_block['5'], self.shortcut['0']
        if tape is None and c1.fused_norm_ok(srcs):
            # inference (see ResnetBlock.run): the shortcut's IN(conv) is the residual of the main branch's second convolution
            s = sc.run_norm(srcs, act=ACT_NONE, want_oct=True, want_xs=False)
            y = c1.run_norm(srcs, act=ACT_RELU)
            return c5.run_norm(y, act=ACT_NONE, residual=s, want_oct=True)
        y = conv_forward(tape, c1, srcs, norm_act=ACT_RELU, raw16=True, out_octet=tape is None and ops.trunk_octet_ok(c1.spec.cout))
        y = conv_forward(tape, c5, y, norm_act=ACT_NONE, raw16=True)
        s = conv_forward(tape, sc, srcs, norm_act=ACT_NONE)            # (the shortcut's raw output is read as a normalised RESIDUAL: fp32)
        return materialize_forward(tape, y, residual=s, consumer=consumer)


class ResnetConditionTriGenerator32_full_ifw(nn.Module):
    """networks.py:1190-1340.  ``norm_layer`` must be instance norm (the only one the model uses,
    geomgm_ifw_fore_model.py:161-209 / base_options.py norm default)."""

    def __init__(self, input_nc, output_nc, ngf=64, norm='instance', use_dropout=False, n_blocks=9,
                 padding_type='reflect', div=3, disp=1):
        super().__init__()
        if norm != 'instance':
            raise NotImplementedError('normalization layer [%s] is not available on the HIP path' % norm)
        if use_dropout:
            raise NotImplementedError('dropout is not available on the HIP path (the model sets no_dropout)')
        if padding_type != 'reflect':
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        assert n_blocks >= 0
        self.n_blocks, self.div, self.disp = n_blocks, div, disp
        self.input_nc, self.output_nc, self.ngf = input_nc, output_nc, ngf
        h = ngf // 2
        con_dim = 16
        dim = ngf * 4
        # registration order == reference state_dict order (model_tri_merge is assigned first, :1251)
        self.model_tri_merge = ConvLayer([ngf * 4, ngf * 4, ngf * 4], dim, 3, 1, 1)
        self.model_tri00 = _seq(_1=ConvLayer([input_nc], h, 7, 1, 3, PAD_REFLECT))
        self.model_tri01 = _seq(_0=ConvLayer([ngf], ngf * 2, 3, 2, 1))
        self.model_tri02 = _seq(_0=ConvLayer([ngf * 2], ngf * 4, 3, 2, 1))
        self.model_tri10 = _seq(_1=ConvLayer([input_nc], ngf, 7, 1, 3, PAD_REFLECT))
        self.model_tri11 = _seq(_0=ConvLayer([ngf], ngf, 3, 2, 1))
        self.model_tri12 = _seq(_0=ConvLayer([ngf * 2], ngf * 4, 3, 2, 1))
        self.model_tri20 = _seq(_1=ConvLayer([input_nc], ngf, 7, 1, 3, PAD_REFLECT))
        self.model_tri21 = _seq(_0=ConvLayer([ngf], ngf * 2, 3, 2, 1))
        self.model_tri22 = _seq(_0=ConvLayer([ngf * 2], ngf * 2, 3, 2, 1))
        blocks = {}
        for i in range(n_blocks):
            if (i + disp) % div == 0:
                blocks[str(i)] = ResnetBlock2([dim, con_dim, con_dim], dim)
            else:
                blocks[str(i)] = ResnetBlock(dim)
        self.model2 = nn.ModuleDict(blocks)
        self.model3 = _seq(_0=ConvLayer([ngf * 4], ngf * 2, 3, 2, 1, transposed=True, output_padding=1),
                           _3=ConvLayer([ngf * 2], ngf, 3, 2, 1, transposed=True, output_padding=1),
                           _7=ConvLayer([ngf], output_nc, 7, 1, 3, PAD_REFLECT))
        self.model_landmark_trans = _seq(_0=ConvLayer([1], 8, 3, 1, 1), _3=ConvLayer([8], con_dim, 3, 2, 1),
                                         _6=ConvLayer([con_dim], con_dim, 3, 2, 1))
        self.land1_cache_key = None      # set by a streaming caller: identity of the (constant) land1 tensor, see _run
        self._land1_cache = None

    def is_block2(self, i):
        return (i + self.disp) % self.div == 0

    def double_feature_warping(self, x, motion, flow, ifmask, level, tape=None, consumer=None):
        return warp_forward(tape, x, motion, flow, ifmask, level, consumer=consumer)

    def forward(self, input, land1, land2, motion, flow, ifmask):
        """G(input, land1, land2, motion, flow, ifmask) -> (B, output_nc, S, S)   (networks.py:1315)."""
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            return generator_apply(self, input, land1, land2, motion, flow, ifmask)
        return self.forward_inference(input, land1, land2, motion, flow, ifmask)

    def forward_inference(self, input, land1, land2, motion, flow, ifmask):
        return self._run(None, False, input, land1, land2, motion, flow, ifmask).data

    def _run(self, tape, input_needs_grad, input, land1, land2, motion, flow, ifmask):
        """The layer schedule.  tape=None: inference; otherwise every step also records its backward."""
        b = input.shape[0]
        inp = Feat(input.contiguous())
        if tape is not None and input_needs_grad:
            tape.track(inp)
        self._last_input_feat = inp if tape is not None else None
        motion, flow, ifmask = motion.contiguous(), flow.contiguous(), ifmask.contiguous()
        cf, dfw = conv_forward, self.double_feature_warping
        # inference: the encoder layers in front of the three warps write the channel-octet layout the warp kernel gathers
        # best (ops.conv2d out_octet); with a tape their outputs stay NCHW for the backward kernels
        oct = tape is None

        def branch1():
            x = cf(tape, self.model_tri00['1'], inp, norm_act=ACT_RELU, out_octet=oct)
            x = dfw(x, motion, flow, ifmask, 0, tape, self.model_tri01['0'])
            x = cf(tape, self.model_tri01['0'], x, norm_act=ACT_RELU)
            return cf(tape, self.model_tri02['0'], x, norm_act=ACT_RELU)

        def branch2():
            x = cf(tape, self.model_tri10['1'], inp, norm_act=ACT_RELU)
            x = cf(tape, self.model_tri11['0'], x, norm_act=ACT_RELU, out_octet=oct)
            x = dfw(x, motion, flow, ifmask, 1, tape, self.model_tri12['0'])
            return cf(tape, self.model_tri12['0'], x, norm_act=ACT_RELU)

        def branch3():
            x = cf(tape, self.model_tri20['1'], inp, norm_act=ACT_RELU)
            x = cf(tape, self.model_tri21['0'], x, norm_act=ACT_RELU)
            x = cf(tape, self.model_tri22['0'], x, norm_act=ACT_RELU, out_octet=oct)
            return dfw(x, motion, flow, ifmask, 2, tape, self.model_tri_merge)

        def landmarks():
            lt = self.model_landmark_trans
            key = self.land1_cache_key
            if tape is None and key is not None:
                # streaming inference: land1 is the photo's landmark map, the same for every frame of a clip -- the caller
                # (GeomCGTIFWTestModel) names it with land1_cache_key and its encoding (networks.py:1331-1332) is reused;
                # only land2 runs through the encoder.  Never on by default: a cached result must not enter a timed step.
                full = (key, tuple(land1.shape), tuple(ops.weight_key(lt[k].weight) for k in ('0', '3', '6')))
                if self._land1_cache is None or self._land1_cache[0] != full:
                    l = cf(None, lt['0'], Feat(land1.contiguous()), norm_act=ACT_RELU)
                    l = cf(None, lt['3'], l, norm_act=ACT_RELU)
                    l = cf(None, lt['6'], l, norm_act=ACT_NONE)
                    l.mean                                           # finalise the statistics once
                    self._land1_cache = (full, l)
                la = self._land1_cache[1]
                lb = cf(None, lt['0'], Feat(land2.contiguous()), norm_act=ACT_RELU)
                lb = cf(None, lt['3'], lb, norm_act=ACT_RELU)
                lb = cf(None, lt['6'], lb, norm_act=ACT_NONE)
                return la, lb
            # land1 / land2 share the encoder weights: one pass over the 2B batch
            lands = Feat(torch.cat([land1, land2], 0).contiguous())
            l = cf(tape, lt['0'], lands, norm_act=ACT_RELU)
            l = cf(tape, lt['3'], l, norm_act=ACT_RELU)
            l = cf(tape, lt['6'], l, norm_act=ACT_NONE)
            return batch_split_forward(tape, l, b)

        # ONE stream: the three branches and the landmark encoder are independent until the merge convolution, but they
        # must not run on side streams -- measured in round 3 (tools/conc_warp.py): a warp kernel that
        # shares compute units with one of the matrix kernels of another stream reads wrong data in lanes 48-63.
        x1, x2, x3 = branch1(), branch2(), branch3()
        l1, l2 = landmarks()
        x = cf(tape, self.model_tri_merge, [x1, x2, x3])
        for i in range(self.n_blocks):
            blk = self.model2[str(i)]
            nxt = self.model3['0'] if i == self.n_blocks - 1 else None
            x = blk.run([x, l1, l2], tape, consumer=nxt) if self.is_block2(i) else blk.run(x, tape, consumer=nxt)
        x = cf(tape, self.model3['0'], x, norm_act=ACT_RELU)
        x = cf(tape, self.model3['3'], x, norm_act=ACT_RELU)
        return cf(tape, self.model3['7'], x, act=ACT_TANH)


class ResnetStyle2Generator(nn.Module):
    """networks.py:573-637 (``resnet_style2_9blocks``): the static drawing generator of the streaming-inference
    model (geomcgt_ifw_test_model.py:225-227, 280-285).  Same ``state_dict`` keys as the reference
    (``model0.{1,4,7}``, ``model.0``, ``model.{3..11}.conv_block.{1,5}``, ``model.{12,15,19}``).  The reference keeps it
    frozen (``.eval()``, loaded from ``checkpoints/static/drawing.pth``), so only the forward pass is built."""

    def __init__(self, input_nc, output_nc, ngf=64, norm='instance', use_dropout=False, n_blocks=6,
                 padding_type='reflect', extra_channel=3, model0_res=0):
        super().__init__()
        if norm != 'instance':
            raise NotImplementedError('normalization layer [%s] is not available on the HIP path' % norm)
        if use_dropout:
            raise NotImplementedError('dropout is not available on the HIP path')
        if padding_type != 'reflect':
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        if model0_res != 0:
            raise NotImplementedError('model0_res != 0 is not used by any shipped configuration')
        assert n_blocks >= 0
        self.n_blocks = n_blocks
        dim = ngf * 4
        self.model0 = _seq(_1=ConvLayer([input_nc], ngf, 7, 1, 3, PAD_REFLECT),
                           _4=ConvLayer([ngf], ngf * 2, 3, 2, 1), _7=ConvLayer([ngf * 2], dim, 3, 2, 1))
        layers = {'0': ConvLayer([dim, extra_channel], dim, 3, 1, 1)}
        for i in range(n_blocks):
            layers[str(3 + i)] = ResnetBlock(dim)
        j = 3 + n_blocks
        layers[str(j)] = ConvLayer([dim], dim // 2, 3, 2, 1, transposed=True, output_padding=1)
        layers[str(j + 3)] = ConvLayer([dim // 2], dim // 4, 3, 2, 1, transposed=True, output_padding=1)
        layers[str(j + 7)] = ConvLayer([ngf], output_nc, 7, 1, 3, PAD_REFLECT)
        self.model = nn.ModuleDict(layers)

    def forward(self, input1, input2):
        """G(input1, input2) -> (B, output_nc, S, S): f1 = model0(input1); model(cat[f1, input2]) (networks.py:632)."""
        with torch.no_grad():
            cf = conv_forward
            x = cf(None, self.model0['1'], Feat(input1.contiguous()), norm_act=ACT_RELU)
            x = cf(None, self.model0['4'], x, norm_act=ACT_RELU)
            x = cf(None, self.model0['7'], x, norm_act=ACT_RELU)
            x = cf(None, self.model['0'], [x, Feat(input2.contiguous())], norm_act=ACT_RELU)
            x = materialize_forward(None, x)
            j = 3 + self.n_blocks
            for i in range(self.n_blocks):
                x = self.model[str(3 + i)].run(x, None, consumer=self.model[str(j)] if i == self.n_blocks - 1 else None)
            x = cf(None, self.model[str(j)], x, norm_act=ACT_RELU)
            x = cf(None, self.model[str(j + 3)], x, norm_act=ACT_RELU)
            return cf(None, self.model[str(j + 7)], x, act=ACT_TANH).data


class NLayerDiscriminator(nn.Module):
    """70x70 PatchGAN, networks.py:2602-2647 (n_layers=3, instance norm)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm='instance'):
        super().__init__()
        if norm != 'instance':
            raise NotImplementedError('normalization layer [%s] is not available on the HIP path' % norm)
        if n_layers != 3:
            raise NotImplementedError('only the 3-layer PatchGAN (netD=basic) is on the HIP path')
        self.input_nc, self.ndf = input_nc, ndf
        self.model = _seq(_0=ConvLayer([input_nc], ndf, 4, 2, 1), _2=ConvLayer([ndf], ndf * 2, 4, 2, 1),
                          _5=ConvLayer([ndf * 2], ndf * 4, 4, 2, 1), _8=ConvLayer([ndf * 4], ndf * 8, 4, 1, 1),
                          _11=ConvLayer([ndf * 8], 1, 4, 1, 1))

    def forward(self, input):
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            return discriminator_apply(self, input)
        return self.forward_inference(input)

    def forward_inference(self, input):
        return self._run(None, False, input).data

    def _run(self, tape, input_needs_grad, input):
        inp = Feat(input.contiguous())
        if tape is not None and input_needs_grad:
            tape.track(inp)
        self._last_input_feat = inp if tape is not None else None
        cf = conv_forward
        x = cf(tape, self.model['0'], inp, act=ACT_LRELU)
        x = cf(tape, self.model['2'], x, norm_act=ACT_LRELU)
        x = cf(tape, self.model['5'], x, norm_act=ACT_LRELU)
        x = cf(tape, self.model['8'], x, norm_act=ACT_LRELU)
        return cf(tape, self.model['11'], x)


# --------------------------------------------------------------------------- registry
def get_norm_layer(norm_type='instance'):
    """networks.py:22-39.  Returned token is what the HIP modules accept as ``norm``."""
    if norm_type in ('batch', 'none'):
        raise NotImplementedError('normalization layer [%s] is not available on the HIP path' % norm_type)
    if norm_type != 'instance':
        raise NotImplementedError('normalization layer [%s] is not found' % norm_type)
    return 'instance'


def get_scheduler(optimizer, opt):
    """networks.py:42-68."""
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.niter) / float(opt.niter_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.niter, eta_min=0)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % opt.lr_policy)


def init_weights(net, init_type='normal', init_gain=0.02):
    """networks.py:71-102: conv weights by ``init_type``, biases 0."""
    for m in net.modules():
        if isinstance(m, ConvLayer):
            if init_type == 'normal':
                nn.init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                nn.init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                nn.init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            nn.init.constant_(m.bias.data, 0.0)
    print('initialize network with %s' % init_type)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:105-120.  The reference wraps in nn.DataParallel; here multi-GPU is one process
    per GPU (animateportrait_amd.parallel), so the net goes to gpu_ids[0] unwrapped -- its state_dict
    equals what the reference saves (base_model.py:152-163 saves ``net.module``)."""
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02,
             gpu_ids=[], model0_res=0, model1_res=0, extra_channel=3, div=3, disp=1, regarch=4):
    """networks.py:123-201."""
    norm = get_norm_layer(norm_type=norm)
    if netG == 'resnet_9blocks_rcatland32_full_ifw':
        net = ResnetConditionTriGenerator32_full_ifw(input_nc, output_nc, ngf, norm=norm, use_dropout=use_dropout,
                                                     n_blocks=9, div=div, disp=disp)
    elif netG == 'resnet_style2_9blocks':                                   # networks.py:155-156
        net = ResnetStyle2Generator(input_nc, output_nc, ngf, norm=norm, use_dropout=use_dropout, n_blocks=9,
                                    model0_res=model0_res, extra_channel=extra_channel)
    elif netG in _REFERENCE_ONLY_G:
        raise NotImplementedError('Generator model name [%s] is outside the MI355X hot path '
                                  '(only %s is built)' % (netG, ', '.join(GENERATOR_NAMES)))
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[],
             n_class=3):
    """networks.py:204-247."""
    norm = get_norm_layer(norm_type=norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm=norm)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm=norm)
    elif netD in ('basic_cls', 'pixel'):
        raise NotImplementedError('Discriminator model name [%s] is outside the MI355X hot path' % netD)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


class GANLoss(nn.Module):
    """networks.py:407-473.  lsgan: mean((pred - label)^2)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        self._labels = (float(target_fake_label), float(target_real_label))   # host copies: no device sync
        if gan_mode not in ('lsgan', 'vanilla', 'wgangp'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)

    def get_target_tensor(self, prediction, target_is_real):
        t = self.real_label if target_is_real else self.fake_label
        return t.expand_as(prediction)

    def __call__(self, prediction, target_is_real):
        if self.gan_mode == 'lsgan':                      # nn.MSELoss vs the expanded label: one fused reduction
            from . import losses
            return losses.lsgan_loss(prediction, self._labels[1 if target_is_real else 0])
        if self.gan_mode == 'vanilla':
            return nn.functional.binary_cross_entropy_with_logits(
                prediction, self.get_target_tensor(prediction, target_is_real))
        return -prediction.mean() if target_is_real else prediction.mean()


class FaceLoss(nn.Module):
    """networks.py:2862-2966 around a frozen feature network: ``forward(imgs1, imgs2, bbox1=, bbox2=)`` crops the
    ``[x1, x2, y1, y2]`` windows into ones-filled squares, resizes them to ``height x width`` (bilinear,
    align_corners=True) and sums the L1 distances of the network's feature list (second argument detached).
    The reference builds Sphere20a from ``checkpoints/sphere20a_20171020.pth`` (absent from its tree); here the
    network is passed in (stock PyTorch-ROCm module), the crop + resize is one HIP launch each way."""

    def __init__(self, net, height=112, width=96):
        super().__init__()
        self.net = net.eval()
        self.height, self.width = height, width

    def crop_head_bbox(self, imgs, bboxs):                                          # :2946-2966
        from . import losses
        n, c = imgs.shape[:2]
        win = bboxs if (torch.is_tensor(bboxs) and bboxs.is_cuda and bboxs.dtype == torch.int32) else \
            losses.windows_to_device(bboxs, n, imgs.device)
        # the reference fills a 3-channel box from the image's channels; a 1-channel drawing was repeated x3 by the
        # caller (geomgm_ifw_fore_model.py:745-750) -- reading channel 0 three times is the same tensor
        return losses.crop_resize(imgs, win, (0, 0, 0) if c == 1 else (0, 1, 2), (self.height, self.width),
                                  losses.RESIZE_BILINEAR_AC)

    def compute_loss(self, img1, img2):                                            # :2926-2940
        f1 = self.net(img1)
        with torch.no_grad():
            f2 = self.net(img2)
        loss = 0.0
        for a, b in zip(f1, f2):
            loss = loss + nn.functional.l1_loss(a, b.detach())
        return loss

    def forward(self, imgs1, imgs2, kps1=None, kps2=None, bbox1=None, bbox2=None):
        if kps1 is not None or kps2 is not None or bbox1 is None or bbox2 is None:
            raise NotImplementedError('FaceLoss on the HIP path takes bbox1 / bbox2 (the only form the model uses, '
                                      'geomgm_ifw_fore_model.py:745-752)')
        return self.compute_loss(self.crop_head_bbox(imgs1, bbox1), self.crop_head_bbox(imgs2, bbox2))
