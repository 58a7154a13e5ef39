"""Reverse pass of the generator / discriminator as an explicit schedule of HIP launches.

The reference relies on torch.autograd over ATen ops (``loss.backward()``,
Module2/models/geomgm_ifw_fore_model.py:586,610,634,780).  Here each network is ONE
``torch.autograd.Function`` to PyTorch; inside, the forward records a tape of layer closures and the
backward replays it in reverse, launching the data-gradient (the forward conv kernel with swapped
roles), weight-gradient, InstanceNorm/activation-backward and warp-backward kernels of libapamd.so.
PyTorch only carries the resulting tensors to the optimiser / the collective.
"""
import os

import torch

from . import ops
from .ops import Feat, ConvSpec, ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT, W_OIHW, W_IOHW


# ctx.needs_input_grad only mirrors requires_grad: a custom Function cannot see whether THIS backward call wants the
# parameter gradients (torch.autograd.grad(loss, [input]) runs the same node and drops them).  Writing them straight into
# the optimiser's flat buffer is therefore opt-in: the train step wraps its ``loss.backward()`` calls in
# ``direct_param_grads()``; everywhere else the gradients go back through autograd.
_DIRECT = 0


class direct_param_grads:
    """``with direct_param_grads(): loss.backward()`` -- the caller states that this backward accumulates into ``p.grad`` of
    every trainable parameter (plain ``.backward()`` of the train step, geomgm_ifw_fore_model.py:586,610,634,780)."""

    def __enter__(self):
        global _DIRECT
        _DIRECT += 1

    def __exit__(self, *exc):
        global _DIRECT
        _DIRECT -= 1
        return False


class Tape:
    def __init__(self):
        self.steps = []
        self.grads = {}          # id(Feat) -> list of (tensor, pad)
        self.keep = []           # keep Feats alive so ids stay unique
        self.param_grads = {}    # Parameter -> tensor
        self.block = None        # contiguous gradient block of the network (see grad_block), or None
        self.block_lo = 0

    def grad_block(self, params, needs=None):
        """When every trainable parameter of the network is a view of ONE optimiser's flat buffer (optim.FlatAdam) and
        the views are adjacent, the backward kernels write the parameter gradients straight into one contiguous block
        laid out like that range; the whole block is then added to the optimiser's flat gradient buffer with one launch
        -- instead of one AccumulateGrad add per parameter and call (~200 launches per train step).

        This bypasses autograd's own accumulation, so it is taken only when that is exactly what autograd would have
        done: the caller opted in (``direct_param_grads()``: a plain ``loss.backward()`` of the train step --
        ``torch.autograd.grad(loss, [input])`` / ``backward(inputs=...)`` must not touch the optimiser's buffers), every
        trainable parameter is wanted (``needs``), and every ``p.grad`` still IS its view of the flat gradient buffer (after
        ``module.zero_grad()`` / ``p.grad = None`` autograd would start from a fresh tensor, not add to stale sums)."""
        if _DIRECT <= 0:
            return False
        live = [p for p in params if p.requires_grad]
        if not live or any(getattr(p, '_flat_owner', None) is None for p in live):
            return False
        if needs is not None and not all(n for p, n in zip(params, needs) if p.requires_grad):
            return False
        owner = live[0]._flat_owner
        if any(p._flat_owner is not owner for p in live):
            return False
        base = owner.flat_grad.data_ptr()
        if any(p.grad is None or p.grad.data_ptr() != base + 4 * p._flat_off for p in live):
            return False
        lo = min(p._flat_off for p in live)
        hi = max(p._flat_off + p.numel() for p in live)
        if hi - lo != sum(p.numel() for p in live):
            return False
        self.block = torch.zeros(hi - lo, dtype=torch.float32, device=owner.flat_grad.device)
        self.block_lo, self.block_owner = lo, owner
        return True

    def slot(self, p):
        """Where the gradient kernel of parameter ``p`` should write (None: allocate; a second contribution to the
        same parameter is computed apart and added)."""
        if self.block is None or p in self.param_grads:
            return None
        o = p._flat_off - self.block_lo
        return self.block[o:o + p.numel()].view(p.shape)

    def flush_block(self):
        if self.block is not None:
            from . import losses
            o = self.block_owner
            losses.axpy_(o.flat_grad[self.block_lo:self.block_lo + self.block.numel()], self.block)
            self.block = None

    def track(self, f):
        self.grads[id(f)] = []
        self.keep.append(f)
        return f

    def tracked(self, f):
        return id(f) in self.grads

    def add(self, f, tensor, pad=0):
        if id(f) in self.grads:
            self.grads[id(f)].append((tensor, pad))

    def take(self, f):
        return self.grads.pop(id(f), [])

    def add_param(self, p, g):
        if p in self.param_grads:
            prev = self.param_grads[p]
            if self.block is not None:
                prev.add_(g)                     # prev is the parameter's slot of the block
            else:
                self.param_grads[p] = prev + g
        else:
            self.param_grads[p] = g

    def backward(self):
        for fn in reversed(self.steps):
            fn()
        self.steps = []


# ------------------------------------------------------------------------------ conv layer
def _dgrad_spec(layer, seg_c):
    """ConvSpec of the operator that computes the data gradient for one input segment (same arithmetic as the
    layer's forward pass)."""
    spec, fold = _dgrad_spec_geometry(layer, seg_c)
    spec.precision = layer.spec.precision
    return spec, fold


def _dgrad_spec_geometry(layer, seg_c):
    s = layer.spec
    k = s.k
    if s.transposed:      # ConvTranspose2d(k, s=2, p): gradient is a stride-2 convolution, weight read as OIHW
        return ConvSpec([s.cout], seg_c, k, 2, s.pad, PAD_ZERO, False, 0, W_OIHW, False), 0
    if s.stride == 1:     # full correlation with flipped taps; reflection-padded layers stay in padded coords
        if s.pad_mode == PAD_REFLECT:
            return ConvSpec([s.cout], seg_c, k, 1, k - 1, PAD_ZERO, False, 0, W_IOHW, True), s.pad
        return ConvSpec([s.cout], seg_c, k, 1, k - 1 - s.pad, PAD_ZERO, False, 0, W_IOHW, True), 0
    # stride 2: transposed convolution; output_padding restores the input size
    return ConvSpec([s.cout], seg_c, k, 2, s.pad, PAD_ZERO, True, 1 if k == 3 else 0, W_IOHW, False), 0


def _split_backward_plan(tape, layer, srcs, out, contribs):
    """Can the InstanceNorm backward of ``out`` write the operands of its consumers itself (ops.instnorm_bwd_split)?  Yes when every
    consumer of the gradient is a bf16 matrix kernel: the weight gradient takes a prepared M-role operand (ops.wgrad_gt_dims) and
    each data gradient stages split copies.  Returns (reduced contributions, gt_dims, want_xs, want_strip) or None."""
    s = layer.spec
    if s.transposed or s.precision == ops.PRECISION_FP32 or ops.DEFAULT_PRECISION == ops.PRECISION_FP32:
        return None
    n, c, h, w = out.data.shape
    pads = sorted(p for _, p in contribs if p > 0)
    if pads and pads[-1] > 1:
        return None
    # the decision depends on shapes and on who wants a gradient, not on values: kept per layer (a handful of C-side plan queries
    # per layer and step otherwise)
    key = (n, c, h, w, bool(pads), layer.weight.requires_grad, tuple(tape.tracked(f) for f in srcs),
           tuple(tuple(f.data.shape) for f in srcs), s.precision, ops.DEFAULT_PRECISION,
           tuple(os.environ.get(v) for v in ('APAMD_NO_INBWD_SPLIT', 'APAMD_NO_BF16X3', 'APAMD_NO_S2D')),
           ops.XS_DIRECT, all(f.xs is not None for f in srcs))
    cache = layer.__dict__.setdefault('_split_bwd_plans', {})
    if key not in cache:
        cache[key] = _split_backward_decision(tape, layer, srcs, out, bool(pads))
    dec = cache[key]
    if dec is None:
        return None
    return (ops._split_contribs(contribs),) + dec


def _split_backward_decision(tape, layer, srcs, out, folded):
    s = layer.spec
    n, c, h, w = out.data.shape
    if not ops.instnorm_bwd_split_ok(out, 1 if folded else 0):
        return None
    gt_dims = None
    wg_xs = False
    if layer.weight.requires_grad:
        # split-bf16 3x3 layers whose sources carry their forward copies: the weight gradient reads the gradient's split copy itself
        # (ops.wgrad g_xs=, ap_conv2d_wgrad_xs) -- no prepared operand; plain bf16 keeps the prepared-operand kernel (faster there)
        wg_xs = (s.precision == ops.PRECISION_BF16X3 and ops.DEFAULT_PRECISION == ops.PRECISION_BF16X3 and
                 ops.wgrad_xs_ok(s.k, s.stride, s.pad, s.pad_mode, (n, c, h, w), srcs, s.precision))
        if not wg_xs:
            gt_dims = ops.wgrad_gt_dims(s.k, s.stride, s.pad, s.pad_mode, (n, c, h, w), srcs, s.precision)
            if gt_dims is None:
                return None
    want_xs = want_strip = False
    probe = Feat(out.data)                 # a plain feature of the gradient's shape
    for i, f in enumerate(srcs):
        if tape.tracked(f):
            spec, fold_pad = _dgrad_spec(layer, s.cin_segments[i])
            if not ops.takes_split(spec, n, h, w):
                return None
            want_xs = True
            want_strip = want_strip or bool(fold_pad and ops.dgrad_strip_eligible(spec, probe))
    if gt_dims is None and not want_xs and not wg_xs:
        return None
    return gt_dims, want_xs or wg_xs, want_strip, wg_xs


def conv_backward(tape, layer, srcs, out, norm, act):
    """out: Feat produced by layer.run(srcs).  Consumes out's gradient contributions."""
    contribs = tape.take(out)
    if not contribs:
        return
    s = layer.spec
    gt = strip = g_xs = db = None
    plan = _split_backward_plan(tape, layer, srcs, out, contribs) if norm else None
    if plan is not None:
        # the gradient only feeds the bf16 matrix kernels: its producer writes their operands, no fp32 dy (ops.instnorm_bwd_split)
        red, gt_dims, want_xs, want_strip, wg_xs = plan
        gfeat, gt, strip = ops.instnorm_bwd_split(red, out, gt_dims, want_xs, want_strip)
        g_xs = gfeat.xs if wg_xs else None
        dy = None
    else:
        if out.data.dtype != torch.float32:      # (a bf16 raw output whose backward takes the fp32 route after all: convert once)
            out = Feat(out.data.float(), out._mean, out._rstd, out.act)
            contribs = [(g.float() if g.dtype != torch.float32 else g, pd) for g, pd in contribs]
        # a stem (no data gradient: its input is an image) whose weight gradient runs on the bf16 matrix pipe: dy has one reader,
        # which rounds it to bf16 -- it is stored that way (ops.instnorm_bwd out_bf16)
        dy16 = (norm and layer.weight.requires_grad and not any(tape.tracked(f) for f in srcs) and
                ops.k7_stem_wgrad_ok(s, tuple(out.data.shape), srcs))
        if norm:
            dy = ops.instnorm_bwd(contribs, out, out_bf16=dy16)
        elif layer.bias.requires_grad and act != ACT_NONE:
            # a plain layer with an activation (the PatchGAN's first): the bias gradient's block sums are formed while dy is written
            dy, db = ops.act_bwd_bias(contribs, out.data, act, tape.slot(layer.bias))
        else:
            dy = ops.act_bwd(contribs, out.data, act)
        gfeat = Feat(dy)
        # split-bf16 layers served by ap_conv2d_wgrad_xs (3x3, the PatchGAN's 4x4): the gradient's split copy -- which the data-gradient
        # convolution stages anyway (cached on the Feat) -- is the weight gradient's operand too: no operand preparation
        if (layer.weight.requires_grad and not s.transposed and s.precision == ops.PRECISION_BF16X3 and
                ops.DEFAULT_PRECISION == ops.PRECISION_BF16X3 and dy.shape[1] % 8 == 0 and
                ops.wgrad_xs_ok(s.k, s.stride, s.pad, s.pad_mode, tuple(dy.shape), srcs, s.precision)):
            g_xs = ops.presplit(gfeat, s.precision)
    # ---- weight gradient
    if layer.weight.requires_grad:
        if s.transposed:
            # dW[ci][co*k*k]: M-role = the layer input (virtual allowed), shifted tensor = dy (stride 2)
            dw = ops.wgrad(s.k, 2, s.pad, PAD_ZERO, srcs[0], [gfeat], layer.weight.shape, precision=s.precision,
                           out=tape.slot(layer.weight))
        else:
            dw = ops.wgrad(s.k, s.stride, s.pad, s.pad_mode, gfeat, srcs, layer.weight.shape, precision=s.precision,
                           out=tape.slot(layer.weight), g_t=gt, g_xs=g_xs)
        tape.add_param(layer.weight, dw)
    if layer.bias.requires_grad:
        # a bias in front of InstanceNorm has an exactly-zero gradient (it is removed by the mean subtraction)
        slot = tape.slot(layer.bias)
        if norm:
            tape.add_param(layer.bias, slot if slot is not None else torch.zeros_like(layer.bias))   # block is zero-filled
        else:
            tape.add_param(layer.bias, db if db is not None else ops.bias_grad(dy, out=slot))
    # ---- data gradients, one launch per input segment that needs one
    c0 = 0
    for i, f in enumerate(srcs):
        c = s.cin_segments[i]
        if tape.tracked(f):
            spec, fold_pad = _dgrad_spec(layer, c)
            if len(srcs) == 1 and s.cout == 1 and ops.final_dgrad_k7_ok(s, gfeat, f):
                # the generator's last layer in plain-bf16 arithmetic: the padded gradient on the bf16 matrix pipe (dgrad_k7.h)
                tape.add(f, ops.final_dgrad_k7(gfeat, layer.weight), fold_pad)
                c0 += c
                continue
            if len(srcs) == 1 and s.cout == 1 and ops.head_dgrad_ok(s, gfeat, f):
                # the PatchGAN's output layer in plain-bf16 arithmetic: its 512-channel gradient as an output stream (dgrad_k7.h)
                tape.add(f, ops.head_dgrad(gfeat, layer.weight, f.data.shape[2], f.data.shape[3]), fold_pad)
                c0 += c
                continue
            w = layer.weight.detach()
            if len(srcs) > 1:
                w = w[c0:c0 + c] if s.transposed else w[:, c0:c0 + c]       # a view: the packer takes strides
            packed = layer.packed_dgrad(i, spec, w)
            if fold_pad and ops.dgrad_strip_eligible(spec, gfeat):
                # 66-column padded gradient: two whole tile columns + a transposed 2-column strip (ops.conv2d_dgrad_strip)
                packed_t = layer.packed_dgrad((i, 'T'), spec, w.transpose(2, 3))
                # (a source whose raw output is stored as bf16 has this gradient as its ONLY contribution and its InstanceNorm
                # backward on the operand-writing route: the gradient is stored as bf16 too)
                g = ops.conv2d_dgrad_strip(spec, gfeat, packed, packed_t, strip, out_bf16=f.data.dtype == torch.bfloat16)
            else:
                g = ops.conv2d(spec, [gfeat], packed, None).data
            tape.add(f, g, fold_pad)
        c0 += c


def conv_forward(tape, layer, srcs, norm_act=None, act=ACT_NONE, out_octet=False, raw16=False):
    """raw16: the raw output may be STORED as bf16 (plain-bf16 train step; the caller vouches that every reader of it is one of
    ap_norm_apply_split, ap_instnorm_bwd_split and the padded-row operand kernel of the weight gradient: the main-branch
    convolutions of the ResNet blocks)."""
    if not isinstance(srcs, (list, tuple)):
        srcs = [srcs]
    raw16 = (raw16 and tape is not None and norm_act is not None and ops.BF16_RAW and ops.DEFAULT_PRECISION == ops.PRECISION_BF16 and
             layer.spec.precision == ops.PRECISION_BF16)
    out = layer.run(srcs, norm_act=norm_act, act=act, out_octet=out_octet and tape is None, out_bf16=raw16)
    if tape is not None:
        tape.track(out)
        norm = norm_act is not None
        tape.steps.append(lambda: conv_backward(tape, layer, srcs, out, norm, act))
    return out


def materialize_forward(tape, f, residual=None, consumer=None):
    """consumer: a ConvLayer that reads the result and is not a c -> c trunk layer; the fp32 tensor is dropped (inference)
    only if that layer, too, stages split copies (at ngf = 16 the first up-convolution, 64 -> 32, reads fp32)."""
    keep = tape is not None or (consumer is not None and not consumer.stages_split(f.data.shape))
    out = ops.materialize(f, residual=residual, keep_fp32=keep)
    if tape is not None:
        tape.track(out)

        def bwd():
            contribs = tape.take(out)
            if not contribs:
                return
            g1, pad, g2 = ops._split_contribs(contribs)
            g = g1 if (pad == 0 and g2 is None) else ops.fold_add(g1, pad, g2)
            tape.add(f, g, 0)
            if residual is not None:
                tape.add(residual, g, 0)
        tape.steps.append(bwd)
    return out


def warp_forward(tape, f, motion, flow, ifmask, level, consumer=None):
    """consumer: the ConvLayer that reads the warped concat.  When it stages split-bf16 sources the warp writes that
    copy itself; in inference (no tape) the fp32 concat is then not written at all."""
    n, _, h, w = f.data.shape
    emit_xs = consumer is not None and ops.takes_split(consumer.spec, n, h, w)
    s2d = emit_xs and len(consumer.spec.cin_segments) == 1 and ops.s2d_eligible(consumer.spec, h, w)
    out = ops.warp_concat(f, motion, flow, ifmask, level, emit_xs=emit_xs, keep_fp32=tape is not None or not emit_xs, s2d=s2d)
    if tape is not None:
        tape.track(out)

        def bwd():
            contribs = tape.take(out)
            if not contribs or not tape.tracked(f):
                return
            g1, pad, g2 = ops._split_contribs(contribs)
            g = g1 if (pad == 0 and g2 is None) else ops.fold_add(g1, pad, g2)
            tape.add(f, ops.warp_concat_bwd(g, motion, flow, ifmask, level), 0)
        tape.steps.append(bwd)
    return out


def batch_split_forward(tape, f, b):
    """l -> (l[:b], l[b:]) for the shared landmark encoder run on the stacked 2B batch."""
    l1, l2 = f.batch_slice(0, b), f.batch_slice(b, 2 * b)
    if tape is not None:
        tape.track(l1)
        tape.track(l2)

        def bwd():
            parts = []
            for part in (l1, l2):
                contribs = tape.take(part)
                g1, pad, g2 = ops._split_contribs(contribs)
                parts.append(g1 if (pad == 0 and g2 is None) else ops.fold_add(g1, pad, g2))
            tape.add(f, torch.cat(parts, 0), 0)
        tape.steps.append(bwd)
    return l1, l2


# ------------------------------------------------------------------------------ autograd.Function wrappers
def _weights_stamp(params):
    """What the backward pass assumes unchanged since forward: it reads ``layer.weight`` live (the packed copies are
    rebuilt from it), so an optimiser step in between would silently give gradients of a different network, where
    torch.autograd raises its saved-tensor version error."""
    return tuple(ops.weight_key(p) for p in params)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, n_inputs, input_needs_grad, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        tape = Tape()
        with torch.no_grad():
            out = net._run(tape, input_needs_grad, *inputs)
        ctx.tape, ctx.net, ctx.out_feat = tape, net, out
        ctx.n_inputs, ctx.params = n_inputs, params
        ctx.stamp = _weights_stamp(params)
        ctx.in_feat = getattr(net, '_last_input_feat', None)
        return out.data

    @staticmethod
    def backward(ctx, gout):
        tape = ctx.tape
        if tape is None:
            raise RuntimeError('animateportrait_amd: backward through this network a second time: the layer tape is '
                               'freed after the first pass (retain_graph is not supported); run forward again')
        if _weights_stamp(ctx.params) != ctx.stamp:
            raise RuntimeError('animateportrait_amd: a parameter of %s was modified (optimizer.step / in-place write) '
                               'between forward and backward; its gradients would belong to different weights'
                               % type(ctx.net).__name__)
        with torch.no_grad():
            direct = tape.grad_block(ctx.params, ctx.needs_input_grad[3 + ctx.n_inputs:])
            tape.add(ctx.out_feat, gout.contiguous(), 0)
            tape.backward()
            gin = [None] * ctx.n_inputs
            if ctx.in_feat is not None and tape.tracked(ctx.in_feat):
                contribs = tape.take(ctx.in_feat)
                if contribs:
                    g1, pad, g2 = ops._split_contribs(contribs)
                    gin[0] = g1 if (pad == 0 and g2 is None) else ops.fold_add(g1, pad, g2)
            if direct:       # already summed into the optimiser's flat gradient buffer (p.grad is a view of it)
                tape.flush_block()
                gp = [None] * len(ctx.params)
            else:
                gp = [tape.param_grads.get(p) for p in ctx.params]
        ctx.tape = None
        return (None, None, None, *gin, *gp)


def generator_apply(net, input, land1, land2, motion, flow, ifmask):
    """The reference generator is differentiable w.r.t. every input; this path defines the gradient w.r.t. the image
    ``input`` (the three stems' data gradients) and refuses the others instead of silently returning zeros: no caller
    on the hot path asks for them (land / motion / flow / ifmask come from the data layer and netF under no_grad,
    geomgm_ifw_fore_model.py:443-505), and ap_warp_concat_bwd has no gradient w.r.t. the sampling grid."""
    for t, name in ((land1, 'land1'), (land2, 'land2'), (motion, 'motion'), (flow, 'flow'), (ifmask, 'ifmask')):
        if t.requires_grad:
            raise NotImplementedError('animateportrait_amd generator: gradient w.r.t. %s is not defined on the HIP '
                                      'path (detach it)' % name)
    params = [p for p in net.parameters()]
    return _NetFn.apply(net, 6, bool(input.requires_grad), input, land1, land2, motion, flow, ifmask, *params)


def discriminator_apply(net, input):
    params = [p for p in net.parameters()]
    return _NetFn.apply(net, 1, bool(input.requires_grad), input, *params)
