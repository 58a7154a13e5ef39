"""Loss reductions, mask compositing, aux-net glue and landmark rasterisers on the MI355X (csrc/losses.hip).

Each differentiable piece is one ``torch.autograd.Function`` whose forward and backward are single HIP launches;
the reference runs them as chains of ATen elementwise ops:
``GANLoss`` lsgan (Module2/models/networks.py:429-430, 455-473), ``L1Loss`` and the lip-line mean
(Module2/models/geomgm_ifw_fore_model.py:715-739), the compositing formulas (:523-543) and ``BaseModel.masked``
(base_model.py:238-247), the window crop + resize in front of MobileFaceNet (``get_lm`` :390-410) and of Sphere20a
(``FaceLoss.crop_head_bbox`` networks.py:2946-2966), ``kp_to_map`` / the tail of ``flow_network_warp`` (:19-84)
and the ``cv2.circle`` landmark maps of the data layer (data/umlvdfw_test_dataset.py:34-41).
"""
import ctypes

import torch

from . import _capi as C
from .ops import _ptr, _stream, _require_device

RED_SQDIFF, RED_L1, RED_WMEAN = 0, 1, 2
COMP_MASK0, COMP_FORE, COMP_BLEND, COMP_COPY = 0, 1, 2, 3
RESIZE_BILINEAR_AC, RESIZE_BICUBIC = 0, 1

_WS = {}


def _workspace(dev):
    """Per-device scratch of the two-stage reductions (stream-ordered reuse: every use is write-then-read inside one
    call on the current stream)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(C.lib().ap_reduce_workspace_floats(), dtype=torch.float32, device=dev)
        _WS[key] = ws
    return ws


class _ReduceMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, op, c, weight):
        a = a.contiguous()
        _require_device(a, 'loss operand')
        if b is not None:
            b = b.contiguous()
            _require_device(b, 'loss operand')
            if b.shape != a.shape:
                raise ValueError('loss operands of shapes %s / %s' % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty((), dtype=torch.float32, device=a.device)
        C.check(C.lib().ap_reduce_mean(op, _ptr(a), _ptr(b), float(c), a.numel(), float(weight), _ptr(_workspace(a.device)),
                                       _ptr(out), _stream()), 'reduce_mean')
        ctx.save_for_backward(a, b)
        ctx.args = (op, float(c), float(weight))
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        op, c, weight = ctx.args
        ga = gb = None
        if ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and op == RED_L1):
            ga = torch.empty_like(a)
            C.check(C.lib().ap_reduce_mean_bwd(op, _ptr(a), _ptr(b), c, a.numel(), weight, _ptr(gout.contiguous()),
                                               _ptr(ga), _stream()), 'reduce_mean_bwd')
            if ctx.needs_input_grad[1] and op == RED_L1:
                gb = -ga
            if not ctx.needs_input_grad[0]:
                ga = None
        elif ctx.needs_input_grad[1]:
            raise NotImplementedError('gradient w.r.t. the weight map of a weighted mean is not defined on the HIP path')
        return ga, gb, None, None, None


def lsgan_loss(pred, target_value, weight=1.0):
    """weight * mean((pred - target_value)^2): GANLoss('lsgan')(pred, is_real) with its 1.0 / 0.0 label."""
    return _ReduceMean.apply(pred, None, RED_SQDIFF, target_value, weight)


def l1_loss(a, b, weight=1.0):
    """weight * nn.L1Loss()(a, b)."""
    return _ReduceMean.apply(a, b, RED_L1, 0.0, weight)


def weighted_mean(a, w, add=0.0, weight=1.0):
    """weight * mean((a + add) * w); w is a constant map (the lip-line mask)."""
    return _ReduceMean.apply(a, w.detach(), RED_WMEAN, add, weight)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, m, s, mode, append):
        a, m = a.contiguous(), m.contiguous()
        n, c, h, w = a.shape
        _require_device(a, 'composite image')
        _require_device(m, 'composite mask')
        if m.shape != (n, 1, h, w):
            raise ValueError('composite: mask of shape %s for an image of shape %s' % (tuple(m.shape), tuple(a.shape)))
        cs = 0
        if mode == COMP_BLEND:
            s = s.contiguous()
            _require_device(s, 'composite background')
            cs = s.shape[1]
            if s.shape[0] != n or s.shape[2:] != a.shape[2:] or cs not in (1, c):
                raise ValueError('composite: background of shape %s' % (tuple(s.shape),))
        out = torch.empty((n, c + int(append), h, w), dtype=torch.float32, device=a.device)
        C.check(C.lib().ap_mask_composite(_ptr(a), _ptr(m), _ptr(s) if mode == COMP_BLEND else None, n, c, cs, h * w, mode,
                                          int(append), _ptr(out), _stream()), 'mask_composite')
        ctx.save_for_backward(m)
        ctx.args = (mode, c)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        mode, c = ctx.args
        ga = None
        if ctx.needs_input_grad[0]:
            g = g.contiguous()
            n, gc, h, w = g.shape
            ga = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
            C.check(C.lib().ap_mask_composite_bwd(_ptr(g), _ptr(m), n, c, gc, h * w, mode, _ptr(ga), _stream()),
                    'mask_composite_bwd')
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError('composite: gradients w.r.t. the mask / background are not defined on the HIP path '
                                      '(every caller passes constants, geomgm_ifw_fore_model.py:523-557)')
        return ga, None, None, None, None


def fore_composite(x, mask):
    """((x/2+.5)*mask + 1 - mask)*2 - 1 (geomgm_ifw_fore_model.py:523-527)."""
    return _Composite.apply(x, mask.detach(), None, COMP_FORE, False)


def bg_blend(fake, static, mask):
    """((fake/2+.5)*mask + (static/2+.5)*(1-mask))*2 - 1 (geomgm_ifw_fore_model.py:541, 543)."""
    return _Composite.apply(fake, mask.detach(), static.detach(), COMP_BLEND, False)


def masked(a, mask, mask_type):
    """BaseModel.masked (base_model.py:238-247)."""
    if mask_type not in (0, 1, 2, 3):
        raise ValueError('mask_type %r' % (mask_type,))
    mode = (COMP_MASK0, COMP_FORE, COMP_COPY, COMP_FORE)[mask_type]
    return _Composite.apply(a, mask.detach(), None, mode, mask_type >= 2)


def axpy_(dst, src, alpha=1.0):
    """dst += alpha * src on flat fp32 buffers (gradient accumulation into an optimiser's flat buffer)."""
    if dst.numel() != src.numel():
        raise ValueError('axpy: %d vs %d elements' % (dst.numel(), src.numel()))
    C.check(C.lib().ap_axpy(_ptr(dst), _ptr(src), dst.numel(), float(alpha), _stream()), 'axpy')
    return dst


def windows_to_device(win, n, device):
    """(n, 4) int windows [x1, x2, y1, y2] (CPU tensor / list, as the data layer hands them over) -> validated int32
    device tensor.  The box is (x2-x1)^2; the reference's slice assignment fails when the window is taller than
    that (geomgm_ifw_fore_model.py:399-401), so that is an error here as well."""
    w = torch.as_tensor(win).detach().to('cpu', torch.int32).reshape(-1, 4)
    if w.shape[0] == 1 and n > 1:
        w = w.expand(n, 4)
    if w.shape[0] != n:
        raise ValueError('expected %d windows, got %d' % (n, w.shape[0]))
    if bool((w[:, 1] <= w[:, 0]).any()) or bool(((w[:, 3] - w[:, 2]) > (w[:, 1] - w[:, 0])).any()):
        raise ValueError('window [x1, x2, y1, y2] must have x2 > x1 and y2 - y1 <= x2 - x1: %s' % w.tolist())
    return w.contiguous().to(device)


def _chmap(channels):
    v = 0
    for k, c in enumerate(channels):
        v |= int(c) << (8 * k)
    return v


class _CropResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, win, channels, oh, ow, mode, scale, shift):
        x = x.contiguous()
        _require_device(x, 'crop_resize input')
        n, c, h, w = x.shape
        if win.dtype != torch.int32 or win.shape != (n, 4) or win.device != x.device:
            raise ValueError('crop_resize: win must be an int32 (N, 4) tensor on the input\'s device')
        oc = len(channels)
        out = torch.empty((n, oc, oh, ow), dtype=torch.float32, device=x.device)
        ctx.args = (n, c, h, w, oc, _chmap(channels), oh, ow, mode, float(scale))
        ctx.save_for_backward(win)
        C.check(C.lib().ap_crop_resize_fwd(_ptr(x), ctypes.c_void_p(win.data_ptr()), n, c, h, w, oc, ctx.args[5], oh, ow,
                                           mode, float(scale), float(shift), _ptr(out), _stream()), 'crop_resize_fwd')
        return out

    @staticmethod
    def backward(ctx, g):
        (win,) = ctx.saved_tensors
        n, c, h, w, oc, chmap, oh, ow, mode, scale = ctx.args
        gx = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
        C.check(C.lib().ap_crop_resize_bwd(_ptr(g.contiguous()), ctypes.c_void_p(win.data_ptr()), n, c, h, w, oc, chmap, oh,
                                           ow, mode, scale, _ptr(gx), _stream()), 'crop_resize_bwd')
        return gx, None, None, None, None, None, None, None


def crop_resize(x, win, channels, size, mode, scale=1.0, shift=0.0):
    """out[:, k] = scale * resize(box(x[:, channels[k]], win)) + shift; box = ones-filled (x2-x1)^2 square holding the
    window's part of the image.  win: int32 (N, 4) device tensor (``windows_to_device``)."""
    return _CropResize.apply(x, win, tuple(channels), int(size[0]), int(size[1]), mode, scale, shift)


# ----------------------------------------------------------------------------------- rasterisers (no gradient)
def kp_to_map(lm, size=224, num=7.0, den=8.0, radius=4.0):
    """kp_to_map_some((size, size), lm * num / den) (geomgm_ifw_fore_model.py:19-51, 72-73): (N, P, 2) (x, y) pixel
    landmarks -> (N, P, size, size) binary disc maps, on the device (the reference makes them with numpy on the host)."""
    lm = lm.detach().float().contiguous()
    _require_device(lm, 'landmarks')
    n, p, _ = lm.shape
    out = torch.empty((n, p, size, size), dtype=torch.float32, device=lm.device)
    C.check(C.lib().ap_kp_to_map(_ptr(lm), n, p, size, num, den, radius, _ptr(out), _stream()), 'kp_to_map')
    return out


def flow_post(flow_out, vis_out, out_size=256, gain=20.0, num=8.0, den=7.0):
    """Tail of flow_network_warp (geomgm_ifw_fore_model.py:75-83) -> (warp_flow (N,2,S',S'), res_mask (N,1,S',S'))."""
    flow_out, vis_out = flow_out.detach().float().contiguous(), vis_out.detach().float().contiguous()
    _require_device(flow_out, 'flow_out')
    _require_device(vis_out, 'vis_out')
    n, two, s, s2 = flow_out.shape
    if two != 2 or s != s2 or vis_out.shape[0] != n or vis_out.shape[2:] != flow_out.shape[2:]:
        raise ValueError('flow_post: flow %s / vis %s' % (tuple(flow_out.shape), tuple(vis_out.shape)))
    wf = torch.empty((n, 2, out_size, out_size), dtype=torch.float32, device=flow_out.device)
    rm = torch.empty((n, 1, out_size, out_size), dtype=torch.float32, device=flow_out.device)
    C.check(C.lib().ap_flow_post(_ptr(flow_out), _ptr(vis_out), n, vis_out.shape[1], s, out_size, gain, num, den, _ptr(wf),
                                 _ptr(rm), _stream()), 'flow_post')
    return wf, rm


def flow_network_warp(netF, real_A, lm1, lm2):
    """flow_network_warp(netF, real_A, lm1, lm2) (geomgm_ifw_fore_model.py:69-84): joint maps of both landmark sets ->
    the frozen FlowUnet ``netF`` (stock PyTorch-ROCm, ``(flow_out, vis_out, _, _) = netF(x)``) -> masked, rescaled,
    resized flow and mask.  ``real_A`` only fixes the device (its 224^2 resize at :71 is unused there too)."""
    with torch.no_grad():
        j = kp_to_map(torch.cat([lm1.to(real_A.device), lm2.to(real_A.device)], 1))     # == cat([j1, j2], 1)
        flow_out, vis_out = netF(j)[:2]
        return flow_post(flow_out, vis_out, real_A.shape[-1])


def landmark_discs(lm, height, width, radius=3, lo=-1.0, hi=1.0):
    """draw2(height, width, lands, radius, op=0) (data/umlvdfw_test_dataset.py:34-41) for a batch: (N, P, 2) (x, y)
    -> (N, 1, height, width) in {lo, hi}."""
    lm = lm.detach().float().contiguous()
    _require_device(lm, 'landmarks')
    n, p, _ = lm.shape
    out = torch.empty((n, 1, height, width), dtype=torch.float32, device=lm.device)
    C.check(C.lib().ap_landmark_discs(_ptr(lm), n, p, height, width, radius, lo, hi, _ptr(out), _stream()),
            'landmark_discs')
    return out


def lip_line_mask(lands, segments, size, thickness):
    """getlipline (geomgm_ifw_fore_model.py:507-515) for a batch: lands (N, P, 2) device (x, y) -> (N, 1, size, size) in
    {0, 1}: the union of cv2.line(..., thickness) over ``segments`` (pairs of landmark indices), OpenCV's integer rule."""
    lands = lands.detach().float().contiguous()
    _require_device(lands, 'landmarks')
    n, p, _ = lands.shape
    k = len(segments)
    a = (ctypes.c_int32 * k)(*[int(s[0]) for s in segments])
    b = (ctypes.c_int32 * k)(*[int(s[1]) for s in segments])
    out = torch.empty((n, 1, size, size), dtype=torch.float32, device=lands.device)
    C.check(C.lib().ap_lip_line_mask(_ptr(lands), n, p, a, b, k, size, int(thickness), _ptr(out), _stream()), 'lip_line_mask')
    return out


def circle_rows(radius):
    """Row half-widths hw[0..radius] of the filled cv2.circle the rasteriser draws (host-side, no GPU needed)."""
    hw = (ctypes.c_int32 * (radius + 1))()
    C.check(C.lib().ap_circle_rows(radius, hw), 'circle_rows')
    return list(hw)
