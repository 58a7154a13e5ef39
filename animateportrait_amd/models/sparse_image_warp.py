"""sparse_image_warp on the MI355X -- same call as Module2/models/sparse_image_warp.py:35-58.

``sparse_image_warp(img_tensor, source_control_point_locations, dest_control_point_locations)`` with
``img_tensor`` NHWC and control points (row, col), returning ``(warped NHWC, dense_flows (b,h,w,2))``.
Batched (the reference only works for b=1).  Order-2 spline only (the only order the model uses);
no gradient is defined: every call site of the train step detaches the result or warps a constant
(geomgm_ifw_fore_model.py:537-565, 738-739).
"""
import ctypes
import os

import torch

from .. import _capi as C
from ..ops import _ptr, _stream, _require_device


# status words of solves not yet inspected: reading one is a device sync, so the check is deferred to the next natural
# sync point (BaseModel.get_current_losses, the tests) unless APAMD_TPS_CHECK=1 asks for it after every call
_PENDING = []
_PENDING_MAX = 256


def check_status():
    """Raise if any spline system solved since the last check was singular (duplicate / collinear control points).
    The reference fails loudly there (torch.solve error -> pdb, sparse_image_warp.py:124-128); ap_tps_solve flags it
    in a device word and writes zero coefficients, which would otherwise be a silent identity warp."""
    global _PENDING
    pending, _PENDING = _PENDING, []
    if pending and int(torch.stack(pending).max()) != 0:
        raise RuntimeError('sparse_image_warp: singular spline system (duplicate or collinear control points); '
                           'the affected warps were written as identity')


def sparse_image_warp(img_tensor, source_control_point_locations, dest_control_point_locations,
                      interpolation_order=2, regularization_weight=0.0, num_boundaries_points=0, return_flow=True):
    if interpolation_order != 2 or regularization_weight != 0.0 or num_boundaries_points != 0:
        raise NotImplementedError('only interpolation_order=2 without regularisation is on the HIP path')
    img = img_tensor.detach()
    b, h, w, c = img.shape
    src = source_control_point_locations.detach().float().contiguous()
    dst = dest_control_point_locations.detach().float().contiguous()
    n = src.shape[1]
    if src.shape != (b, n, 2) or dst.shape != (b, n, 2):
        raise ValueError('control points must be (b, n, 2); got %s / %s' % (tuple(src.shape), tuple(dst.shape)))
    nchw = img.permute(0, 3, 1, 2).contiguous()
    for t, name in ((nchw, 'img_tensor'), (src, 'source points'), (dst, 'dest points')):
        _require_device(t, name)
    coef = torch.empty((b, n + 3, 2), dtype=torch.float32, device=img.device)
    status = torch.zeros((), dtype=torch.int32, device=img.device)
    lib = C.lib()
    C.check(lib.ap_tps_solve(_ptr(src), _ptr(dst), b, n, _ptr(coef), ctypes.c_void_p(status.data_ptr()), _stream()),
            'tps_solve')
    _PENDING.append(status)
    if os.environ.get('APAMD_TPS_CHECK') == '1' or len(_PENDING) >= _PENDING_MAX:
        check_status()
    out = torch.empty_like(nchw)
    flow = torch.empty((b, h, w, 2), dtype=torch.float32, device=img.device) if return_flow else None
    C.check(lib.ap_tps_warp(_ptr(nchw), _ptr(dst), _ptr(coef), b, n, c, h, w, _ptr(out), _ptr(flow), _stream()),
            'tps_warp')
    return out.permute(0, 2, 3, 1), flow


def warp_nchw(img, src_rc, dst_rc):
    """Convenience for the model: NCHW in, NCHW out, no flow."""
    return sparse_image_warp(img.permute(0, 2, 3, 1), src_rc, dst_rc, return_flow=False)[0].permute(0, 3, 1, 2).contiguous()
