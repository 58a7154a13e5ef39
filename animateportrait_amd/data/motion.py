"""Landmark -> motion grid on the device: ``cal_motion256`` of the reference's data layer
(Module2/data/umlvd_ifw_dataset.py:60-74, umlvdfw_test_dataset.py:67-81), which builds the ``warp_motion`` input of the
generator with ``scipy.interpolate.griddata(destination, source, 256x256 grid, method='linear')`` on the CPU for every
frame (~50 ms -- two orders of magnitude more than the generator needs for the frame).

Same definition here: Delaunay triangulation of the 68 destination landmarks + the reference's 8 border points
(``scipy.spatial.Delaunay``, the triangulation griddata itself builds; 76 points, a fraction of a millisecond on the
host), then ONE kernel launch (``ap_motion_grid``) rasterises the piecewise-linear map for the whole batch and writes
the normalised ``(N, S, S, 2)`` grid where the generator reads it.  SURVEY.md section 8f, row N3.
"""
import ctypes

import numpy as np
import torch

from .. import _capi as C

# umlvd_ifw_dataset.py:62 (row, col); the repeated corners are the reference's
EDGES = np.array([[0, 0], [255, 255], [0, 255], [255, 0], [0, 255], [255, 0], [255, 255], [255, 255]], dtype=np.float64)


def triangulate(dest):
    """Delaunay simplices (T, 3) int32 of the (P, 2) destination points, as griddata(linear) builds them."""
    from scipy.spatial import Delaunay
    return np.ascontiguousarray(Delaunay(dest).simplices.astype(np.int32))


def cal_motion256(lm2d0, lm2d, device=None, size=256):
    """lm2d0 / lm2d: source / destination landmarks, (68, 2) or (N, 68, 2), as (x, y) pixels (the txt files).
    Returns the (N, size, size, 2) float32 motion grid on ``device`` (what the reference's dataset yields as
    ``warp_motion``, umlvdfw_test_dataset.py:149-151)."""
    device = torch.device(device if device is not None else 'cuda:0')
    if device.type != 'cuda':
        raise RuntimeError('animateportrait_amd: cal_motion256 rasterises on the MI355X; there is no CPU path')
    # (C-contiguous copies: an expanded / broadcast view would carry its zero strides through every step below and
    # reach the kernel as a non-contiguous device tensor)
    a0 = np.array(lm2d0.cpu() if torch.is_tensor(lm2d0) else lm2d0, dtype=np.float64, order='C')
    a1 = np.array(lm2d.cpu() if torch.is_tensor(lm2d) else lm2d, dtype=np.float64, order='C')
    if a0.ndim == 2:
        a0, a1 = a0[None], a1[None]
    n = a0.shape[0]
    edges = EDGES * ((size - 1) / 255.0)
    dst = np.concatenate([a1[:, :, [1, 0]], np.broadcast_to(edges, (n,) + edges.shape)], 1)     # (row, col)
    src = np.concatenate([a0[:, :, [1, 0]], np.broadcast_to(edges, (n,) + edges.shape)], 1)
    tris = [triangulate(dst[i]) for i in range(n)]
    tmax = max(t.shape[0] for t in tris)
    tri = np.full((n, tmax, 3), -1, dtype=np.int32)
    for i, t in enumerate(tris):
        tri[i, :t.shape[0]] = t
    pts = torch.from_numpy(np.ascontiguousarray(dst, dtype=np.float32)).to(device).contiguous()
    val = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).to(device).contiguous()
    trid = torch.from_numpy(np.ascontiguousarray(tri)).to(device).contiguous()
    out = torch.empty((n, size, size, 2), dtype=torch.float32, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    C.check(C.lib().ap_motion_grid(ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(val.data_ptr()),
                                   ctypes.c_void_p(trid.data_ptr()), n, dst.shape[1], tmax, size,
                                   ctypes.c_void_p(out.data_ptr()), stream), 'motion_grid')
    return out
