"""Oracle (test infrastructure): feature-warping gathers of the generator.

Two restatements of each op are kept side by side:

* ``*_torch``  : the same ``torch.nn.functional`` call the reference makes.
* ``*_formula``: the explicit index arithmetic, which
  is what the HIP kernels implement.  ``tests/test_oracle_golden.py`` checks the
  two against each other and against the reference goldens.

Reference: Module2/models/networks.py:1298-1313 (double_feature_warping),
Module2/intrinsic_flow_models/modules.py:596-625 (warp_acc_flow).
"""
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- resize ----
def resize_bilinear_ac_torch(x, size):
    """F.interpolate(mode='bilinear', align_corners=True) (networks.py:1301-1310)."""
    return F.interpolate(x, size=(size, size), mode='bilinear', align_corners=True)


def resize_bilinear_ac_formula(x, size):
    """src = dst*(N-1)/(S-1); i0=floor(src); i1=min(i0+1,N-1); lam=src-i0."""
    n, c, h, w = x.shape

    def axis(nin, nout):
        d = torch.arange(nout, dtype=torch.float32)
        scale = (nin - 1) / (nout - 1) if nout > 1 else 0.0
        src = d * scale
        i0 = src.floor().long().clamp(max=nin - 1)
        i1 = (i0 + 1).clamp(max=nin - 1)
        lam = src - i0.float()
        return i0, i1, lam

    y0, y1, ly = axis(h, size)
    x0, x1, lx = axis(w, size)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly)[:, None] + bot * ly[:, None]


# ----------------------------------------------------------- grid_sample ----
def grid_sample_torch(x, grid):
    """F.grid_sample(x, grid) with the defaults the reference relies on
    (bilinear, zeros, align_corners=False) (networks.py:1311)."""
    return F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def grid_sample_formula(x, grid):
    """ix = ((gx+1)*W-1)/2, iy likewise; 4 taps, out-of-range taps read 0."""
    n, c, h, w = x.shape
    gx, gy = grid[..., 0], grid[..., 1]
    ix = ((gx + 1) * w - 1) / 2
    iy = ((gy + 1) * h - 1) / 2
    x0 = ix.floor()
    y0 = iy.floor()
    wx1 = ix - x0
    wy1 = iy - y0
    wx0 = 1 - wx1
    wy0 = 1 - wy1
    out = torch.zeros(n, c, grid.shape[1], grid.shape[2], dtype=x.dtype)
    flat = x.reshape(n, c, h * w)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = (x0 + dx).long()
            yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).reshape(n, 1, -1).expand(n, c, -1)
            v = flat.gather(2, idx).reshape(n, c, grid.shape[1], grid.shape[2])
            out = out + v * (wy * wx * ok)[:, None]
    return out


# --------------------------------------------------------- warp_acc_flow ----
def _flow_to_grid(flow):
    """modules.py:606-618: meshgrid + flow, then 2*g/max(size-1,1)-1."""
    n, _, h, w = flow.shape
    xx = torch.arange(w, dtype=torch.float32).view(1, 1, w).expand(n, h, w)
    yy = torch.arange(h, dtype=torch.float32).view(1, h, 1).expand(n, h, w)
    gx = 2.0 * (xx + flow[:, 0]) / max(w - 1, 1) - 1.0
    gy = 2.0 * (yy + flow[:, 1]) / max(h - 1, 1) - 1.0
    return torch.stack([gx, gy], dim=-1)


def warp_acc_flow(x, flow, mask=None, mask_value=-1.0, formula=False):
    """modules.py:596-625.  flow channel 0 = dx, 1 = dy in pixels of x's grid."""
    grid = _flow_to_grid(flow)
    out = grid_sample_formula(x, grid) if formula else grid_sample_torch(x, grid)
    if mask is not None:
        out = torch.where(mask > 0.5, out, torch.full((), mask_value, dtype=out.dtype))
    return out


# ------------------------------------------------ double_feature_warping ----
def double_feature_warping(x, motion, flow, ifmask, level, formula=False):
    """networks.py:1298-1313.  level 0/1/2 works at 256/128/64 px: motion
    (B,256,256,2), flow (B,2,256,256) and ifmask (B,1,256,256) are resized with
    align_corners=True (flow pre-divided by 2**level), then
    cat[grid_sample(x, motion), warp_acc_flow(x, flow, mask)]."""
    rs = resize_bilinear_ac_formula if formula else resize_bilinear_ac_torch
    if level in (1, 2):
        size = 128 if level == 1 else 64
        motion = rs(motion.permute(0, 3, 1, 2), size).permute(0, 2, 3, 1)
        flow = rs(flow / (2 ** level), size)
        ifmask = rs(ifmask, size)
    gs = grid_sample_formula if formula else grid_sample_torch
    x1 = gs(x, motion)
    x2 = warp_acc_flow(x, flow, mask=ifmask, formula=formula)
    return torch.cat([x1, x2], 1)


# ------------------------------------------------------------ reflection ----
def reflect_index(i, n):
    """nn.ReflectionPad2d index rule: i<0 -> -i ; i>=n -> 2(n-1)-i."""
    if i < 0:
        return -i
    if i >= n:
        return 2 * (n - 1) - i
    return i
