"""Oracle (test infrastructure): polyharmonic-spline (order 2, TPS-like) sparse image warp.

Restates Module2/models/sparse_image_warp.py:35-58 and callees, batched
("reference b=1 applied per sample"; the reference itself only runs b=1, its
bmm at :201-203 does not broadcast).  Layout follows the reference: image is
NHWC, control points are (row, col).
"""
import torch


def phi2(r):
    """sparse_image_warp.py:157-175, order 2: 0.5*r*log(max(r,1e-10)) on SQUARED distances."""
    return 0.5 * r * torch.log(torch.clamp(r, min=1e-10))


def cross_sq_dist(x, y):
    """:134-154 -- |x|^2 - 2 x.y + |y|^2 in fp32 (the cancellation is part of the numerics)."""
    xn = (x * x).sum(-1).unsqueeze(2)
    yn = (y * y).sum(-1).unsqueeze(1)
    return xn - 2 * torch.bmm(x, y.transpose(1, 2)) + yn


def solve_interpolation(c, f):
    """:93-132.  c (b,n,2) control points, f (b,n,k) values -> w (b,n,k), v (b,3,k).
    The reference's bottom-right block is 0*randn/1e10 == exact zeros."""
    b, n, d = c.shape
    k = f.shape[-1]
    a = phi2(cross_sq_dist(c, c))
    bm = torch.cat([c, torch.ones_like(c[:, :, :1])], 2)
    left = torch.cat([a, bm.transpose(1, 2)], 1)
    right = torch.cat([bm, torch.zeros(b, d + 1, d + 1, dtype=c.dtype)], 1)
    lhs = torch.cat([left, right], 2)
    rhs = torch.cat([f, torch.zeros(b, d + 1, k, dtype=c.dtype)], 1)
    x = torch.linalg.solve(lhs, rhs)  # torch.solve(rhs, lhs) in torch 1.8 (:125): LU with partial pivoting
    return x[:, :n], x[:, n:]


def apply_interpolation(q, c, w, v):
    """:186-217.  q (b,m,2) query points -> (b,m,k)."""
    rbf = torch.bmm(phi2(cross_sq_dist(q, c)), w)
    qp = torch.cat([q, torch.ones_like(q[..., :1])], 2)
    return rbf + torch.bmm(qp, v)


def flat_grid(h, w, dtype=torch.float32):
    """:71-75: (row, col) of every pixel, row-major."""
    yy, xx = torch.meshgrid(torch.arange(h, dtype=dtype), torch.arange(w, dtype=dtype), indexing='ij')
    return torch.stack([yy, xx], -1).reshape(h * w, 2)


def interpolate_bilinear(img, q):
    """:267-361.  img (b,h,w,c); q (b,m,2) as (row, col).  floor clamped to
    [0,size-2], alpha clamped to [0,1] (edge replicate)."""
    b, h, w, ch = img.shape
    fl, al = [], []
    for dim, size in ((0, h), (1, w)):
        qq = q[..., dim]
        f = torch.clamp(torch.floor(qq), min=0.0, max=float(size - 2))
        fl.append(f.long())
        al.append(torch.clamp(qq - f, 0.0, 1.0).unsqueeze(2))
    flat = img.reshape(b, h * w, ch)

    def g(y, x):
        idx = (y * w + x).unsqueeze(2).expand(b, -1, ch)
        return flat.gather(1, idx)

    tl = g(fl[0], fl[1])
    tr = g(fl[0], fl[1] + 1)
    bl = g(fl[0] + 1, fl[1])
    br = g(fl[0] + 1, fl[1] + 1)
    top = al[1] * (tr - tl) + tl
    bot = al[1] * (br - bl) + bl
    return al[0] * (bot - top) + top


def dense_image_warp(img, flow):
    """:220-264: out(q) = bilinear(img, q - flow(q))."""
    b, h, w, ch = img.shape
    q = flat_grid(h, w, img.dtype).unsqueeze(0) - flow.reshape(b, h * w, 2)
    return interpolate_bilinear(img, q).reshape(b, h, w, ch)


def sparse_image_warp(img, src, dst):
    """:35-58.  img (b,h,w,c) NHWC; src/dst (b,n,2) (row,col).
    Returns (warped (b,h,w,c), dense_flow (b,h,w,2))."""
    b, h, w, _ = img.shape
    wts, v = solve_interpolation(dst, dst - src)
    q = flat_grid(h, w, img.dtype).unsqueeze(0).expand(b, -1, -1)
    flow = apply_interpolation(q, dst, wts, v).reshape(b, h, w, 2)
    return dense_image_warp(img, flow), flow
