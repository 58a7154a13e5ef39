"""CPU emulation of the precision of Winograd F(2x2,3x3) on split-bf16 operands for the generator's wide 3x3
stride-1 layers, against the exact-fp32 oracle (go / no-go input for tools/conv_wino.h).

Every eligible F.conv2d of oracle.generator (3x3, stride 1, >= 256 input channels: the 15 + 3 + 3 ResNet-block
convolutions and the merge convolution) is replaced by
    direct : xh*wh + xh*wl + xl*wh on the bf16 heads / tails (what conv_bf16x3 computes today), or
    wino   : V = B^T d B and U = G g G^T in fp32, both split into bf16 head + tail, M = the same three products
             accumulated in fp32 over the channels, Y = A^T M A in fp32.
Prints L-inf of the generator output against the unmodified oracle (budget 1e-3, bench bar 5e-4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import generator as og
from animateportrait_amd.synthetic import make_generator_inputs, generator_args

MODE = None
_conv2d = F.conv2d


def split(x):
    h = x.bfloat16().float()
    l = (x - h).bfloat16().float()
    return h, l


def prod3(xh, xl, wh, wl, f):
    return f(xh, wh) + (f(xh, wl) + f(xl, wh))


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino(x, w):
    """x (N,C,H+2,W+2) already padded, w (O,C,3,3) -> (N,O,H,W), H and W even."""
    n, c, hp, wp = x.shape
    th, tw = (hp - 2) // 2, (wp - 2) // 2
    d = x.unfold(2, 4, 2).unfold(3, 4, 2)                       # (N,C,th,tw,4,4)
    V = torch.einsum('ai,nctuij,bj->abnctu', BT, d, BT)         # (4,4,N,C,th,tw)
    U = torch.einsum('ai,ocij,bj->aboc', G, w, G)               # (4,4,O,C)
    Vh, Vl = split(V)
    Uh, Ul = split(U)
    f = lambda v, u: torch.einsum('abnctu,aboc->abnotu', v, u)
    M = prod3(Vh, Vl, Uh, Ul, f)
    return torch.einsum('ia,abnotu,jb->notiuj', AT, M, AT).reshape(n, w.shape[0], 2 * th, 2 * tw)


def patched(x, w, b=None, stride=1, padding=0, *a, **k):
    if MODE and w.shape[2:] == (3, 3) and stride == 1 and w.shape[1] >= 256:
        if padding:
            x = F.pad(x, (padding,) * 4)
        if MODE == 'direct':
            xh, xl = split(x)
            wh, wl = split(w)
            y = prod3(xh, xl, wh, wl, lambda u, v: _conv2d(u, v))
        else:
            y = wino(x, w)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return _conv2d(x, w, b, stride, padding, *a, **k)


def main():
    global MODE
    torch.set_num_threads(8)
    F.conv2d = patched
    og.F.conv2d = patched
    sd = og.init_params(og.generator_param_shapes(3, 1, 64, 9, 3, 3), seed=1234)
    for seed in (1234, 7):
        args = generator_args(make_generator_inputs(1, seed=seed))
        with torch.no_grad():
            MODE = None
            ref = og.generator_forward(sd, *args, div=3, disp=3)
            ref64 = og.generator_forward({k: v.double() for k, v in sd.items()}, *[a.double() for a in args], div=3, disp=3)
            print('seed %d  fp32 oracle vs fp64: %.3e' % (seed, float((ref.double() - ref64).abs().max())), flush=True)
            for m in ('direct', 'wino'):
                MODE = m
                y = og.generator_forward(sd, *args, div=3, disp=3)
                print('seed %d  %-6s  L-inf vs fp32 oracle %.3e   vs fp64 %.3e' % (
                    seed, m, float((y - ref).abs().max()), float((y.double() - ref64).abs().max())), flush=True)


if __name__ == '__main__':
    main()
