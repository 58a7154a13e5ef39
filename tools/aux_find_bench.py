#!/usr/bin/env python3
"""Do the frozen aux networks run faster when MIOpen may time its algorithms (torch.backends.cudnn.benchmark = True: "find" mode)
instead of taking its immediate-mode pick?  One process per setting (MIOpen caches its decisions):
    AUX_BENCHMARK=0 python tools/aux_find_bench.py;  AUX_BENCHMARK=1 python tools/aux_find_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animateportrait_amd import aux_nets, standins

torch.backends.cudnn.benchmark = os.environ.get('AUX_BENCHMARK', '0') == '1'
dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


B = 16
mf = aux_nets._frozen(aux_nets.MobileFaceNet(), dev)
sp = aux_nets._frozen(aux_nets.Sphere20a(), dev)
mo = aux_nets._frozen(aux_nets.MODNet(), dev)
sl = standins.StandinLandmarkNet().to(dev)
sf = standins.StandinFaceNet().to(dev)


def fb(net, x, pick):
    def run():
        xi = x.detach().requires_grad_(True)
        o = pick(net(xi))
        (o.float().sum() if torch.is_tensor(o) else sum(t.float().sum() for t in o)).backward()
    return run


x1 = torch.rand(2 * B, 3, 112, 112, device=dev)
x2 = torch.rand(B, 3, 112, 96, device=dev)
x3 = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
print('cudnn.benchmark =', torch.backends.cudnn.benchmark)
print('  MobileFaceNet fwd+bwd [32,3,112,112]  %7.2f ms' % timeit(fb(mf, x1, lambda o: o[0])))
print('  Sphere20a fwd+bwd [16,3,112,96]       %7.2f ms' % timeit(fb(sp, x2, tuple)))
with torch.no_grad():
    print('  Sphere20a fwd [16,3,112,96]           %7.2f ms' % timeit(lambda: sp(x2)))
    print('  MODNet fwd [16,3,256,256]             %7.2f ms' % timeit(lambda: mo(x3)))
print('  stand-in landmark net fwd+bwd         %7.2f ms' % timeit(fb(sl, x1, lambda o: o[0] if isinstance(o, (tuple, list)) else o)))
print('  stand-in face net fwd+bwd             %7.2f ms' % timeit(fb(sf, x2, lambda o: tuple(o) if isinstance(o, (tuple, list)) else o)))
