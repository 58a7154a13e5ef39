#!/usr/bin/env python3
"""What the three frozen aux networks of the train step cost on the MI355X at the step's shapes (B = 16), and what the
torch-level levers buy: BatchNorm folding (frozen, eval), channels_last, bf16 autocast, one hipGraph per call.
    python tools/aux_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animateportrait_amd import aux_nets

dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def fold_bn(net):
    from torch.nn.utils.fusion import fuse_conv_bn_eval
    import copy
    net = copy.deepcopy(net).eval()

    def walk(m):
        names = list(m._modules.keys())
        for a, b in zip(names, names[1:]):
            ma, mb = m._modules[a], m._modules[b]
            if isinstance(ma, torch.nn.Conv2d) and isinstance(mb, torch.nn.BatchNorm2d):
                m._modules[a] = fuse_conv_bn_eval(ma, mb)
                m._modules[b] = torch.nn.Identity()
        for c in m._modules.values():
            if c is not None:
                walk(c)
    walk(net)
    return net


def bench(name, net, x, backward, out_of):
    rows = []
    for tag, n2, cl, ac in (('fp32', net, False, False), ('fp32 folded', fold_bn(net), False, False),
                            ('fp32 folded channels_last', fold_bn(net), True, False),
                            ('bf16 autocast folded', fold_bn(net), False, True),
                            ('bf16 autocast folded channels_last', fold_bn(net), True, True)):
        n2 = n2.to(dev)
        if cl:
            n2 = n2.to(memory_format=torch.channels_last)
        xx = x.clone().to(memory_format=torch.channels_last) if cl else x.clone()

        def run():
            xi = xx.detach().requires_grad_(backward)
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=ac):
                y = out_of(n2(xi))
            if backward:
                y.float().sum().backward()
            return y
        rows.append((tag, timeit(run)))
    print(name, 'backward' if backward else 'forward only')
    for tag, ms in rows:
        print('   %-40s %7.2f ms' % (tag, ms))


B = 16
with torch.no_grad():
    pass
mf = aux_nets._frozen(aux_nets.MobileFaceNet(), dev)
sp = aux_nets._frozen(aux_nets.Sphere20a(), dev)
mo = aux_nets._frozen(aux_nets.MODNet(), dev)
bench('MobileFaceNet 2 x [16,3,112,112]', mf, torch.rand(2 * B, 3, 112, 112, device=dev), True, lambda o: o[0])
bench('Sphere20a fake [16,3,112,96] (fwd + dgrad)', sp, torch.rand(B, 3, 112, 96, device=dev), True, lambda o: sum(f.float().abs().mean() for f in o))
bench('Sphere20a static [16,3,112,96] (fwd)', sp, torch.rand(B, 3, 112, 96, device=dev), False, lambda o: o[-1])
bench('MODNet [16,3,256,256] (fwd)', mo, torch.rand(B, 3, 256, 256, device=dev) * 2 - 1, False, lambda o: o[2])


# ---- one hipGraph per call: forward + backward of the landmark net (make_graphed_callables), forward of the matting net
def graphed_fwd_bwd(net, x, out_of, cl, ac):
    n2 = fold_bn(net).to(dev)
    if cl:
        n2 = n2.to(memory_format=torch.channels_last)

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.n = n2

        def forward(self, t):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=ac):
                return out_of(self.n(t)).float()
    sample = x.clone().requires_grad_(True)
    g = torch.cuda.make_graphed_callables(Wrap(), (sample,))

    def run():
        xi = x.detach().clone().requires_grad_(True)
        y = g(xi)
        y.sum().backward()
        return xi.grad
    return timeit(run)


def graphed_fwd(net, x, out_of, cl, ac):
    n2 = fold_bn(net).to(dev)
    if cl:
        n2 = n2.to(memory_format=torch.channels_last)
    static_x = x.clone().to(memory_format=torch.channels_last) if cl else x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=ac):
                out_of(n2(static_x))
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr), torch.no_grad():
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=ac):
            y = out_of(n2(static_x))

    def run():
        static_x.copy_(x)
        gr.replay()
        return y
    return timeit(run)


print('one hipGraph per call')
for cl, ac in ((False, False), (True, False), (True, True)):
    tag = ('channels_last ' if cl else '') + ('bf16 autocast' if ac else 'fp32')
    try:
        print('   MobileFaceNet fwd+bwd graphed, %-28s %7.2f ms' % (tag, graphed_fwd_bwd(mf, torch.rand(2 * B, 3, 112, 112, device=dev), lambda o: o[0], cl, ac)))
    except Exception as e:
        print('   MobileFaceNet graphed', tag, 'FAILED', repr(e)[:200])
    try:
        print('   Sphere20a fwd+dgrad graphed,  %-28s %7.2f ms' % (tag, graphed_fwd_bwd(sp, torch.rand(B, 3, 112, 96, device=dev), lambda o: sum(f.float().abs().mean() for f in o), cl, ac)))
    except Exception as e:
        print('   Sphere20a graphed', tag, 'FAILED', repr(e)[:200])
    try:
        print('   MODNet fwd graphed,           %-28s %7.2f ms' % (tag, graphed_fwd(mo, torch.rand(B, 3, 256, 256, device=dev) * 2 - 1, lambda o: o[2], cl, ac)))
    except Exception as e:
        print('   MODNet graphed', tag, 'FAILED', repr(e)[:200])

# ---- where MODNet's time goes
from torch.profiler import profile, ProfilerActivity
n2 = fold_bn(mo).to(dev).to(memory_format=torch.channels_last)
xx = (torch.rand(B, 3, 256, 256, device=dev) * 2 - 1).to(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        n2(xx, True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        n2(xx, True)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=70))
n3 = fold_bn(mf).to(dev)
x3 = torch.rand(2 * B, 3, 112, 112, device=dev)
for _ in range(3):
    xi = x3.clone().requires_grad_(True); n3(xi)[0].sum().backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    xi = x3.clone().requires_grad_(True); n3(xi)[0].sum().backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=70))
