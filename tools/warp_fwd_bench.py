"""Forward warp kernel at the generator's three levels (B = 16, ngf = 64), NCHW vs channel-octet input, and the three producer
layers with NCHW vs octet output:  python tools/warp_fwd_bench.py [B]"""
import sys
import torch
sys.path.insert(0, '.')
from animateportrait_amd import ops
from animateportrait_amd.networks import ConvLayer
from animateportrait_amd.synthetic import make_generator_inputs


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device('cuda:0')
    d = make_generator_inputs(b, seed=3)
    mo, fl, mk = d['motion'].to(dev), d['flow'].to(dev), d['ifmask'].to(dev)
    s = mo.shape[1]
    for level, c in ((0, 32), (1, 64), (2, 128)):
        h = s >> level
        x = torch.randn(b, c, h, h, device=dev)
        m = x.mean((2, 3)).reshape(-1)
        r = 1.0 / torch.sqrt(x.var((2, 3), unbiased=False).reshape(-1) + 1e-5)
        xo = x.view(b, c // 8, 8, h * h).permute(0, 1, 3, 2).contiguous()

        def feat(octet):
            if not octet:
                return ops.Feat(x, m, r, ops.ACT_RELU)
            f = ops.Feat(torch.empty(1, device=dev).expand(x.shape), m, r, ops.ACT_RELU)
            f.oct = xo
            return f
        kw = dict(emit_xs=True, keep_fp32=False, s2d=level < 2 and "nos2d" not in sys.argv)
        t0 = timeit(lambda: ops.warp_concat(feat(False), mo, fl, mk, level, **kw))
        t1 = timeit(lambda: ops.warp_concat(feat(True), mo, fl, mk, level, **kw))
        # checksum of the octet path's split output: equal between the default gather and APAMD_WARP_GATHER=quad
        chk = ops.warp_concat(feat(True), mo, fl, mk, level, emit_xs=False).data
        print('   octet-input fp32 output: sum %.10e  |sum| %.10e (gather: %s)' % (float(chk.double().sum()), float(chk.double().abs().sum()),
                                                                                   __import__('os').environ.get('APAMD_WARP_GATHER', 'lane')))
        inb, outb = x.numel() * 4 / 1e6, 2 * x.numel() * 4 / 1e6
        print('warp level %d C=%3d %3dx%3d: NCHW %.1f us, octet %.1f us  (alg. %.0f MB in + %.0f MB out: %.2f TB/s)'
              % (level, c, h, h, t0, t1, inb, outb, (inb + outb) / t1))        # MB / us = TB/s
    torch.manual_seed(0)
    for name, cin, cout, k, stride, pad, pm, h in (('tri00 stem 3->32 7x7', 3, 32, 7, 1, 3, ops.PAD_REFLECT, 256),
                                                  ('tri11 64->64 3x3 s2', 64, 64, 3, 2, 1, ops.PAD_ZERO, 256),
                                                  ('tri22 128->128 3x3 s2', 128, 128, 3, 2, 1, ops.PAD_ZERO, 128)):
        l = ConvLayer([cin], cout, k, stride, pad, pm).to(dev)
        torch.nn.init.normal_(l.weight, 0.0, 0.02)
        x = torch.randn(b, cin, h, h, device=dev)
        f = ops.Feat(x)
        if k == 3:
            f = ops.presplit_s2d(f)
            f0 = ops.Feat(x)
            f0.s2d = f
            f = f0
        t0 = timeit(lambda: l.run([f], norm_act=ops.ACT_RELU))
        t1 = timeit(lambda: l.run([f], norm_act=ops.ACT_RELU, out_octet=True))
        print('%s: NCHW %.1f us, octet %.1f us' % (name, t0, t1))


if __name__ == '__main__':
    main()
