"""Victim = the warp kernel alone (level 2, split output), aggressor = the trunk 3x3 convolution looping on another stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animateportrait_amd import ops
from animateportrait_amd.networks import ConvLayer
from animateportrait_amd.ops import Feat
from animateportrait_amd.synthetic import make_generator_inputs, generator_args
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
LEVEL = int(os.environ.get('LEVEL', '2'))
C, S = {0: 32, 1: 64, 2: 128}[LEVEL], 256 >> LEVEL
args = [t.to(dev)[:8].contiguous() for t in generator_args(make_generator_inputs(16, seed=1234))]
motion, flow, ifmask = args[3], args[4], args[5]
if os.environ.get('SMALL_DIRECT'):
    # the geometry of level 2 (128 channels, 64 x 64) through the level-0 code path: inputs given at 64 x 64
    import torch.nn.functional as F
    LEVEL, C, S = 0, 128, 64
    motion = F.interpolate(motion.permute(0, 3, 1, 2), size=(64, 64), mode='bilinear', align_corners=True).permute(0, 2, 3, 1).contiguous()
    flow = F.interpolate(flow, size=(64, 64), mode='bilinear', align_corners=True).contiguous()
    ifmask = F.interpolate(ifmask, size=(64, 64), mode='bilinear', align_corners=True).contiguous()
L = ConvLayer([256], 256, 3, 1, 1, ops.PAD_REFLECT).to(dev)
with torch.no_grad():
    L.weight.copy_(torch.randn(L.weight.shape, generator=g) * 0.02)
xa = Feat(torch.randn(8, 256, 64, 64, generator=g).to(dev))
x = torch.randn(8, C, S, S, generator=g).to(dev)
mean = torch.randn(8 * C, generator=g).to(dev) * 0.1
rstd = (torch.rand(8 * C, generator=g).to(dev) + 0.5)
f = Feat(x, mean, rstd, act=ops.ACT_RELU)
variant = os.environ.get('VICTIM', 'split')


def victim():
    if variant == 'split':
        return ops.warp_concat(f, motion, flow, ifmask, LEVEL, emit_xs=True, keep_fp32=False).xs
    return ops.warp_concat(f, motion, flow, ifmask, LEVEL, emit_xs=False).data


with torch.no_grad():
    ref = victim().clone()
    L.run(xa, norm_act=ops.ACT_RELU)
torch.cuda.synchronize()
AGGR = os.environ.get('AGGR', 'conv')
import ctypes
SYN = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'xcdvis', 'libaggr.so')) if AGGR.startswith('synth') else None
ma = torch.randn(4096, 4096, generator=g).to(dev).to(torch.bfloat16)
mb = torch.randn(4096, 4096, generator=g).to(dev).to(torch.bfloat16)
mf = torch.randn(4096, 4096, generator=g).to(dev)
big = torch.randn(64 << 20, generator=g).to(dev)


def aggressor():
    if AGGR == 'conv':
        return L.run(xa, norm_act=ops.ACT_RELU)
    if AGGR == 'gemm_bf16':
        return ma @ mb
    if AGGR == 'gemm_f32':
        return mf @ mf
    if AGGR == 'stream':
        return big * 1.5
    if AGGR == 'norm_split':
        return ops._norm_apply_split(Feat(xa.data, mean[:8 * 256].contiguous() if False else torch.zeros(8 * 256, device=dev), torch.ones(8 * 256, device=dev), act=ops.ACT_RELU), None, want_y=True, want_xs=True)
    if AGGR.startswith('synth:'):
        rc = SYN.aggr_launch(int(AGGR[6:]), ctypes.c_void_p(big.data_ptr()), ctypes.c_void_p(mf.data_ptr()), 60,
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        return None
    raise SystemExit('AGGR?')


with torch.no_grad():
    aggressor()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
tot = 0
for rep in range(5):
    with torch.no_grad():
        with torch.cuda.stream(sb):
            keep = [aggressor() for _ in range(40)]
        with torch.cuda.stream(sa):
            outs = [victim() for _ in range(40)]
    torch.cuda.synchronize()
    nbad = sum(int(not torch.equal(o, ref)) for o in outs)
    tot += nbad
    del keep
print('LEVEL %d victim %s aggressor %s ABLATE=%s: wrong warp launches %d of 200' % (LEVEL, variant, AGGR, os.environ.get('APAMD_ABLATE', '-'), tot))

if variant != 'split' and tot:
    # which half of the concat goes wrong: channels [0, C) follow `motion`, [C, 2C) follow `flow` / `ifmask`
    bad = [o for o in outs if not torch.equal(o, ref)]
    if not bad:
        with torch.no_grad():
            with torch.cuda.stream(sb):
                keep = [aggressor() for _ in range(40)]
            with torch.cuda.stream(sa):
                outs = [victim() for _ in range(40)]
        torch.cuda.synchronize()
        bad = [o for o in outs if not torch.equal(o, ref)]
    for o in bad[:3]:
        d = (o != ref)
        idx = d.nonzero()
        print('  differing values', int(d.sum()), ' motion half:', int(d[:, :C].sum()), ' flow half:', int(d[:, C:].sum()),
              ' samples', sorted(set(idx[:, 0].tolist())), ' x range', int(idx[:, 3].min()), int(idx[:, 3].max()),
              ' distinct (n, y, x-block-of-16):', len(set((a, b, c // 16) for a, _, b, c in idx.tolist())))
        n0, c0, y0, x0 = idx[0].tolist()
        print('  first: n %d c %d y %d x %d got %.6f want %.6f; same pixel other channels differ: %s' % (
            n0, c0, y0, x0, float(o[n0, c0, y0, x0]), float(ref[n0, c0, y0, x0]), d[n0, :, y0, x0].nonzero().flatten().tolist()[:40]))
