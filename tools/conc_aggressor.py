"""Which kernel, looping on a second HIP stream, makes a generator forward on the first stream go wrong?"""
import os, sys, io, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from animateportrait_amd import ops
from animateportrait_amd.ops import Feat
from animateportrait_amd.synthetic import make_generator_inputs, generator_args
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    G = bench.build_generator(dev)
args = [t.to(dev)[:8].contiguous() for t in generator_args(make_generator_inputs(16, seed=1234))]
g = torch.Generator().manual_seed(3)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x256 = Feat(r(8, 256, 64, 64))
x256b = Feat(r(8, 256, 64, 64))
x64 = Feat(r(8, 64, 256, 256))
x3 = Feat(r(8, 3, 256, 256))
x128 = Feat(r(8, 128, 128, 128))
l1 = Feat(r(16, 1, 256, 256))
with torch.no_grad():
    want = G(*args).clone()
    raw = G.model2['1'].conv_block['1'].run(x256, norm_act=ops.ACT_RELU)
torch.cuda.synchronize()
aggr = {
    'trunk 3x3 conv (bf16x3 MFMA, LDS-DMA)': lambda: G.model2['1'].conv_block['1'].run(x256, norm_act=ops.ACT_RELU),
    'norm_split pass': lambda: ops._norm_apply_split(Feat(raw.data, pending=raw.pending, act=ops.ACT_RELU) if False else raw, x256b, want_y=True, want_xs=True),
    'transposed conv 256->128 (conv_ph4)': lambda: G.model3['0'].run(x256, norm_act=ops.ACT_RELU),
    'transposed conv 128->64 (conv_ph4)': lambda: G.model3['3'].run(x128, norm_act=ops.ACT_RELU),
    'final 7x7 conv (conv_direct, vector ALU)': lambda: G.model3['7'].run(x64, act=ops.ACT_TANH),
    'stem 7x7 (row form)': lambda: G.model_tri10['1'].run(x3, norm_act=ops.ACT_RELU),
    'stride-2 3x3 conv 128->256 (space-to-depth)': lambda: G.model_tri12['0'].run(x128, norm_act=ops.ACT_RELU),
    'landmark conv (conv_small)': lambda: G.model_landmark_trans['0'].run(l1, norm_act=ops.ACT_RELU),
    'torch elementwise (x*1.5)': lambda: x64.data * 1.5,
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in aggr.items():
    with torch.no_grad():
        fn()                                   # pack weights etc. on the main stream first
    torch.cuda.synchronize()
    bad = []
    for rep in range(4):
        with torch.no_grad():
            with torch.cuda.stream(sb):
                keep = [fn() for _ in range(60)]
            with torch.cuda.stream(sa):
                y = G(*args)
        torch.cuda.synchronize()
        bad.append(float((y - want).abs().max()))
        del keep
    print('%-48s forward on the other stream: max err per rep %s' % (name, ['%.1e' % b for b in bad]))
