"""Per-sample independence across batch sizes: G(x[:b]) against G(x)[:b] for b = 1..16 (one stream)."""
import os, sys, io, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from animateportrait_amd.synthetic import make_generator_inputs, generator_args
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    G = bench.build_generator(dev)
args = [t.to(dev) for t in generator_args(make_generator_inputs(16, seed=1234))]
with torch.no_grad():
    full = G(*args)
    for b in range(1, 17):
        y = G(*[a[:b].contiguous() for a in args])
        y2 = G(*[a[16 - b:].contiguous() for a in args])
        print(b, 'head max diff %.3e' % float((y - full[:b]).abs().max()), 'tail max diff %.3e' % float((y2 - full[16 - b:]).abs().max()))
