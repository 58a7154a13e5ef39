"""Run a short chain of generator ops on two HIP streams at once (different inputs) and compare every intermediate with
the one-stream result.  All intermediates are kept alive, so the caching allocator never recycles a block during the test."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animateportrait_amd import ops
from animateportrait_amd.networks import ConvLayer
from animateportrait_amd.ops import Feat

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
L1 = ConvLayer([256], 256, 3, 1, 1, ops.PAD_REFLECT).to(dev)
L2 = ConvLayer([256], 256, 3, 1, 1, ops.PAD_REFLECT).to(dev)
L3 = ConvLayer([256], 128, 3, 2, 1, ops.PAD_ZERO).to(dev)
with torch.no_grad():
    for l in (L1, L2, L3):
        l.weight.copy_(torch.randn(l.weight.shape, generator=g) * 0.02)
xs = [torch.randn(8, 256, 64, 64, generator=g).to(dev) for _ in range(2)]


def chain(x, keep):
    f0 = Feat(x)
    a = L1.run(f0, norm_act=ops.ACT_RELU)
    keep += [f0, a]
    b = L2.run(a, norm_act=ops.ACT_NONE)
    keep += [b]
    y, sp = ops._norm_apply_split(b, f0, want_y=True, want_xs=True)
    keep += [y, sp]
    c = L3.run(Feat(y), norm_act=ops.ACT_RELU)
    keep += [c]
    return [('conv1 raw', a.data), ('conv1 mean', a.mean), ('conv1 xs', a.xs), ('conv2 raw', b.data), ('y', y), ('split', sp), ('conv3 raw', c.data),
            ('conv3 mean', c.mean)]


keep = []
with torch.no_grad():
    want = [chain(x, keep) for x in xs]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
for rep in range(4):
    got = []
    main = torch.cuda.current_stream()
    with torch.no_grad():
        for st, x in zip(streams, xs):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for _ in range(3):
                    r = chain(x, keep)
                got.append(r)
    torch.cuda.synchronize()
    for i in range(2):
        bad = [(n, float((a.float() - b.float()).abs().max()) if a.dtype != torch.uint8 else int((a != b).sum()))
               for (n, a), (_, b) in zip(got[i], want[i]) if a is not None and not torch.equal(a, b)]
        print('rep', rep, 'stream', i, 'differs:', bad)
