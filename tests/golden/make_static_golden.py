#!/usr/bin/env python3
"""Golden vectors of the static drawing generator (``resnet_style2_9blocks``), produced by
the REFERENCE's own ``networks.define_G`` / ``ResnetStyle2Generator`` imported read-only from /root/reference/Module2.

    python tests/golden/make_static_golden.py          (build container only)

Fixtures hold seeds, outputs and checksums -- no reference source.  Weights: ``define_G`` builds the net (running the
reference's init_weights), then the seeded N(0, 0.02) tensors of ``oracle.generator.init_params`` are loaded strictly,
so the key names / shapes of ``oracle.static_generator.static_param_shapes`` are verified against the reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, save, sha  # noqa: E402


def main():
    import warnings
    warnings.filterwarnings('ignore')
    networks, _, _, _ = import_reference()
    from oracle import generator as og, static_generator as os_
    torch.set_num_threads(8)
    out = {}
    for tag, ngf, size in (('ngf8', 8, 64), ('ngf64', 64, 128)):
        G = networks.define_G(3, 1, ngf, 'resnet_style2_9blocks', 'instance', use_dropout=False, gpu_ids=[])
        sd = og.init_params(os_.static_param_shapes(3, 1, ngf), seed=4321)
        G.load_state_dict(sd, strict=True)
        assert list(G.state_dict().keys()) == list(sd.keys())
        G.eval()
        g = torch.Generator().manual_seed(77 + ngf)
        n = 2
        x = torch.rand(n, 3, size, size, generator=g) * 2 - 1
        style = os_.style_code(n, size // 4)
        with torch.no_grad():
            y = G(x, style)
            y1 = G(x[:1], style[:1])
        assert torch.allclose(y1, y[:1], atol=1e-5)
        out['y_' + tag] = y
        out['seed_' + tag] = np.int64(77 + ngf)
        out['weights_sha256_' + tag] = sha(sd)
        out['size_' + tag] = np.int64(size)
    save('static_gen.npz', **out)


if __name__ == '__main__':
    main()
