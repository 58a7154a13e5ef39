#!/usr/bin/env python3
"""Golden for the Module1 content network mirror (tests/golden/module1.npz): state_dict keys / shapes of the
reference's ``Audio2landmark_content`` (Module1/src/models/model_audio2landmark.py:28-90) and its output for seeded
weights and inputs.  Run in the build container:  python tests/golden/make_module1_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def seeded_state(keys_shapes, seed=77):
    """weights both sides load: N(0, 0.05) per tensor in key order; BatchNorm running_var = 1 + |.|, counters 0"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape, dtype in keys_shapes:
        if 'num_batches_tracked' in k:
            sd[k] = torch.zeros(shape, dtype=torch.int64)
        else:
            t = torch.randn(shape, generator=g) * 0.05
            sd[k] = (1.0 + t.abs()) if k.endswith('running_var') else t
    return sd


def main():
    sys.path.insert(0, '/root/reference/Module1')
    from src.models.model_audio2landmark import Audio2landmark_content
    from make_golden import save
    net = Audio2landmark_content(num_window_frames=18, in_size=80, use_prior_net=False, hidden_size=256, num_layers=3,
                                 drop_out=0, bidirectional=False)
    ks = [(k, tuple(v.shape), str(v.dtype)) for k, v in net.state_dict().items()]
    net.load_state_dict(seeded_state(ks), strict=True)
    net.eval()
    g = torch.Generator().manual_seed(3)
    au = torch.randn(6, 18, 80, generator=g)
    fid = torch.randn(1, 204, generator=g) * 0.1
    with torch.no_grad():
        out, _ = net(au, fid)
    save('module1.npz', keys=np.array([k for k, _, _ in ks]), shapes=np.array([str(s) for _, s, _ in ks]),
         dtypes=np.array([d for _, _, d in ks]), au=au, fid=fid, out=out)


if __name__ == '__main__':
    main()
