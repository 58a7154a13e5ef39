#!/usr/bin/env python3
"""Golden for the FlowUnet_v2 mirror (tests/golden/flowunet.npz): state_dict keys / shapes of the reference class
(Module2/intrinsic_flow_models/networks.py:647-744) at a small configuration and its outputs for seeded weights.
Shims: the stubs of make_golden.py plus ``np.int`` (removed from numpy 2; the reference uses it at :670).
Run in the build container:  python tests/golden/make_flowunet_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

CONFIG = dict(input_nc=136, nf=8, max_nf=24, start_scale=2, num_scales=3, n_residual_blocks=2, norm='batch')


def main():
    import warnings
    warnings.filterwarnings('ignore')
    from make_golden import import_reference, save
    from make_module1_golden import seeded_state
    np.int = int
    _, _, ifm, _ = import_reference()
    net = ifm.FlowUnet_v2(**CONFIG)
    ks = [(k, tuple(v.shape), str(v.dtype)) for k, v in net.state_dict().items()]
    net.load_state_dict(seeded_state(ks, seed=55), strict=True)
    net.eval()
    x = (torch.rand(1, 136, 64, 64, generator=torch.Generator().manual_seed(6)) > 0.97).float()
    with torch.no_grad():
        flow, vis, pyr, feat = net(x)
    save('flowunet.npz', keys=np.array([k for k, _, _ in ks]), shapes=np.array([str(s) for _, s, _ in ks]),
         dtypes=np.array([d for _, _, d in ks]), flow=flow, vis=vis, pyr1=pyr[1], feat_sum=feat.double().sum())


if __name__ == '__main__':
    main()
