"""FlowUnetV2 (animateportrait_amd/flow_unet.py) against the reference class's golden (tests/golden/flowunet.npz, made
by tests/golden/make_flowunet_golden.py from Module2/intrinsic_flow_models/networks.py:647-744): state_dict keys,
shapes and registration order, outputs for seeded weights, and the train_opt.json / .pth loader."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from make_flowunet_golden import CONFIG           # noqa: E402
from make_module1_golden import seeded_state      # noqa: E402

from animateportrait_amd import flow_unet         # noqa: E402


def _golden():
    return np.load(os.path.join(HERE, 'golden', 'flowunet.npz'))


def _seeded_net():
    g = _golden()
    net = flow_unet.FlowUnetV2(**CONFIG)
    ks = [(k, tuple(v.shape), str(v.dtype)) for k, v in net.state_dict().items()]
    assert [k for k, _, _ in ks] == [str(k) for k in g['keys']]
    assert [str(s) for _, s, _ in ks] == [str(s) for s in g['shapes']]
    assert [d for _, _, d in ks] == [str(d) for d in g['dtypes']]
    net.load_state_dict(seeded_state(ks, seed=55), strict=True)
    return net.eval(), g


def test_flowunet_matches_reference_class():
    net, g = _seeded_net()
    x = (torch.rand(1, 136, 64, 64, generator=torch.Generator().manual_seed(6)) > 0.97).float()
    with torch.no_grad():
        flow, vis, pyr, feat = net(x)
    for name, got in (('flow', flow), ('vis', vis), ('pyr1', pyr[1])):
        ref = torch.from_numpy(g[name])
        assert got.shape == ref.shape
        assert (got - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-6, name
    assert abs(float(feat.double().sum()) - float(g['feat_sum'])) <= 1e-4 * abs(float(g['feat_sum'])) + 1e-4


def test_load_flow_network_reads_opt_and_checkpoint(tmp_path):
    net, _ = _seeded_net()
    d = tmp_path / 'FlowReg_id_flow_faces'
    d.mkdir()
    opt = dict(which_model='unet_v2', input_type1='joint', input_type2='joint', joint_nc=68, seg_nc=7,
               nf=CONFIG['nf'], max_nf=CONFIG['max_nf'], start_scale=CONFIG['start_scale'],
               num_scale=CONFIG['num_scales'], norm=CONFIG['norm'])
    json.dump(opt, open(d / 'train_opt.json', 'w'))
    torch.save(net.state_dict(), d / 'best_net_netF.pth')
    got = flow_unet.load_flow_network(checkpoints_dir=str(tmp_path))
    assert not got.training and not any(p.requires_grad for p in got.parameters())
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), got.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert flow_unet.input_dim(opt, 'joint+seg') == 75
