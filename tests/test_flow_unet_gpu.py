"""GPU (-m gpu): FlowUnet_v2 on the HIP convolution kernels (animateportrait_amd/flow_unet_hip.py: BatchNorm folded, loader
activations, two-segment concat, 1x1 / stride-2 / pixel-shuffle layers) against the reference class's golden
(tests/golden/flowunet.npz, made from Module2/intrinsic_flow_models/networks.py:647-744) and, at a width where the wide
layers take the split-bf16 path, against the stock-PyTorch mirror that is pinned to the reference class on the CPU."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def test_flowunet_hip_matches_reference_golden(dev):
    from test_flow_unet_cpu import _seeded_net
    from animateportrait_amd.flow_unet_hip import FlowUnetV2Hip
    net, g = _seeded_net()
    hip = FlowUnetV2Hip(net).to(dev)
    x = (torch.rand(1, 136, 64, 64, generator=torch.Generator().manual_seed(6)) > 0.97).float()
    flow, vis, pyr, feat = hip(x.to(dev))
    for name, got in (('flow', flow), ('vis', vis), ('pyr1', pyr[1])):
        ref = torch.from_numpy(g[name])
        assert got.shape == ref.shape
        assert float((got.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 2e-6, name
    assert abs(float(feat.double().sum()) - float(g['feat_sum'])) <= 1e-4 * abs(float(g['feat_sum'])) + 1e-3


def test_pixel_shuffle_and_1x1_layers(dev):
    from animateportrait_amd import ops
    from animateportrait_amd.networks import ConvLayer
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 12, 7, 9, generator=g)
    assert torch.equal(ops.pixel_shuffle2(x.to(dev)).cpu(), torch.nn.functional.pixel_shuffle(x, 2))
    for cin, cout, h, w in ((136, 64, 20, 24), (64, 64, 33, 17), (24, 8, 16, 16)):
        layer = ConvLayer([cin], cout, 1, 1, 0).to(dev)
        wt, b = torch.randn(cout, cin, 1, 1, generator=g) * 0.1, torch.randn(cout, generator=g)
        with torch.no_grad():
            layer.weight.copy_(wt); layer.bias.copy_(b)
        xi = torch.randn(2, cin, h, w, generator=g)
        got = layer.run([ops.Feat(xi.to(dev), act=ops.ACT_RELU)]).data.cpu()
        ref = torch.nn.functional.conv2d(torch.relu(xi).double(), wt.double(), b.double())
        assert float((got.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (cin, cout)


@pytest.mark.parametrize('cfg', [dict(input_nc=136, nf=32, max_nf=64, start_scale=2, num_scales=3, n_residual_blocks=2, norm='batch'),
                                 dict(input_nc=136, nf=64, max_nf=256, start_scale=2, num_scales=4, n_residual_blocks=2, norm='batch')])
def test_flowunet_hip_wide_vs_torch_mirror(dev, cfg):
    """Widths at which the 3x3 layers run on the split-bf16 matrix kernels; random BatchNorm statistics so that the folding
    matters; 224 x 224 joint maps as flow_network_warp feeds them (geomgm_ifw_fore_model.py:69-84)."""
    from animateportrait_amd import flow_unet
    from animateportrait_amd.flow_unet_hip import FlowUnetV2Hip
    torch.manual_seed(4)
    net = flow_unet.FlowUnetV2(**cfg).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    x = (torch.rand(2, 136, 224, 224, generator=torch.Generator().manual_seed(8)) > 0.985).float()
    with torch.no_grad():
        rf, rv, rp, rfeat = net(x)
    flow, vis, pyr, feat = FlowUnetV2Hip(net).to(dev)(x.to(dev))
    for name, got, ref in (('flow', flow, rf), ('vis', vis, rv), ('pyr_last', pyr[-1], rp[-1]), ('feat', feat, rfeat)):
        assert got.shape == ref.shape, name
        assert float((got.cpu() - ref).abs().max()) <= 5e-4 * float(ref.abs().max()) + 1e-5, (name, float((got.cpu() - ref).abs().max()), float(ref.abs().max()))


def test_models_attach_the_hip_flow_network_from_its_checkpoint(dev, tmp_path):
    """``load_flow_network()`` of both models (geomgm_ifw_fore_model.py:57-68, :386): with the checkpoint directory present,
    ``BaseModel.attach_flow_network`` puts FlowUnet_v2 on the HIP kernels, and ``flow_network_warp`` (:69-84) through the
    model's set_input == the oracle's CPU composition with the stock-PyTorch network."""
    import contextlib
    import io
    import json
    from test_flow_unet_cpu import _seeded_net, CONFIG
    from animateportrait_amd.flow_unet_hip import FlowUnetV2Hip
    from animateportrait_amd.options.base_options import TestOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.synthetic import make_landmarks
    from oracle import aux_glue as oa
    net, _ = _seeded_net()
    d = tmp_path / 'FlowReg_id_flow_faces'
    d.mkdir()
    json.dump(dict(which_model='unet_v2', input_type1='joint', input_type2='joint', joint_nc=68, seg_nc=7, nf=CONFIG['nf'],
                   max_nf=CONFIG['max_nf'], start_scale=CONFIG['start_scale'], num_scale=CONFIG['num_scales'],
                   norm=CONFIG['norm']), open(d / 'train_opt.json', 'w'))
    torch.save(net.state_dict(), d / 'best_net_netF.pth')
    opt = TestOptions().parse(['--model', 'geomcgt_ifw_test', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode',
                               'synthetic', '--name', 'x', '--output_nc', '1', '--ngf', '8', '--netg_resb_div', '3',
                               '--netg_resb_disp', '3', '--gpu_ids', '0'])
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
    model.FLOW_CHECKPOINT_DIR = str(tmp_path)
    model.attach_flow_network()
    assert isinstance(model.aux['netF'], FlowUnetV2Hip) and model.aux['netF'].heads_only and model.aux['netF'].use_graph
    g = torch.Generator().manual_seed(2)
    lm1, lm2 = make_landmarks(2, g), make_landmarks(2, g)
    photo = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    from animateportrait_amd import losses
    flow, mask = losses.flow_network_warp(model.aux['netF'], photo.to(dev), lm1.to(dev), lm2.to(dev))
    rflow, rmask = oa.flow_network_warp(net, photo, lm1, lm2)
    # the visibility argmax may flip where two logits tie within rounding: compare where the masks agree (nearly everywhere)
    agree = (mask.cpu() - rmask).abs() < 1e-4
    assert float(agree.float().mean()) > 0.999
    assert float(((flow.cpu() - rflow).abs() * agree).max()) <= 2e-4 * float(rflow.abs().max()) + 1e-5
    # the merged 5-output head == the two separate heads of the full forward
    x = (torch.rand(2, 136, 64, 64, generator=g) > 0.97).float().to(dev)
    fa, va, _, _ = model.aux['netF'](x)
    fa2, va2, _, _ = model.aux['netF'](x * 0 + (torch.rand(x.shape, device=dev) > 0.9).float())   # a replay on other data
    fa3, va3, _, _ = model.aux['netF'](x)                         # and back: the graph is a function of its input only
    assert torch.equal(fa, fa3) and torch.equal(va, va3) and not torch.equal(fa, fa2)
    model.aux['netF'].heads_only = False
    model.aux['netF'].use_graph = False
    fb, vb, pyr, _ = model.aux['netF'](x)
    assert len(pyr) == CONFIG['num_scales'] and torch.allclose(fa, fb, atol=1e-6) and torch.allclose(va, vb, atol=1e-6)
