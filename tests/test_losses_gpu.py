"""GPU (-m gpu): loss reductions, mask compositing, the crop + resize glue in front of the frozen aux nets and the
landmark rasterisers (csrc/losses.hip) against the reference goldens (aux.npz, losses.npz), the oracle and torch
autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import linf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def test_ganloss_lsgan_vs_reference_golden_and_autograd(dev, golden):
    from animateportrait_amd import networks as N
    gd = golden('losses.npz')
    crit = N.GANLoss('lsgan')
    p = gd['pred'].to(dev).requires_grad_(True)
    lr, lf = crit(p, True), crit(p, False)
    assert abs(float(lr) - float(gd['gan_real'])) < 1e-6 and abs(float(lf) - float(gd['gan_fake'])) < 1e-6
    (0.5 * lr + 2.0 * lf).backward()
    pr = gd['pred'].clone().requires_grad_(True)
    (0.5 * ((pr - 1.0) ** 2).mean() + 2.0 * (pr ** 2).mean()).backward()
    assert linf(p.grad, pr.grad) < 1e-9


@pytest.mark.parametrize('n', [1, 900, 28800, 16 * 256 * 256 + 3])
def test_reductions_vs_torch(dev, n):
    """ap_reduce_mean ops 0/1/2 and their backward against float64 torch on the same values."""
    from animateportrait_amd import losses
    g = torch.Generator().manual_seed(n)
    a = torch.randn(n, generator=g)
    b = torch.randn(n, generator=g)
    w = (torch.rand(n, generator=g) > 0.7).float()
    for name, fn, ref in (
            ('lsgan', lambda x: losses.lsgan_loss(x, 1.0, 0.5), lambda x: 0.5 * ((x - 1.0) ** 2).mean()),
            ('l1', lambda x: losses.l1_loss(x, b.to(x.device) if x.is_cuda else b, 10.0), lambda x: 10.0 * (x - b.double()).abs().mean()),
            ('wmean', lambda x: losses.weighted_mean(x, w.to(x.device), 1.0, 50.0), lambda x: 50.0 * ((x + 1.0) * w.double()).mean())):
        x = a.to(dev).requires_grad_(True)
        y = fn(x)
        (y * 3.0).backward()
        xr = a.double().requires_grad_(True)
        yr = ref(xr)
        (yr * 3.0).backward()
        assert abs(float(y) - float(yr)) <= 2e-6 * abs(float(yr)) + 1e-7, (name, float(y), float(yr))
        assert linf(x.grad, xr.grad) <= 1e-6 * float(xr.grad.abs().max()) + 1e-12, name
    # determinism: the two-stage sum has a fixed order
    x = a.to(dev)
    assert float(losses.lsgan_loss(x, 0.0)) == float(losses.lsgan_loss(x, 0.0))
    # L1 gradient also flows to the second operand when it asks for one
    xa, xb = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    losses.l1_loss(xa, xb).backward()
    assert torch.equal(xa.grad, -xb.grad)


@pytest.mark.parametrize('c', [1, 3])
def test_mask_compositing_bit_exact_vs_reference_formulas(dev, c, golden):
    """masked() types 0-3 (base_model.py:238-247), foreground-on-white and background blend
    (geomgm_ifw_fore_model.py:523-543): bit-identical to the ATen chain on the same device, equal to the oracle (which is
    pinned to the reference-derived losses.npz), gradients = autograd's."""
    from animateportrait_amd import losses
    from oracle import losses as ol
    g = torch.Generator().manual_seed(5 + c)
    a = (torch.rand(2, c, 64, 48, generator=g) * 2 - 1)
    m = torch.rand(2, 1, 64, 48, generator=g)
    m[:, :, :8] = 0.0
    m[:, :, 8:16] = 1.0
    s = (torch.rand(2, 1, 64, 48, generator=g) * 2 - 1)
    ad, md, sd = a.to(dev), m.to(dev), s.to(dev)
    for t in range(4):
        x = ad.clone().requires_grad_(True)
        y = losses.masked(x, md, t)
        xr = ad.clone().requires_grad_(True)
        yr = ol.masked(xr, md, t)                       # the same formulas as ATen ops on the GPU
        assert torch.equal(y, yr), t
        assert torch.equal(y.cpu(), ol.masked(a, m, t)), t
        go = torch.randn(y.shape, generator=g).to(dev)
        y.backward(go)
        yr.backward(go)
        assert linf(x.grad, xr.grad) < 1e-7, t
    x = ad.clone().requires_grad_(True)
    y = losses.fore_composite(x, md)
    assert torch.equal(y, ol.fore_composite(ad, md))
    y2 = losses.bg_blend(x, sd, md)
    assert torch.equal(y2, ol.bg_blend(ad, sd, md))
    xr = ad.clone().requires_grad_(True)
    (ol.bg_blend(xr, sd, md) * 2 + ol.fore_composite(xr, md)).sum().backward()
    (y2 * 2 + y).sum().backward()
    assert linf(x.grad, xr.grad) < 1e-6
    if c == 1:
        gd = golden('losses.npz')
        for t in range(4):      # outputs of the reference's own BaseModel.masked (make_golden.py)
            assert linf(losses.masked(gd['A'].to(dev), gd['M'].to(dev), t), gd['masked%d' % t]) < 1e-6


def test_get_lm_crop_bicubic_vs_reference_golden(dev, golden):
    """ap_crop_resize (bicubic) + the model's get_lm against GeomGMIFWForeModel.get_lm outputs (aux.npz)."""
    from animateportrait_amd import losses, standins
    from animateportrait_amd.models.geomgm_ifw_fore_model import GeomGMIFWForeModel
    import types
    gd = golden('aux.npz')
    net = standins.StandinLandmarkNet().to(dev)
    me = types.SimpleNamespace(aux={'landmarks': net})
    for c in (1, 3):
        xs, wins, outs = [], [], []
        for i in range(3):
            seed = int(gd['lm_c%d_%d_seed' % (c, i)])
            x = torch.rand(1, c, 256, 256, generator=torch.Generator().manual_seed(seed)) * 2 - 1
            win = gd['lm_c%d_%d_win' % (c, i)]
            wdev = losses.windows_to_device(win, 1, dev)
            box = losses.crop_resize(x.to(dev), wdev, (2, 1, 0) if c == 3 else (0, 0, 0), (112, 112), losses.RESIZE_BICUBIC,
                                     0.5, 0.5)
            ref = gd['lm_c%d_%d_box' % (c, i)]
            assert linf(box[:, :ref.shape[1]], ref) < 2e-6, (c, i)
            assert np.allclose(box.double().sum(dim=(0, 2, 3)).cpu().numpy(), gd['lm_c%d_%d_boxsum' % (c, i)], rtol=1e-6)
            lm = GeomGMIFWForeModel.get_lm(me, x.to(dev), win)
            assert linf(lm, gd['lm_c%d_%d_out' % (c, i)]) < 2e-4, (c, i)                  # pixels
            xs.append(x); wins.append(torch.as_tensor(win)); outs.append(gd['lm_c%d_%d_out' % (c, i)])
        # batched (the reference is b=1 only): per-sample windows, same numbers
        lm = GeomGMIFWForeModel.get_lm(me, torch.cat(xs, 0).to(dev), torch.cat(wins, 0))
        assert linf(lm, torch.cat(outs, 0)) < 2e-4


@pytest.mark.parametrize('mode', [0, 1])
def test_crop_resize_backward_vs_autograd(dev, mode):
    from animateportrait_amd import losses
    from oracle import aux_glue as oa
    g = torch.Generator().manual_seed(40 + mode)
    x = torch.rand(2, 1, 96, 96, generator=g) * 2 - 1
    wins = torch.tensor([[10, 80, 12, 70], [-8, 60, 30, 96]], dtype=torch.int32)
    size = (40, 40) if mode == 1 else (37, 29)
    go = torch.randn(2, 3, *size, generator=g)
    xd = x.to(dev).requires_grad_(True)
    y = losses.crop_resize(xd, losses.windows_to_device(wins, 2, dev), (0, 0, 0), size, mode, 0.5, 0.5)
    y.backward(go.to(dev))
    xr = x.double().requires_grad_(True)
    outs = []
    for i in range(2):
        box = oa.crop_box(xr[i:i + 1], wins[i]).repeat(1, 3, 1, 1)
        if mode == 1:
            outs.append((F.interpolate(box, size=size, mode='bicubic', align_corners=False) + 1) * 0.5)
        else:
            outs.append((F.interpolate(box, size=size, mode='bilinear', align_corners=True) + 1) * 0.5)
    yr = torch.cat(outs, 0)
    yr.backward(go.double())
    assert linf(y, yr) < 1e-5            # fp32 sample coordinates vs the fp64 evaluation
    assert linf(xd.grad, xr.grad) < 1e-5 * max(1.0, float(xr.grad.abs().max()))
    with pytest.raises(ValueError):
        losses.windows_to_device([[10, 10, 0, 5]], 1, dev)          # empty window
    with pytest.raises(ValueError):
        losses.windows_to_device([[10, 20, 0, 50]], 1, dev)         # taller than the box: the reference's slice fails


def test_faceloss_vs_reference_golden(dev, golden):
    """networks.FaceLoss (device crop + bilinear 112x96, stand-in features) against the reference FaceLoss.forward:
    head crops, loss and the gradient w.r.t. the drawing."""
    from animateportrait_amd import networks as N, standins
    gd = golden('aux.npz')
    fl = N.FaceLoss(standins.StandinFaceNet().to(dev))
    a = (torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(int(gd['fl_seed_a']))) * 2 - 1).to(dev)
    b = (torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(int(gd['fl_seed_b']))) * 2 - 1).to(dev)
    a.requires_grad_(True)
    h1 = fl.crop_head_bbox(a, gd['fl_bb1'])
    assert linf(h1[:, :1], gd['fl_head1']) < 2e-6 and torch.equal(h1[:, 0], h1[:, 2])
    assert linf(fl.crop_head_bbox(b, gd['fl_bb2'])[:, :1], gd['fl_head2']) < 2e-6
    loss = fl(a, b, bbox1=torch.as_tensor(gd['fl_bb1']), bbox2=torch.as_tensor(gd['fl_bb2']))
    assert abs(float(loss.detach()) - float(gd['fl_loss'])) < 2e-6
    loss.backward()
    scale = float(gd['fl_grad_a_sub'].abs().max())
    assert linf(a.grad[:, :, ::4, ::4], gd['fl_grad_a_sub']) < 1e-4 * scale
    assert abs(float(a.grad.double().abs().sum()) - float(gd['fl_grad_a_abs'])) < 1e-4 * float(gd['fl_grad_a_abs'])
    # a 3-channel input (the x3 repeat the reference's caller makes) gives the same crops
    assert torch.equal(fl.crop_head_bbox(a.detach().repeat(1, 3, 1, 1), gd['fl_bb1']), h1.detach())


def test_kp_to_map_and_flow_post_vs_reference_golden(dev, golden):
    """ap_kp_to_map bit-exact against kp_to_map_some; flow_network_warp (device pre / post around the stand-in
    FlowUnet) against the reference function's output (geomgm_ifw_fore_model.py:19-84)."""
    from animateportrait_amd import losses, standins
    gd = golden('aux.npz')
    lm1, lm2 = gd['kp_lm1'].to(dev), gd['kp_lm2'].to(dev)
    j1 = losses.kp_to_map(lm1)
    assert j1.shape == (1, 68, 224, 224)
    assert np.array_equal(j1.cpu().numpy().astype(np.uint8), gd['kp_j1'])
    from oracle import aux_glue as oa
    odd = torch.tensor([[[-8.0 / 7.0, 10.0], [16.0, 16.0], [20.0, 20.0], [-8.0, 3.0], [300.0, 100.0]]])
    want = oa.kp_to_map_some((224, 224), odd.numpy() * 7 / 8)        # incl. the x == -1 blanking rule (:33-34)
    got = losses.kp_to_map(odd.to(dev)).cpu()
    assert torch.equal(got, want) and float(got[0, 1].sum()) == 49.0 and float(got[0, 4].sum()) == 0.0
    netF = standins.StandinFlowNet().to(dev)
    wf, rm = losses.flow_network_warp(netF, torch.zeros(1, 3, 256, 256, device=dev), lm1, lm2)
    assert linf(wf[:, :, ::2, ::2], gd['fw_flow_sub']) < 2e-4          # flows of +-20 px through a conv on another device
    assert linf(rm[:, :, ::2, ::2], gd['fw_mask_sub']) < 1e-6 or \
        float((rm[:, :, ::2, ::2].cpu() - gd['fw_mask_sub']).abs().gt(1e-6).float().mean()) < 1e-3   # argmax ties
    assert abs(float(wf.double().abs().sum()) - float(gd['fw_flow_abs'])) < 1e-3 * float(gd['fw_flow_abs'])
    # the post stage alone, on identical inputs, against the oracle's expressions
    g = torch.Generator().manual_seed(8)
    fo, vo = torch.randn(2, 2, 224, 224, generator=g), torch.randn(2, 3, 224, 224, generator=g)
    wf2, rm2 = losses.flow_post(fo.to(dev), vo.to(dev))
    mask = (vo.argmax(dim=1, keepdim=True).float() < 2).float()
    ref_f = F.interpolate(fo * 20. * mask / 7 * 8, size=(256, 256), mode='bilinear', align_corners=True)
    ref_m = F.interpolate(mask, size=(256, 256), mode='bilinear', align_corners=True)
    assert linf(wf2, ref_f) < 2e-5 and linf(rm2, ref_m) < 1e-6


def test_flow_network_warp_with_the_flowunet_mirror(dev):
    """The FlowUnet_v2 mirror (flow_unet.py, pinned on the CPU by test_flow_unet_cpu.py) on the device behind the device
    pre / post stages == the oracle's flow_network_warp around the same module on the CPU."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from make_flowunet_golden import CONFIG
    from make_module1_golden import seeded_state
    from animateportrait_amd import flow_unet, losses
    from oracle import aux_glue as oa
    net = flow_unet.FlowUnetV2(**CONFIG)
    net.load_state_dict(seeded_state([(k, tuple(v.shape), str(v.dtype)) for k, v in net.state_dict().items()], seed=55))
    net.eval()
    g = torch.Generator().manual_seed(12)
    lm1 = torch.rand(2, 68, 2, generator=g) * 200 + 28
    lm2 = lm1 + torch.randn(2, 68, 2, generator=g) * 3
    want_f, want_m = oa.flow_network_warp(net, torch.zeros(2, 3, 256, 256), lm1, lm2)
    import copy
    wf, rm = losses.flow_network_warp(copy.deepcopy(net).to(dev), torch.zeros(2, 3, 256, 256, device=dev), lm1, lm2)
    assert wf.shape == (2, 2, 256, 256) and rm.shape == (2, 1, 256, 256)
    tie = float((rm.cpu() - want_m).abs().gt(1e-6).float().mean())          # visibility argmax ties flip mask pixels
    assert tie < 2e-3
    if tie == 0.0:
        assert linf(wf, want_f) < 1e-3 * max(1.0, float(want_f.abs().max()))
    else:
        assert float((wf.cpu() - want_f).abs().gt(1e-3 * max(1.0, float(want_f.abs().max()))).float().mean()) < 4e-3


def test_landmark_discs_vs_oracle_rule(dev):
    """ap_landmark_discs == the oracle's draw2(op=0) (OpenCV's filled-circle rows; cv2 itself is absent: unpinned)."""
    from animateportrait_amd import losses
    from oracle import aux_glue as oa
    g = torch.Generator().manual_seed(3)
    lm = torch.rand(3, 68, 2, generator=g) * 270 - 7            # some discs clipped by / outside the frame
    lm[0, 0] = torch.tensor([7.5, 8.5])                          # half-to-even rounding
    for r in (1, 3, 4):
        out = losses.landmark_discs(lm.to(dev), 256, 256, r)
        for i in range(3):
            assert torch.equal(out[i].cpu(), oa.draw2(256, 256, lm[i].numpy(), r)), (r, i)
    assert set(out.unique().tolist()) == {-1.0, 1.0}


def test_lip_line_mask_is_opencv_thickline_rule(dev):
    """ap_lip_line_mask (getlipline, geomgm_ifw_fore_model.py:507-515) against the restatement of OpenCV's ThickLine /
    FillConvexPoly / Line2 / Circle (oracle/cv_raster.py) -- bit exact: random lip loops incl. points outside the frame,
    thickness 2 (256 px) and 4 (512 px), and the committed literal rasters of tests/golden/opencv_rules.json."""
    import json
    import os
    from animateportrait_amd import losses
    from animateportrait_amd.models.geomgm_ifw_fore_model import LIP_SEGMENTS
    from oracle import cv_raster as cr
    g = torch.Generator().manual_seed(12)
    for size, th in ((256, 2), (512, 4), (64, 3)):
        lands = torch.rand(3, 68, 2, generator=g) * (size * 1.2) - size * 0.1          # some points leave the frame
        lands[1, 48:68] = lands[1, 48:68].round()                                       # integer coordinates too
        lands[2, 50] = lands[2, 51]                                                     # a zero-length segment
        got = losses.lip_line_mask(lands.to(dev), LIP_SEGMENTS, size, th).cpu()
        for i in range(3):
            want = cr.lip_line_mask(size, lands[i].double().numpy(), LIP_SEGMENTS, th)
            assert got.shape == (3, 1, size, size) and np.array_equal(got[i, 0].numpy(), want), (size, th, i, float(np.abs(got[i, 0].numpy() - want).sum()))
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'opencv_rules.json')))
    for ln in d['lines']:
        pts = torch.tensor([[ln['p0'], ln['p1']]], dtype=torch.float32) + 0.4          # the binding truncates
        got = losses.lip_line_mask(pts.to(dev), [(0, 1)], d['canvas'], ln['thickness']).cpu()[0, 0].numpy()
        want = np.zeros((d['canvas'], d['canvas']), np.float32)
        for y, a, b in ln['runs']:
            want[y, a:b + 1] = 1.0
        assert np.array_equal(got, want), ln
