"""CPU: the oracle (oracle/) against the golden vectors captured from the reference's
own network code (tests/golden/make_golden.py).  This is the pin of the oracle."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import linf, sd_sha
from oracle import generator as og, discriminator as od, warp as ow, losses as ol, tps as ot
from animateportrait_amd.synthetic import make_generator_inputs, generator_args

TOL = 2e-5   # fp32 CPU vs fp32 CPU, same library: only summation-order noise


def sub(gd, prefix):
    return {k[len(prefix):]: v for k, v in gd.items() if k.startswith(prefix)}


def test_resnet_blocks(golden):
    gd = golden('ops_small.npz')
    sd = {'b.' + k: v for k, v in sub(gd, 'rb_conv_block').items()}
    sd = {k.replace('b.', 'b.conv_block', 1): v for k, v in sd.items()}
    y1 = F.relu(og.inorm(og.conv_reflect(sd, 'b.conv_block.1', gd['rb_x'], 1)))
    assert linf(y1, gd['rb_first']) < TOL
    assert linf(og.resnet_block(sd, 'b', gd['rb_x']), gd['rb_y']) < TOL
    sd2 = {'b.' + k[4:]: v for k, v in gd.items() if k.startswith('rb2_') and k not in ('rb2_x', 'rb2_y')}
    assert linf(og.resnet_block2(sd2, 'b', gd['rb2_x']), gd['rb2_y']) < TOL


def test_generator_pieces(golden):
    gd = golden('ops_small.npz')
    sd = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234)
    assert linf(og.stem7(sd, 'model_tri10.1', gd['stem_x']), gd['stem10_y']) < TOL
    assert linf(og.stem7(sd, 'model_tri00.1', gd['stem_x']), gd['stem00_y']) < TOL
    assert linf(og.down3(sd, 'model_tri01.0', gd['down_x']), gd['down01_y']) < TOL
    up = F.relu(og.inorm(og.deconv(sd, 'model3.0', gd['up_x'])))
    assert linf(up, gd['up_y']) < TOL
    up2 = F.relu(og.inorm(og.deconv(sd, 'model3.3', up)))
    assert linf(up2, gd['up2_y']) < TOL
    assert linf(torch.tanh(og.conv_reflect(sd, 'model3.7', up2, 3)), gd['final_y']) < TOL
    assert linf(og.landmark_trans(sd, gd['land_x']), gd['land_y']) < TOL


@pytest.mark.parametrize('formula', [False, True])
def test_gathers(golden, formula):
    gd = golden('ops_small.npz')
    gs = ow.grid_sample_formula if formula else ow.grid_sample_torch
    assert linf(gs(gd['gs_x'], gd['gs_grid']), gd['gs_y']) < TOL
    assert linf(ow.warp_acc_flow(gd['gs_x'], gd['wf_flow'], gd['wf_mask'], formula=formula), gd['wf_y']) < TOL
    assert linf(ow.warp_acc_flow(gd['gs_x'], gd['wf_flow'], None, formula=formula), gd['wf_y_nomask']) < TOL


def test_resize_formula():
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1))
    for s in (128, 64):
        assert linf(ow.resize_bilinear_ac_formula(x, s), ow.resize_bilinear_ac_torch(x, s)) < 1e-5


@pytest.mark.parametrize('formula', [False, True])
def test_double_feature_warping(golden, formula):
    gd = golden('dfw.npz')
    d = make_generator_inputs(1, seed=int(gd['seed']))
    for level, size in ((0, 256), (1, 128), (2, 64)):
        x = torch.randn(1, 2, size, size, generator=torch.Generator().manual_seed(100 + level))
        y = ow.double_feature_warping(x, d['motion'], d['flow'], d['ifmask'], level, formula=formula)
        # formula path: mask>0.5 decisions can flip on exact ties only; demand near-total agreement
        diff = (y - gd['y%d' % level]).abs()
        assert float((diff > 1e-4).float().mean()) < (1e-4 if formula else 1e-9), level


def test_generator_ngf8(golden):
    gd = golden('gen_ngf8.npz')
    d = make_generator_inputs(2, seed=1234)
    assert sd_sha({k: d[k] for k in ('input', 'land1', 'land2', 'motion', 'flow', 'ifmask')}) == str(gd['inputs_sha256'])
    sd = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234)
    assert sd_sha(sd) == str(gd['weights_sha256'])
    y = og.generator_forward(sd, *generator_args(d), div=3, disp=3)
    assert linf(y, gd['y_disp3']) < 1e-4
    sd1 = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 1), seed=1234)
    y = og.generator_forward(sd1, *generator_args(d), div=3, disp=1)
    assert linf(y, gd['y_disp1']) < 1e-4


def test_generator_ngf8_grads(golden):
    gd = golden('gen_ngf8.npz')
    d = make_generator_inputs(2, seed=1234)
    sd = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234)
    for v in sd.values():
        v.requires_grad_(True)
    y = og.generator_forward(sd, *generator_args(d), div=3, disp=3)
    up = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    (y * up).sum().backward()
    norms = np.array([float(sd[k].grad.double().norm()) for k in sd])
    ref = gd['grad_norms']
    big = ref > 1e-3     # biases in front of IN have pure-noise grads
    assert np.allclose(norms[big], ref[big], rtol=2e-3)
    for k in gd:
        if k.startswith('grad_model'):
            g = sd[k[5:]].grad
            assert linf(g, gd[k]) <= 2e-3 * float(gd[k].abs().max()) + 1e-6, k


@pytest.mark.slow
def test_generator_ngf64(golden):
    gd = golden('gen_ngf64.npz')
    d = make_generator_inputs(2, seed=1234)
    sd = og.init_params(og.generator_param_shapes(3, 1, 64, 9, 3, 3), seed=1234)
    assert sd_sha(sd) == str(gd['weights_sha256'])
    assert sum(v.numel() for v in sd.values()) == int(gd['n_params']) == 15925553
    with torch.no_grad():
        y = og.generator_forward(sd, *generator_args(d), div=3, disp=3)
    assert linf(y, gd['y']) < 1e-4


def test_static_generator(golden):
    """ResnetStyle2Generator (networks.py:573-637) against the outputs of the reference's own
    class (tests/golden/make_static_golden.py)."""
    from oracle import static_generator as osg
    gd = golden('static_gen.npz')
    for tag, ngf in (('ngf8', 8), ('ngf64', 64)):
        size = int(gd['size_' + tag])
        x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(int(gd['seed_' + tag]))) * 2 - 1
        sd = og.init_params(osg.static_param_shapes(3, 1, ngf), seed=4321)
        assert sd_sha(sd) == str(gd['weights_sha256_' + tag])
        with torch.no_grad():
            y = osg.static_forward(sd, x, osg.style_code(2, size // 4))
        assert linf(y, gd['y_' + tag]) < 1e-5


def test_motion_grid(golden):
    """cal_motion256 (griddata over the landmark triangulation) against the output of the
    reference's own function (tests/golden/make_motion_golden.py)."""
    from oracle import motion as om
    gd = golden('motion.npz')
    for i in range(2):
        got = om.cal_motion256(gd['lm0_%d' % i].numpy(), gd['lm_%d' % i].numpy())
        assert got.shape == (256, 256, 2) and float(np.abs(got - gd['motion_%d' % i].numpy()).max()) == 0.0


def test_patchgan(golden):
    gd = golden('patchgan.npz')
    for cin in (1, 2):
        sd = og.init_params(od.patchgan_param_shapes(cin, 8), seed=4321 + cin)
        x = (torch.rand(2, cin, 256, 256, generator=torch.Generator().manual_seed(900 + cin)) * 2 - 1).requires_grad_(True)
        for v in sd.values():
            v.requires_grad_(True)
        y = od.patchgan_forward(sd, x)
        assert y.shape == (2, 1, 30, 30)
        assert linf(y, gd['y8_c%d' % cin]) < TOL
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
        (y * up).sum().backward()
        assert linf(x.grad, gd['dx8_c%d' % cin]) < 1e-4 * float(gd['dx8_c%d' % cin].abs().max()) + 1e-7
        for k in sd:
            ref = gd['g8_c%d_%s' % (cin, k)]
            assert linf(sd[k].grad, ref) <= 2e-3 * float(ref.abs().max()) + 1e-5, k
        sd64 = og.init_params(od.patchgan_param_shapes(cin, 64), seed=4321 + cin)
        with torch.no_grad():
            y = od.patchgan_forward(sd64, x[:1].detach())
        assert linf(y, gd['y64_c%d' % cin]) < 1e-4


def test_losses_and_masks(golden):
    gd = golden('losses.npz')
    assert abs(float(ol.gan_loss_lsgan(gd['pred'], True)) - float(gd['gan_real'])) < 1e-6
    assert abs(float(ol.gan_loss_lsgan(gd['pred'], False)) - float(gd['gan_fake'])) < 1e-6
    for mt in range(4):
        assert linf(ol.masked(gd['A'], gd['M'], mt), gd['masked%d' % mt]) < 1e-6


def test_tps(golden):
    gd = golden('tps.npz')
    img = gd['s_img'].clone().requires_grad_(True)
    w, fl = ot.sparse_image_warp(img, gd['s_src'], gd['s_dst'])
    assert linf(fl, gd['s_flow']) < 2e-3     # fp32 LU of an ill-conditioned system: solver-order noise
    assert linf(w, gd['s_warped']) < 2e-3
    (w * gd['s_up']).sum().backward()
    assert linf(img.grad, gd['s_dimg']) < 5e-3
    for tag, n in (('m68', 68), ('m76', 76)):
        tg = torch.Generator().manual_seed(int(gd[tag + '_seed']))
        src = torch.rand(1, n, 2, generator=tg) * (256 * 0.8) + 256 * 0.1
        dst = src + torch.randn(1, n, 2, generator=tg) * 4.0
        img = torch.rand(1, 256, 256, 1, generator=tg) * 2 - 1
        assert torch.equal(src, gd[tag + '_src']) and torch.equal(dst, gd[tag + '_dst'])
        w, fl = ot.sparse_image_warp(img, src, dst)
        assert linf(fl[:, ::8, ::8], gd[tag + '_flow_sub']) < 5e-2
        assert float((w - gd[tag + '_warped']).abs().mean()) < 2e-3


def test_tps_batched_equals_per_sample():
    tg = torch.Generator().manual_seed(3)
    src = torch.rand(3, 20, 2, generator=tg) * 24 + 4
    dst = src + torch.randn(3, 20, 2, generator=tg)
    img = torch.rand(3, 32, 32, 2, generator=tg)
    w, fl = ot.sparse_image_warp(img, src, dst)
    for i in range(3):
        wi, fi = ot.sparse_image_warp(img[i:i + 1], src[i:i + 1], dst[i:i + 1])
        assert linf(w[i:i + 1], wi) < 1e-4 and linf(fl[i:i + 1], fi) < 1e-4


# ------------------------------------------------------------------ glue around the frozen aux nets (aux.npz)
def _aux_inputs(gd, c, i):
    seed = int(gd['lm_c%d_%d_seed' % (c, i)])
    x = torch.rand(1, c, 256, 256, generator=torch.Generator().manual_seed(seed)) * 2 - 1
    return x, gd['lm_c%d_%d_win' % (c, i)]


def test_get_lm_crop_bicubic_vs_reference(golden):
    """oracle.aux_glue.get_lm == GeomGMIFWForeModel.get_lm (geomgm_ifw_fore_model.py:390-415) run by
    make_aux_golden.py with the stand-in landmark net: the tensor fed to the net and the re-projected landmarks."""
    from oracle import aux_glue as oa
    from animateportrait_amd import standins
    gd = golden('aux.npz')
    net = standins.StandinLandmarkNet()
    for c in (1, 3):
        for i in range(3):
            x, win = _aux_inputs(gd, c, i)
            box = oa.get_lm_box(x, win)
            ref = gd['lm_c%d_%d_box' % (c, i)]
            assert linf(box[:, :ref.shape[1]], ref) < 1e-6
            assert np.allclose(box.double().sum(dim=(0, 2, 3)).numpy(), gd['lm_c%d_%d_boxsum' % (c, i)], rtol=1e-9)
            assert linf(oa.get_lm(net, x, win), gd['lm_c%d_%d_out' % (c, i)]) < 1e-4     # pixels


def test_faceloss_crop_and_feature_l1_vs_reference(golden):
    """oracle.aux_glue.face_loss == networks.FaceLoss.forward (networks.py:2881-2966) with the stand-in feature net."""
    from oracle import aux_glue as oa
    from animateportrait_amd import standins
    gd = golden('aux.npz')
    net = standins.StandinFaceNet()
    a = (torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(int(gd['fl_seed_a']))) * 2 - 1)
    b = (torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(int(gd['fl_seed_b']))) * 2 - 1)
    a.requires_grad_(True)
    h1 = oa.crop_head_bbox(a.repeat(1, 3, 1, 1), gd['fl_bb1'])
    h2 = oa.crop_head_bbox(b.repeat(1, 3, 1, 1), gd['fl_bb2'])
    assert linf(h1[:, :1], gd['fl_head1']) < 1e-6 and linf(h2[:, :1], gd['fl_head2']) < 1e-6
    loss = oa.face_loss(net, a.repeat(1, 3, 1, 1), b.repeat(1, 3, 1, 1), gd['fl_bb1'], gd['fl_bb2'])
    assert abs(float(loss) - float(gd['fl_loss'])) < 1e-6
    loss.backward()
    assert linf(a.grad[:, :, ::4, ::4], gd['fl_grad_a_sub']) < 1e-8
    assert abs(float(a.grad.double().abs().sum()) - float(gd['fl_grad_a_abs'])) < 1e-6 * float(gd['fl_grad_a_abs'])


def test_kp_to_map_and_flow_network_warp_vs_reference(golden):
    """oracle.aux_glue == kp_to_map_some / flow_network_warp (geomgm_ifw_fore_model.py:19-51, 69-84)."""
    from oracle import aux_glue as oa
    from animateportrait_amd import standins
    gd = golden('aux.npz')
    j1 = oa.kp_to_map_some((224, 224), gd['kp_lm1'].numpy() * 7 / 8)
    assert np.array_equal(j1.numpy().astype(np.uint8), gd['kp_j1'])
    wf, rm = oa.flow_network_warp(standins.StandinFlowNet(), torch.zeros(1, 3, 256, 256), gd['kp_lm1'], gd['kp_lm2'])
    assert linf(wf[:, :, ::2, ::2], gd['fw_flow_sub']) < 1e-5 and linf(rm[:, :, ::2, ::2], gd['fw_mask_sub']) < 1e-6
    assert abs(float(wf.double().abs().sum()) - float(gd['fw_flow_abs'])) < 1e-6 * float(gd['fw_flow_abs'])
    assert float(rm.double().sum()) == pytest.approx(float(gd['fw_mask_sum']), rel=1e-7)


def test_cv2_filled_circle_rule_table():
    """draw2(op=0) uses cv2.circle(..., -1).  cv2 is absent here (parity unpinned for this one function): the oracle
    restates OpenCV's octant walk; this table is what that rule gives for the small radii -- the r=1 plus, the r=2
    diamond, and the r=3 disc of the reference's landmark maps (row half-widths for |dy| = 0..r) -- and the C
    library's host-side table (what ap_landmark_discs rasterises) agrees with the oracle for every radius."""
    from oracle import aux_glue as oa
    from animateportrait_amd import losses
    table = {0: [0], 1: [1, 0], 2: [2, 1, 0], 3: [3, 2, 2, 0], 4: [4, 3, 3, 2, 0], 5: [5, 4, 4, 4, 3, 0]}
    for r, hw in table.items():
        assert oa.cv2_filled_circle_rows(r) == hw, r
    for r in range(0, 32):
        assert losses.circle_rows(r) == oa.cv2_filled_circle_rows(r), r
    img = oa.draw2(16, 16, [[7.5, 8.0], [0.2, 15.0]], 3)       # np.round: 7.5 -> 8 (half to even); clipping at the border
    pix = (img[0] > 0).numpy()
    assert pix.sum() == 29 + 11 and pix[8, 5:12].all() and pix[5, 8] and not pix[5, 7]
