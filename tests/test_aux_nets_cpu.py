"""CPU: the stock-PyTorch mirrors of the three frozen aux networks (animateportrait_amd/aux_nets.py) against the golden made
from the REFERENCE's own classes (tests/golden/make_auxnets_golden.py; Module2/models/mobilefacenet.py:104-159,
facenet.py:200-282, modnet.py:204-236 + backbones/): state_dict key lists -- names, shapes, dtypes, order, so a reference
checkpoint loads strictly -- outputs for seeded weights, and the checkpoint loaders / auto-attach of the models."""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from make_auxnets_golden import SEEDS, inputs, keys_of, seeded_state_scaled      # noqa: E402

from animateportrait_amd import aux_nets                                          # noqa: E402


def _golden():
    return np.load(os.path.join(HERE, 'golden', 'auxnets.npz'))


def _seeded(tag, net):
    g = _golden()
    ks = keys_of(net)
    assert [k for k, _, _ in ks] == [str(k) for k in g[tag + '_keys']], tag
    assert [str(s) for _, s, _ in ks] == [str(s) for s in g[tag + '_shapes']], tag
    assert [d for _, _, d in ks] == [str(d) for d in g[tag + '_dtypes']], tag
    net.load_state_dict(seeded_state_scaled(ks, SEEDS[tag]), strict=True)
    return net.eval(), g


def _close(got, ref, rel=2e-5):
    ref = torch.from_numpy(np.asarray(ref))
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert float((got - ref).abs().max()) <= rel * float(ref.abs().max()) + 1e-6, float((got - ref).abs().max())


def test_mobilefacenet_matches_reference_class():
    net, g = _seeded('mobilefacenet', aux_nets.MobileFaceNet((112, 112), 136))
    with torch.no_grad():
        emb, feat = net(inputs()['mobilefacenet'])
    _close(emb, g['mobilefacenet_out'])
    _close(feat[:, ::16], g['mobilefacenet_feat_sub'])
    assert abs(float(feat.double().abs().sum()) - float(g['mobilefacenet_feat_abs'])) <= 1e-5 * float(g['mobilefacenet_feat_abs'])


def test_sphere20a_matches_reference_class():
    net, g = _seeded('sphere20a', aux_nets.Sphere20a())
    with torch.no_grad():
        feats = net(inputs()['sphere20a'])
    assert len(feats) == 5
    for i, f in enumerate(feats):
        _close(f if f.dim() == 2 else f[:, ::16, ::2, ::2], g['sphere20a_f%d' % i])
        assert abs(float(f.double().abs().sum()) - float(g['sphere20a_f%d_abs' % i])) <= 1e-5 * float(g['sphere20a_f%d_abs' % i])


def test_modnet_matches_reference_class():
    net, g = _seeded('modnet', aux_nets.MODNet(backbone_pretrained=False))
    x = inputs()['modnet']
    with torch.no_grad():
        sem, det, matte = net(x, False)
        s2, d2, matte_inf = net(x, True)
    assert s2 is None and d2 is None and torch.equal(matte, matte_inf)
    _close(matte, g['modnet_matte'])
    _close(sem, g['modnet_semantic'])
    _close(det[:, :, ::2, ::2], g['modnet_detail_sub'])
    # the backbone is one object under two prefixes, as in the reference checkpoint
    sd = net.state_dict()
    assert sd['backbone.model.features.0.0.weight'].data_ptr() == sd['lr_branch.backbone.model.features.0.0.weight'].data_ptr()


def test_checkpoint_loaders_and_auto_attach(tmp_path):
    """Files written the way the published checkpoints are laid out (geomgm_ifw_fore_model.py:362-376, networks.py:3044-3053)
    load strictly, come back frozen / in eval mode, and attach_aux_networks fills exactly the empty aux slots."""
    mf, _ = _seeded('mobilefacenet', aux_nets.MobileFaceNet((112, 112), 136))
    sp, _ = _seeded('sphere20a', aux_nets.Sphere20a())
    mo, _ = _seeded('modnet', aux_nets.MODNet())
    torch.save({'state_dict': mf.state_dict(), 'epoch': 3}, tmp_path / aux_nets.MOBILEFACENET_CKPT)
    torch.save({'module.' + k: v for k, v in mo.state_dict().items()}, tmp_path / aux_nets.MODNET_CKPT)
    sph = dict(sp.state_dict())
    sph['fc6.weight'] = torch.zeros(512, 10574)              # the classifier head the reference drops
    torch.save(sph, tmp_path / 'sphere20a_20171020.pth')
    dev = torch.device('cpu')
    x = inputs()
    for load, path, want, xin, pick in ((aux_nets.load_mobilefacenet, aux_nets.MOBILEFACENET_CKPT, mf, x['mobilefacenet'], lambda o: o[0]),
                                        (aux_nets.load_sphere20a, 'sphere20a_20171020.pth', sp, x['sphere20a'], lambda o: o[-1]),
                                        (aux_nets.load_modnet, aux_nets.MODNET_CKPT, mo, x['modnet'], lambda o: o[2])):
        got = load(str(tmp_path / path), dev, fold=False)
        assert not got.training and not any(p.requires_grad for p in got.parameters())
        for (ka, va), (kb, vb) in zip(want.state_dict().items(), got.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
        # the default loader folds the (frozen) BatchNorms into the convolutions: same function
        folded = load(str(tmp_path / path), dev)
        nbn = lambda net: sum(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())          # noqa: E731
        assert nbn(folded) == (0 if load is not aux_nets.load_modnet else nbn(want) - 52)       # MODNet keeps its half-BN IBNorm layers
        with torch.no_grad():
            a, b = pick(want(xin)), pick(folded(xin))
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-6
    opt = types.SimpleNamespace(face_recog_model=str(tmp_path / 'sphere20a_20171020.pth'), identity_loss=2)
    model = types.SimpleNamespace(aux={'modnet': None, 'landmarks': None, 'faceloss': None, 'netF': None}, opt=opt, device=dev, isTrain=True)
    assert aux_nets.attach_aux_networks(model, str(tmp_path), verbose=False) == ['modnet', 'landmarks', 'faceloss']
    assert isinstance(model.aux['modnet'], aux_nets.MODNet) and isinstance(model.aux['landmarks'].net, aux_nets.MobileFaceNet)
    assert isinstance(model.aux['faceloss'].net.net, aux_nets.Sphere20a) and model.aux['netF'] is None
    assert aux_nets.attach_aux_networks(model, str(tmp_path), verbose=False) == []        # slots already filled
    test_model = types.SimpleNamespace(aux={'modnet': None, 'netF': None}, opt=opt, device=dev, isTrain=False)
    assert aux_nets.attach_aux_networks(test_model, str(tmp_path), verbose=False) == ['modnet']
    empty = types.SimpleNamespace(aux={'modnet': None, 'landmarks': None, 'faceloss': None}, opt=opt, device=dev, isTrain=True)
    opt2 = types.SimpleNamespace(face_recog_model='./checkpoints/none.pth', identity_loss=2)
    empty.opt = opt2
    assert aux_nets.attach_aux_networks(empty, str(tmp_path / 'nowhere'), verbose=False) == []
