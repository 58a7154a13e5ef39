"""CPU: host side of the in-process clip pipeline: the Module1 content network mirror
against the reference class (module1.npz), the landmark txt format, the Savitzky-Golay smoothing."""
import os
import sys

import numpy as np
import torch

from conftest import linf, GOLDEN

sys.path.insert(0, GOLDEN)


def test_module1_content_network_matches_reference(golden):
    from animateportrait_amd.module1 import Audio2LandmarkContent
    from make_module1_golden import seeded_state
    gd = golden('module1.npz')
    net = Audio2LandmarkContent()
    mine = [(k, str(tuple(v.shape)), str(v.dtype)) for k, v in net.state_dict().items()]
    ref = list(zip(gd['keys'].tolist(), gd['shapes'].tolist(), gd['dtypes'].tolist()))
    assert mine == ref                      # same keys, order, shapes: the reference checkpoint loads strictly
    net.load_state_dict(seeded_state([(k, eval(s), d) for k, s, d in ref]), strict=True)
    net.eval()
    with torch.no_grad():
        out, fid = net(gd['au'], gd['fid'])
    assert linf(out, gd['out']) < 1e-6 and fid.shape == (6, 204)


def _seeded(net, gd, tag, seed):
    from make_module1_golden import seeded_state
    mine = [(k, str(tuple(v.shape)), str(v.dtype)) for k, v in net.state_dict().items()]
    ref = list(zip(gd[tag + 'keys'].tolist(), gd[tag + 'shapes'].tolist(), gd[tag + 'dtypes'].tolist()))
    assert mine == ref                      # same keys, order, shapes: the reference checkpoint loads strictly
    sd = seeded_state([(k, eval(s), d) for k, s, d in ref], seed=seed)
    sd = {k: (net.state_dict()[k] if k.endswith('pe.pe') else v) for k, v in sd.items()}   # constant table (as the golden)
    net.load_state_dict(sd, strict=True)
    return net.eval()


def test_module1_speaker_aware_network_and_test_configuration_match_reference(golden):
    """``Audio2LandmarkPos`` against ``Audio2landmark_pos`` and the content net in the configuration
    ``Audio2landmark_model`` builds (use_prior_net=True): keys / order / shapes, outputs for seeded weights."""
    from animateportrait_amd.module1 import Audio2LandmarkContent, Audio2LandmarkPos
    gd = golden('module1.npz')
    netc = _seeded(Audio2LandmarkContent(use_prior_net=True, drop_out=0.5), gd, 'c_', 78)
    netg = _seeded(Audio2LandmarkPos(drop_out=0.5), gd, 'g_', 79)
    with torch.no_grad():
        assert linf(netc(gd['au'], gd['fid'])[0], gd['c_out']) < 1e-6
        pred, face, spk = netg(gd['au'], gd['g_emb'], gd['fid'].repeat(6, 1), None, torch.zeros(6, 128))
    assert linf(pred, gd['g_out']) < 1e-6 and face.shape == (1, 204) and spk.shape == (6, 128)


def test_module1_clip_pipeline_matches_reference_methods(golden):
    """predict_landmarks_speaker_aware == the reference's __train_face_and_pos__ / __calib_baseline_pred_fls__ /
    __solve_inverse_lip2__ + the loop of __train_pass__ on a 600-window clip (two segments); the inverted-lip fix and
    add_naive_eye on their own against the reference functions."""
    from animateportrait_amd import module1 as m1
    gd = golden('module1.npz')
    netc = _seeded(m1.Audio2LandmarkContent(use_prior_net=True, drop_out=0.5), gd, 'c_', 78)
    netg = _seeded(m1.Audio2LandmarkPos(drop_out=0.5), gd, 'g_', 79)
    gp = torch.Generator().manual_seed(int(gd['p_seed']))
    au = torch.randn(int(gd['p_T']), 18, 80, generator=gp)
    spk = torch.randn(256, generator=gp)
    fl = m1.predict_landmarks_speaker_aware(netg, netc, au, spk, gd['fid'].view(-1))
    assert fl.shape == (600, 204)
    ref = gd['p_sub'].numpy()
    assert np.abs(fl[::16] - ref).max() < 2e-5 * np.abs(ref).max()
    assert abs(fl.sum() - float(gd['p_sum'])) < 1e-4 * float(gd['p_abs'])
    lips = gd['lip_in'].numpy().astype(np.float64)
    fixed = m1.solve_inverse_lip(lips.copy())
    assert np.abs(fixed - gd['lip_out'].numpy()).max() < 1e-12 and np.abs(fixed - lips).max() > 0.1   # (some frames were fixed)
    np.random.seed(5)
    eyes = m1.add_naive_eye(gd['eye_in'].numpy().copy())
    assert np.abs(eyes - gd['eye_out'].numpy()).max() < 1e-6
    img = m1.to_image_landmarks(fl, scale=0.01, shift=(-128.0, -120.0), rng=np.random.RandomState(1))
    assert img.shape == (600, 68, 3) and np.isfinite(img).all()


def test_predict_landmarks_postprocessing():
    from animateportrait_amd.module1 import Audio2LandmarkContent, predict_landmarks
    from scipy.signal import savgol_filter
    torch.manual_seed(1)
    net = Audio2LandmarkContent().eval()
    au = torch.randn(40, 18, 80)
    fid = torch.randn(204) * 0.2
    lm = predict_landmarks(net, au, fid, scale=0.01, shift=(-128.0, -120.0), segment=16)
    assert lm.shape == (40, 68, 2) and np.isfinite(lm).all()
    with torch.no_grad():
        raw = (net(au, fid.view(1, -1))[0] + fid.view(1, -1)).view(-1, 68, 3).numpy()
    raw[:, :, :2] = -raw[:, :, :2] / 0.01 - np.array([-128.0, -120.0], dtype=np.float32)
    flat = raw.reshape(40, 204).astype(np.float64)
    flat[:, :144] = savgol_filter(flat[:, :144], 15, 3, axis=0)          # main_end2end_module2.py:269-270
    flat[:, 144:] = savgol_filter(flat[:, 144:], 5, 3, axis=0)
    assert np.abs(lm - flat.reshape(40, 68, 3)[:, :, :2]).max() < 1e-3


def test_landmark_txt_roundtrip(tmp_path):
    from animateportrait_amd import stream
    rng = np.random.default_rng(0)
    ori = rng.uniform(20, 230, (68, 2)).astype(np.float32)
    seq = ori[None] + rng.normal(0, 2, (7, 68, 2)).astype(np.float32)
    d = str(tmp_path)
    stream.write_landmark_txt(os.path.join(d, 'ori.txt'), ori)
    for k, lm in enumerate(seq):
        stream.write_landmark_txt(os.path.join(d, '%05d.txt' % k), lm)
    o2, s2 = stream.load_landmark_dir(d)
    assert np.allclose(o2, ori, atol=1e-4) and s2.shape == (7, 68, 2) and np.allclose(s2, seq, atol=1e-4)
    assert stream.window_of(ori)[1] - stream.window_of(ori)[0] > 0


def test_module1_checkpoint_loader_and_face_normalisation(tmp_path, golden):
    """load_module1 (train_audio2landmark.py:55-79: 'G' minus comb_mlp, 'model_g_face_id') and the photo-landmark
    normalisation of main_end2end_module2.py:196-204 + util/utils.py:348-359 with its inverse (:311-315)."""
    from animateportrait_amd import module1 as m1
    gd = golden('module1.npz')
    netc = _seeded(m1.Audio2LandmarkContent(use_prior_net=True, drop_out=0.5), gd, 'c_', 78)
    netg = _seeded(m1.Audio2LandmarkPos(drop_out=0.5), gd, 'g_', 79)
    g_sd = dict(netg.state_dict())
    g_sd['comb_mlp.0.weight'] = torch.zeros(3, 3)                       # entries the reference drops (:64)
    torch.save({'G': g_sd}, tmp_path / 'g.pth')
    torch.save({'model_g_face_id': netc.state_dict()}, tmp_path / 'c.pth')
    lg, lc = m1.load_module1(str(tmp_path / 'g.pth'), str(tmp_path / 'c.pth'))
    assert not lg.training and not lc.training and not any(p.requires_grad for p in lg.parameters())
    for a, b in ((netg, lg), (netc, lc)):
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
    rng = np.random.RandomState(0)
    shape = np.stack([np.linspace(60, 200, 68) + rng.randn(68), 100 + 40 * np.sin(np.arange(68)) + rng.randn(68), rng.randn(68)], 1)
    shape[0, 0], shape[16, 0] = 200.0, 60.0                             # x[0] - x[16] sets the scale
    fid, scale, shift = m1.adjust_and_norm_input_face(shape, std_face_z=np.arange(68) / 10.0)
    assert abs(scale - 1.6 / 140.0) < 1e-12 and np.allclose(shift, [-130.0, -0.5 * (shape[0, 1] + shape[16, 1])])
    assert np.allclose(fid[:, 2], np.arange(68) / 100.0) and abs(fid[0, 0] + 0.8) < 1e-9 and abs(fid[16, 0] - 0.8) < 1e-9
    back = m1.photo_landmarks_in_pixels(fid, scale, shift)
    adj = shape[:, :2].copy()
    adj[49:54, 1] += 1.; adj[55:60, 1] -= 1.; adj[[37, 38, 43, 44], 1] -= 2; adj[[40, 41, 46, 47], 1] += 2
    assert np.abs(back - adj).max() < 1e-4
