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
