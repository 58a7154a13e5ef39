"""GPU (-m gpu): the audio half of BASELINE configs[4] on the device -- both Module1 landmark networks and the
``Audio2landmark_model.test`` post-processing against the reference-made golden (tests/golden/module1.npz), then the
reference's example clip (tests/golden/female12.wav) from samples to frames: mel windows (animateportrait_amd/audio.py,
pinned to the reference's own extraction on CPU) -> Module1 on the MI355X -> image landmarks -> ClipStreamer."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import linf, GOLDEN

sys.path.insert(0, GOLDEN)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def _nets(gd, dev):
    from animateportrait_amd import module1 as m1
    from test_stream_cpu import _seeded
    netc = _seeded(m1.Audio2LandmarkContent(use_prior_net=True, drop_out=0.5), gd, 'c_', 78)
    netg = _seeded(m1.Audio2LandmarkPos(drop_out=0.5), gd, 'g_', 79)
    return netc.to(dev), netg.to(dev)


def test_module1_networks_on_device_match_reference_golden(dev, golden):
    """Module1/src/models/model_audio2landmark.py:28-90 (content) and :296-386 (speaker-aware) with seeded weights: the
    device outputs against the outputs of the reference's own classes (CPU fp32)."""
    gd = golden('module1.npz')
    netc, netg = _nets(gd, dev)
    with torch.no_grad():
        c = netc(gd['au'].to(dev), gd['fid'].to(dev))[0]
        pred, face, spk = netg(gd['au'].to(dev), gd['g_emb'].to(dev), gd['fid'].repeat(6, 1).to(dev), None,
                               torch.zeros(6, 128, device=dev))
    assert linf(c, gd['c_out']) < 2e-5 * float(gd['c_out'].abs().max()) + 1e-6
    assert linf(pred, gd['g_out']) < 2e-5 * float(gd['g_out'].abs().max()) + 1e-6


def test_module1_clip_pipeline_on_device_matches_reference_methods(dev, golden):
    """predict_landmarks_speaker_aware on the device == the reference's ``test`` loop (600 windows, two segments,
    train_audio2landmark.py:247-309) to 1e-4 of the landmark range."""
    from animateportrait_amd import module1 as m1
    gd = golden('module1.npz')
    netc, netg = _nets(gd, dev)
    gp = torch.Generator().manual_seed(int(gd['p_seed']))
    au = torch.randn(int(gd['p_T']), 18, 80, generator=gp)
    spk = torch.randn(256, generator=gp)
    fl = m1.predict_landmarks_speaker_aware(netg, netc, au, spk, gd['fid'].view(-1))
    ref = gd['p_sub'].numpy()
    assert fl.shape == (600, 204) and np.abs(fl[::16] - ref).max() < 1e-4 * np.abs(ref).max()


def test_example_clip_from_samples_to_frames(dev, golden):
    """examples/female12.wav -> mel (62.5 frames/s) -> 18-frame windows -> Module1 (device) -> landmarks in image pixels ->
    motion grids / landmark maps / generator (ClipStreamer).  Module1 on the device agrees with the same networks on the
    CPU; the frames are those of the streamer driven by the CPU landmarks."""
    import contextlib
    import io
    from animateportrait_amd import audio, module1 as m1, stream
    from animateportrait_amd.options.base_options import TestOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.synthetic import make_landmarks
    gd = golden('module1.npz')
    netc, netg = _nets(gd, dev)
    w = audio.clip_audio_features(os.path.join(GOLDEN, 'female12.wav'), max_frames=64)
    assert w.shape == (64, 18, 80)
    spk = torch.randn(256, generator=torch.Generator().manual_seed(3))
    fid = gd['fid'].view(-1)
    fl = m1.predict_landmarks_speaker_aware(netg, netc, w, spk, fid)
    fl_cpu = m1.predict_landmarks_speaker_aware(netg.cpu(), netc.cpu(), w, spk, fid)
    assert fl.shape == (64, 204) and np.abs(fl - fl_cpu).max() < 1e-4 * np.abs(fl_cpu).max()

    def to_pixels(f):
        # Module1's normalised face -> displacements around the photo's landmarks (random weights do not draw a face: the
        # motion of the prediction is kept, its mean shape replaced; main_end2end_module2.py:262-272 otherwise)
        img = m1.to_image_landmarks(f, scale=0.01, shift=(-128.0, -128.0), rng=np.random.RandomState(0))[:, :, :2]
        d = img - img.mean(0, keepdims=True)
        return lm0.numpy()[None] + np.clip(d, -6.0, 6.0)
    lm0 = make_landmarks(1, torch.Generator().manual_seed(9))[0]
    seq, seq_cpu = to_pixels(fl), to_pixels(fl_cpu)
    assert np.abs(seq - seq_cpu).max() < 2e-2            # px
    opt = TestOptions().parse(['--model', 'geomcgt_ifw_test', '--netG', 'resnet_9blocks_rcatland32_full_ifw',
                               '--dataset_mode', 'synthetic', '--name', 'drawing_stream', '--output_nc', '1', '--ngf', '8',
                               '--netg_resb_div', '3', '--netg_resb_disp', '3', '--gpu_ids', '0'])
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(1)
        model = create_model(opt)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 256), torch.linspace(-1, 1, 256), indexing='ij')
    photo = torch.stack([torch.sin(3 * xx + yy), torch.cos(2 * yy - xx), xx * yy], 0).unsqueeze(0).contiguous()
    matte = (((yy / 0.8) ** 2 + (xx / 0.6) ** 2) < 1).float().view(1, 1, 256, 256)
    frames = stream.ClipStreamer(model, batch=16).run(photo, lm0, seq[:24], matte=matte)
    assert frames.shape == (24, 1, 256, 256) and bool(torch.isfinite(frames).all()) and float(frames.std()) > 1e-3
    frames_cpu = stream.ClipStreamer(model, batch=16).run(photo, lm0, seq_cpu[:24], matte=matte)
    assert float((frames - frames_cpu).abs().mean()) < 1e-3


def test_end2end_cli_from_wav(dev, golden, tmp_path, monkeypatch):
    """``python -m animateportrait_amd.end2end --wav ... --photo_landmarks ...``: the whole of main_end2end_module2.py:181-343
    in one process -- Module1 checkpoints loaded by path (the reference's key layout), mel windows from the wav, landmarks,
    frames written (64 frames: the reference's add_naive_eye needs more than 45)."""
    from PIL import Image
    from animateportrait_amd import audio, end2end, module1 as m1, stream
    from animateportrait_amd.synthetic import make_landmarks
    from oracle import generator as og, static_generator as osg
    gd = golden('module1.npz')
    netc, netg = _nets(gd, torch.device('cpu'))
    monkeypatch.chdir(tmp_path)
    os.makedirs('checkpoints/e2e'); os.makedirs('checkpoints/static'); os.makedirs('m1')
    torch.save({'G': netg.state_dict()}, 'm1/g.pth')
    torch.save({'model_g_face_id': netc.state_dict()}, 'm1/c.pth')
    torch.save(og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234), 'checkpoints/e2e/7_net_G_A.pth')
    torch.save(og.init_params(osg.static_param_shapes(3, 1, 64), seed=4321), 'checkpoints/static/drawing.pth')
    yy, xx = np.meshgrid(np.linspace(-1, 1, 256), np.linspace(-1, 1, 256), indexing='ij')
    Image.fromarray(((np.stack([np.sin(3 * xx + yy), np.cos(2 * yy - xx), xx * yy], -1) + 1) * 127.5).astype(np.uint8)).save('photo.png')
    Image.fromarray(((((yy / 0.8) ** 2 + (xx / 0.6) ** 2) < 1) * 255).astype(np.uint8)).save('matte.png')
    lm0 = make_landmarks(1, torch.Generator().manual_seed(9))[0].numpy()
    lm0[0, 0], lm0[16, 0] = 190.0, 70.0              # jaw ends set the normalisation scale (util/utils.py:349)
    np.savetxt('photo_lm.txt', np.concatenate([lm0, np.zeros((68, 1))], 1))
    np.savetxt('spk.txt', np.abs(np.random.RandomState(3).randn(256)) * 0.1)
    np.savetxt('trg.txt', np.abs(np.random.RandomState(4).randn(256)) * 0.1)
    from animateportrait_amd import autovc
    torch.manual_seed(5)
    vc = autovc.Generator(16, 256, 512, 16)
    torch.save({'model': vc.state_dict()}, 'm1/autovc.pth')
    wav = os.path.join(GOLDEN, 'female12.wav')
    base = ['--photo', 'photo.png', '--matte', 'matte.png', '--wav', wav, '--photo_landmarks', 'photo_lm.txt',
            '--load_a2l_G_name', 'm1/g.pth', '--load_a2l_C_name', 'm1/c.pth', '--max_frames', '64',
            '--out', 'out', '--batch', '8', '--name', 'e2e', '--epoch', '7', '--ngf', '8', '--checkpoints_dir', 'checkpoints']
    # the reference feeds Module1 the AutoVC-converted spectrogram and the clip's speaker embedding: neither is optional
    for bad in (base, base + ['--speaker_emb', 'spk.txt']):            # no embedding / no converter checkpoint at the default path
        with pytest.raises(SystemExit):
            end2end.main(bad)
    argv = base + ['--speaker_emb', 'spk.txt', '--load_AUTOVC_name', 'm1/autovc.pth', '--autovc_target_emb', 'trg.txt']
    np.random.seed(0)
    assert end2end.main(argv) == 0
    files = sorted(os.listdir('out/frames'))
    assert files == ['%05d.png' % k for k in range(64)]
    got = np.stack([np.asarray(Image.open(os.path.join('out/frames', f))) for f in files])
    assert got.shape == (64, 256, 256, 3) and got.std() > 1.0
    # the landmark sequence the CLI derived: finite, inside a sane range around the photo's landmarks
    fid, scale, shift = m1.adjust_and_norm_input_face(np.loadtxt('photo_lm.txt'))
    g2, c2 = m1.load_module1('m1/g.pth', 'm1/c.pth', dev)
    vc2 = autovc.load_generator('m1/autovc.pth', dev)
    conv = lambda mel: autovc.convert_mel(vc2, mel, None, np.loadtxt('spk.txt'), np.loadtxt('trg.txt'), dev)       # noqa: E731
    fl = m1.predict_landmarks_speaker_aware(g2, c2, audio.clip_audio_features(wav, max_frames=64, converter=conv),
                                            np.loadtxt('spk.txt'), fid.reshape(-1))
    np.random.seed(0)
    seq = m1.to_image_landmarks(fl, scale=scale, shift=shift)[:, :, :2]
    assert np.isfinite(seq).all() and np.abs(seq.mean(0) - m1.photo_landmarks_in_pixels(fid, scale, shift)).max() < 400



@pytest.mark.parametrize('cfg', [(512, 16, 2, True, 96), (545, 512, 3, False, 96), (80, 256, 1, False, 40), (40, 48, 2, True, 33)])
def test_lstm_recurrence_kernel_equals_the_library_lstm(dev, cfg):
    """ap_lstm_recurrence (csrc/lstm.hip) through lstm_hip.lstm_forward against nn.LSTM on the device: the AutoVC encoder's 2-layer
    bidirectional LSTM (16 hidden units), its 3-layer decoder LSTM (512), and two other sizes of both kernels."""
    from animateportrait_amd import lstm_hip
    i, h, layers, bi, t = cfg
    torch.manual_seed(i + h)
    lstm = torch.nn.LSTM(i, h, layers, batch_first=True, bidirectional=bi).to(dev).eval()
    x = torch.randn(1, t, i, device=dev)
    with torch.no_grad():
        assert lstm_hip.supported(lstm, x)
        ref = lstm(x)[0]
        got = lstm_hip.lstm_forward(lstm, x)
    torch.cuda.synchronize()
    lstm_hip.check_timeouts()
    assert got.shape == ref.shape
    assert linf(got, ref) < 2e-5


def test_autovc_converter_on_the_lstm_kernels_matches_the_reference_golden(dev):
    """convert_mel on the device (LSTM time loops in ap_lstm_recurrence) against the output of the reference's own Generator
    (tests/golden/autovc.npz, made by make_autovc_golden.py on the CPU with the same seeded weights and inputs)."""
    from animateportrait_amd import autovc
    from make_autovc_golden import SEED, make_inputs
    from make_auxnets_golden import seeded_state_scaled, keys_of
    g = np.load(os.path.join(GOLDEN, 'autovc.npz'))
    G = autovc.Generator(16, 256, 512, 16).eval()
    G.load_state_dict(seeded_state_scaled(keys_of(G), SEED), strict=True)
    G = G.to(dev)
    mel, f0, e_src, e_trg = make_inputs()
    got = autovc.convert_mel(G, mel, f0, e_src, e_trg, dev)
    ref = g['converted']
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-4 * float(np.abs(ref).max()) + 1e-5
