"""CPU: the AutoVC converter mirror (animateportrait_amd/autovc.py) against the golden made from the REFERENCE's own Generator
and quantize_f0_interp (tests/golden/make_autovc_golden.py), the converter loop, the checkpoint loader, and its place in the
clip's audio features."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from make_autovc_golden import SEED, make_inputs                                   # noqa: E402
from make_auxnets_golden import seeded_state_scaled, keys_of                        # noqa: E402

from animateportrait_amd import autovc, audio                                       # noqa: E402


def _net():
    g = np.load(os.path.join(HERE, 'golden', 'autovc.npz'))
    G = autovc.Generator(16, 256, 512, 16).eval()
    ks = keys_of(G)
    assert [k for k, _, _ in ks] == [str(k) for k in g['keys']]
    assert [str(s) for _, s, _ in ks] == [str(s) for s in g['shapes']]
    G.load_state_dict(seeded_state_scaled(ks, SEED), strict=True)
    return G, g


def _close(got, ref, rel=2e-5):
    got, ref = torch.as_tensor(got), torch.from_numpy(np.asarray(ref))
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= rel * float(ref.abs().max()) + 1e-6


def test_generator_and_f0_quantiser_match_reference():
    G, g = _net()
    mel, f0, e_src, e_trg = make_inputs()
    f0q = autovc.quantize_f0_interp(f0)
    assert np.array_equal(f0q.argmax(1), g['f0q_index']) and np.array_equal(f0q.sum(1), g['f0q_rowsum'])
    assert f0q[10:25, 0].all() and f0q[40, 256] == 1 and f0q[41, 1] == 1             # unvoiced -> bin 0; 1.0 -> 256; 0.0 -> 1
    x, pad = autovc._pad_seq(mel.astype('float32'))
    f, _ = autovc._pad_seq(f0q)
    to = lambda a: torch.from_numpy(a[np.newaxis].astype('float32'))                 # noqa: E731
    with torch.no_grad():
        m, mp, codes = G(to(x), to(e_src), to(f), to(e_trg), to(f))
    assert pad == 21
    _close(m, g['mel_out']); _close(mp, g['mel_postnet']); _close(codes, g['codes'])  # noqa: E702
    _close(autovc.convert_mel(G, mel, f0, e_src, e_trg), g['converted'])
    with pytest.raises(ValueError):
        autovc.quantize_f0_interp(np.array([0.5, 1.5]))


def test_long_clip_is_converted_in_4096_frame_pieces():
    """:248-266: pieces of 4096 frames, only the last one padded; a piece boundary restarts the LSTM state."""
    G, _ = _net()
    rng = np.random.RandomState(1)
    mel = rng.rand(4096 + 40, 80).astype(np.float32)
    e1, e2 = rng.rand(256).astype(np.float32) * 0.1, rng.rand(256).astype(np.float32) * 0.1
    full = autovc.convert_mel(G, mel, None, e1, e2)
    assert full.shape == mel.shape
    a = autovc.convert_mel(G, mel[:4096], None, e1, e2)
    b = autovc.convert_mel(G, mel[4096:], None, e1, e2)
    assert np.allclose(full[:4096], a, atol=1e-5) and np.allclose(full[4096:], b, atol=1e-5)


def test_checkpoint_loader_and_clip_features(tmp_path):
    G, _ = _net()
    torch.save({'model': G.state_dict(), 'iter': 1}, tmp_path / 'ckpt_autovc.pth')
    got = autovc.load_generator(str(tmp_path / 'ckpt_autovc.pth'), torch.device('cpu'))
    assert not got.training and all(torch.equal(a, b) for a, b in zip(G.state_dict().values(), got.state_dict().values()))
    np.savetxt(tmp_path / 'emb.txt', np.arange(256) / 256.0)
    assert autovc.load_target_embedding(str(tmp_path / 'emb.txt')).shape == (256,)
    with pytest.raises(FileNotFoundError):
        autovc.load_target_embedding(str(tmp_path / 'none.txt'))
    wav = os.path.join(HERE, 'golden', 'female12.wav')
    raw = audio.clip_audio_features(wav, max_frames=40)
    e = np.zeros(256, dtype=np.float32)
    conv = audio.clip_audio_features(wav, max_frames=40, converter=lambda mel: autovc.convert_mel(got, mel, None, e, e))
    assert conv.shape == raw.shape == (40, 18, 80) and conv.dtype == np.float32
    assert not np.allclose(conv, raw)
