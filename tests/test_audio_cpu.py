"""CPU: the audio front end (animateportrait_amd/audio.py) against the reference's own mel extraction run on the reference's
example clip (tests/golden/audio.npz, made by tests/golden/make_audio_golden.py from
Module1/src/autovc/retrain_version/vocoder_spec/extract_f0_func.py:95-131), plus the properties that define the mel
filter bank (the one table librosa's absence leaves unpinned)."""
import os

import numpy as np

from conftest import GOLDEN

from animateportrait_amd import audio


def _g():
    return np.load(os.path.join(GOLDEN, 'audio.npz'))


def test_mel_spectrogram_matches_reference_function_on_the_example_clip():
    g = _g()
    x, sr = audio.read_wav(os.path.join(GOLDEN, 'female12.wav'))
    assert sr == 16000 and x.shape == (209319, 2) and x.dtype == np.float64 and np.abs(x).max() <= 1.0
    S = audio.mel_spectrogram(x)
    assert S.shape == (818, 80) == g['S'].shape
    assert np.abs(S - g['S']).max() < 2e-6
    assert abs(S.sum() - float(g['S_sum'])) < 1e-6 * abs(float(g['S_sum']))
    # floor of the dB mapping: (20 log10(1e-5) - 16 + 100) / 100
    assert abs(S.min() - (-0.16)) < 1e-9 or S.min() > -0.16


def test_stft_magnitude_matches_reference_pystft():
    g = _g()
    m = audio.stft_magnitude(np.linspace(-1, 1, 4000) ** 3)
    assert m.shape == g['stft_probe'].shape == (16, 513)
    assert np.abs(m - g['stft_probe']).max() < 1e-5 * np.abs(g['stft_probe']).max()


def test_padding_rule_for_multiples_of_the_hop():
    """A length that is a multiple of 256 gets one extra 1e-6 sample (:116-117): one more frame than floor(T / 256)."""
    x = np.random.RandomState(1).randn(256 * 40) * 0.1
    assert audio.mel_spectrogram(x).shape[0] == 41 and audio.mel_spectrogram(x[:-1]).shape[0] == 40


def test_windows_are_the_collate_of_the_reference():
    g = _g()
    w = audio.window_frames(g['S'][:40])
    assert w.shape == (22, 18, 80) and np.array_equal(w, g['windows'])
    assert audio.window_frames(g['S'][:18]).shape == (0, 18, 80)          # range(0, T - 18): nothing for T <= 18
    full = audio.clip_audio_features(os.path.join(GOLDEN, 'female12.wav'), normalize=False)
    assert full.shape == (800, 18, 80) and full.dtype == np.float32 and np.array_equal(full[:22], g['windows'])


def test_mel_filterbank_properties():
    """Slaney scale anchors (linear 200/3 Hz per mel below 1 kHz, 27 log-spaced steps from 1 kHz to 6.4 kHz), triangles
    that peak between their neighbours' peaks, Slaney area normalisation (every filter integrates to 1 over frequency)."""
    assert abs(float(audio._hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(audio._hz_to_mel(6400.0)) - 42.0) < 1e-9
    assert abs(float(audio._mel_to_hz(audio._hz_to_mel(3333.0))) - 3333.0) < 1e-9
    w = audio.mel_filterbank()
    assert w.shape == (80, 513) and w.dtype == np.float32 and (w >= 0).all()
    assert np.array_equal(w, _g()['mel_basis'])
    f = np.linspace(0, 8000, 513)
    peaks = f[w.argmax(1)]
    assert (np.diff(peaks) > 0).all() and peaks[0] > 90 and peaks[-1] < 7600
    assert w[:, f < 90].sum() == 0 and w[:, f > 7600].sum() == 0
    area = w.sum(1) * (f[1] - f[0])
    assert np.abs(area[20:] - 1.0).max() < 0.05           # (narrow low bands are under-sampled by the 15.6 Hz bins)


def test_loudness_normalisation_hits_the_target():
    x = np.random.RandomState(2).randn(16000, 2) * 0.01
    y = audio.normalize_loudness(x, -20.0)
    q = np.round(y * 32768)
    db = 20 * np.log10(np.sqrt(np.mean(q ** 2)) / 32768)
    assert abs(db - (-20.0)) < 0.05 and np.abs(y).max() <= 1.0
    assert audio.resample_to_16k(np.zeros(22050), 22050).shape == (16000,)


def test_loudness_normalisation_matches_audioop_sample_for_sample():
    """pydub's ``match_target_amplitude`` (AutoVC_mel_Convertor_retrain_version.py:13-15, :222-224) is ``audioop.rms`` +
    ``audioop.mul`` on the 16-bit samples; audioop is in this interpreter's standard library, so the restatement is pinned
    bit for bit: truncated integer rms, gain 10^(dB/20), clamp then FLOOR (not round-to-nearest)."""
    import audioop
    import math
    rs = np.random.RandomState(5)
    for scale, ch in ((0.01, 1), (0.3, 2), (0.9, 1)):            # quiet / loud (the gain is < 1) / near full scale
        q = np.clip(np.round(rs.randn(4000, ch).squeeze() * scale * 32768), -32768, 32767).astype(np.int16)
        data = q.tobytes()
        rms = audioop.rms(data, 2)
        gain = 10 ** ((-20.0 - 20 * math.log(rms / 32768.0, 10)) / 20)                 # pydub: ratio_to_db / db_to_float
        want = np.frombuffer(audioop.mul(data, 2, gain), dtype=np.int16).reshape(q.shape)
        got = audio.normalize_loudness(q.astype(np.float64) / 32768.0, -20.0) * 32768.0
        assert np.array_equal(got, want.astype(np.float64)), (scale, ch, np.abs(got - want).max())
