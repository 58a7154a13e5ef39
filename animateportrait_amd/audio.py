"""Audio front end of the end-to-end clip (BASELINE configs[4] "MFCC/mel audio features"):
wav -> 80-bin mel spectrogram at 62.5 frames/s -> 18-frame windows for the Module1 landmark networks.

Host-side numpy / scipy, as in the reference (the mel stage runs once per clip on the CPU there too):
  * ``mel_spectrogram``  = the spectral half of ``extract_f0_func_audiofile``
    (Module1/src/autovc/retrain_version/vocoder_spec/extract_f0_func.py:14-27, 95-131): first channel, the 1e-6 sample
    appended when the length is a multiple of 256, 5th-order Butterworth high-pass at 30 Hz run forward-backward
    (``filtfilt``), x 0.95 + RandomState(0) dither of +-5e-7, reflect-padded 1024-point STFT with hop 256 and a periodic
    Hann window, 80-band mel projection (90 .. 7600 Hz), ``20 log10(max(1e-5, .)) - 16`` mapped to ``(dB + 100) / 100``;
  * ``normalize_loudness`` = ``match_target_amplitude(sound, -20.0)`` of the converter
    (Module1/src/autovc/AutoVC_mel_Convertor_retrain_version.py:222-224; pydub semantics on int16 samples);
  * ``window_frames``     = the test-time collate of ``Audio2landmark_Dataset``
    (Module1/src/dataset/audio2landmark/audio2landmark_dataset.py:73-78): windows ``au[i : i + 18]`` for
    ``i in range(0, T - 18, 1)``.
The AutoVC content converter between the mel spectrogram and the windows (``Generator(16, 256, 512, 16)``, :205-208) lives in
``autovc.py`` (round 4) and enters ``clip_audio_features`` as the ``converter`` callable.  NOT here, because the packages are
not in this image: the RAPT f0 track the converter consumes (pysptk) and the speaker embedding (resemblyzer) -- both are
arguments of ``autovc.convert_mel``.
The mel filter bank restates librosa 0.7's ``filters.mel`` (Slaney scale, Slaney area normalisation, the version the
reference pins); librosa is absent from this image, so that table is **parity unpinned** -- everything around it is pinned
to the reference's own function (tests/golden/make_audio_golden.py).
"""
import numpy as np
from scipy import signal
from scipy.io import wavfile

SAMPLE_RATE = 16000           # main_end2end_module2.py:214 resamples every input to 16 kHz
N_FFT, HOP, N_MELS = 1024, 256, 80
FMIN, FMAX = 90.0, 7600.0
WINDOW_FRAMES = 18            # num_window_frames of the test configuration (train_audio2landmark.py:37-41)


def read_wav(path):
    """``soundfile.read``: float64 samples in [-1, 1), (T,) or (T, channels), and the sample rate."""
    sr, x = wavfile.read(path)
    if x.dtype == np.int16:
        x = x.astype(np.float64) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float64) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float64) - 128.0) / 128.0
    else:
        x = x.astype(np.float64)
    return x, sr


def resample_to_16k(x, sr):
    """The ``ffmpeg -ar 16000`` step (main_end2end_module2.py:214) for inputs at another rate (polyphase FIR)."""
    if sr == SAMPLE_RATE:
        return x
    g = np.gcd(int(sr), SAMPLE_RATE)
    return signal.resample_poly(x, SAMPLE_RATE // g, int(sr) // g, axis=0)


def normalize_loudness(x, target_dbfs=-20.0):
    """pydub ``sound.apply_gain(target_dBFS - sound.dBFS)`` on 16-bit samples: dBFS = 20 log10(rms / 32768) with the rms
    over ALL samples of all channels (audioop.rms), gain applied in the integer domain as audioop.mul does (clamp, then floor)."""
    q = np.clip(np.round(np.asarray(x, dtype=np.float64) * 32768.0), -32768, 32767)
    rms = int(np.sqrt(np.mean(q.astype(np.float64) ** 2)))              # audioop.rms truncates to an integer
    if rms == 0:
        return q / 32768.0
    gain = 10.0 ** ((target_dbfs - 20.0 * np.log10(rms / 32768.0)) / 20.0)
    # audioop.mul (what AudioSegment.apply_gain runs): fbound() = clamp to [minval, maxval] -- anything below minval + 1 becomes
    # minval -- THEN floor toward -inf; not round-to-nearest
    v = q * gain
    v = np.where(v > 32767.0, 32767.0, np.where(v < -32767.0, -32768.0, v))
    return np.floor(v) / 32768.0


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, min_log_hz) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """(n_mels, 1 + n_fft / 2) triangular filters on the Slaney mel scale, each scaled by 2 / (its band width in Hz)."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def stft_magnitude(x, n_fft=N_FFT, hop=HOP):
    """|rfft| of Hann-windowed frames of the reflect-padded signal: (frames, 1 + n_fft / 2)  (pySTFT(...).T, :14-27)."""
    x = np.pad(np.asarray(x, dtype=np.float64), n_fft // 2, mode='reflect')
    n = (x.shape[0] - (n_fft - hop)) // hop
    frames = np.lib.stride_tricks.sliding_window_view(x, n_fft)[::hop][:n]
    win = signal.get_window('hann', n_fft, fftbins=True)
    return np.abs(np.fft.rfft(frames * win, n=n_fft, axis=1))


def mel_spectrogram(x, mel_basis=None):
    """x: samples at 16 kHz, (T,) or (T, channels) -> S (frames, 80) float64 in about [0, 1]."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim >= 2:
        x = x[:, 0]
    if x.shape[0] % 256 == 0:
        x = np.concatenate((x, np.array([1e-06])), axis=0)
    b, a = signal.butter(5, 30.0 / (0.5 * SAMPLE_RATE), btype='high', analog=False)
    y = signal.filtfilt(b, a, x)
    wav = y * 0.95 + (np.random.RandomState(0).rand(y.shape[0]) - 0.5) * 1e-06
    basis = mel_filterbank() if mel_basis is None else mel_basis
    d_mel = np.dot(stft_magnitude(wav), basis.T)
    min_level = np.exp(-100 / 20 * np.log(10))
    d_db = 20 * np.log10(np.maximum(min_level, d_mel)) - 16
    return (d_db + 100) / 100


def window_frames(au, num_window_frames=WINDOW_FRAMES, step=1):
    """(T, 80) -> (max(T - 18, 0) / step, 18, 80): the windows the landmark networks see, one per output frame."""
    au = np.asarray(au)
    n = au.shape[0] - num_window_frames
    if n <= 0:
        return np.zeros((0, num_window_frames) + au.shape[1:], dtype=au.dtype)
    return np.stack([au[i:i + num_window_frames] for i in range(0, n, step)])


def clip_audio_features(path, max_frames=None, normalize=True, converter=None):
    """wav file -> float32 windows (F, 18, 80) for ``module1.predict_landmarks*``; F output frames at 62.5 fps.
    converter: the AutoVC stage between the mel spectrogram and the windows (main_end2end_module2.py:218-224), a callable
    (T, 80) mel -> (T, 80) mel -- ``lambda mel: autovc.convert_mel(G, mel, f0, emb_src, emb_trg)``.  None hands the RAW mel to
    the windows, which is NOT what checkpoints trained by the reference expect (they saw converted spectrograms)."""
    x, sr = read_wav(path)
    x = resample_to_16k(x, sr)
    if normalize:
        x = normalize_loudness(x)
    s = mel_spectrogram(x).astype(np.float32)
    if converter is not None:
        s = np.asarray(converter(s), dtype=np.float32)
    w = window_frames(s)
    return w if max_frames is None else w[:max_frames]
