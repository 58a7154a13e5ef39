#!/usr/bin/env python3
"""Golden for the mel stage (tests/golden/audio.npz): the reference's own ``extract_f0_func_audiofile``
(Module1/src/autovc/retrain_version/vocoder_spec/extract_f0_func.py:95-131) run in the build container on the reference's
example clip (examples/female12.wav, copied to tests/golden/female12.wav as a data fixture).

The module imports four packages this image lacks; they are stubbed at import time:
  soundfile.read          -> scipy.io.wavfile (int16 / 32768, what libsndfile returns)
  librosa.filters.mel     -> animateportrait_amd.audio.mel_filterbank  (the restated Slaney filter bank: NOT pinned by this
                             golden -- it is an input of the reference function here; stored in the fixture as `mel_basis`)
  pysptk.sptk.rapt        -> a constant track (the f0 output is not used by this build)
  pyworld, pdb            -> empty modules
Everything else -- the 1e-6 padding rule, filtfilt, the dither, pySTFT, the dB mapping -- is the reference's code.
Also stores the windows the reference's collate makes of the first 40 frames (audio2landmark_dataset.py:73-78 restated in one
line here: the class needs a pickle dump directory).
Run in the build container:  python tests/golden/make_audio_golden.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from scipy.io import wavfile
    from animateportrait_amd import audio

    def sf_read(path):
        sr, x = wavfile.read(path)
        return x.astype(np.float64) / 32768.0, sr
    sys.modules['soundfile'] = types.SimpleNamespace(read=sf_read)
    lib = types.ModuleType('librosa')
    lib.filters = types.ModuleType('librosa.filters')
    lib.filters.mel = lambda sr, n_fft, fmin, fmax, n_mels: audio.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lib.util = types.ModuleType('librosa.util')
    sys.modules.update({'librosa': lib, 'librosa.filters': lib.filters, 'librosa.util': lib.util})
    sp = types.ModuleType('pysptk')
    sp.sptk = types.SimpleNamespace(rapt=lambda x, fs, hop, min, max, otype: np.full((len(x) + hop - 1) // hop + 1, 150.0))
    sys.modules.update({'pysptk': sp, 'pysptk.sptk': sp.sptk, 'pyworld': types.ModuleType('pyworld')})
    sys.path.insert(0, '/root/reference/Module1')
    from src.autovc.retrain_version.vocoder_spec import extract_f0_func as ref
    ref.speaker_normalization = lambda f0, idx, m, s: f0           # f0 branch unused
    wav = os.path.join(HERE, 'female12.wav')
    S, _ = ref.extract_f0_func_audiofile(wav, 'F')
    stft = ref.pySTFT(np.linspace(-1, 1, 4000) ** 3).T
    np.savez_compressed(os.path.join(HERE, 'audio.npz'), S=S.astype(np.float32), S_sum=np.float64(S.sum()),
                        stft_probe=stft.astype(np.float32), mel_basis=audio.mel_filterbank(),
                        windows=np.stack([S[i:i + 18] for i in range(0, 40 - 18, 1)]).astype(np.float32))
    print('S', S.shape, float(S.min()), float(S.max()))


if __name__ == '__main__':
    main()
