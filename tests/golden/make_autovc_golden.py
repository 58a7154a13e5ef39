#!/usr/bin/env python3
"""Golden for the AutoVC converter mirror (tests/golden/autovc.npz), from the REFERENCE's own ``Generator``
(Module1/src/autovc/retrain_version/model_vc_37_1.py:165-205) and ``quantize_f0_interp`` (Module1/src/autovc/utils.py:132-144)
run in the build container: state_dict key list and the three outputs for seeded weights / inputs.

    python tests/golden/make_autovc_golden.py

Import shims: librosa / pysptk / pyworld (imported at the top of src/autovc/utils.py, unused by quantize_f0_interp) are empty
modules.  The checkpoint is not in the reference tree: both sides load ``seeded_state_scaled``.  The driver loop of
``convert_single_wav_to_autovc_input`` needs pydub / soundfile / pysptk / resemblyzer and cannot run here; the golden holds what
that loop computes from the reference pieces (pad to 32, G(x, e_src, f0, e_trg, f0)[1], strip the pad) for a 75-frame clip."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
SEED = 401


def make_inputs():
    g = torch.Generator().manual_seed(400)
    mel = torch.rand(75, 80, generator=g).numpy()
    f0 = torch.rand(75, generator=g).numpy().astype(np.float64)
    f0[10:25] = -1e10                                     # an unvoiced stretch
    f0[40] = 1.0
    f0[41] = 0.0
    e_src = (torch.rand(256, generator=g) * 0.2).numpy()
    e_trg = (torch.rand(256, generator=g) * 0.2).numpy()
    return mel, f0, e_src, e_trg


def main():
    import warnings
    warnings.filterwarnings('ignore')
    for name in ('librosa', 'pysptk', 'pyworld'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, '/root/reference/Module1')
    from src.autovc.retrain_version.model_vc_37_1 import Generator
    from src.autovc.utils import quantize_f0_interp
    from make_golden import save
    from make_auxnets_golden import seeded_state_scaled, keys_of
    torch.set_num_threads(8)
    G = Generator(16, 256, 512, 16).eval()
    ks = keys_of(G)
    G.load_state_dict(seeded_state_scaled(ks, SEED), strict=True)
    mel, f0, e_src, e_trg = make_inputs()
    f0q = quantize_f0_interp(f0)
    pad = 96 - 75
    x = np.pad(mel, ((0, pad), (0, 0)), 'constant').astype('float32')
    f = np.pad(f0q, ((0, pad), (0, 0)), 'constant').astype('float32')
    to = lambda a: torch.from_numpy(a[np.newaxis].astype('float32'))       # noqa: E731
    with torch.no_grad():
        m, mp, codes = G(to(x), to(e_src), to(f), to(e_trg), to(f))
        enc = G(to(x), to(e_src), enc_on=True)
    assert torch.equal(enc, codes)
    save('autovc.npz', keys=np.array([k for k, _, _ in ks]), shapes=np.array([str(s) for _, s, _ in ks]),
         f0q_index=f0q.argmax(1).astype(np.int16), f0q_rowsum=f0q.sum(1), mel_out=m, mel_postnet=mp, codes=codes,
         converted=mp[0, :-pad])
    print(m.abs().max(), mp.abs().max(), codes.shape)


if __name__ == '__main__':
    main()
