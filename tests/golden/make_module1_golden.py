#!/usr/bin/env python3
"""Golden for the Module1 mirrors (tests/golden/module1.npz): state_dict keys / shapes of the reference's
``Audio2landmark_content`` and ``Audio2landmark_pos`` (Module1/src/models/model_audio2landmark.py:28-90, 296-386), their
outputs for seeded weights and inputs, and the clip pipeline of ``Audio2landmark_model.test``
(Module1/src/approaches/train_audio2landmark.py:101-141, 235-245, 594-617) run through the reference's own methods.
Shims: an empty ``cv2`` module and ``nn.Module.cuda`` as the identity (no GPU in the build container).
Run in the build container:  python tests/golden/make_module1_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def seeded_state(keys_shapes, seed=77):
    """weights both sides load: N(0, 0.05) per tensor in key order; BatchNorm running_var = 1 + |.|, counters 0"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape, dtype in keys_shapes:
        if 'num_batches_tracked' in k:
            sd[k] = torch.zeros(shape, dtype=torch.int64)
        else:
            t = torch.randn(shape, generator=g) * 0.05
            sd[k] = (1.0 + t.abs()) if k.endswith('running_var') else t
    return sd


def main():
    import types
    import warnings
    warnings.filterwarnings('ignore')
    sys.path.insert(0, '/root/reference/Module1')
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    torch.nn.Module.cuda = lambda self, *a, **k: self          # the reference's DecoderLayer calls .cuda() (:245); no GPU here
    from src.models.model_audio2landmark import Audio2landmark_content, Audio2landmark_pos
    import src.approaches.train_audio2landmark as approach
    from util.utils import add_naive_eye
    from make_golden import save
    out = {}

    def describe(net, tag):
        ks = [(k, tuple(v.shape), str(v.dtype)) for k, v in net.state_dict().items()]
        out[tag + 'keys'] = np.array([k for k, _, _ in ks])
        out[tag + 'shapes'] = np.array([str(s) for _, s, _ in ks])
        out[tag + 'dtypes'] = np.array([d for _, _, d in ks])
        return ks

    # ---- content net, the configuration of round 2's first golden (use_prior_net=False) ...
    net = Audio2landmark_content(num_window_frames=18, in_size=80, use_prior_net=False, hidden_size=256, num_layers=3,
                                 drop_out=0, bidirectional=False)
    ks = describe(net, '')
    net.load_state_dict(seeded_state(ks), strict=True)
    net.eval()
    g = torch.Generator().manual_seed(3)
    au = torch.randn(6, 18, 80, generator=g)
    fid = torch.randn(1, 204, generator=g) * 0.1
    with torch.no_grad():
        out.update(au=au, fid=fid, out=net(au, fid)[0])
    # ---- ... and the one Audio2landmark_model builds (train_audio2landmark.py:71-73: use_prior_net=True, drop_out=0.5)
    netc = Audio2landmark_content(num_window_frames=18, in_size=80, use_prior_net=True, bidirectional=False, drop_out=0.5)
    ksc = describe(netc, 'c_')
    netc.load_state_dict(seeded_state(ksc, seed=78), strict=True)
    netc.eval()
    with torch.no_grad():
        out['c_out'] = netc(au, fid)[0]
    # ---- speaker-aware pose net (:55-59)
    netg = Audio2landmark_pos(drop_out=0.5, spk_emb_enc_size=128, c_enc_hidden_size=256, transformer_d_model=32, N=2,
                              heads=2, z_size=128, audio_dim=256)
    ksg = describe(netg, 'g_')
    sdg = seeded_state(ksg, seed=79)
    sdg = {k: (netg.state_dict()[k] if k.endswith('pe.pe') else v) for k, v in sdg.items()}   # the constant table stays
    netg.load_state_dict(sdg, strict=True)
    netg.eval()
    emb6 = torch.randn(6, 256, generator=g)
    with torch.no_grad():
        out.update(g_emb=emb6, g_out=netg(au, emb6, fid.repeat(6, 1), fid.repeat(6, 1), torch.zeros(6, 128))[0])
    # ---- the clip pipeline of Audio2landmark_model.test on T = 600 windows (two segments): the reference's own
    # methods, called on a namespace that carries what they read; the loop around them is __train_pass__ :277-309
    cls = approach.Audio2landmark_model
    approach.device = torch.device('cpu')
    me = types.SimpleNamespace(G=netg, C=netc, std_face_id=fid.clone(),
                               opt_parser=types.SimpleNamespace(amp_pos=0.5, amp_lip_x=2.0, amp_lip_y=2.0))
    for name in ('__calib_baseline_pred_fls__', '__solve_inverse_lip2__', '__train_face_and_pos__'):
        setattr(me, name, types.MethodType(getattr(cls, name), me))
    T = 600
    gp = torch.Generator().manual_seed(17)
    au_t = torch.randn(T, 18, 80, generator=gp)
    spk = torch.randn(256, generator=gp)
    fl_in = torch.zeros(T, 18, 204)
    emb_t = spk.view(1, -1).repeat(T, 1)
    segs = []
    with torch.no_grad():
        for j in range(0, T, 512):
            pred, face = me.__train_face_and_pos__(fl_in[j:j + 512], au_t[j:j + 512], emb_t[j:j + 512], me.std_face_id)
            segs.append(me.__solve_inverse_lip2__((pred + face).data.cpu().numpy()))
    fake = np.concatenate(segs)
    fake[:, 27 * 3:28 * 3] = fake[:, 28 * 3:29 * 3] * 2 - fake[:, 29 * 3:30 * 3]
    from scipy.signal import savgol_filter
    fake = savgol_filter(fake, 5, 3, axis=0)
    out.update(p_seed=np.int64(17), p_T=np.int64(T), p_sub=fake[::16].astype(np.float32), p_sum=np.float64(fake.sum()),
               p_abs=np.float64(np.abs(fake).sum()))
    # ---- the inverted-lip fix on frames that need it, and add_naive_eye with numpy's global generator seeded
    rng = np.random.RandomState(4)
    lips = rng.randn(12, 204).astype(np.float64)
    out.update(lip_in=lips, lip_out=me.__solve_inverse_lip2__(lips.copy()))
    fl3 = rng.randn(260, 68, 1)
    np.random.seed(5)
    out.update(eye_in=fl3.astype(np.float32), eye_out=add_naive_eye(fl3.astype(np.float32).copy()))
    save('module1.npz', **out)


if __name__ == '__main__':
    main()
