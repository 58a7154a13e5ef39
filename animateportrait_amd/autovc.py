"""The AutoVC content converter between the mel spectrogram and the landmark networks' windows:
``Generator(16, 256, 512, 16)`` of Module1/src/autovc/retrain_version/model_vc_37_1.py:165-205 (encoder :48-88, decoder
:92-112, postnet :117-160) as stock PyTorch-ROCm modules with the reference's ``state_dict`` keys, and the driver loop of
``AutoVC_mel_Convertor.convert_single_wav_to_autovc_input`` (AutoVC_mel_Convertor_retrain_version.py:199-276, "long split
version": 4096-frame pieces, each padded to a multiple of 32 frames, source speaker -> target speaker with the source f0).

The reference feeds Module1 the CONVERTED spectrogram ``x_identic_psnt`` (main_end2end_module2.py:218-224); the Module1
checkpoints were trained on it.  What is not in this image and enters as arguments: the RAPT f0 track (pysptk; ``f0_norm`` of
extract_f0_func_audiofile, -1e10 where unvoiced), the resemblyzer speaker embedding (256-d), and the target-speaker embedding
(the reference reads ``src/autovc/retrain_version/obama_emb.txt``: a data file of the user's checkout, ``load_target_embedding``).
Small LSTM / conv1d network (~0.1 GFLOP per frame): library kernels through torch, except the LSTM recurrences on the device,
whose time loops run inside one kernel each (lstm_hip.py, csrc/lstm.hip).
Pinned to the reference class and functions by tests/golden/make_autovc_golden.py.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

DIM_ENC = DIM_DEC = 512
DIM_FREQ, DIM_F0, GROUPS = 80, 257, 32
PIECE, PAD_BASE = 4096, 32                     # AutoVC_mel_Convertor_retrain_version.py:250, :201-205
TARGET_EMB_PATHS = ('src/autovc/retrain_version/obama_emb.txt', 'Module1/src/autovc/retrain_version/obama_emb.txt')   # :211-214


def _lstm(lstm, x):
    """``lstm(x)[0]``; on the device (inference) with the time loop in one kernel per layer and direction (lstm_hip.py)."""
    if x.is_cuda:
        from . import lstm_hip
        return lstm_hip.lstm_forward(lstm, x.contiguous())
    return lstm(x)[0]


class _Linear(nn.Module):
    """key: linear_layer.{weight, bias}"""

    def __init__(self, cin, cout):
        super().__init__()
        self.linear_layer = nn.Linear(cin, cout)

    def forward(self, x):
        return self.linear_layer(x)


class _Conv(nn.Module):
    """key: conv.{weight, bias}; 5-tap 'same' conv1d"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, kernel_size=5, stride=1, padding=2)

    def forward(self, x):
        return self.conv(x)


def _conv_gn(cin, cout, groups):
    return nn.Sequential(_Conv(cin, cout), nn.GroupNorm(groups, cout))


class _Encoder(nn.Module):
    def __init__(self, dim_neck, dim_emb, freq):
        super().__init__()
        self.dim_neck, self.freq = dim_neck, freq
        self.convolutions = nn.ModuleList([_conv_gn(DIM_FREQ + dim_emb if i == 0 else DIM_ENC, DIM_ENC, GROUPS) for i in range(3)])
        self.lstm = nn.LSTM(DIM_ENC, dim_neck, 2, batch_first=True, bidirectional=True)

    def forward(self, x):
        for conv in self.convolutions:
            x = F.relu(conv(x))
        out = _lstm(self.lstm, x.transpose(1, 2))
        fwd, bwd = out[:, :, :self.dim_neck], out[:, :, self.dim_neck:]
        # one code per ``freq`` frames: the forward state at the END of the segment, the backward state at its START
        return [torch.cat((fwd[:, i + self.freq - 1], bwd[:, i]), -1) for i in range(0, out.size(1), self.freq)]


class _Decoder(nn.Module):
    def __init__(self, dim_neck, dim_emb):
        super().__init__()
        self.lstm = nn.LSTM(dim_neck * 2 + dim_emb + DIM_F0, DIM_DEC, 3, batch_first=True)
        self.linear_projection = _Linear(DIM_DEC, DIM_FREQ)

    def forward(self, x):
        return self.linear_projection(_lstm(self.lstm, x))


class _Postnet(nn.Module):
    def __init__(self):
        super().__init__()
        chans = [DIM_FREQ, 512, 512, 512, 512, DIM_FREQ]
        self.convolutions = nn.ModuleList([_conv_gn(chans[i], chans[i + 1], GROUPS if i < 4 else 5) for i in range(5)])

    def forward(self, x):
        for conv in self.convolutions[:-1]:
            x = torch.tanh(conv(x))
        return self.convolutions[-1](x)


class Generator(nn.Module):
    """forward(x (B, T, 80), c_org (B, E), f0_org, c_trg (B, E), f0_trg (B, T, 257)) -> (mel (B, T, 80), mel + postnet, codes);
    T must be a multiple of ``freq``.  ``dim_pre`` is accepted and unused, as in the reference."""

    def __init__(self, dim_neck=16, dim_emb=256, dim_pre=512, freq=16):
        super().__init__()
        self.encoder = _Encoder(dim_neck, dim_emb, freq)
        self.decoder = _Decoder(dim_neck, dim_emb)
        self.postnet = _Postnet()
        self.freq = freq

    def forward(self, x, c_org, f0_org=None, c_trg=None, f0_trg=None, enc_on=False):
        t = x.size(1)
        codes = self.encoder(torch.cat((x.transpose(2, 1), c_org.unsqueeze(-1).expand(-1, -1, t)), 1))
        if enc_on:
            return torch.cat(codes, -1)
        code_exp = torch.cat([c.unsqueeze(1).expand(-1, self.freq, -1) for c in codes], 1)
        mel = self.decoder(torch.cat((code_exp, c_trg.unsqueeze(1).expand(-1, t, -1), f0_trg), -1))
        return mel, mel + self.postnet(mel.transpose(2, 1)).transpose(2, 1), torch.cat(codes, -1)


def quantize_f0_interp(x, num_bins=256):
    """src/autovc/utils.py:132-144: normalised log-f0 in [0, 1] (negative = unvoiced) -> one-hot (T, 257); bin 0 = unvoiced."""
    x = np.asarray(x).astype(float).copy()
    if x.ndim != 1:
        raise ValueError('quantize_f0_interp: a 1-d f0 track')
    uv = x < 0
    x[uv] = 0.0
    if not ((x >= 0).all() and (x <= 1).all()):
        raise ValueError('quantize_f0_interp: voiced values must lie in [0, 1]')
    idx = np.round(x * (num_bins - 1)) + 1
    idx[uv] = 0.0
    enc = np.zeros((len(x), num_bins + 1), dtype=np.float32)
    enc[np.arange(len(x)), idx.astype(np.int32)] = 1.0
    return enc


def _pad_seq(x, base=PAD_BASE):
    pad = int(base * math.ceil(float(x.shape[0]) / base)) - x.shape[0]
    return np.pad(x, ((0, pad), (0, 0)), 'constant'), pad


def convert_mel(G, mel, f0_norm, emb_src, emb_trg, device=None):
    """AutoVC_mel_Convertor_retrain_version.py:246-274: (T, 80) mel + (T,) normalised f0 (-1e10 / negative where unvoiced; None:
    an all-unvoiced track) + the two 256-d speaker embeddings -> the converted (T, 80) mel ``x_identic_psnt``."""
    device = device or next(G.parameters()).device
    mel = np.asarray(mel)
    f0 = np.full(mel.shape[0], -1e10) if f0_norm is None else np.asarray(f0_norm)
    if f0.shape[0] != mel.shape[0]:
        raise ValueError('convert_mel: %d mel frames, %d f0 values' % (mel.shape[0], f0.shape[0]))
    f0q = quantize_f0_interp(f0)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)[np.newaxis].astype('float32')).to(device)      # noqa: E731
    e_src, e_trg = to(np.asarray(emb_src).reshape(-1)), to(np.asarray(emb_trg).reshape(-1))
    out, pad = [], 0
    with torch.no_grad():
        for i in range(0, mel.shape[0], PIECE):
            x, pad = _pad_seq(mel[i:i + PIECE].astype('float32'))
            f, _ = _pad_seq(f0q[i:i + PIECE].astype('float32'))
            out.append(G(to(x), e_src, to(f), e_trg, to(f))[1])
    y = torch.cat(out, 1)[0]
    res = (y if pad == 0 else y[:-pad]).cpu().numpy()        # only the last piece can carry padding (4096 = 128 * 32)
    if torch.device(device).type == 'cuda':
        from . import lstm_hip
        lstm_hip.check_timeouts()                            # (after the copy's synchronisation: a recurrence whose workgroups never met fails loudly)
    return res


def load_generator(path, device):
    """:206-209: ``Generator(16, 256, 512, 16).eval()`` with the checkpoint's 'model' entry, strict."""
    G = Generator(16, 256, 512, 16)
    G.load_state_dict(torch.load(path, map_location='cpu')['model'], strict=True)
    G = G.to(device).eval()
    for p in G.parameters():
        p.requires_grad_(False)
    return G


def load_target_embedding(path=None):
    """The target-speaker embedding the converter maps every voice to (:211-215), from the user's reference checkout."""
    for p in ([path] if path else TARGET_EMB_PATHS):
        if os.path.exists(p):
            return np.loadtxt(p).astype(np.float32).reshape(-1)
    raise FileNotFoundError('AutoVC target-speaker embedding not found (looked for %s)' % ', '.join([path] if path else TARGET_EMB_PATHS))
