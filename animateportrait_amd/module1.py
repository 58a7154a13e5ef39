"""Module1's audio -> landmark networks and clip-level landmark pipeline.

Mirrors of the reference's two networks, attribute for attribute, so that its checkpoints load with ``strict=True``:

* ``Audio2LandmarkContent`` = ``Audio2landmark_content`` (Module1/src/models/model_audio2landmark.py:28-90; checkpoint
  ``ckpt_content_branch.pth['model_g_face_id']``, train_audio2landmark.py:71-79): a 3-layer LSTM over 18-frame mel
  windows (behind the ``fc_prior`` MLP when ``use_prior_net``) + a 3-layer MLP, ~1.5 M parameters;
* ``Audio2LandmarkPos`` = ``Audio2landmark_pos`` (:296-386; checkpoint ``ckpt_speaker_branch.pth['G']``,
  train_audio2landmark.py:55-66): LSTM audio encoder, speaker-embedding MLP, a 2-layer / 2-head self-attention encoder
  (``Embedder`` / ``PositionalEncoder`` / ``EncoderLayer`` / ``Norm``, :94-262) over the windows of a segment, output
  MLP.  The reference also constructs a ``Decoder`` it never runs; it is built here too because its tensors are in the
  checkpoint.

Both run ONCE per clip over all windows: stock PyTorch-ROCm, as the hot-path scope prescribes for Module1 (no hand
kernels).  ``predict_landmarks_speaker_aware`` is ``Audio2landmark_model.test`` (train_audio2landmark.py:101-141,
235-245, 247-309, 594-617) followed by the clip-level post-processing of main_end2end_module2.py:262-272; the simpler
``predict_landmarks`` is the content branch alone (``__train_face_wo_pos__``).  NOT built: the AutoVC mel / speaker
embedding front end (librosa / pysptk / pyworld / resemblyzer are not in this image, and neither are the checkpoints);
callers pass the (T, 18, 80) mel windows and the 256-d speaker embedding the reference's ``au_data`` / ``au_emb`` hold.
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .stream import smooth_landmarks

FACE_ID_FEAT_SIZE = 204       # 68 x 3, model_audio2landmark.py:23


def _mlp3(cin, h1, h2, cout, slope):
    return nn.Sequential(nn.Linear(cin, h1), nn.LeakyReLU(slope), nn.Linear(h1, h2), nn.LeakyReLU(slope), nn.Linear(h2, cout))


class Audio2LandmarkContent(nn.Module):
    def __init__(self, num_window_frames=18, in_size=80, lstm_size=161, use_prior_net=False, hidden_size=256, num_layers=3,
                 drop_out=0.0):
        super().__init__()
        # (the reference assigns fc_prior and fc to the same Sequential first and then replaces fc, :33-38 / :62-70:
        # both exist in its state_dict, registered in the order fc_prior, fc, bilstm)
        self.fc_prior = nn.Sequential(nn.Linear(in_size, 256), nn.BatchNorm1d(256), nn.LeakyReLU(0.2),
                                      nn.Linear(256, lstm_size))
        self.fc = self.fc_prior
        self.use_prior_net = use_prior_net
        self.bilstm = nn.LSTM(input_size=lstm_size if use_prior_net else in_size, hidden_size=hidden_size,
                              num_layers=num_layers, dropout=drop_out, bidirectional=False, batch_first=True)   # :41-55
        self.fc = nn.Sequential(nn.Linear(hidden_size + FACE_ID_FEAT_SIZE, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2),
                                nn.Linear(512, 256), nn.BatchNorm1d(256), nn.LeakyReLU(0.2), nn.Linear(256, 204))
        self.in_size, self.lstm_size, self.num_window_frames = in_size, lstm_size, num_window_frames

    def forward(self, au, face_id):                                              # :74-88
        x = au
        if self.use_prior_net:
            x = self.fc_prior(x.contiguous().view(-1, self.in_size)).view(-1, self.num_window_frames, self.lstm_size)
        output, _ = self.bilstm(x)
        output = output[:, -1, :]
        if face_id.shape[0] == 1:
            face_id = face_id.repeat(output.shape[0], 1)
        return self.fc(torch.cat((output, face_id), dim=1)), face_id


# ---------------------------------------------------------------------------------------------- self-attention pieces
class Embedder(nn.Module):                                                       # :94-99
    def __init__(self, feat_size, d_model):
        super().__init__()
        self.embed = nn.Linear(feat_size, d_model)

    def forward(self, x):
        return self.embed(x)


class PositionalEncoder(nn.Module):                                              # :102-127 (its own sin / cos exponents)
    def __init__(self, d_model, max_seq_len=512):
        super().__init__()
        self.d_model = d_model
        pos = torch.arange(max_seq_len, dtype=torch.float64).unsqueeze(1)
        i = torch.arange(0, d_model, 2, dtype=torch.float64).unsqueeze(0)
        pe = torch.zeros(max_seq_len, d_model, dtype=torch.float64)
        pe[:, 0::2] = torch.sin(pos / (10000 ** ((2 * i) / d_model)))
        pe[:, 1::2] = torch.cos(pos / (10000 ** ((2 * (i + 1)) / d_model)))
        self.register_buffer('pe', pe.float().unsqueeze(0))

    def forward(self, x):
        return x * math.sqrt(self.d_model) + self.pe[:, :x.size(1)]


class MultiHeadAttention(nn.Module):                                             # :130-183
    def __init__(self, heads, d_model, dropout=0.1):
        super().__init__()
        self.d_model, self.d_k, self.h = d_model, d_model // heads, heads
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.out = nn.Linear(d_model, d_model)

    def forward(self, q, k, v, mask=None):
        bs = q.size(0)
        k = self.k_linear(k).view(bs, -1, self.h, self.d_k).transpose(1, 2)
        q = self.q_linear(q).view(bs, -1, self.h, self.d_k).transpose(1, 2)
        v = self.v_linear(v).view(bs, -1, self.h, self.d_k).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(self.d_k)
        if mask is not None:
            scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
        scores = self.dropout(F.softmax(scores, dim=-1))
        return self.out(torch.matmul(scores, v).transpose(1, 2).contiguous().view(bs, -1, self.d_model))


class FeedForward(nn.Module):                                                    # :185-195
    def __init__(self, d_model, d_ff=2048, dropout=0.1):
        super().__init__()
        self.linear_1 = nn.Linear(d_model, d_ff)
        self.dropout = nn.Dropout(dropout)
        self.linear_2 = nn.Linear(d_ff, d_model)

    def forward(self, x):
        return self.linear_2(self.dropout(F.relu(self.linear_1(x))))


class Norm(nn.Module):                                                           # :198-211 (std unbiased, eps on the std)
    def __init__(self, d_model, eps=1e-6):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(d_model))
        self.bias = nn.Parameter(torch.zeros(d_model))
        self.eps = eps

    def forward(self, x):
        return self.alpha * (x - x.mean(dim=-1, keepdim=True)) / (x.std(dim=-1, keepdim=True) + self.eps) + self.bias


class EncoderLayer(nn.Module):                                                   # :214-229
    def __init__(self, d_model, heads, dropout=0.1):
        super().__init__()
        self.norm_1, self.norm_2 = Norm(d_model), Norm(d_model)
        self.attn = MultiHeadAttention(heads, d_model)
        self.ff = FeedForward(d_model)
        self.dropout_1, self.dropout_2 = nn.Dropout(dropout), nn.Dropout(dropout)

    def forward(self, x, mask):
        x2 = self.norm_1(x)
        x = x + self.dropout_1(self.attn(x2, x2, x2, mask))
        return x + self.dropout_2(self.ff(self.norm_2(x)))


class DecoderLayer(nn.Module):                                                   # :233-255 (constructed, never run)
    def __init__(self, d_model, heads, dropout=0.1):
        super().__init__()
        self.norm_1, self.norm_2, self.norm_3 = Norm(d_model), Norm(d_model), Norm(d_model)
        self.dropout_1, self.dropout_2, self.dropout_3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.attn_1 = MultiHeadAttention(heads, d_model)
        self.attn_2 = MultiHeadAttention(heads, d_model)
        self.ff = FeedForward(d_model)

    def forward(self, x, e_outputs, src_mask, trg_mask):
        x2 = self.norm_1(x)
        x = x + self.dropout_1(self.attn_1(x2, x2, x2, trg_mask))
        x = x + self.dropout_2(self.attn_2(self.norm_2(x), e_outputs, e_outputs, src_mask))
        return x + self.dropout_3(self.ff(self.norm_3(x)))


class _Stack(nn.Module):                                                         # Encoder / Decoder, :262-293
    def __init__(self, layer, d_model, n, in_size):
        super().__init__()
        self.N = n
        self.embed = Embedder(in_size, d_model)
        self.pe = PositionalEncoder(d_model)
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
        self.norm = Norm(d_model)


class Encoder(_Stack):
    def __init__(self, d_model, n, heads, in_size):
        super().__init__(EncoderLayer(d_model, heads), d_model, n, in_size)

    def forward(self, x, mask=None):
        x = self.pe(self.embed(x))
        for layer in self.layers:
            x = layer(x, mask)
        return self.norm(x)


class Decoder(_Stack):
    def __init__(self, d_model, n, heads, in_size):
        super().__init__(DecoderLayer(d_model, heads), d_model, n, in_size)

    def forward(self, x, e_outputs, src_mask=None, trg_mask=None):
        x = self.pe(self.embed(x))
        for layer in self.layers:
            x = layer(x, e_outputs, src_mask, trg_mask)
        return self.norm(x)


class Audio2LandmarkPos(nn.Module):
    def __init__(self, audio_feat_size=80, c_enc_hidden_size=256, num_layers=3, drop_out=0.0, spk_feat_size=256,
                 spk_emb_enc_size=128, transformer_d_model=32, N=2, heads=2, z_size=128, audio_dim=256):
        super().__init__()
        self.audio_content_encoder = nn.LSTM(input_size=audio_feat_size, hidden_size=c_enc_hidden_size,
                                             num_layers=num_layers, dropout=drop_out, bidirectional=False, batch_first=True)
        self.use_audio_projection = audio_dim != c_enc_hidden_size
        if self.use_audio_projection:
            self.audio_projection = _mlp3(c_enc_hidden_size, 256, 128, audio_dim, 0.02)
        self.spk_emb_encoder = _mlp3(spk_feat_size, 256, 128, spk_emb_enc_size, 0.02)
        d_model = transformer_d_model * heads
        self.encoder = Encoder(d_model, N, heads, in_size=audio_dim + spk_emb_enc_size + z_size)
        self.decoder = Decoder(d_model, N, heads, in_size=204)
        self.out = _mlp3(d_model + z_size, 512, 256, 204, 0.02)

    def forward(self, au, emb, face_id, fls=None, z=None):                       # :354-386 (add_z_spk=False)
        audio_encode = self.audio_content_encoder(au)[0][:, -1, :]
        if self.use_audio_projection:
            audio_encode = self.audio_projection(audio_encode)
        spk_encode = self.spk_emb_encoder(emb)
        src_feat = torch.cat((audio_encode, spk_encode, z), dim=1).unsqueeze(0)  # the segment's windows = ONE sequence
        e_outputs = torch.cat((self.encoder(src_feat)[0], z), dim=1)
        return self.out(e_outputs), face_id[0:1, :], spk_encode


# ------------------------------------------------------------------------------------------------- clip-level pipeline
def _savgol(x, window, order=3):
    from scipy.signal import savgol_filter
    return savgol_filter(x, window, order, axis=0)


def close_pose_branch_mouth(fl, ratio=0.99):
    """train_audio2landmark.py:119-130 on (T, 204): pull the upper / lower lip contours of the pose branch together."""
    fl = fl.reshape((-1, 68, 3))
    index1, index2 = list(range(59, 54, -1)), list(range(67, 64, -1))
    mean_out = 0.5 * fl[:, 49:54] + 0.5 * fl[:, index1]
    fl[:, 49:54] = mean_out * ratio + fl[:, 49:54] * (1 - ratio)
    fl[:, index1] = mean_out * ratio + fl[:, index1] * (1 - ratio)
    mean_in = 0.5 * (fl[:, 61:64] + fl[:, index2])
    fl[:, 61:64] = mean_in * ratio + fl[:, 61:64] * (1 - ratio)
    fl[:, index2] = mean_in * ratio + fl[:, index2] * (1 - ratio)
    return fl.reshape(-1, 204)


def calib_baseline(pred, amp_lip_x=2.0, amp_lip_y=2.0, ratio=0.5):
    """__calib_baseline_pred_fls__ (:235-245): per coordinate, subtract the mean of its K smallest values over the
    segment, then amplify the mouth's x / y."""
    x = np.array(pred, dtype=np.float32, copy=True)
    k = int(x.shape[0] * ratio)
    for c in range(204):
        idx = np.argpartition(x[:, c], k)
        x[:, c] = x[:, c] - np.mean(x[idx[:k], c])
    x[:, 48 * 3::3] *= amp_lip_x
    x[:, 48 * 3 + 1::3] *= amp_lip_y
    return x


def _signed_area(pts):
    """util/geo_math.py:27-39: fan of signed triangles from the first vertex"""
    ab, ac = pts[1:-1] - pts[0], pts[2:] - pts[0]
    return float(0.5 * np.sum(ab[:, 0] * ac[:, 1] - ab[:, 1] * ac[:, 0]))


def solve_inverse_lip(fl):
    """__solve_inverse_lip2__ (:594-617) on (T, 204), frame by frame (frame j reads the already fixed frame j - 1)."""
    for j in range(fl.shape[0]):
        if _signed_area(fl[j].reshape(68, 3)[60:68, 0:2]) < 0:
            for a, b in ((63, 65), (62, 66), (61, 67)):
                fl[j, b * 3:b * 3 + 3] = 0.5 * (fl[j, a * 3:a * 3 + 3] + fl[j, b * 3:b * 3 + 3])
                fl[j, a * 3:a * 3 + 3] = fl[j, b * 3:b * 3 + 3]
            p = max(j - 1, 0)
            for dst, src in (((55, 59), (64, 68)), ((59, 60), (60, 61)), ((49, 54), (60, 65))):
                d = slice(dst[0] * 3 + 1, dst[1] * 3 + 1, 3)
                s = slice(src[0] * 3 + 1, src[1] * 3 + 1, 3)
                fl[j, d] = fl[j, s] + fl[p, d] - fl[p, s]
    return fl


def add_naive_eye(fl, rng=np.random):
    """util/utils.py:361-393 on (T, 68, C): eyelids slightly closed throughout, plus blinks at random times (the
    reference draws them from numpy's global generator)."""
    pairs = ((37, 41), (38, 40), (43, 47), (44, 46))
    r = 0.95
    for a, b in pairs:
        fa, fb = fl[:, a].copy(), fl[:, b].copy()
        fl[:, a], fl[:, b] = r * fa + (1 - r) * fb, (1 - r) * fa + r * fb
    k1, k2, length = 10, 15, fl.shape[0]
    stamps, t = [30], 30
    while t < length - 1 - k2:
        t += 60
        t += rng.randint(30, 90)
        if t < length - 1 - k2:
            stamps.append(t)
    lids = [37, 38, 40, 41, 43, 44, 46, 47]
    for t in stamps:
        for a, b in pairs:
            v = 0.25 * fl[t, a] + 0.75 * fl[t, b]
            fl[t, a], fl[t, b] = v, v.copy()
        for t0 in range(t - k1 + 1, t):
            w = (t - t0) / 1. / k1
            fl[t0, lids] = w * fl[t - k1, lids] + (1 - w) * fl[t, lids]
        for t0 in range(t + 1, t + k2):
            w = (t + k2 - 1 - t0) / 1. / k2
            fl[t0, lids] = w * fl[t, lids] + (1 - w) * fl[t + k2, lids]
    return fl


@torch.no_grad()
def predict_landmarks_speaker_aware(net_g, net_c, au_windows, spk_emb, face_id, amp_pos=0.5, amp_lip_x=2.0, amp_lip_y=2.0,
                                    segment=512, smooth_win=31, close_mouth_ratio=0.99):
    """``Audio2landmark_model.test`` (train_audio2landmark.py:247-309 with __train_face_and_pos__ :101-141):
    au_windows (T, 18, 80), spk_emb (256,), face_id (204,) -> (T, 204) landmarks in Module1's normalised frame.
    Per 512-window segment: pose branch G (speaker embedding x 3, z = 0) -> Savitzky-Golay over the segment -> lips
    pulled together -> x amp_pos; content branch C on the first 18 frames -> calibrated; sum + face id; inverted-lip fix.
    Then the nose-top extrapolation and a (5, 3) Savitzky-Golay over the clip."""
    net_g.eval()
    net_c.eval()
    dev = next(net_g.parameters()).device
    au = torch.as_tensor(au_windows, dtype=torch.float32, device=dev)
    emb = torch.as_tensor(spk_emb, dtype=torch.float32, device=dev).view(1, -1).expand(au.shape[0], -1)
    fid = torch.as_tensor(face_id, dtype=torch.float32, device=dev).view(1, FACE_ID_FEAT_SIZE)
    out = []
    for j in range(0, au.shape[0], segment):
        a, e = au[j:j + segment], emb[j:j + segment]
        if a.shape[0] < 10:                                                      # :288-289
            continue
        z = torch.zeros(a.shape[0], 128, device=dev)
        dis = net_g(a, e * 3.0, fid.repeat(a.shape[0], 1), None, z)[0].cpu().numpy()
        dis = _savgol(dis, int(min(dis.shape[0] - 1, smooth_win) // 2 * 2 + 1))
        dis = close_pose_branch_mouth(dis, close_mouth_ratio).astype(np.float32) * np.float32(amp_pos)
        base = net_c(a[:, 0:18, :], fid)[0].cpu().numpy()
        seg = dis + calib_baseline(base, amp_lip_x, amp_lip_y) + fid.cpu().numpy()
        out.append(solve_inverse_lip(seg))
    fl = np.concatenate(out)
    fl[:, 27 * 3:28 * 3] = fl[:, 28 * 3:29 * 3] * 2 - fl[:, 29 * 3:30 * 3]       # revise the nose top point (:300)
    return _savgol(fl, 5)


def to_image_landmarks(fl, scale=1.0, shift=(0.0, 0.0), eyes=True, smooth=True, rng=np.random):
    """main_end2end_module2.py:262-272 on the (T, 204) output above: flip / scale / shift x and y into image pixels,
    ``add_naive_eye``, the two clip-level Savitzky-Golay filters.  Returns (T, 68, 3)."""
    fl = np.array(fl, dtype=np.float64).reshape((-1, 68, 3))
    fl[:, :, 0:2] = -fl[:, :, 0:2]
    fl[:, :, 0:2] = fl[:, :, 0:2] / scale - np.asarray(shift, dtype=np.float64)
    if eyes:
        fl = add_naive_eye(fl, rng)
    return smooth_landmarks(fl) if smooth else fl.astype(np.float32)


@torch.no_grad()
def predict_landmarks(net, au_windows, face_id, scale=1.0, shift=(0.0, 0.0), segment=512, smooth=True):
    """Content branch only (``__train_face_wo_pos__``): segments of 512 windows through the content net
    (train_audio2landmark.py:279-296), ``fl = displacement + face_id`` (:298), then main_end2end_module2.py:264-271.
    Returns (T, 68, 2) image-pixel landmarks (x, y)."""
    net.eval()
    dev = next(net.parameters()).device
    au = torch.as_tensor(au_windows, dtype=torch.float32, device=dev)
    fid = torch.as_tensor(face_id, dtype=torch.float32, device=dev).view(1, FACE_ID_FEAT_SIZE)
    outs = []
    for j in range(0, au.shape[0], segment):
        dis, f = net(au[j:j + segment][:, 0:18, :], fid)
        outs.append(dis + f)
    fl = torch.cat(outs, 0).cpu().numpy()
    return to_image_landmarks(fl, scale, shift, eyes=False, smooth=smooth)[:, :, :2]


# ------------------------------------------------------------------------------------------------ clip driver pieces
def load_module1(speaker_ckpt, content_ckpt, device=None):
    """The two checkpoints ``Audio2landmark_model.__init__`` loads (train_audio2landmark.py:55-79):
    ``ckpt_speaker_branch.pth`` -> ``G`` (key 'G', minus the ``comb_mlp`` entries) and ``ckpt_content_branch.pth`` -> ``C``
    (key 'model_g_face_id'); both in eval mode on ``device``.  Returns (net_g, net_c)."""
    net_g = Audio2LandmarkPos(drop_out=0.5)                                          # :55-59
    ck = torch.load(speaker_ckpt, map_location='cpu')
    sd = net_g.state_dict()
    sd.update({k: v for k, v in ck['G'].items() if k.split('.')[0] not in ['comb_mlp']})   # :64-66
    net_g.load_state_dict(sd)
    net_c = Audio2LandmarkContent(use_prior_net=True, drop_out=0.5)                  # :71-73
    net_c.load_state_dict(torch.load(content_ckpt, map_location='cpu')['model_g_face_id'])   # :76-77
    for n in (net_g, net_c):
        n.eval()
        for q in n.parameters():
            q.requires_grad_(False)
    return (net_g.to(device), net_c.to(device)) if device is not None else (net_g, net_c)


def adjust_and_norm_input_face(shape_3d, std_face_z=None):
    """main_end2end_module2.py:196-204 + util/utils.py:348-359 on the photo's (68, 3) landmarks (pixels): the manual lip /
    eye adjustment, then the normalisation to Module1's frame.  Returns (face_id (68, 3), scale, shift (2,)); ``std_face_z``
    is column z of STD_FACE_LANDMARKS.txt (the reference replaces the detected depth by it x 0.1; zeros when absent)."""
    f = np.array(shape_3d, dtype=np.float64, copy=True).reshape(68, 3)
    f[49:54, 1] += 1.
    f[55:60, 1] -= 1.
    f[[37, 38, 43, 44], 1] -= 2
    f[[40, 41, 46, 47], 1] += 2
    scale = 1.6 / (f[0, 0] - f[16, 0])
    shift = -0.5 * (f[0, 0:2] + f[16, 0:2])
    f[:, 0:2] = (f[:, 0:2] + shift) * scale
    f[:, 2] = 0.0 if std_face_z is None else np.asarray(std_face_z, dtype=np.float64).reshape(68) * 0.1
    f[:, 0:2] = -f[:, 0:2]
    return f, float(scale), shift


def photo_landmarks_in_pixels(face_id, scale, shift):
    """main_end2end_module2.py:311-315 ('save ori'): the normalised face back in image pixels, (68, 2)."""
    f = np.array(face_id, dtype=np.float64, copy=True).reshape(68, 3)
    f[:, 0:2] = -f[:, 0:2] / scale - np.asarray(shift, dtype=np.float64)
    return f[:, :2].astype(np.float32)
