"""Module1's audio -> landmark content network for the in-process clip pipeline.

``Audio2LandmarkContent`` mirrors ``Audio2landmark_content`` (Module1/src/models/model_audio2landmark.py:28-90): the
same layers under the same attribute names, so the reference's content checkpoint
(``ckpt_content_branch.pth`` -> ``['model_g_face_id']``, train_audio2landmark.py:60-66) loads with ``strict=True``.
It is a 3-layer LSTM(80 -> 256) over 18-frame mel windows plus a 3-layer MLP: ~1.5 M parameters, run ONCE per clip over
all windows -- stock PyTorch-ROCm, as the hot-path scope prescribes for Module1 (no hand kernels).

``predict_landmarks`` is the content-branch part of ``Audio2landmark_model.__train_face_and_pos__`` / ``test``
(train_audio2landmark.py:101-141, 247-352) followed by the clip-level post-processing of
main_end2end_module2.py:262-272: displacement + face id -> (T, 68, 3) -> sign flip, scale / shift to image pixels ->
Savitzky-Golay smoothing.  NOT built: the speaker-aware pose branch (``Audio2landmark_pos``, :296-386: head motion) and
the AutoVC mel front end (librosa / pysptk / pyworld / resemblyzer are not in this image, and neither are the
checkpoints); callers pass the (T, 18, 80) mel windows the reference's ``au_data`` holds.
"""
import numpy as np
import torch
import torch.nn as nn

from .stream import smooth_landmarks

FACE_ID_FEAT_SIZE = 204       # 68 x 3, model_audio2landmark.py:23


class Audio2LandmarkContent(nn.Module):
    def __init__(self, num_window_frames=18, in_size=80, lstm_size=161, hidden_size=256, num_layers=3):
        super().__init__()
        # (the reference assigns fc_prior and fc to the same Sequential first and then replaces fc, :33-38 / :62-70:
        # both exist in its state_dict)
        self.fc_prior = nn.Sequential(nn.Linear(in_size, 256), nn.BatchNorm1d(256), nn.LeakyReLU(0.2),
                                      nn.Linear(256, lstm_size))
        self.fc = self.fc_prior                 # registration order of the reference's state_dict: fc_prior, fc, bilstm
        self.bilstm = nn.LSTM(input_size=in_size, hidden_size=hidden_size, num_layers=num_layers, dropout=0,
                              bidirectional=False, batch_first=True)             # use_prior_net=False branch, :49-55
        self.fc = nn.Sequential(nn.Linear(hidden_size + FACE_ID_FEAT_SIZE, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2),
                                nn.Linear(512, 256), nn.BatchNorm1d(256), nn.LeakyReLU(0.2), nn.Linear(256, 204))
        self.in_size, self.num_window_frames = in_size, num_window_frames

    def forward(self, au, face_id):                                              # :74-88
        output, _ = self.bilstm(au)
        output = output[:, -1, :]
        if face_id.shape[0] == 1:
            face_id = face_id.repeat(output.shape[0], 1)
        return self.fc(torch.cat((output, face_id), dim=1)), face_id


@torch.no_grad()
def predict_landmarks(net, au_windows, face_id, scale=1.0, shift=(0.0, 0.0), segment=512, smooth=True):
    """au_windows: (T, 18, 80) mel windows; face_id: (204,) the photo's 3-D landmarks in Module1's normalised frame.
    Returns (T, 68, 2) image-pixel landmarks (x, y): segments of 512 windows through the content net
    (train_audio2landmark.py:279-296), ``fl = displacement + face_id`` (:298), then main_end2end_module2.py:264-271:
    ``fl[:, :, :2] = -fl[:, :, :2] / scale - shift`` and the two Savitzky-Golay filters."""
    net.eval()
    dev = next(net.parameters()).device
    au = torch.as_tensor(au_windows, dtype=torch.float32, device=dev)
    fid = torch.as_tensor(face_id, dtype=torch.float32, device=dev).view(1, FACE_ID_FEAT_SIZE)
    outs = []
    for j in range(0, au.shape[0], segment):
        seg = au[j:j + segment]
        dis, f = net(seg[:, 0:18, :], fid)
        outs.append(dis + f)
    fl = torch.cat(outs, 0).view(-1, 68, 3).cpu().numpy()
    fl[:, :, 0:2] = -fl[:, :, 0:2]
    fl[:, :, 0:2] = fl[:, :, 0:2] / scale - np.asarray(shift, dtype=np.float32)
    if smooth:
        fl = smooth_landmarks(fl)
    return fl[:, :, :2]
