"""In-process clip streaming: landmark sequence -> motion grids -> landmark maps -> (netF) -> generator -> frames.

What the reference does for a clip (main_end2end_module2.py:181-343): Module1 predicts a (T, 68, 3) landmark
sequence from the audio (``Audio2landmark_model.test`` :254-256), smooths it with two Savitzky-Golay filters
(:262-272), WRITES one txt + one PNG per frame under ``Data/`` (:294-331), then shells out to
``python test.py --model geomcgt_ifw_test`` (:96-110) whose dataset re-reads those files, builds ``warp_motion`` with
``scipy.griddata`` and the landmark maps with ``cv2.circle`` per frame on the CPU
(Module2/data/umlvdfw_test_dataset.py:114-167), runs the model at batch size 1 and writes PNGs; ffmpeg assembles them
at 62.5 fps (:123-124).  A 10 s clip is 625 generator frames.

Here the same stages run in one process, batched, with no file / subprocess / PNG round trip:

    landmarks (T, 68, 2)  --cal_motion256 (device rasteriser, data/motion.py)-->  warp_motion (B, 256, 256, 2)
                          --landmark_discs (device, losses.py)---------------->  tB_lm (B, 1, 256, 256)
                          --flow_network_warp (device pre/post + frozen netF)->  iw_flow, if_mask
                          --GeomCGTIFWTestModel.forward (static drawing cached per photo)-->  fake_B (B, 1, 256, 256)

Landmark sources: a directory in the reference's ``Alm_txt`` layout (``%05d.txt`` frames + ``ori.txt``, 68 lines
"x y" each, written at :300-303 / :324-328 and read at umlvdfw_test_dataset.py:123-129), an array, or
``Audio2LandmarkContent`` (the Module1 content network mirrored in ``animateportrait_amd/module1.py``).
"""
import glob
import os
import time

import numpy as np
import torch

from . import losses, ops
from .data.motion import cal_motion256


def read_landmark_txt(path):
    """68 lines 'x y' -> (68, 2) float32 (umlvdfw_test_dataset.py:123-124)."""
    rows = [l.split() for l in open(path).read().splitlines() if l.strip()]
    lm = np.array([[float(r[0]), float(r[1])] for r in rows], dtype=np.float32)
    if lm.shape != (68, 2):
        raise ValueError('%s: expected 68 "x y" lines, got %s' % (path, lm.shape))
    return lm


def write_landmark_txt(path, lm):
    """the writer of main_end2end_module2.py:300-303: print(x, y) per landmark"""
    with open(path, 'w') as f:
        for x, y in np.asarray(lm):
            print(float(x), float(y), file=f)


def load_landmark_dir(d, scale=1.0):
    """``Data/Alm_txt/MTCNN/<db>_MTCNN/``: ori.txt (the photo's landmarks) + 00000.txt, 00001.txt, ... (the clip).
    scale: the reference writes 512-px coordinates for 256-px photos (:292-296); pass 0.5 to bring them back."""
    ori = read_landmark_txt(os.path.join(d, 'ori.txt')) * scale
    frames = sorted(glob.glob(os.path.join(d, '[0-9]*.txt')))
    if not frames:
        raise FileNotFoundError('no frame landmark files (%%05d.txt) in %s' % d)
    return ori, np.stack([read_landmark_txt(f) for f in frames]) * scale


def smooth_landmarks(fl):
    """The 'additional smooth' of main_end2end_module2.py:268-271 on a (T, 68, C) sequence: Savitzky-Golay, window 15
    for the 48 non-lip points and 5 for the lips, order 3, along time."""
    from scipy.signal import savgol_filter
    t, p, c = fl.shape
    flat = np.asarray(fl, dtype=np.float64).reshape(t, p * c).copy()
    if t >= 15:
        flat[:, :48 * c] = savgol_filter(flat[:, :48 * c], 15, 3, axis=0)
    if t >= 5:
        flat[:, 48 * c:] = savgol_filter(flat[:, 48 * c:], 5, 3, axis=0)
    return flat.reshape(t, p, c).astype(np.float32)


def window_of(lm, size=256, margin=0.15):
    """[x1, x2, y1, y2] square window around a landmark set (what trans_lm derives from the crop parameters,
    umlvdfw_test_dataset.py:12-31; only used for the landmark visualisation of the test model)."""
    lo, hi = lm.min(0), lm.max(0)
    c = (lo + hi) / 2
    half = (hi - lo).max() * (0.5 + margin)
    x1, y1 = int(round(c[0] - half)), int(round(c[1] - half))
    s = int(round(2 * half))
    return [x1, x1 + s, y1, y1 + s]


class ClipStreamer:
    """Drives a ``GeomCGTIFWTestModel`` over a landmark sequence in batches of ``batch`` frames."""

    def __init__(self, model, batch=16):
        self.model, self.batch = model, int(batch)
        self.device = model.device
        self.timing = {}

    def _tick(self, key, t0):
        torch.cuda.synchronize(self.device)
        self.timing[key] = self.timing.get(key, 0.0) + time.perf_counter() - t0

    @torch.no_grad()
    def run(self, photo, photo_lm, landmark_seq, matte=None, profile=False):
        """photo: (1, 3, S, S) in [-1, 1]; photo_lm: (68, 2) px (x, y); landmark_seq: (T, 68, 2) px.
        matte: (1, 1, S, S) in [0, 1] when the model has no matting net.  Returns (T, output_nc, S, S) on the device.
        profile=True synchronises after every stage and fills ``self.timing`` (seconds per stage)."""
        dev, m = self.device, self.model
        photo = photo.to(dev).float().contiguous()
        s = photo.shape[-1]
        lm0 = torch.as_tensor(photo_lm, dtype=torch.float32)
        seq = torch.as_tensor(landmark_seq, dtype=torch.float32)
        t_total = seq.shape[0]
        a_lm = losses.landmark_discs(lm0.view(1, 68, 2).to(dev), s, s, 5 if s == 512 else 3)       # draw2 op=0 radius rule
        out = torch.empty((t_total, m.opt.output_nc, s, s), dtype=torch.float32, device=dev)
        self.timing = {}
        photo_b, alm_b, matte_b = {}, {}, {}
        ClipStreamer._clips = getattr(ClipStreamer, '_clips', 0) + 1
        clip_id = ('clip', ClipStreamer._clips)      # names this run's constant photo landmark map for the generator's cache
        for lo in range(0, t_total, self.batch):
            hi = min(lo + self.batch, t_total)
            b = hi - lo
            if b not in photo_b:                     # one expanded copy per batch size: the model caches per photo tensor
                photo_b[b] = photo.expand(b, -1, -1, -1).contiguous()
                alm_b[b] = a_lm.expand(b, -1, -1, -1).contiguous()
                matte_b[b] = None if matte is None else matte.to(dev).float().expand(b, -1, -1, -1).contiguous()
            lm_t = seq[lo:hi]
            t0 = time.perf_counter()
            motion = cal_motion256(lm0.unsqueeze(0).expand(b, -1, -1).numpy(), lm_t.numpy(), device=dev, size=s)
            if profile:
                self._tick('motion_grid', t0)
            t0 = time.perf_counter()
            lm_dev = lm_t.to(dev)
            tb_lm = losses.landmark_discs(lm_dev, s, s, 5 if s == 512 else 3)
            if profile:
                self._tick('landmark_maps', t0)
            data = {'A': photo_b[b], 'warp_motion': motion, 'A_lm': alm_b[b], 'tB_lm': tb_lm,
                    'A_lm_68': lm0.view(1, 68, 2).expand(b, -1, -1).to(dev), 'tB_lm_68': lm_dev,
                    'image_paths': ['%05d' % i for i in range(lo, hi)], 'clip_id': clip_id}
            if matte is not None:
                data['matte'] = matte_b[b]
            if m.aux['netF'] is None:                # no intrinsic-flow network: no flow, nothing masked out
                data['iw_flow'] = torch.zeros((b, 2, s, s), device=dev)
                data['if_mask'] = torch.ones((b, 1, s, s), device=dev)
            t0 = time.perf_counter()
            m.set_input(data)                        # runs flow_network_warp when aux['netF'] is set
            if profile:
                self._tick('set_input_netF', t0)
            t0 = time.perf_counter()
            m.test()
            out[lo:hi] = m.fake_B
            if profile:
                self._tick('generator', t0)
        if ops.FUSED_NORM:
            ops.check_fused_norm()                   # opt-in in-kernel InstanceNorm: a timed-out exchange invalidates the clip
        return out


def reference_style_cpu_clip(oracle_frame_fn, photo_lm, landmark_seq, workdir):
    """The reference's data path for the same clip, for the CPU wall-clock beside ``ClipStreamer`` (bench.py --stream):
    per frame, write the landmark txt (main_end2end_module2.py:300-303), read it back (umlvdfw_test_dataset.py:123-129),
    build warp_motion with scipy.griddata (``oracle.motion.cal_motion256``) and run ``oracle_frame_fn(lm, motion)`` --
    the oracle's generator -- at batch size 1.  Test / bench infrastructure: the callables come from ``oracle``."""
    os.makedirs(workdir, exist_ok=True)
    write_landmark_txt(os.path.join(workdir, 'ori.txt'), photo_lm)
    for k, lm in enumerate(landmark_seq):
        write_landmark_txt(os.path.join(workdir, '%05d.txt' % k), lm)
    ori, seq = load_landmark_dir(workdir)
    return [oracle_frame_fn(ori, lm) for lm in seq]
