"""Module2 side of the end-to-end driver, in one process -- counterpart of ``test_gan_new`` and the frame / video steps of
main_end2end_module2.py:90-124, 294-343.

The reference writes one landmark txt + one landmark PNG per frame, shells out to ``python test.py --model
geomcgt_ifw_test`` (which re-reads them, builds the motion grids with scipy on the CPU and runs the model at batch 1),
copies the PNGs it wrote and calls ffmpeg at 62.5 fps.  Here: photo + a landmark clip in, frames (and the video, when
ffmpeg exists) out, through ``stream.ClipStreamer`` in batches.

    python -m animateportrait_amd.end2end --photo face.png --landmarks Data/Alm_txt/MTCNN/<db>_MTCNN \\
        --landmark_scale 0.5 --name formal/drawing --epoch 70 --out output/<db> [--audio a.wav]

Landmark sources: ``--landmarks DIR`` (the reference's ``Alm_txt`` layout: ``ori.txt`` + ``%05d.txt``, 512-px coordinates
for a 256-px photo -> ``--landmark_scale 0.5``), ``--landmarks_npy FILE`` ((T + 1, 68, 2): row 0 = the photo's), or the audio
itself: ``--wav FILE --photo_landmarks TXT`` runs the audio half of main_end2end_module2.py:181-272 in process -- mel
windows (audio.py) -> the two Module1 networks (``--load_a2l_G_name`` / ``--load_a2l_C_name`` checkpoints) -> the landmark
post-processing -> image-pixel landmarks -- with the photo's detected (68, 3) landmarks (``face_alignment`` output, a txt of 68
rows) and a speaker embedding (``--speaker_emb`` txt of 256 numbers; the resemblyzer / AutoVC front end is not part of this
build) as inputs.  The matte comes from ``--matte PNG`` (white = foreground) unless the model has its matting network.
"""
import argparse
import os
import shutil
import subprocess
import sys

import numpy as np
import torch

from . import stream
from .models import create_model
from .options.base_options import TestOptions


def tensor2im(t):
    """util/util.py:9-29 for one (C, H, W) frame in [-1, 1] -> uint8 (H, W, 3)."""
    a = t.detach().float().cpu().numpy()
    if a.shape[0] == 1:
        a = np.tile(a, (3, 1, 1))
    elif a.shape[0] == 2:
        a = np.concatenate([a, a[1:2]], 0)
    return ((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(np.uint8)


def load_photo(path, size):
    from PIL import Image
    im = Image.open(path).convert('RGB')
    if im.size != (size, size):
        im = im.resize((size, size), Image.BICUBIC)
    a = np.asarray(im, dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).unsqueeze(0) * 2 - 1                # transforms.Normalize(0.5, 0.5)


def load_matte(path, size):
    from PIL import Image
    im = Image.open(path).convert('L')
    if im.size != (size, size):
        im = im.resize((size, size), Image.BILINEAR)
    return torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).view(1, 1, size, size)


def landmarks_from_audio(a, device):
    """main_end2end_module2.py:205-272 in process (mel -> AutoVC conversion -> both Module1 networks -> post-processing), with
    the clip's f0 track and speaker embedding as inputs: returns (photo landmarks (68, 2), clip (T, 68, 2)) px."""
    from . import audio, module1
    shape = np.loadtxt(a.photo_landmarks).reshape(68, -1)
    if shape.shape[1] == 2:
        shape = np.concatenate([shape, np.zeros((68, 1))], 1)
    std_z = np.loadtxt(a.std_face).reshape(68, 3)[:, 2] if a.std_face else None
    face_id, scale, shift = module1.adjust_and_norm_input_face(shape, std_z)
    net_g, net_c = module1.load_module1(a.load_a2l_G_name, a.load_a2l_C_name, device)
    emb = np.loadtxt(a.speaker_emb).reshape(-1).astype(np.float32)
    converter = None
    if not a.no_autovc:
        # main_end2end_module2.py:218-224: Module1 sees the AutoVC-converted spectrogram, not the raw mel
        from . import autovc
        G = autovc.load_generator(a.load_AUTOVC_name, device)
        emb_trg = autovc.load_target_embedding(a.autovc_target_emb)
        f0 = np.load(a.f0_npy) if a.f0_npy else None
        if f0 is None:
            print('WARNING: no --f0_npy: the converter runs on an all-unvoiced f0 track (the reference extracts RAPT f0 with '
                  'pysptk, which is not in this image)')
        converter = lambda mel: autovc.convert_mel(G, mel, None if f0 is None else f0[:mel.shape[0]], emb, emb_trg, device)   # noqa: E731
    windows = audio.clip_audio_features(a.wav, max_frames=a.max_frames, converter=converter)
    fl = module1.predict_landmarks_speaker_aware(net_g, net_c, windows, emb, face_id.reshape(-1))
    seq = module1.to_image_landmarks(fl, scale=scale, shift=shift)[:, :, :2]
    return module1.photo_landmarks_in_pixels(face_id, scale, shift), seq.astype(np.float32)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--photo', required=True)
    ap.add_argument('--landmarks', default=None, help='directory in the Alm_txt layout')
    ap.add_argument('--landmarks_npy', default=None)
    ap.add_argument('--landmark_scale', type=float, default=1.0)
    ap.add_argument('--matte', default=None)
    ap.add_argument('--out', required=True)
    ap.add_argument('--fps', type=float, default=62.5)                              # main_end2end_module2.py:343
    ap.add_argument('--audio', default=None)
    ap.add_argument('--wav', default=None, help='drive the clip from this audio file (needs --photo_landmarks)')
    ap.add_argument('--photo_landmarks', default=None, help='txt, 68 rows "x y z": the photo\'s detected landmarks in pixels')
    ap.add_argument('--speaker_emb', default=None, help='txt with the 256-d resemblyzer speaker embedding of the clip (required with --wav)')
    ap.add_argument('--load_AUTOVC_name', default='Module1/checkpoints/ckpt_autovc.pth',
                    help='AutoVC converter checkpoint (main_end2end_module2.py:43); Module1 is fed the converted spectrogram')
    ap.add_argument('--autovc_target_emb', default=None, help='target-speaker embedding txt (default: the reference checkout\'s obama_emb.txt)')
    ap.add_argument('--f0_npy', default=None, help='normalised RAPT f0 track of the clip (extract_f0_func_audiofile), one value per mel frame')
    ap.add_argument('--no_autovc', action='store_true',
                    help='feed the RAW mel spectrogram to Module1 (NOT what the reference does: its checkpoints were trained on '
                         'AutoVC-converted spectrograms); for experiments with networks trained that way')
    ap.add_argument('--std_face', default=None, help='STD_FACE_LANDMARKS.txt (its depth column replaces the detected one)')
    ap.add_argument('--load_a2l_G_name', default='Module1/checkpoints/ckpt_speaker_branch.pth')
    ap.add_argument('--load_a2l_C_name', default='Module1/checkpoints/ckpt_content_branch.pth')
    ap.add_argument('--max_frames', type=int, default=None)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=256)
    a, rest = ap.parse_known_args(argv)
    if sum(x is not None for x in (a.landmarks, a.landmarks_npy, a.wav)) != 1:
        ap.error('exactly one of --landmarks / --landmarks_npy / --wav')
    if a.wav is not None and a.photo_landmarks is None:
        ap.error('--wav needs --photo_landmarks')
    if a.wav is not None and a.speaker_emb is None:
        ap.error('--wav needs --speaker_emb (the 256-d resemblyzer embedding the speaker-aware branch is conditioned on, '
                 'main_end2end_module2.py:215-217); a zero vector is not a neutral default')
    if a.wav is not None and not a.no_autovc and not os.path.exists(a.load_AUTOVC_name):
        ap.error('--wav: AutoVC checkpoint %s not found; the reference converts the spectrogram before Module1 '
                 '(pass --load_AUTOVC_name, or --no_autovc to feed the raw mel on purpose)' % a.load_AUTOVC_name)
    # the model's own options: the test settings of test_gan_new (:95-104) unless given
    defaults = ['--model', 'geomcgt_ifw_test', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--netg_resb_div', '3',
                '--netg_resb_disp', '3', '--output_nc', '1', '--dataset_mode', 'synthetic', '--blendbg', '1', '--gpu_ids', '0']
    opt = TestOptions().parse(defaults + rest)
    torch.cuda.set_device(opt.gpu_ids[0])
    model = create_model(opt)
    model.setup(opt)                    # '<epoch>_net_G_A.pth' + the static drawing generator; missing files are errors
    model.eval()
    if a.wav is not None:
        lm0, seq = landmarks_from_audio(a, torch.device('cuda', opt.gpu_ids[0]))
        if a.audio is None:
            a.audio = a.wav
    elif a.landmarks is not None:
        lm0, seq = stream.load_landmark_dir(a.landmarks, a.landmark_scale)
    else:
        arr = np.load(a.landmarks_npy).astype(np.float32) * a.landmark_scale
        lm0, seq = arr[0], arr[1:]
    photo = load_photo(a.photo, a.size)
    matte = load_matte(a.matte, a.size) if a.matte else None
    if matte is None and model.aux.get('modnet') is None:
        raise SystemExit('no matting network is attached (aux["modnet"]): pass --matte PNG')
    frames = stream.ClipStreamer(model, batch=a.batch).run(photo, lm0, seq, matte=matte)
    fdir = os.path.join(a.out, 'frames')
    os.makedirs(fdir, exist_ok=True)
    from PIL import Image
    for k in range(frames.shape[0]):
        Image.fromarray(tensor2im(frames[k])).save(os.path.join(fdir, '%05d.png' % k))
    print('wrote %d frames to %s' % (frames.shape[0], fdir))
    ffmpeg = shutil.which('ffmpeg')
    if ffmpeg is None:
        print('ffmpeg not found: frames only (the reference assembles them at %.1f fps, :118-121)' % a.fps)
        return 0
    video = os.path.join(a.out, 'output.mp4')
    subprocess.check_call([ffmpeg, '-loglevel', 'panic', '-framerate', str(a.fps), '-i', os.path.join(fdir, '%05d.png'),
                           '-c:v', 'libx264', '-y', '-vf', 'format=yuv420p', video])
    if a.audio:
        subprocess.check_call([ffmpeg, '-loglevel', 'panic', '-i', video, '-i', a.audio, '-vcodec', 'copy', '-acodec', 'copy',
                               '-y', video.replace('.mp4', '.mov')])
    print('output is', video)
    return 0


if __name__ == '__main__':
    sys.exit(main())
