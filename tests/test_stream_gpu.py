"""GPU (-m gpu): the in-process clip pipeline (ClipStreamer: landmarks -> motion grids -> landmark maps -> netF pre/post
-> GeomCGTIFWTestModel) against the oracle's per-frame composition of the reference's data path
(scipy.griddata motion, cv2-rule discs, flow_network_warp, static drawing, generator, blend)."""
import os

import numpy as np
import pytest
import torch

from conftest import linf

pytestmark = pytest.mark.gpu


def test_clip_streamer_vs_oracle_frames():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import contextlib
    import io
    from animateportrait_amd import standins, stream
    from animateportrait_amd.options.base_options import TestOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.synthetic import make_landmarks
    from oracle import generator as og, static_generator as osg, motion as om, aux_glue as oa
    dev = torch.device('cuda:0')
    opt = TestOptions().parse(['--model', 'geomcgt_ifw_test', '--netG', 'resnet_9blocks_rcatland32_full_ifw',
                               '--dataset_mode', 'synthetic', '--name', 'drawing_stream', '--output_nc', '1', '--ngf', '8',
                               '--netg_resb_div', '3', '--netg_resb_disp', '3', '--gpu_ids', '0'])
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
    sd_g = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234)
    sd_s = og.init_params(osg.static_param_shapes(3, 1, 64), seed=4321)
    model.netG_A.load_state_dict(sd_g, strict=True)
    model.net_staticG.load_state_dict(sd_s, strict=True)
    netF = standins.StandinFlowNet()
    model.aux['netF'] = standins.StandinFlowNet().to(dev)
    # a smooth photo (sampling a noise image at positions that differ by 1e-3 px would dominate the comparison)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 256), torch.linspace(-1, 1, 256), indexing='ij')
    photo = torch.stack([torch.sin(3 * xx + yy), torch.cos(2 * yy - xx), xx * yy], 0).unsqueeze(0).contiguous()
    matte = (((yy / 0.8) ** 2 + (xx / 0.6) ** 2) < 1).float().view(1, 1, 256, 256) * 0.9
    g = torch.Generator().manual_seed(9)
    lm0 = make_landmarks(1, g)[0]
    t = torch.arange(5).view(5, 1, 1).float()
    seq = (lm0.unsqueeze(0) + 3.0 * torch.sin(0.7 * t + lm0.unsqueeze(0) / 40.0)).round() + 0.25   # (5, 68, 2)
    streamer = stream.ClipStreamer(model, batch=2)
    out = streamer.run(photo, lm0, seq, matte=matte, profile=True).cpu()
    assert out.shape == (5, 1, 256, 256) and set(streamer.timing) == {'motion_grid', 'landmark_maps', 'set_input_netF', 'generator'}
    from animateportrait_amd import losses
    n_bad, n_pix, worst_mean = 0, 0, 0.0
    with torch.no_grad():
        static = osg.static_drawing(sd_s, photo)
        a_lm = oa.draw2(256, 256, lm0.numpy(), 3).unsqueeze(0)
        for k in range(5):
            motion = torch.from_numpy(om.cal_motion256(lm0.numpy(), seq[k].numpy())).unsqueeze(0)
            tb_lm = oa.draw2(256, 256, seq[k].numpy(), 3).unsqueeze(0)
            flow, ifm = oa.flow_network_warp(netF, photo, lm0.unsqueeze(0), seq[k:k + 1])
            fake, _, _, _ = osg.streaming_forward(lambda *a: og.generator_forward(sd_g, *a, div=3, disp=3), photo, matte,
                                                  static, a_lm, tb_lm, motion, flow, ifm)
            # (1) the model on the ORACLE's motion grid (same sampling positions): the path's own budget, every pixel
            data = {'A': photo.to(dev), 'warp_motion': motion.to(dev), 'A_lm': a_lm.to(dev), 'tB_lm': tb_lm.to(dev),
                    'A_lm_68': lm0.view(1, 68, 2).to(dev), 'tB_lm_68': seq[k:k + 1].to(dev), 'matte': matte.to(dev),
                    'image_paths': ['%05d' % k]}
            model.set_input(data)
            model.test()
            assert linf(model.fake_B, fake) < 1e-3, (k, linf(model.fake_B, fake))
            # (2) the streamer's frame (device motion rasteriser: grid within 2e-5 = 2.5e-3 px of scipy.griddata's,
            # tests/test_gpu_parity.py).  The generated drawing has hard edges (mask > 0.5 thresholds in warp_acc_flow
            # and in the blend), so a sampling position that moves by 1e-3 px flips isolated pixels: counted, not averaged
            # away -- at most 0.05 % of the pixels of a frame may leave the 1e-3 budget, and the mean stays < 3e-4
            err = (out[k:k + 1] - fake).abs()
            n_bad += int((err > 1e-3).sum())
            n_pix += err.numel()
            worst_mean = max(worst_mean, float(err.mean()))
            assert int((err > 1e-3).sum()) <= 0.0005 * err.numel(), (k, int((err > 1e-3).sum()), float(err.max()))
            assert float(err.mean()) < 3e-4, (k, float(err.mean()))
    print('stream vs oracle: %d of %d pixels beyond 1e-3 (device motion grid), worst frame mean %.2e' % (n_bad, n_pix, worst_mean))
    # same clip in one batch and without stage synchronisation: identical frames
    out2 = stream.ClipStreamer(model, batch=8).run(photo, lm0, seq, matte=matte).cpu()
    assert linf(out2, out) < 1e-4          # another batch size picks another tile shape: fp32 summation order
    # the photo's landmark map is encoded once per clip (generator land1 cache): another photo landmark set must not see
    # the previous clip's encoding
    lm0b = lm0 + torch.tensor([3.0, -2.0])
    out3 = stream.ClipStreamer(model, batch=8).run(photo, lm0b, seq, matte=matte).cpu()
    model.netG_A._land1_cache = None
    out4 = stream.ClipStreamer(model, batch=8).run(photo, lm0b, seq, matte=matte).cpu()
    assert torch.equal(out3, out4) and linf(out3, out2) > 1e-4


def test_end2end_cli_writes_the_clip(tmp_path, monkeypatch):
    """animateportrait_amd.end2end (the in-process test_gan_new of main_end2end_module2.py:90-124): checkpoints loaded by
    name, photo / matte PNGs and an Alm_txt landmark directory in, PNG frames out == ClipStreamer on the same inputs."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from PIL import Image
    from animateportrait_amd import end2end, stream
    from animateportrait_amd.synthetic import make_landmarks
    from oracle import generator as og, static_generator as osg
    monkeypatch.chdir(tmp_path)
    os.makedirs('checkpoints/e2e')
    os.makedirs('checkpoints/static')
    torch.save(og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=1234), 'checkpoints/e2e/7_net_G_A.pth')
    torch.save(og.init_params(osg.static_param_shapes(3, 1, 64), seed=4321), 'checkpoints/static/drawing.pth')
    yy, xx = np.meshgrid(np.linspace(-1, 1, 256), np.linspace(-1, 1, 256), indexing='ij')
    photo = np.stack([np.sin(3 * xx + yy), np.cos(2 * yy - xx), xx * yy], -1)
    Image.fromarray(((photo + 1) * 127.5).astype(np.uint8)).save('photo.png')
    Image.fromarray(((((yy / 0.8) ** 2 + (xx / 0.6) ** 2) < 1) * 255).astype(np.uint8)).save('matte.png')
    g = torch.Generator().manual_seed(9)
    lm0 = make_landmarks(1, g)[0]
    t = torch.arange(5).view(5, 1, 1).float()
    seq = lm0.unsqueeze(0) + 2.0 * torch.sin(0.3 * t + lm0.unsqueeze(0) / 40.0)
    os.makedirs('lm')
    stream.write_landmark_txt('lm/ori.txt', (lm0 * 2).numpy())                    # the reference writes 512-px coordinates
    for k in range(5):
        stream.write_landmark_txt('lm/%05d.txt' % k, (seq[k] * 2).numpy())
    argv = ['--photo', 'photo.png', '--matte', 'matte.png', '--landmarks', 'lm', '--landmark_scale', '0.5', '--out', 'out',
            '--batch', '4', '--name', 'e2e', '--epoch', '7', '--ngf', '8', '--checkpoints_dir', 'checkpoints']
    assert end2end.main(argv) == 0
    files = sorted(os.listdir('out/frames'))
    assert files == ['%05d.png' % k for k in range(5)]
    got = np.stack([np.asarray(Image.open(os.path.join('out/frames', f))) for f in files])
    assert got.shape == (5, 256, 256, 3) and got.std() > 1.0
    # the same clip through the streamer directly
    from animateportrait_amd.options.base_options import TestOptions
    from animateportrait_amd.models import create_model
    opt = TestOptions().parse(['--model', 'geomcgt_ifw_test', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--netg_resb_div',
                               '3', '--netg_resb_disp', '3', '--output_nc', '1', '--dataset_mode', 'synthetic', '--blendbg', '1',
                               '--gpu_ids', '0', '--name', 'e2e', '--epoch', '7', '--ngf', '8', '--checkpoints_dir', 'checkpoints'])
    model = create_model(opt)
    model.setup(opt)
    model.eval()
    lm_ori, lm_seq = stream.load_landmark_dir('lm', 0.5)
    frames = stream.ClipStreamer(model, batch=2).run(end2end.load_photo('photo.png', 256), lm_ori, lm_seq,
                                                     matte=end2end.load_matte('matte.png', 256))
    want = np.stack([end2end.tensor2im(frames[k]) for k in range(5)])
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1        # (batch 4 vs 2: same frames)
