#!/usr/bin/env python3
"""Golden of the COMPOSED train step: tests/golden/train_step.npz, produced by running the
REFERENCE's own model class -- ``GeomGMIFWForeModel.__init__ / set_input / forward / backward_G / backward_D_A{,_l,_le,_ll,
_coh} / optimize_parameters`` (Module2/models/geomgm_ifw_fore_model.py:211-388, :443-505, :517-565, :677-780, :637-672,
:782-819), its own option parser (options/train_options.py + modify_commandline_options, README flags readme.md:65), its own
``define_G / define_D / GANLoss / ImagePool / sparse_image_warp / FaceLoss / torch.optim.Adam`` -- in the build container:

    python tests/golden/make_train_golden.py

What is NOT the reference's (absent from this image or from the reference tree) and what stands in for it:
* no GPU: ``Tensor.cuda`` / ``Module.cuda`` are the identity; ``opt.gpu_ids`` / ``gpu_ids_p`` are ``[0]`` lists that report
  ``len() == 0`` so that ``init_net`` stays on the CPU while ``self.gpu = opt.gpu_ids[0]`` still works;
* the four frozen nets' checkpoints: the fixed-seed stand-ins of animateportrait_amd/standins.py (same call contracts) are
  what ``MobileFaceNet(...)``, ``MODNet(...)``, ``Sphere20a()``, ``load_flow_network()`` return, and ``torch.load`` of the
  three checkpoint paths returns the stand-ins' own state_dicts (so the reference's load_state_dict calls run for real);
* cv2 (``getlipline``, :507-515): ``cv2.line`` = oracle/cv_raster.thick_line (OpenCV's ThickLine rule restated; float end
  points truncated) -- the one piece of this golden that is the oracle's, everything around it is the reference's;
* width: ngf = ndf = 8 (fixtures stay small), batch 1 (the reference's train step is b=1 only).

Stored per quantity: the reference evaluated in fp64 (``torch.set_default_dtype(float64)``, modules ``.double()``) = the
truth, kept as float32 arrays, and the distance of the reference's own fp32 evaluation from it (``*_noise``) = the bar a
second fp32 implementation can be held to.  Nothing of the reference's source is stored.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

README_ARGV = ('--dataroot drawing --name training/drawing1 --model geomgm_ifw_fore --netG resnet_9blocks_rcatland32_full_ifw '
               '--netg_resb_div 3 --netg_resb_disp 3 --output_nc 1 --display_env training_drawing1 --lr 0.00005 '
               '--lambda_geom 50 --lambda_geom_lipline 50 --more_weight_for_lip 2 --lambda_face 3.0 --lambda_warp_inter 10 '
               '--blendbg 1 --select_target12_thre 0.0 --niter 70 --niter_decay 0').split()
WIDTH = 8
BATCH_SEED, G_SEED, D_SEED0 = 5, 11, 20
DNAMES = ['D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh']
G_TERMS = ['G_A', 'G_A_l', 'G_A_le', 'G_A_ll', 'G_A_coh', 'geom_B', 'geom_B_lipline', 'warp_B', 'warp_inter1', 'iden_B', 'G']
WINS = {'winA': [36, 220, 30, 210], 'winB': [-12, 200, 24, 230], 'winB2': [40, 260, 50, 256], 'winBr': [32, 224, 32, 224]}
# kept at every second pixel (the full-resolution tensors are the frames and the four TPS-warped constants)
SUBSAMPLED = ('iw_flow', 'if_mask', 'iw_flow2', 'if_mask2', 'mask', 'real_A_fore', 'fake_B_l', 'fake_B2_l', 'real_B_l', 'fake_B_le',
              'real_B_le', 'fake_B2_ll', 'real_B_ll', 'liplinemask1')
SEQ_LR, SEQ_STEPS = 1e-3, 3          # the optimize_parameters sequence runs at a large lr so that a step moves the losses


class NoGPU(list):
    """[0] that reports no GPUs: ``len()`` / ``bool()`` say empty (init_net, BaseModel.device), ``[0]`` still indexes."""
    def __len__(self):
        return 0


class Cast(torch.nn.Module):
    """A frozen stand-in net that casts whatever it is fed to its own dtype (the reference builds some inputs as float32
    whatever the model's dtype: numpy joint maps, ``.float()`` masks)."""
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, *a):
        dt = next(self.net.parameters()).dtype
        return self.net(*[t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t for t in a])


def make_batch():
    """animateportrait_amd.data.synthetic_dataset.make_train_batch(1, seed) + the keys only the reference's set_input reads."""
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    b = make_train_batch(1, seed=BATCH_SEED)
    for k, w in WINS.items():
        b[k] = torch.IntTensor([w])
    return b


def reference_batch(b, dtype):
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t       # noqa: E731
    r = {k: cast(v) for k, v in b.items()}
    z = torch.zeros(1, 3, 256, 256, dtype=dtype)
    r.update(realA_static_warp=z, realA_static_warp2=z, B_lm=r['A_lm'], B_lm_68=r['A_lm_68'], A_paths=['a'], B_paths=['b'],
             B1_lm_68=r['tB_lm_68'], B2_lm_68=r['tB2_lm_68'], winBr1=b['winBr'], winBr2=b['winBr'], B1_path=['b1'])
    return r


def install_shims():
    from make_golden import import_reference
    cv2 = types.ModuleType('cv2')
    from oracle import cv_raster

    def line(img, p0, p1, color, thickness):
        work = np.zeros(img.shape, dtype=np.uint8)
        cv_raster.thick_line(work, (int(p0[0]), int(p0[1])), (int(p1[0]), int(p1[1])), thickness, 255)
        img[work > 0] = color
        return img
    cv2.line = line
    sys.modules['cv2'] = cv2
    networks, _, _, _ = import_reference()
    # sparse_image_warp.py:125 -- torch.solve(rhs, lhs); in the fp64 run the reference's own ``.float()`` casts leave rhs float32
    torch.solve = lambda rhs, lhs: (torch.linalg.solve(lhs, rhs.to(lhs.dtype)), None)
    # ... and its other ``.float()`` casts (:103, :109, :121 ...) would mix dtypes: while the spline runs on fp64 inputs,
    # ``Tensor.float`` keeps fp64 tensors as they are, so the reference's own arithmetic is evaluated in double
    from models import sparse_image_warp as siw
    siw_real, float_real = siw.sparse_image_warp, torch.Tensor.float

    def siw_any_dtype(image, src, dst, *a, **k):
        if torch.float64 not in (image.dtype, src.dtype, dst.dtype):
            return siw_real(image, src, dst, *a, **k)
        torch.Tensor.float = lambda self, *aa, **kk: self if self.dtype == torch.float64 else float_real(self, *aa, **kk)
        try:
            return siw_real(image.double(), src.double(), dst.double(), *a, **k)
        finally:
            torch.Tensor.float = float_real
    siw.sparse_image_warp = siw_any_dtype
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from models import geomgm_ifw_fore_model as ref
    from animateportrait_amd import standins
    ref.MobileFaceNet = lambda *a, **k: Cast(standins.StandinLandmarkNet())
    ref.MODNet = lambda *a, **k: Cast(standins.StandinMatteNet())
    ref.load_flow_network = lambda *a, **k: Cast(standins.StandinFlowNet())
    networks.Sphere20a = lambda *a, **k: Cast(standins.StandinFaceNet())
    networks.FaceLoss.load_sphere_model = lambda self, path: None
    real_load = torch.load

    def load(path, *a, **k):
        p = str(path)
        if 'mobilefacenet_model_best' in p:
            return {'state_dict': Cast(standins.StandinLandmarkNet()).state_dict()}
        if 'modnet_photographic_portrait_matting' in p:
            return {'module.' + kk: v for kk, v in Cast(standins.StandinMatteNet()).state_dict().items()}
        return real_load(path, *a, **k)
    torch.load = load
    return networks, ref


def parse_options(ref, lr=None):
    from options.train_options import TrainOptions
    p = TrainOptions().initialize(argparse.ArgumentParser())
    p = ref.GeomGMIFWForeModel.modify_commandline_options(p, True)
    opt = p.parse_args(README_ARGV + ['--ngf', str(WIDTH), '--ndf', str(WIDTH)] + (['--lr', str(lr)] if lr else []))
    opt.isTrain = True                               # what TrainOptions.parse() sets (train_options.py)
    opt.gpu_ids, opt.gpu_ids_p = NoGPU([0]), NoGPU([0])
    return opt


def build_model(ref, dtype, lr=None):
    from oracle import generator as og, discriminator as od
    cwd = os.getcwd()
    os.chdir('/root/reference/Module2')               # './faceLmarkLookup.npy' (:384)
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            m = ref.GeomGMIFWForeModel(parse_options(ref, lr))
    finally:
        os.chdir(cwd)
    sdG = og.init_params(og.generator_param_shapes(3, 1, WIDTH, 9, 3, 3), seed=G_SEED)
    assert list(sdG) == list(m.netG_A.state_dict()), 'generator state_dict keys / order'
    m.netG_A.load_state_dict(sdG, strict=True)
    for i, name in enumerate(DNAMES):
        sd = og.init_params(od.patchgan_param_shapes(1 if name == 'D_A' else 2, WIDTH), seed=D_SEED0 + i)
        getattr(m, 'net' + name).load_state_dict(sd, strict=True)
    if dtype == torch.float64:
        for n in ['netG_A'] + ['net' + d for d in DNAMES] + ['mobilefacenet', 'modnet', 'faceidenloss', 'netF', 'criterionGAN']:
            getattr(m, n).double()
        m.edges = m.edges.double()
    return m


def run(ref, dtype):
    try:
        out = {}
        b = reference_batch(make_batch(), dtype)
        # models are BUILT under the float32 default (the stand-ins draw their fixed-seed weights with the default dtype) and
        # converted; the default dtype only governs the run (``torch.ones`` boxes, ``torch.Tensor`` constants)
        m = build_model(ref, dtype)
        first = m
        torch.set_default_dtype(dtype)
        m.set_input(b)
        out['iw_flow'], out['if_mask'], out['iw_flow2'], out['if_mask2'] = m.iw_flow, m.real_A_if_mask, m.iw_flow2, m.real_A_if_mask2
        m.forward()
        for k in ('mask', 'mask1', 'mask2', 'real_A_fore', 'fake_B_fore', 'fake_B2_fore', 'fake_B', 'fake_B2', 'fakeB_static_warp',
                  'fake_B_l', 'fake_B2_l', 'real_B_l', 'fake_B_le', 'real_B_le', 'fake_B2_ll', 'real_B_ll'):
            out[k] = getattr(m, k).detach().clone()
        nets_D = [getattr(m, 'net' + d) for d in DNAMES]
        m.set_requires_grad(nets_D, False)
        m.optimizer_G.zero_grad()
        m.backward_G()
        out['fake_B_warp'] = m.fake_B_warp.detach().clone()
        out['liplinemask1'] = m.liplinemask1.detach().clone()
        out['fake_B_lm_68'] = m.fake_B_lm_68.detach().clone()
        for k in G_TERMS:
            out['loss_' + k] = getattr(m, 'loss_' + k).detach().clone()
        for k, p in m.netG_A.named_parameters():
            out['gG/' + k] = p.grad.detach().clone()
        assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for d in nets_D for p in d.parameters()), 'D grads in the G step'
        m.set_requires_grad(nets_D, True)
        m.optimizer_D.zero_grad()
        m.backward_D_A(); m.backward_D_A_l(); m.backward_D_A_le(); m.backward_D_A_ll(); m.backward_D_A_coh()   # noqa: E702
        for d in DNAMES:
            out['loss_' + d] = getattr(m, 'loss_' + d).detach().clone()
            for k, p in getattr(m, 'net' + d).named_parameters():
                out['gD/%s/%s' % (d, k)] = p.grad.detach().clone()
        # ---- the step sequence: optimize_parameters x SEQ_STEPS from the same initial state at a large lr
        torch.set_default_dtype(torch.float32)
        m = build_model(ref, dtype, lr=SEQ_LR)
        torch.set_default_dtype(dtype)
        for it in range(SEQ_STEPS):
            m.set_input(b)
            m.optimize_parameters()
            for k in G_TERMS + DNAMES:
                out['seq%d_loss_%s' % (it, k)] = getattr(m, 'loss_' + k).detach().clone()
        out['seq_w_tri_merge'] = m.netG_A.model_tri_merge.weight.detach().clone()
        out['seq_w_D_A_coh_8'] = m.netD_A_coh.model[8].weight.detach().clone()
        out['opt_G_betas'] = torch.tensor(m.optimizer_G.param_groups[0]['betas'])
        out['opt_D_nparams'] = torch.tensor(float(sum(len(g['params']) for g in m.optimizer_D.param_groups)))
        return out, first
    finally:
        torch.set_default_dtype(torch.float32)


def rel(a, b64):
    return float((a.double() - b64).abs().max() / b64.abs().max().clamp_min(1e-30))


def main():
    import warnings
    warnings.filterwarnings('ignore')
    torch.set_num_threads(8)
    from make_golden import save
    networks, ref = install_shims()
    r64, m64 = run(ref, torch.float64)
    r32, _ = run(ref, torch.float32)
    out = dict(batch_seed=np.int64(BATCH_SEED), g_seed=np.int64(G_SEED), d_seed0=np.int64(D_SEED0), width=np.int64(WIDTH),
               seq_lr=np.float64(SEQ_LR), seq_steps=np.int64(SEQ_STEPS),
               **{k: np.array(v, dtype=np.int32) for k, v in WINS.items()})
    opt = m64.opt
    for k in ('lr', 'beta1', 'lambda_G_A_l', 'lambda_G_A_coh', 'lambda_geom', 'lambda_geom_lipline', 'lambda_warp', 'lambda_warp_inter',
              'lambda_face', 'more_weight_for_lip', 'identity_loss', 'warp_loss', 'blendbg', 'mask_type', 'coherent', 'coh_use_more',
              'check_fakeb2_in_backwardD', 'pool_size', 'netg_resb_div', 'netg_resb_disp'):
        out['opt_' + k] = np.float64(getattr(opt, k))
    for k, v in r64.items():
        if k.startswith('loss_') or k.startswith('seq') and 'loss' in k or k.startswith('opt_'):
            out[k] = v.double().numpy()
            if k in r32 and not k.startswith('opt_'):
                out[k + '_f32'] = r32[k].double().numpy()
        elif k in SUBSAMPLED:
            out[k + '_sub2'] = v[..., ::2, ::2].float().numpy()
            out[k + '_noise'] = np.float64(rel(r32[k], v.double()))
        else:
            out[k] = v.float().numpy()
            out[k + '_noise'] = np.float64(rel(r32[k], v.double()))
    save('train_step.npz', **out)
    for k in G_TERMS + DNAMES:
        print('%-16s f64 %.8f  f32 %.8f' % (k, float(r64['loss_' + k]), float(r32['loss_' + k])))
    for it in range(SEQ_STEPS):
        print('step %d  G %.6f (f32 %.6f)  D_A %.6f  D_A_coh %.6f' % (it, float(r64['seq%d_loss_G' % it]), float(r32['seq%d_loss_G' % it]),
                                                                      float(r64['seq%d_loss_D_A' % it]), float(r64['seq%d_loss_D_A_coh' % it])))
    worst = sorted(((out[k], k[:-6]) for k in out if k.endswith('_noise')), reverse=True)[:8]
    print('largest fp32-vs-fp64 distances of the reference itself:', [(k, '%.2e' % v) for v, k in worst])


if __name__ == '__main__':
    main()
