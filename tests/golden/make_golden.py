#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
network code (imported read-only from /root/reference/Module2) on seeded inputs.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

Nothing of the reference's source is stored: fixtures are inputs (or the seed that
regenerates them), outputs, and checksums.  Import-time stubs follow SURVEY.md
section 8c: empty ``torchvision`` / ``skimage`` modules, and ``torch.solve`` mapped to
``torch.linalg.solve`` for sparse_image_warp.py:125.

Weights: the reference has no pretrained weights in the tree, so each net is built
with ``networks.define_G/define_D`` (which runs the reference's init_weights) and
then loaded -- strictly, so key names and shapes are verified -- with the seeded
N(0, 0.02) / zero-bias tensors from ``oracle.generator.init_params``.  The GPU box
regenerates the same tensors from the seed; their sha256 is stored in the fixture.
"""
import hashlib
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/Module2'


def import_reference():
    for name in ['torchvision', 'torchvision.models', 'torchvision.transforms', 'skimage', 'skimage.measure']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['skimage.measure'].compare_ssim = None
    sys.modules['skimage.measure'].compare_psnr = None
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    # torch.solve still exists in torch 2.x but only raises; the reference then drops into pdb (:127-128)
    torch.solve = lambda rhs, lhs: (torch.linalg.solve(lhs, rhs), None)
    sys.path.insert(0, REF)
    from models import networks, sparse_image_warp  # noqa
    from intrinsic_flow_models import networks as ifm  # noqa
    from util.image_pool import ImagePool  # noqa
    return networks, sparse_image_warp, ifm, ImagePool


def sha(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def main():
    import warnings
    warnings.filterwarnings('ignore')
    networks, siw, ifm, ImagePool = import_reference()
    from oracle import generator as og, discriminator as od
    from animateportrait_amd.synthetic import make_generator_inputs, generator_args
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(1234)

    def rn(*s):
        return torch.randn(*s, generator=g)

    # ---------------------------------------------------------------- G1-G3 per-op
    ops = {}
    norm = networks.get_norm_layer('instance')
    rb = networks.ResnetBlock(16, 'reflect', norm, False, True)
    networks.init_weights(rb, 'normal', 0.02)
    with torch.no_grad():
        for p in rb.parameters():
            p.copy_(rn(*p.shape) * (0.05 if p.dim() > 1 else 0.5))   # non-zero biases on purpose
    x = rn(2, 16, 16, 16)
    ops['rb_x'] = x
    for k, v in rb.state_dict().items():
        ops['rb_' + k] = v
    with torch.no_grad():
        ops['rb_first'] = rb.conv_block[0:4](x)        # reflect pad + conv + IN + ReLU
        ops['rb_y'] = rb(x)
    rb2 = networks.ResnetBlock2(18, 16, 'reflect', norm, False, True)
    with torch.no_grad():
        for p in rb2.parameters():
            p.copy_(rn(*p.shape) * (0.05 if p.dim() > 1 else 0.5))
    x = rn(2, 18, 16, 16)
    ops['rb2_x'] = x
    for k, v in rb2.state_dict().items():
        ops['rb2_' + k] = v
    with torch.no_grad():
        ops['rb2_y'] = rb2(x)

    # small generator pieces (ngf=8)
    G8 = networks.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [], div=3, disp=3)
    shapes8 = og.generator_param_shapes(3, 1, 8, 9, 3, 3)
    sd8 = og.init_params(shapes8, seed=1234)
    assert list(sd8.keys()) == list(G8.state_dict().keys()), 'state_dict key order mismatch'
    G8.load_state_dict(sd8, strict=True)
    with torch.no_grad():
        x = rn(1, 3, 40, 48)
        ops['stem_x'] = x
        ops['stem10_y'] = G8.model_tri10(x)            # reflect7x7 3->8 + IN + ReLU
        ops['stem00_y'] = G8.model_tri00(x)            # 3->4
        x = rn(2, 8, 32, 32)
        ops['down_x'] = x
        ops['down01_y'] = G8.model_tri01(x)            # 3x3 s2 8->16 + IN + ReLU
        x = rn(1, 32, 16, 24)
        ops['up_x'] = x
        ops['up_y'] = G8.model3[0:3](x)                # deconv 32->16 + IN + ReLU
        ops['up2_y'] = G8.model3[3:6](ops['up_y'])     # deconv 16->8
        ops['final_y'] = G8.model3[6:9](ops['up2_y'])  # reflect 7x7 8->1 + tanh
        x = (torch.rand(2, 1, 32, 32, generator=g) > 0.9).float() * 2 - 1
        ops['land_x'] = x
        ops['land_y'] = G8.model_landmark_trans(x)
    # G4/G5 gathers
    x = rn(2, 8, 32, 32)
    grid = (torch.rand(2, 32, 32, 2, generator=g) * 2.6 - 1.3)
    flow = rn(2, 2, 32, 32) * 12.0
    mask = torch.rand(2, 1, 32, 32, generator=g)
    import torch.nn.functional as F
    ops['gs_x'], ops['gs_grid'], ops['wf_flow'], ops['wf_mask'] = x, grid, flow, mask
    with torch.no_grad():
        ops['gs_y'] = F.grid_sample(x, grid)
        ops['wf_y'] = ifm.warp_acc_flow(x, flow, mask=mask)
        ops['wf_y_nomask'] = ifm.warp_acc_flow(x, flow)
    save('ops_small.npz', **ops)

    # ---------------------------------------------------------------- G6 double_feature_warping
    d = make_generator_inputs(1, seed=77)
    dfw = {'seed': np.int64(77)}
    with torch.no_grad():
        for level, size in ((0, 256), (1, 128), (2, 64)):
            xg = torch.Generator().manual_seed(100 + level)
            x = torch.randn(1, 2, size, size, generator=xg)
            dfw['y%d' % level] = G8.double_feature_warping(x, d['motion'], d['flow'], d['ifmask'], level)
    save('dfw.npz', **dfw)

    # ---------------------------------------------------------------- G7 whole generator ngf=8
    d = make_generator_inputs(2, seed=1234)
    args = generator_args(d)
    out = {'weights_sha256': sha(sd8), 'inputs_sha256': sha({k: d[k] for k in ('input', 'land1', 'land2', 'motion', 'flow', 'ifmask')})}
    G8.zero_grad()
    y = G8(*args)
    out['y_disp3'] = y
    up = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    (y * up).sum().backward()
    gsd = {k: p.grad for k, p in G8.named_parameters()}
    out['grad_norms'] = np.array([float(gsd[k].double().norm()) for k in sd8.keys()])
    for k in ('model_tri00.1.weight', 'model_tri_merge.weight', 'model2.0.conv_block.1.weight', 'model2.0.shortcut.0.weight',
              'model2.4.conv_block.5.weight', 'model3.0.weight', 'model3.3.weight', 'model3.7.weight', 'model3.7.bias',
              'model_landmark_trans.0.weight', 'model_landmark_trans.6.weight', 'model_tri12.0.weight'):
        out['grad_' + k] = gsd[k]
    # fp32 weight gradients of this net are only accurate to ~1e-2 of their scale (cancellation over 131k pixels
    # after InstanceNorm): store the reference evaluated in fp64 as the truth, plus the fp32 reference's own
    # distance to it, which is the accuracy bar for the HIP path.
    G8d = networks.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [], div=3, disp=3)
    G8d.load_state_dict(sd8, strict=True)
    G8d = G8d.double()
    yd = G8d(*[a.double() for a in args])
    (yd * up.double()).sum().backward()
    g64 = {k: p.grad for k, p in G8d.named_parameters()}
    out['grad64_norms'] = np.array([float(g64[k].norm()) for k in sd8.keys()])
    out['grad32_relerr'] = np.array([float((gsd[k].double() - g64[k]).abs().max() / g64[k].abs().max().clamp_min(1e-30))
                                     for k in sd8.keys()])
    for k in list(out.keys()):
        if k.startswith('grad_model'):
            out['grad64_' + k[5:]] = g64[k[5:]].float()
    # disp=1 variant (model default, geomgm_ifw_fore_model.py:167): blocks 2,5,8 are ResnetBlock2
    G8b = networks.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [], div=3, disp=1)
    sd8b = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 1), seed=1234)
    G8b.load_state_dict(sd8b, strict=True)
    with torch.no_grad():
        out['y_disp1'] = G8b(*args)
    save('gen_ngf8.npz', **out)

    # ---------------------------------------------------------------- G8 whole generator ngf=64
    G64 = networks.define_G(3, 1, 64, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [], div=3, disp=3)
    sd64 = og.init_params(og.generator_param_shapes(3, 1, 64, 9, 3, 3), seed=1234)
    G64.load_state_dict(sd64, strict=True)
    with torch.no_grad():
        y2 = G64(*args)
        y1 = G64(*[a[:1] for a in args])
    assert torch.equal(y1, y2[:1]), 'reference G is not sample-independent?'
    save('gen_ngf64.npz', y=y2, weights_sha256=sha(sd64), n_params=np.int64(sum(v.numel() for v in sd64.values())))

    # ---------------------------------------------------------------- G9 PatchGAN
    pg = {}
    for cin in (1, 2):
        D8 = networks.define_D(cin, 8, 'basic', 3, 'instance', 'normal', 0.02, [])
        sdd = og.init_params(od.patchgan_param_shapes(cin, 8), seed=4321 + cin)
        assert list(sdd.keys()) == list(D8.state_dict().keys())
        D8.load_state_dict(sdd, strict=True)
        x = (torch.rand(2, cin, 256, 256, generator=torch.Generator().manual_seed(900 + cin)) * 2 - 1).requires_grad_(True)
        y = D8(x)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
        (y * up).sum().backward()
        pg['y8_c%d' % cin] = y
        pg['dx8_c%d' % cin] = x.grad
        for k, p in D8.named_parameters():
            pg['g8_c%d_%s' % (cin, k)] = p.grad
        D64 = networks.define_D(cin, 64, 'basic', 3, 'instance', 'normal', 0.02, [])
        sdd = og.init_params(od.patchgan_param_shapes(cin, 64), seed=4321 + cin)
        D64.load_state_dict(sdd, strict=True)
        with torch.no_grad():
            pg['y64_c%d' % cin] = D64(x[:1].detach())
    save('patchgan.npz', **pg)

    # ---------------------------------------------------------------- G10/G11 losses + masks
    ls = {}
    crit = networks.GANLoss('lsgan')
    p = rn(2, 1, 30, 30)
    ls['pred'] = p
    ls['gan_real'] = crit(p, True)
    ls['gan_fake'] = crit(p, False)
    a = torch.rand(2, 1, 64, 64, generator=g) * 2 - 1
    m = (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float()
    ls['A'], ls['M'] = a, m

    class _O:
        pass
    from models.base_model import BaseModel
    for mt in range(4):
        o = _O(); o.opt = _O(); o.opt.mask_type = mt
        ls['masked%d' % mt] = BaseModel.masked(o, a, m)
    save('losses.npz', **ls)

    # ---------------------------------------------------------------- G12 TPS (b=1 only in the reference)
    tp = {}
    for tag, size, n in (('s', 32, 12), ('m68', 256, 68), ('m76', 256, 76)):
        tg = torch.Generator().manual_seed(31 + n)
        src = torch.rand(1, n, 2, generator=tg) * (size * 0.8) + size * 0.1
        dst = src + torch.randn(1, n, 2, generator=tg) * (size / 64.0)
        img = torch.rand(1, size, size, 1, generator=tg) * 2 - 1
        img.requires_grad_(True)
        wimg, dflow = siw.sparse_image_warp(img, src, dst)
        up = torch.randn(wimg.shape, generator=tg)
        (wimg * up).sum().backward()
        tp[tag + '_src'], tp[tag + '_dst'] = src, dst
        tp[tag + '_warped'] = wimg
        if size == 32:
            tp[tag + '_img'] = img
            tp[tag + '_up'] = up
            tp[tag + '_dimg'] = img.grad
            tp[tag + '_flow'] = dflow
        else:
            tp[tag + '_seed'] = np.int64(31 + n)
            tp[tag + '_flow_sub'] = dflow[:, ::8, ::8]
            # the fp32 solve of this ill-conditioned system is itself only good to ~0.05 px: store the algorithm
            # evaluated in fp64 (the reference code hard-casts to float32, so this uses the oracle restatement,
            # which test_oracle_golden pins to the reference) and the reference's own distance to it
            from oracle import tps as ot
            w64, f64 = ot.sparse_image_warp(img.detach().double(), src.double(), dst.double())
            tp[tag + '_flow64_sub'] = f64[:, ::8, ::8].float()
            tp[tag + '_warped64'] = w64.float()
            tp[tag + '_ref32_flow_err'] = np.float64((dflow.double() - f64).abs().max())
            tp[tag + '_ref32_warp_err'] = np.float64((wimg.detach().double() - w64).abs().mean())
    save('tps.npz', **tp)

    # ---------------------------------------------------------------- G14 Adam + LambdaLR
    ad = {}
    w = rn(1000).requires_grad_(True)
    ad['w0'] = w.detach().clone()
    opt = torch.optim.Adam([w], lr=5e-5, betas=(0.5, 0.999))
    for it in range(3):
        opt.zero_grad()
        gr = torch.randn(1000, generator=torch.Generator().manual_seed(50 + it))
        w.grad = gr.clone()
        opt.step()
        ad['w%d' % (it + 1)] = w.detach().clone()

    class _Opt:
        lr_policy = 'linear'; epoch_count = 1; niter = 3; niter_decay = 4
    sched = networks.get_scheduler(torch.optim.Adam([w], lr=1.0), _Opt)
    facs = []
    for ep in range(8):
        facs.append(sched.get_last_lr()[0])
        sched.optimizer.step(); sched.step()
    ad['lr_factors'] = np.array(facs)
    save('adam.npz', **ad)

    # ---------------------------------------------------------------- G15 ImagePool
    random.seed(0)
    pool = ImagePool(50)
    seq = []
    for i in range(60):
        r = pool.query(torch.full((1, 1, 1, 1), float(i)))
        seq.append(float(r.item()))
    save('imagepool.npz', returned=np.array(seq))


if __name__ == '__main__':
    main()
