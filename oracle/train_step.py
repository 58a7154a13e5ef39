"""Oracle (test infrastructure): one ``geomgm_ifw_fore`` train step, restated on CPU PyTorch.

Follows Module2/models/geomgm_ifw_fore_model.py: forward :517-565, backward_G :677-780,
backward_D_basic3 :613-635, backward_D_basic2 :589-611, with the batched semantics the build defines
("reference b=1 applied per sample, mean-reduced losses averaged over the batch").  The frozen auxiliary
networks (MODNet / MobileFaceNet / Sphere20a / FlowUnet) are not in the reference tree; their outputs are part
of the batch (``mask``, ``iw_flow``, ``if_mask`` ...).  The geometry (:704-713) and identity (:741-752) terms run
when ``aux`` supplies a landmark regressor / a face feature network (tests pass the fixed-seed stand-ins of
``animateportrait_amd/standins.py`` to both sides), through ``oracle.aux_glue`` -- which
is pinned to the reference's own ``get_lm`` / ``FaceLoss`` (tests/golden/aux.npz); without ``aux`` they are left
out, exactly as the product model does when no aux callable is registered.

``overrides``: the three TPS-warped constants (``mask1``, ``mask2``, ``fakeB_static_warp``) and ``fake_B_warp`` may be
supplied by the caller.  The fp32 spline solve is ill-conditioned (0.1-0.2 px between any two fp32 evaluations, see
tests/test_train_gpu.py::test_tps_vs_reference_golden); the composed-gradient test feeds the product's warps here
so that every other difference is held to fp32-rounding level.

Pinned (round 4) to the reference's OWN model class: tests/golden/make_train_golden.py instantiates
``GeomGMIFWForeModel`` in the build container (cv2 / .cuda() / checkpoint shims, stand-in aux nets, ngf = ndf = 8, b = 1) and
runs its ``set_input / forward / backward_G / backward_D_* / optimize_parameters``; tests/test_train_golden_cpu.py holds this
file to that golden (every loss term, every gradient tensor, three optimiser steps).  The pieces (G, D, GANLoss, masked,
sparse_image_warp) are in addition pinned one by one (tests/golden/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import generator as og, discriminator as od, losses as ol, tps as ot, aux_glue as oa


class Opt:
    """README training values (readme.md:65) on top of the model defaults (:161-209)."""
    use_mask = use_eye_mask = use_lip_mask = 1
    mask_type = 3
    blendbg = 1
    coherent = 1
    coh_use_more = 2
    check_fakeb2_in_backwardD = 1
    warp_loss = 2
    lambda_G_A_l = 0.5
    lambda_G_A_coh = 0.5
    lambda_geom_lipline = 50.0
    lambda_warp = 5.0
    lambda_warp_inter = 10.0
    lambda_geom = 50.0
    more_weight_for_lip = 2
    lambda_face = 3.0
    identity_loss = 2
    crop_size = 256
    div, disp = 3, 3
    thickness = 2


LIP_SEGMENTS = [(i, i + 1) for i in range(48, 59)] + [(59, 48)] + [(i, i + 1) for i in range(60, 67)] + [(67, 60)]


def lipline(lands, size, thickness):
    """getlipline (:507-515): union of the 20 lip segments drawn with cv2.line(thickness), per sample, by OpenCV's
    ThickLine rule as restated in oracle/cv_raster.py (float landmark coordinates truncated to int)."""
    from . import cv_raster
    ln = lands.detach().cpu().double().numpy()
    m = np.stack([cv_raster.lip_line_mask(size, ln[i], LIP_SEGMENTS, thickness) for i in range(ln.shape[0])])
    return torch.from_numpy(m).to(lands.dtype if lands.dtype.is_floating_point else torch.float32).unsqueeze(1)


def warp_nchw(img, src_xy, dst_xy):
    """sparse_image_warp with (x,y) landmarks -> (row,col) as the model passes them (lm[:,:,[1,0]], :537)."""
    w, _ = ot.sparse_image_warp(img.permute(0, 2, 3, 1), src_xy[:, :, [1, 0]], dst_xy[:, :, [1, 0]])
    return w.permute(0, 3, 1, 2)


def forward(sdG, batch, opt=Opt, overrides=None):
    """:517-565 -> dict of tensors (fake_B, fake_B2, local crops, warped static drawing ...)."""
    o = {}
    ov = overrides or {}
    mask = (batch['mask'] > 0.5).to(batch['mask'].dtype)
    real_A_fore = ol.fore_composite(batch['A'], mask)
    gen = lambda tlm, mo, fl, im: og.generator_forward(sdG, real_A_fore, batch['A_lm'], tlm, mo, fl, im,  # noqa: E731
                                                       div=opt.div, disp=opt.disp)
    fake_B = gen(batch['tB_lm'], batch['warp_motion'], batch['iw_flow'], batch['if_mask'])
    fake_B2 = gen(batch['tB2_lm'], batch['warp_motion2'], batch['iw_flow2'], batch['if_mask2'])
    o['fake_B_fore'], o['fake_B2_fore'] = fake_B, fake_B2
    if opt.blendbg:
        mask1 = ov['mask1'] if 'mask1' in ov else warp_nchw(mask, batch['A_lm_68'], batch['tB_lm_68'])
        mask2 = ov['mask2'] if 'mask2' in ov else warp_nchw(mask, batch['A_lm_68'], batch['tB2_lm_68'])
        fake_B = ol.bg_blend(fake_B, batch['fakeB_static'], mask1)
        fake_B2 = ol.bg_blend(fake_B2, batch['fakeB_static'], mask2)
    o['fake_B'], o['fake_B2'] = fake_B, fake_B2
    for suf in ('', 'e', 'l'):
        o['fake_B_l' + suf] = ol.masked(fake_B, batch['B_mask' + suf], opt.mask_type)
        o['fake_B2_l' + suf] = ol.masked(fake_B2, batch['B2_mask' + suf], opt.mask_type)
        o['real_B_l' + suf] = ol.masked(batch['B'], batch['Br_mask' + suf], opt.mask_type)
    # blendbg=1: real_A_lm_681 is the plain 68-point set (no edge points), :534-536 then :558-565
    o['fakeB_static_warp'] = ov['fakeB_static_warp'] if 'fakeB_static_warp' in ov else \
        warp_nchw(batch['fakeB_static'], batch['A_lm_68'], batch['tB_lm_68'])
    return o


def g_loss(sdD, o, batch, opt=Opt, aux=None, overrides=None):
    """:677-780.  sdD: dict name -> discriminator params; aux: {'landmarks': net, 'faceloss': net} (optional)."""
    gan = ol.gan_loss_lsgan
    D = od.patchgan_forward
    terms = {}
    terms['G_A'] = gan(D(sdD['D_A'], o['fake_B']), True) + gan(D(sdD['D_A'], o['fake_B2']), True)
    for name, suf in (('D_A_l', ''), ('D_A_le', 'e'), ('D_A_ll', 'l')):
        terms['G_A_l' + suf] = (gan(D(sdD[name], o['fake_B_l' + suf]), True)
                                + gan(D(sdD[name], o['fake_B2_l' + suf]), True)) * opt.lambda_G_A_l
    terms['G_A_coh'] = gan(D(sdD['D_A_coh'], torch.cat((o['fake_B'], o['fake_B2']), 1)), True) * opt.lambda_G_A_coh
    m1 = lipline(batch['tB_lm_68'], opt.crop_size, opt.thickness)
    m2 = lipline(batch['tB2_lm_68'], opt.crop_size, opt.thickness)
    terms['geom_B_lipline'] = (torch.mean((o['fake_B'] + 1) * m1) + torch.mean((o['fake_B2'] + 1) * m2)) * opt.lambda_geom_lipline
    terms['warp_B'] = F.l1_loss(o['fake_B'], o['fakeB_static_warp']) * opt.lambda_warp
    ov = overrides or {}
    fake_B_warp = ov['fake_B_warp'] if 'fake_B_warp' in ov else \
        warp_nchw(o['fake_B'].detach(), batch['tB_lm_68'], batch['tB2_lm_68'])
    terms['warp_inter1'] = F.l1_loss(o['fake_B2'], fake_B_warp) * opt.lambda_warp_inter
    aux = aux or {}
    cs = opt.crop_size
    if aux.get('landmarks') is not None:                                           # :704-713
        mse = F.mse_loss
        lm1 = oa.get_lm(aux['landmarks'], o['fake_B'], batch['winB'])
        lm2 = oa.get_lm(aux['landmarks'], o['fake_B2'], batch['winB2'])
        t1, t2 = batch['tB_lm_68'][:, :68].to(lm1.dtype), batch['tB2_lm_68'][:, :68].to(lm1.dtype)
        if opt.more_weight_for_lip != 2:
            g = mse(lm1 / cs, t1 / cs) + mse(lm2 / cs, t2 / cs)
        else:
            g = (mse(lm1[:, :48] / cs, t1[:, :48] / cs) + 2 * mse(lm1[:, 48:68] / cs, t1[:, 48:68] / cs)
                 + mse(lm2[:, :48] / cs, t2[:, :48] / cs) + 2 * mse(lm2[:, 48:68] / cs, t2[:, 48:68] / cs))
        terms['geom_B'] = g * opt.lambda_geom
    if aux.get('faceloss') is not None and opt.identity_loss == 2:                 # :748-752
        rep = lambda x: x.repeat(1, 3, 1, 1)                                       # noqa: E731
        terms['iden_B'] = torch.mean(oa.face_loss(aux['faceloss'], rep(o['fake_B']), rep(batch['fakeB_static']),
                                                  batch['winB'], batch['winA'])) * opt.lambda_face
    terms['G'] = sum(terms.values())
    return terms


def d_losses(sdD, o, batch, opt=Opt):
    """backward_D_A / _l / _le / _ll (basic3) and _coh (basic2) with an empty image pool (query returns its input)."""
    gan = ol.gan_loss_lsgan
    D = od.patchgan_forward
    out = {}

    def basic3(name, real, f1, f2):
        return ol.d_loss_basic3(D(sdD[name], real), D(sdD[name], f1.detach()), D(sdD[name], f2.detach()))
    out['D_A'] = basic3('D_A', batch['B'], o['fake_B'], o['fake_B2'])
    for name, suf in (('D_A_l', ''), ('D_A_le', 'e'), ('D_A_ll', 'l')):
        out[name] = basic3(name, o['real_B_l' + suf], o['fake_B_l' + suf], o['fake_B2_l' + suf])
    real = torch.cat((batch['B1'], batch['B2']), 1)
    fake = torch.cat((o['fake_B'], o['fake_B2']), 1).detach()
    other = torch.cat((batch['B3'], batch['B4']), 1)
    out['D_A_coh'] = ol.d_loss_basic2(D(sdD['D_A_coh'], real), D(sdD['D_A_coh'], fake), D(sdD['D_A_coh'], other))
    return out


def set_input_aux(batch, aux):
    """The two frozen-net stages of set_input / forward that produce batch-level tensors (:503-505, :519-520): intrinsic flow
    + visibility mask from netF for both targets, matte from MODNet.  Returns a copy of ``batch`` with ``iw_flow, if_mask,
    iw_flow2, if_mask2, mask`` filled in from ``aux['netF']`` / ``aux['modnet']`` (keys that are absent stay as they were)."""
    b = dict(batch)
    if aux.get('netF') is not None:
        b['iw_flow'], b['if_mask'] = oa.flow_network_warp(aux['netF'], b['A'], b['A_lm_68'][:, :68], b['tB_lm_68'][:, :68])
        b['iw_flow2'], b['if_mask2'] = oa.flow_network_warp(aux['netF'], b['A'], b['A_lm_68'][:, :68], b['tB2_lm_68'][:, :68])
    if aux.get('modnet') is not None:
        with torch.no_grad():
            b['mask'] = aux['modnet'](b['A'], True)[2]
    return b


class TrainState:
    """Leaf tensors + the two Adam optimisers of :346-360 (G; the concatenation of the D parameter lists in model_names order)."""

    def __init__(self, sdG, sdD, lr=5e-5, beta1=0.5):
        self.G = {k: v.clone().requires_grad_(True) for k, v in sdG.items()}
        self.D = {n: {k: v.clone().requires_grad_(True) for k, v in sd.items()} for n, sd in sdD.items()}
        self.opt_G = torch.optim.Adam(list(self.G.values()), lr=lr, betas=(beta1, 0.999))
        self.opt_D = torch.optim.Adam([v for n in ('D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh') if n in self.D
                                       for v in self.D[n].values()], lr=lr, betas=(beta1, 0.999))

    def d_requires_grad(self, flag):
        for sd in self.D.values():
            for v in sd.values():
                v.requires_grad_(flag)


def optimize_parameters(state, batch, opt=Opt, aux=None):
    """:782-819: forward; G step with the D's frozen (zero_grad, backward_G, Adam step); then the D step on the frames of THAT
    forward (detached: generated before G's update), each D loss backpropagated, one Adam step over all D parameters.
    Returns (g terms, d losses) as detached scalars."""
    o = forward(state.G, batch, opt)
    state.d_requires_grad(False)
    state.opt_G.zero_grad()
    terms = g_loss(state.D, o, batch, opt, aux)
    terms['G'].backward()
    state.opt_G.step()
    state.d_requires_grad(True)
    state.opt_D.zero_grad()
    dl = d_losses(state.D, o, batch, opt)
    for v in dl.values():
        v.backward()
    state.opt_D.step()
    return {k: v.detach() for k, v in terms.items()}, {k: v.detach() for k, v in dl.items()}
