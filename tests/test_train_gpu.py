"""GPU (-m gpu): train-step pieces (TPS warp, Adam, full G/D step) against goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import linf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def test_tps_vs_reference_golden(dev, golden):
    from animateportrait_amd.models.sparse_image_warp import sparse_image_warp
    gd = golden('tps.npz')
    w, fl = sparse_image_warp(gd['s_img'].to(dev), gd['s_src'].to(dev), gd['s_dst'].to(dev))
    assert linf(fl, gd['s_flow']) < 5e-3          # fp32 LU of an ill-conditioned (n+3)^2 system
    assert linf(w, gd['s_warped']) < 5e-3
    for tag, n in (('m68', 68), ('m76', 76)):
        tg = torch.Generator().manual_seed(int(gd[tag + '_seed']))
        src = torch.rand(1, n, 2, generator=tg) * (256 * 0.8) + 256 * 0.1
        dst = src + torch.randn(1, n, 2, generator=tg) * 4.0
        img = torch.rand(1, 256, 256, 1, generator=tg) * 2 - 1
        w, fl = sparse_image_warp(img.to(dev), src.to(dev), dst.to(dev))
        # 256x256 / 68-76 points: the fp32 solve is ill-conditioned (the reference's own fp32 result is 0.14-0.23 px
        # away from the fp64 evaluation).  Bar: as close to the fp64 truth as the reference is.
        e_flow = linf(fl[:, ::8, ::8], gd[tag + '_flow64_sub'])
        e_warp = float((w.cpu() - gd[tag + '_warped64']).abs().mean())
        assert e_flow <= 2.0 * float(gd[tag + '_ref32_flow_err']) + 1e-2, (tag, e_flow)
        assert e_warp <= 2.0 * float(gd[tag + '_ref32_warp_err']) + 1e-3, (tag, e_warp)


def test_tps_batched_vs_oracle(dev):
    from animateportrait_amd.models.sparse_image_warp import sparse_image_warp
    from oracle import tps as ot
    tg = torch.Generator().manual_seed(3)
    src = torch.rand(4, 30, 2, generator=tg) * 48 + 8
    dst = src + torch.randn(4, 30, 2, generator=tg)
    img = torch.rand(4, 64, 64, 2, generator=tg)
    w, fl = sparse_image_warp(img.to(dev), src.to(dev), dst.to(dev))
    wo, fo = ot.sparse_image_warp(img, src, dst)
    w64, f64 = ot.sparse_image_warp(img.double(), src.double(), dst.double())
    # accuracy bar = the fp32 CPU evaluation's own distance to the fp64 evaluation (ill-conditioned solve)
    assert linf(fl, f64) <= 2.0 * linf(fo, f64) + 1e-3
    assert linf(w, w64) <= 2.0 * linf(wo, w64) + 1e-3


def test_adam_vs_torch_golden(dev, golden):
    from animateportrait_amd.optim import FlatAdam
    gd = golden('adam.npz')
    w = torch.nn.Parameter(gd['w0'].clone().to(dev))
    opt = FlatAdam([w], lr=5e-5, betas=(0.5, 0.999))
    for it in range(3):
        opt.zero_grad()
        w.grad.copy_(torch.randn(1000, generator=torch.Generator().manual_seed(50 + it)).to(dev))
        opt.step()
        assert linf(w.detach(), gd['w%d' % (it + 1)]) < 2e-7


def _make_model(dev, ngf=8, ndf=8):
    from animateportrait_amd.options.base_options import TrainOptions
    from animateportrait_amd.models import create_model
    argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
            '--output_nc', '1', '--ngf', str(ngf), '--ndf', str(ndf), '--netg_resb_div', '3', '--netg_resb_disp', '3',
            '--lr', '0.00005', '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2',
            '--lambda_face', '3.0', '--lambda_warp_inter', '10', '--blendbg', '1', '--select_target12_thre', '0.0',
            '--niter', '70', '--niter_decay', '0', '--batch_size', '2', '--gpu_ids', '0']   # readme.md:65
    opt = TrainOptions().parse(argv)
    return create_model(opt), opt


def _relerr(a, b64):
    b64 = b64.double()
    return float((a.detach().cpu().double() - b64).abs().max() / b64.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('precision,width,nb', [('bf16x3', 8, 2), ('fp32', 8, 2), ('bf16x3', 64, 1), ('fp32', 64, 1)])
def test_train_step_losses_and_grads_vs_oracle(dev, precision, width, nb, monkeypatch):
    """One G step and one D step of the drawing configuration (ngf=ndf=8, B=2; and at the FULL width of BASELINE
    configs[2], ngf=ndf=64, B=1 -- the CPU oracle's fp64 pass takes ~30 s there) WITH the geometry and identity
    branches (stand-in landmark / face nets): every loss term, and every gradient tensor
    ELEMENTWISE, against the CPU oracle composition evaluated in fp64.  Bar per tensor: as close to the fp64 truth as
    the oracle's own fp32 evaluation is (x3), which is how the reference itself would fare -- weight gradients behind
    InstanceNorm are cancellation-dominated.  The TPS-warped constants are handed from the product to the oracle
    (their own parity: test_tps_*), so every remaining difference is arithmetic of the path under test.
    Two arithmetic modes: exact fp32 MFMA (APAMD_PRECISION=fp32; bar = 3x the oracle's fp32 noise + 5e-5: the
    composition -- tape accumulation over 2B batches and five D's, reflection-pad folds, fused losses -- is exact) and
    the default split-bf16 mode (fp32-class products, ~2^-16 each: + 2e-3 for the generator, the gradient
    counterpart of the 1e-3 L-inf output budget; + 1e-2 for the discriminators' weight gradients: on a mostly-white
    masked crop sum_pix g * a cancels to ~1e-3 of sum |g * a| -- the activation plane is almost constant and the
    InstanceNorm gradient sums to zero -- so the 2^-16 product error is amplified; fp32 mode shows the schedule
    itself is exact)."""
    from animateportrait_amd import ops
    monkeypatch.setattr(ops, 'DEFAULT_PRECISION', ops.PRECISION_FP32 if precision == 'fp32' else ops.PRECISION_BF16X3)
    # Full width: the row-wise flip allowance below exempts up to 1 % of a tensor's rows.  A ReLU flip depends on the data; a
    # wrong row (a mis-indexed tile, a dropped partial sum) does not.  The step therefore runs on TWO batches and no
    # (tensor, row) may be exempt in both.
    exempt = [_train_step_vs_oracle(dev, precision, width, nb, seed) for seed in ((5, 6) if width == 64 else (5,))]
    if len(exempt) == 2:
        both = exempt[0] & exempt[1]
        assert not both, 'rows outside the no-flip bar on two different batches: %s' % sorted(both)[:10]


def _train_step_vs_oracle(dev, precision, width, nb, batch_seed):
    from animateportrait_amd import networks as N, standins, ops
    floor = 5e-5 if precision == 'fp32' else 2e-3
    floor_d = 5e-5 if precision == 'fp32' else 1e-2
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    from oracle import generator as og, discriminator as od, train_step as ts
    torch.manual_seed(0)
    model, opt = _make_model(dev, width, width)
    sdG = og.init_params(og.generator_param_shapes(3, 1, width, 9, 3, 3), seed=11)
    model.netG_A.load_state_dict(sdG, strict=True)
    sdD = {}
    dnames = ['D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh']
    for i, name in enumerate(dnames):
        sdD[name] = og.init_params(od.patchgan_param_shapes(1 if name == 'D_A' else 2, width), seed=20 + i)
        getattr(model, 'net' + name).load_state_dict(sdD[name], strict=True)
    model.aux['landmarks'] = standins.StandinLandmarkNet().to(dev)
    model.aux['faceloss'] = N.FaceLoss(standins.StandinFaceNet().to(dev))
    batch = make_train_batch(nb, seed=batch_seed)
    batch['winB'] = torch.tensor([[32, 224, 32, 224], [-12, 200, 24, 230]])[-nb:]  # one window leaves the frame
    batch['winB2'] = torch.tensor([[30, 226, 28, 220], [40, 260, 50, 256]])[-nb:]
    batch['winA'] = torch.tensor([[36, 220, 30, 210], [20, 230, 20, 228]])[-nb:]
    # ---------------- product
    model.set_input(batch)
    model.forward()
    nets_D = [getattr(model, 'net' + n) for n in model.model_names[1:]]
    model.set_requires_grad(nets_D, False)
    model.optimizer_G.zero_grad()
    model.backward_G()
    gG = {k: p.grad.detach().clone().cpu() for k, p in model.netG_A.named_parameters()}
    model.set_requires_grad(nets_D, True)
    model.optimizer_D.zero_grad()
    model.backward_D_A(); model.backward_D_A_l(); model.backward_D_A_le(); model.backward_D_A_ll(); model.backward_D_A_coh()
    gD = {n: {k: p.grad.detach().clone().cpu() for k, p in getattr(model, 'net' + n).named_parameters()} for n in dnames}
    from animateportrait_amd.models.sparse_image_warp import check_status
    check_status()
    ov32 = {k: getattr(model, k).detach().cpu() for k in ('mask1', 'mask2', 'fakeB_static_warp', 'fake_B_warp')}
    # the D step is checked on the product's own generated frames (their parity is asserted above at 1e-3; a 1e-4
    # input difference would otherwise dominate D gradients whose fp32 noise is 1e-6)
    fakes32 = {k: getattr(model, k).detach().cpu() for k in ('fake_B', 'fake_B2', 'fake_B_l', 'fake_B2_l', 'fake_B_le',
                                                              'fake_B2_le', 'fake_B_ll', 'fake_B2_ll')}
    # ---------------- oracle, fp32 (what the reference's arithmetic gives) and fp64 (the truth)
    res = {}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        cast = lambda t: t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t     # noqa: E731
        sG = {k: cast(v).clone().requires_grad_(True) for k, v in sdG.items()}
        sD = {n: {k: cast(v).clone() for k, v in sd.items()} for n, sd in sdD.items()}
        b = {k: cast(v) for k, v in batch.items()}
        ov = {k: cast(v) for k, v in ov32.items()}
        aux = {'landmarks': standins.StandinLandmarkNet().to(dt), 'faceloss': standins.StandinFaceNet().to(dt)}
        o = ts.forward(sG, b, overrides=ov)
        terms = ts.g_loss(sD, o, b, aux=aux, overrides=ov)
        terms['G'].backward()
        for sd in sD.values():
            for v in sd.values():
                v.requires_grad_(True)
        od_ = dict(o)
        od_.update({k: cast(v) for k, v in fakes32.items()})
        dl = ts.d_losses(sD, od_, b)
        sum(dl.values()).backward()
        res[tag] = dict(o=o, terms=terms, dl=dl, gG={k: v.grad for k, v in sG.items()},
                        gD={n: {k: v.grad for k, v in sd.items()} for n, sd in sD.items()})
    r32, r64 = res['f32'], res['f64']
    # ---------------- forward tensors and loss terms
    assert linf(model.fake_B_fore, r64['o']['fake_B_fore']) < 1e-3 and linf(model.fake_B2_fore, r64['o']['fake_B2_fore']) < 1e-3
    assert linf(model.fake_B, r64['o']['fake_B']) < 1e-3 and linf(model.fake_B2, r64['o']['fake_B2']) < 1e-3
    assert linf(model.fake_B_l, r64['o']['fake_B_l']) < 1e-3 and linf(model.real_B_ll, r64['o']['real_B_ll']) < 1e-6
    for k in ('G_A', 'G_A_l', 'G_A_le', 'G_A_ll', 'G_A_coh', 'geom_B', 'geom_B_lipline', 'warp_B', 'warp_inter1',
              'iden_B', 'G'):
        a, t = float(getattr(model, 'loss_' + k)), float(r64['terms'][k])
        assert abs(a - t) <= 1e-3 * abs(t) + 1e-5, (k, a, t)
    for name in dnames:
        a, t = float(getattr(model, 'loss_' + name)), float(r64['dl'][name])
        assert abs(a - t) <= 1e-3 * abs(t) + 1e-6, (name, a, t)
    # ---------------- gradients, elementwise, every tensor
    bad, all_, exempt_rows = [], [], set()
    # Full width: 40 M activations sit in front of a (leaky) ReLU, so a handful of them are within rounding of zero
    # and take the other branch in ANY two evaluations (the oracle's own fp32 pass against its fp64 pass shows ~1e-2
    # L-inf on the generator for that reason; measured on D_A_coh: one flipped pixel in one 31x31 plane moves that
    # output channel's weight-gradient row by 3e-2 of the tensor maximum, every other row by 2e-6, and the layers below
    # it by a diffuse 5e-4; round 3 saw 6e-2 on one row of D_A_ll's 128 -> 256 layer, median row error 3e-6).  Tensors with
    # >= 32 output-channel rows are therefore checked ROW BY ROW: all rows but 1 % (>= 1) hold HALF the flip allowance (the
    # diffuse part), the exempt rows stay below 0.15, and so does the per-row median.  Small tensors keep the L-inf allowance.
    flip = 2e-2 if width == 64 else 0.0

    def rowmed(a, b64):
        if a.dim() < 2:
            return _relerr(a, b64)
        d = (a.detach().cpu().double() - b64.double()).abs().flatten(1).amax(1)
        return float(d.median() / b64.double().abs().max().clamp_min(1e-30))

    def check(tag, k, mine, g32, g64):
        if float(g64.abs().max()) < 1e-12 or (k.endswith('.bias') and float(mine.abs().max()) == 0.0):
            # bias in front of InstanceNorm: exact zero here, pure rounding noise in the reference
            assert float(g64.abs().max()) < 1e-6 * max(1.0, float(r64['gG']['model_tri_merge.weight'].abs().max())), (tag, k)
            return
        e, noise = _relerr(mine, g64), _relerr(g32, g64)
        me, mnoise = rowmed(mine, g64), rowmed(g32, g64)
        all_.append((tag, k, round(e, 6), round(noise, 6), round(me, 6), round(mnoise, 6)))
        fl = floor if tag == 'G' else floor_d
        if flip and mine.dim() >= 2 and mine.shape[0] >= 32:
            # full width, a tensor with many output-channel rows: a flipped (leaky) ReLU pixel hits the rows of the planes it
            # sits in and nothing else -- so instead of loosening L-inf for the whole tensor, every row must hold the
            # no-flip bar except at most 1 % of them (>= 1), and those stay below the size of a few flips
            d = (mine.detach().cpu().double() - g64.double()).abs().flatten(1).amax(1) / g64.double().abs().max().clamp_min(1e-30)
            over = d > 3.0 * noise + fl + 0.5 * flip
            exempt_rows.update((tag, k, int(r)) for r in over.nonzero().flatten().tolist())
            if int(over.sum()) > max(1, mine.shape[0] // 100) or float(d.max()) > 0.15 or me > 3.0 * mnoise + fl + 0.5 * flip:
                bad.append((tag, k, e, noise, me, mnoise, int(over.sum())))
        elif e > 3.0 * noise + fl + flip or me > 3.0 * mnoise + fl + 0.5 * flip:
            bad.append((tag, k, e, noise, me, mnoise))
    for k in sdG:
        check('G', k, gG[k], r32['gG'][k], r64['gG'][k])
    for n in dnames:
        for k in sdD[n]:
            check(n, k, gD[n][k], r32['gD'][n][k], r64['gD'][n][k])
    if os.environ.get('APAMD_TEST_DUMP'):
        open(os.environ['APAMD_TEST_DUMP'], 'a').write('%s %d seed %d %r\n' % (precision, width, batch_seed, all_))
    assert not bad, bad
    return exempt_rows


def test_optimize_parameters_runs_and_updates(dev):
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    model, opt = _make_model(dev)
    w0 = model.netG_A.model_tri_merge.weight.detach().clone()
    d0 = model.netD_A_coh.model['8'].weight.detach().clone()
    for it in range(2):
        model.set_input(make_train_batch(2, seed=40 + it))
        model.optimize_parameters()
    losses = model.get_current_losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    assert float((model.netG_A.model_tri_merge.weight - w0).abs().max()) > 0
    assert float((model.netD_A_coh.model['8'].weight - d0).abs().max()) > 0
    # parameters moved by about lr per Adam step
    assert float((model.netG_A.model_tri_merge.weight - w0).abs().max()) < 3 * 2 * 5e-5


def test_checkpoint_roundtrip(dev, tmp_path):
    model, opt = _make_model(dev)
    model.save_dir = str(tmp_path)
    model.save_networks('latest')
    sd = torch.load(str(tmp_path / 'latest_net_G_A.pth'))
    assert list(sd.keys())[0] == 'model_tri_merge.weight' and all(v.device.type == 'cpu' for v in sd.values())
    w = model.netG_A.model3['7'].weight.detach().clone()
    with torch.no_grad():
        model.netG_A.model3['7'].weight.add_(1.0)
    model.load_networks('latest')
    assert torch.equal(model.netG_A.model3['7'].weight, w)


# ------------------------------------------------------------------ data-parallel equivalence, entry points, guards
def _flat_grads(model):
    return model.optimizer_G.flat_grad.clone(), model.optimizer_D.flat_grad.clone()


def _backward_both(model, batch):
    """The gradient half of optimize_parameters (:782-819) without the optimiser steps."""
    model.set_input(batch)
    model.forward()
    nets_D = [getattr(model, 'net' + n) for n in model.model_names[1:]]
    model.set_requires_grad(nets_D, False)
    model.optimizer_G.zero_grad()
    model.backward_G()
    model.set_requires_grad(nets_D, True)
    model.optimizer_D.zero_grad()
    model.backward_D_A(); model.backward_D_A_l(); model.backward_D_A_le(); model.backward_D_A_ll(); model.backward_D_A_coh()
    return _flat_grads(model)


def test_dp_equivalence_shard_grads_average_to_big_batch_grads(dev):
    """SURVEY.md section 4 / VERDICT r1 #1b: the data-parallel step is exact.  backward of a B=4 batch == the mean of
    the backward passes of its two B=2 shards (what the two ranks of a DP run all-reduce), for G and all five D's.
    The image pool of D_A_coh is empty (< pool_size images seen), so query() returns its input on every call."""
    from animateportrait_amd import parallel
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    torch.manual_seed(3)
    model, opt = _make_model(dev)
    batch = make_train_batch(4, seed=77)
    gG, gD = _backward_both(model, batch)
    accG, accD = torch.zeros_like(gG), torch.zeros_like(gD)
    for r in range(2):
        a, b = _backward_both(model, parallel.shard_batch(batch, r, 2))
        accG += a / 2
        accD += b / 2
    for name, big, avg, opt_ in (('G', gG, accG, model.optimizer_G), ('D', gD, accD, model.optimizer_D)):
        assert float(big.abs().max()) > 0
        off = 0
        for p in opt_._params:
            k = p.numel()
            x, y = big[off:off + k].double(), avg[off:off + k].double()
            off += k
            scale = float(x.abs().max())
            if scale == 0.0:            # biases in front of InstanceNorm: exact zeros on both sides
                assert float(y.abs().max()) == 0.0
                continue
            # same per-sample arithmetic, different summation order over the batch: fp32 rounding only
            assert float((x - y).abs().max()) <= 2e-4 * scale + 1e-9, (name, tuple(p.shape), float((x - y).abs().max()), scale)
        rel = float((big.double() - avg.double()).norm() / big.double().norm())
        assert rel < 5e-5, (name, rel)


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
def test_train_step_at_the_reported_size_equals_mean_of_single_sample_steps(dev, precision, monkeypatch):
    """VERDICT r4 #5a: the train step at the size bench.py reports -- ngf = ndf = 64, B = 16 (the generator runs 2B = 32 images:
    1024 tiles = four rounds of the persistent grid, the weight gradients split 32 images, the persistent InstanceNorm backward
    walks several items per CU) -- in both arithmetic modes.  No oracle needed: InstanceNorm keeps samples independent, so the
    flat G and D gradients of the B = 16 backward are the mean of the sixteen B = 1 backward passes (each of which is what
    test_train_step_losses_and_grads_vs_oracle pins at this width), and so are the loss values.  Which kernels ran at B = 16 is
    read off the launch profiler.

    The bar.  B = 1, 2 and 4 take the same kernels and agree to 1e-6 (tools/batch_grad_check.py); from 2B = 8 images on the 3x3
    layers take the 16-row tile, whose InstanceNorm partial sums are formed over other pixel sets: statistics differ by ~1e-7,
    the split-bf16 representation of an activation (head + tail, 2^-17) then rounds differently, and the backward through
    InstanceNorm amplifies that rounding ~300x (sums that cancel to 1e-3 of their terms) -- every split-bf16 evaluation of these
    gradients carries that noise, 1-2e-3 per layer against exact fp32 (tools/layer_batch_check.py).  So the reference is the
    EXACT gradient (the same model in fp32 arithmetic, mean of the sixteen single-sample passes: that mode is batch-independent to
    5e-6, tools/batch_net_check.py) and the claim is: the B = 16 evaluation is no further from it than the single-sample
    evaluations in the same arithmetic are (x 1.5).  A wrong plan at B = 16 -- a dropped round of tiles, a mis-split image range --
    is O(1)."""
    from animateportrait_amd import ops, parallel
    from animateportrait_amd.options.base_options import TrainOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    B = 16
    batch = make_train_batch(B, seed=1234)

    def build(prec):
        monkeypatch.setattr(ops, 'DEFAULT_PRECISION', {'bf16': ops.PRECISION_BF16, 'bf16x3': ops.PRECISION_BF16X3, 'fp32': ops.PRECISION_FP32}[prec])
        argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
                '--output_nc', '1', '--ngf', '64', '--ndf', '64', '--netg_resb_div', '3', '--netg_resb_disp', '3',
                '--lr', '0.00005', '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2',
                '--lambda_face', '3.0', '--lambda_warp_inter', '10', '--blendbg', '1', '--select_target12_thre', '0.0',
                '--niter', '70', '--niter_decay', '0', '--batch_size', '16', '--gpu_ids', '0', '--precision', prec]
        torch.manual_seed(1234)
        return create_model(TrainOptions().parse(argv))

    def mean_of_singles(model):
        accG = accD = None
        losses = {}
        for r in range(B):
            model.fake_B_pool = type(model.fake_B_pool)(model.opt.pool_size)      # (an empty pool returns its input: no history)
            a, b = _backward_both(model, parallel.shard_batch(batch, r, B))
            accG = a.double() / B if accG is None else accG + a.double() / B
            accD = b.double() / B if accD is None else accD + b.double() / B
            for k, v in model.get_current_losses().items():
                losses[k] = losses.get(k, 0.0) + v / B
        return accG, accD, losses

    exact = build('fp32')
    refG, refD, _ = mean_of_singles(exact)
    w_exact = exact.optimizer_G.flat.clone()
    del exact
    torch.cuda.empty_cache()
    model = build(precision)
    assert torch.equal(model.optimizer_G.flat, w_exact)                 # same seed, same initialisation
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    try:
        gG, gD = _backward_both(model, batch)
    finally:
        ops.PROFILER = None
    names = {r[0].replace(' ', '') for r in prof.records}
    big_losses = dict(model.get_current_losses())
    tall = 'Bf3Cfg<1,3,1,2,4,4>' + ('bf16' if precision == 'bf16' else '')
    assert tall in names, sorted(names)                               # the 16-row tile of the 3x3 stride-1 kernel
    # the persistent InstanceNorm backward (1024 threads per 64 x 64 item) and the 3x3 weight gradient on its prepared operand
    # (split-bf16: the weight gradient reads both operands as split copies, ap_conv2d_wgrad_xs)
    wg = prof.calls.get('wgrad_bf16x3<3> (prepared operand)', 0) + prof.calls.get('wgrad_xs<3> (split copies)', 0)
    assert prof.calls.get('instnorm_bwd_split<1024>', 0) >= 20 and wg >= 20, prof.calls
    accG, accD, acc_losses = mean_of_singles(model)
    ltol = 2e-4 if precision == 'bf16x3' else 3e-3       # (plain bf16: an activation that rounds to the other bf16 value moves by 2^-9)
    for k, v in big_losses.items():
        assert abs(v - acc_losses[k]) <= ltol * abs(acc_losses[k]) + 1e-6, (k, v, acc_losses[k])
    rows = []
    for name, big, one, ref, opt_ in (('G', gG, accG, refG, model.optimizer_G), ('D', gD, accD, refD, model.optimizer_D)):
        assert float(big.abs().max()) > 0
        off = 0
        for p in opt_._params:
            k = p.numel()
            x, y, r = big[off:off + k].double(), one[off:off + k], ref[off:off + k]
            off += k
            if float(r.abs().max()) == 0.0:            # biases in front of InstanceNorm: exact zeros everywhere
                assert float(x.abs().max()) == 0.0
                continue
            e16, e1 = float((x - r).norm() / r.norm()), float((y - r).norm() / r.norm())
            if k < 8:
                # a one-element tensor (the last layer's bias: a sum of 2 M gradient values that cancels to a few percent) has no
                # norm to average over: both errors are single draws of the same ~2e-2 noise (measured with the PatchGAN's first
                # layer on either kernel: B = 16 2.1e-2 / 2.3e-2, mean of singles 1.4e-3 / 1.1e-2) -- it is held to that level,
                # not to a ratio of two draws
                assert e16 <= max(3.0 * e1, 6e-2), (name, tuple(p.shape), e16, e1)
                continue
            rows.append((e16 / max(e1, 1e-4), e16, e1, name, tuple(p.shape)))
        e16, e1 = float((big.double() - ref).norm() / ref.norm()), float((one - ref).norm() / ref.norm())
        print('%s %s: distance from the exact gradient: B=16 %.2e, mean of 16 x B=1 %.2e' % (precision, name, e16, e1))
        assert e16 <= 1.5 * max(e1, 1e-4), (name, e16, e1)
    rows.sort(reverse=True)
    for r in rows[:5]:
        print('   worst tensors: %.2f x (B=16 %.2e, B=1 %.2e)  %s %s' % r)
    assert rows[0][0] <= 3.0, rows[0]                               # single tensors scatter more than the whole gradient


_ENTRY_FLAGS = ['--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic', '--output_nc', '1',
                '--ngf', '8', '--ndf', '8', '--netg_resb_div', '3', '--netg_resb_disp', '3', '--gpu_ids', '0']


def test_train_and_test_entry_points_end_to_end(dev, tmp_path, capsys):
    """Module2/train.py:7-64 and test.py:38-66 counterparts, run as the CLI runs them: two epochs of two synthetic
    batches, checkpoints written with the reference's file names, then inference from the saved generator."""
    from animateportrait_amd import train, test
    ck, res = str(tmp_path / 'ck'), str(tmp_path / 'res')
    train.main(['--model', 'geomgm_ifw_fore', '--name', 'e2e_drawing', '--checkpoints_dir', ck, '--batch_size', '2',
                '--synthetic_batches', '2', '--niter', '2', '--niter_decay', '0', '--save_epoch_freq', '1',
                '--print_freq', '2', '--lr', '0.00005', '--lambda_geom_lipline', '50', '--lambda_warp_inter', '10',
                '--blendbg', '1'] + _ENTRY_FLAGS)
    out = capsys.readouterr().out
    assert 'End of epoch 2 / 2' in out and 'G_A:' in out and 'nan' not in out.lower()
    import os
    for name in ('G_A', 'D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh'):
        for ep in ('latest', '1', '2'):
            assert os.path.exists(os.path.join(ck, 'e2e_drawing', '%s_net_%s.pth' % (ep, name))), (ep, name)
    # a mistyped checkpoint name is an error (test.py:48 -> load_networks), not a silent random-weights run
    with pytest.raises(FileNotFoundError):
        test.main(['--model', 'geomgm_ifw_fore', '--name', 'no_such_run', '--checkpoints_dir', ck, '--results_dir', res,
                   '--batch_size', '2', '--synthetic_batches', '1'] + _ENTRY_FLAGS)
    test.main(['--model', 'geomgm_ifw_fore', '--name', 'e2e_drawing', '--checkpoints_dir', ck, '--results_dir', res,
               '--batch_size', '2', '--synthetic_batches', '2', '--num_test', '3', '--epoch', '2'] + _ENTRY_FLAGS)
    frames = sorted(os.listdir(os.path.join(res, 'e2e_drawing', 'test_2', 'images')))
    assert len(frames) >= 3 and all(f.endswith('_fake_B.npy') for f in frames)
    y = np.load(os.path.join(res, 'e2e_drawing', 'test_2', 'images', frames[0]))
    assert y.shape == (1, 256, 256) and np.isfinite(y).all() and np.abs(y).max() <= 1.0
    # the saved generator reproduces the frame: same weights -> same output as the CLI wrote
    from animateportrait_amd import networks
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    G = networks.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [0],
                          div=3, disp=3)
    G.load_state_dict(torch.load(os.path.join(ck, 'e2e_drawing', '2_net_G_A.pth')), strict=True)
    b = make_train_batch(2, seed=1234)
    mask = (b['mask'] > 0.5).float()
    fore = ((b['A'] / 2 + 0.5) * mask + 1 - mask) * 2 - 1
    with torch.no_grad():
        ref = G(*[t.to(dev).contiguous() for t in (fore, b['A_lm'], b['tB_lm'], b['warp_motion'], b['iw_flow'], b['if_mask'])])
    assert linf(ref[0], torch.from_numpy(y)) < 1e-6


def test_streaming_test_model_entry_point_requires_static_checkpoint(dev, tmp_path):
    """geomcgt_ifw_test (main_end2end_module2.py:96-97): without checkpoints/static/drawing.pth setup() fails as the
    reference's __init__ does (:226), unless --allow_random_init asks for the smoke mode."""
    from animateportrait_amd.options.base_options import TestOptions
    from animateportrait_amd.models import create_model
    argv = ['--model', 'geomcgt_ifw_test', '--name', 'x_drawing', '--checkpoints_dir', str(tmp_path),
            '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--output_nc', '1', '--ngf', '8', '--gpu_ids', '0']
    model = create_model(TestOptions().parse(argv))
    with pytest.raises(FileNotFoundError, match='drawing.pth'):
        model.setup(model.opt)


def test_singular_tps_system_is_reported(dev):
    """ADVICE r1 / sparse_image_warp.py:124-128: duplicate control points make the spline system singular; the
    reference fails loudly, ap_tps_solve flags it -- the wrapper must surface the flag at the next sync point."""
    from animateportrait_amd.models import sparse_image_warp as siw
    siw.check_status()
    src = torch.tensor([[[10., 10.], [10., 10.], [30., 40.], [50., 20.]]], device=dev)      # two coincident points
    dst = src + 1.0
    img = torch.rand(1, 64, 64, 1, device=dev)
    siw.sparse_image_warp(img, src, dst)
    with pytest.raises(RuntimeError, match='singular'):
        siw.check_status()
    siw.check_status()                      # the flag is consumed
    tg = torch.Generator().manual_seed(1)
    good = torch.rand(1, 20, 2, generator=tg).to(dev) * 48 + 8
    siw.sparse_image_warp(img, good, good + 0.5)
    siw.check_status()                      # a regular system raises nothing


def test_autograd_guards_and_generator_input_gradient(dev):
    """ADVICE r1 (autograd.py): (a) d(out)/d(input) of the generator is defined and matches the oracle's autograd;
    (b) an optimiser step between forward and backward raises instead of giving gradients of other weights;
    (c) a second backward raises a clear error; (d) inputs without a defined gradient are refused."""
    from animateportrait_amd import networks, ops
    from animateportrait_amd.synthetic import make_generator_inputs, generator_args
    from oracle import generator as og
    sd = og.init_params(og.generator_param_shapes(3, 1, 8, 9, 3, 3), seed=5)
    G = networks.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [0],
                          div=3, disp=3)
    G.load_state_dict(sd, strict=True)
    d = make_generator_inputs(1, seed=21)
    args = [t.to(dev) for t in generator_args(d)]
    gout = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(2))
    # (a)
    x = args[0].clone().requires_grad_(True)
    y = G(x, *args[1:])
    y.backward(gout.to(dev))
    xr = d['input'].clone().double().requires_grad_(True)
    sd64 = {k: v.double() for k, v in sd.items()}
    yr = og.generator_forward(sd64, xr, *[t.double() for t in generator_args(d)[1:]], div=3, disp=3)
    yr.backward(gout.double())
    x32 = d['input'].clone().requires_grad_(True)
    og.generator_forward(sd, x32, *generator_args(d)[1:], div=3, disp=3).backward(gout)
    ref_err = linf(x32.grad, xr.grad)
    scale = float(xr.grad.abs().max())
    assert x.grad is not None and linf(x.grad, xr.grad) <= max(3.0 * ref_err, 2e-3 * scale), (linf(x.grad, xr.grad), ref_err, scale)
    # (c)
    with pytest.raises(RuntimeError, match='second time'):
        y.backward(gout.to(dev))
    # (b)
    y2 = G(*args)
    with torch.no_grad():
        G.model_tri_merge.weight.mul_(1.0)          # in-place write bumps the version
    with pytest.raises(RuntimeError, match='modified'):
        y2.backward(gout.to(dev))
    y3 = G(*args)
    ops.WEIGHTS_EPOCH += 1                          # what FlatAdam.step does
    with pytest.raises(RuntimeError, match='modified'):
        y3.backward(gout.to(dev))
    # (d)
    with pytest.raises(NotImplementedError, match='flow'):
        G(args[0], args[1], args[2], args[3], args[4].clone().requires_grad_(True), args[5])


def test_gradient_block_path_respects_autograd_contract(dev):
    """the direct path of _NetFn.backward (parameter gradients written into the optimiser's flat buffer) is taken
    only when autograd asked for every trainable parameter AND every p.grad still is its flat view.
    * torch.autograd.grad(loss, [input]) must not touch the optimiser's gradient buffer;
    * torch.autograd.grad(loss, params) returns real tensors;
    * after module.zero_grad() (p.grad = None) stale sums in the flat buffer must not be added to."""
    from animateportrait_amd import networks
    from animateportrait_amd.optim import FlatAdam
    torch.manual_seed(2)
    D = networks.define_D(1, 8, 'basic', 3, 'instance', 'normal', 0.02, [0])
    opt = FlatAdam(D.parameters(), lr=1e-3)
    x = torch.randn(2, 1, 64, 64, device=dev, requires_grad=True)
    # reference gradients through the train step's flow (the model wraps its .backward() calls the same way)
    from animateportrait_amd.autograd import direct_param_grads
    opt.zero_grad()
    with direct_param_grads():
        D(x).square().mean().backward()
    ref = opt.flat_grad.clone()
    # ... and without the opt-in the same numbers arrive through autograd's own accumulation
    opt.zero_grad()
    x.grad = None
    D(x).square().mean().backward()
    assert torch.allclose(opt.flat_grad, ref, rtol=1e-5, atol=1e-7 * float(ref.abs().max()))
    ref_x = x.grad.clone()
    assert float(ref.abs().max()) > 0
    # (1) gradient w.r.t. the input only: the flat buffer stays as it is
    opt.zero_grad()
    (gx,) = torch.autograd.grad(D(x).square().mean(), [x])
    assert float(opt.flat_grad.abs().max()) == 0.0 and torch.allclose(gx, ref_x, rtol=1e-5, atol=1e-8)
    # (2) gradients w.r.t. the parameters as returned tensors
    params = [p for p in D.parameters()]
    gs = torch.autograd.grad(D(x).square().mean(), params)
    assert all(g is not None for g in gs) and float(opt.flat_grad.abs().max()) == 0.0
    flat = torch.cat([g.reshape(-1) for g in gs])
    assert torch.allclose(flat, ref, rtol=1e-4, atol=1e-7 * float(ref.abs().max()))
    # (3) module.zero_grad() drops the views; stale contents of the flat buffer must not leak into the next backward
    opt.flat_grad.fill_(7.0)
    D.zero_grad()                                         # set_to_none
    assert all(p.grad is None for p in D.parameters())
    with direct_param_grads():                            # even when opted in: p.grad is no longer the flat view
        D(x).square().mean().backward()
    got = torch.cat([p.grad.reshape(-1) for p in D.parameters()])
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-7 * float(ref.abs().max()))
    opt.step()                                            # _rebind copies the fresh gradients into the flat buffer
    assert all(p.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * p._flat_off for p in D.parameters())


# ------------------------------------------------------------------------------------------------------------------
# The product model against the golden made by the REFERENCE's own model class (tests/golden/make_train_golden.py)
def _golden_model(dev, gd, lr=None):
    from test_train_golden_cpu import golden_setup, DNAMES
    from animateportrait_amd import networks as N, standins
    batch, sdG, sdD, aux, Opt = golden_setup(gd, torch.float32)
    model, opt = _make_model(dev, int(gd['width']), int(gd['width']))
    if lr is not None:
        for o in model.optimizers:
            for g in o.param_groups:
                g['lr'] = lr
    model.netG_A.load_state_dict(sdG, strict=True)
    for n in DNAMES:
        getattr(model, 'net' + n).load_state_dict(sdD[n], strict=True)
    model.aux['landmarks'] = standins.StandinLandmarkNet().to(dev)
    model.aux['faceloss'] = N.FaceLoss(standins.StandinFaceNet().to(dev))
    return model, batch, aux


def test_aux_stages_of_set_input_vs_reference_class_golden(dev, golden):
    """set_input's frozen-net stages on the device (netF pre / post through ap_kp_to_map / ap_flow_post, MODNet matte
    threshold) against what the reference class computed (:503-505, :519-520).  argmax / > 0.5 are discontinuous: a few
    pixels may take the other side."""
    from animateportrait_amd import standins
    gd = golden('train_step.npz')
    model, batch, _ = _golden_model(dev, gd)
    model.aux['netF'] = standins.StandinFlowNet().to(dev)
    model.aux['modnet'] = standins.StandinMatteNet().to(dev)
    model.set_input(batch)
    for mine, key in ((model.iw_flow, 'iw_flow'), (model.iw_flow2, 'iw_flow2'), (model.real_A_if_mask, 'if_mask'),
                      (model.real_A_if_mask2, 'if_mask2')):
        d = (mine.cpu()[..., ::2, ::2] - gd[key + '_sub2']).abs() / float(gd[key + '_sub2'].abs().max())
        assert float((d > 1e-4).float().mean()) < 2e-3, (key, float(d.max()), float((d > 1e-4).float().mean()))
    assert float((model.mask.cpu()[..., ::2, ::2] != gd['mask_sub2']).float().mean()) < 1e-3


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_train_step_vs_reference_class_golden(dev, golden, precision, monkeypatch):
    """forward / backward_G / backward_D_* of the product model == the reference class's own methods (golden: the reference in
    fp64, with the distance of its fp32 evaluation as the per-tensor noise bar): the frames, all eleven G loss values, the five D
    losses, and EVERY gradient tensor elementwise.  The four TPS-warped constants are replayed from the golden (the fp32 spline
    solve is ill-conditioned; the product's own warps are checked against the golden's to the bar of test_tps_*), the
    discontinuous aux stages (previous test) are evaluated on the host."""
    from animateportrait_amd import ops
    from oracle import train_step as ts
    from test_train_golden_cpu import G_TERMS, DNAMES
    import animateportrait_amd.models.geomgm_ifw_fore_model as pm
    monkeypatch.setattr(ops, 'DEFAULT_PRECISION', ops.PRECISION_FP32 if precision == 'fp32' else ops.PRECISION_BF16X3)
    floor = 5e-5 if precision == 'fp32' else 2e-3
    floor_d = 5e-5 if precision == 'fp32' else 1e-2
    gd = golden('train_step.npz')
    model, batch, aux = _golden_model(dev, gd)
    batch = ts.set_input_aux({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()},
                             {k: (v.double() if k in ('netF', 'modnet') else None) for k, v in aux.items()})
    batch = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    replay = [torch.cat([gd['mask1'], gd['mask2']], 0), gd['fakeB_static_warp'], gd['fake_B_warp']]
    own, real_warp = [], pm.warp_nchw

    def warp(img, src, dst):
        w = real_warp(img, src, dst)
        ref = replay[len(own)].to(w.device)
        own.append(float((w - ref).abs().mean()))
        return ref
    monkeypatch.setattr(pm, 'warp_nchw', warp)
    model.set_input(batch)
    model.forward()
    for k in ('fake_B_fore', 'fake_B2_fore', 'fake_B', 'fake_B2'):
        assert linf(getattr(model, k), gd[k]) < 1e-3, k
    for k in ('fake_B_l', 'fake_B2_l', 'real_B_l', 'fake_B_le', 'real_B_le', 'fake_B2_ll', 'real_B_ll'):
        assert linf(getattr(model, k)[..., ::2, ::2], gd[k + '_sub2']) < 1e-3, k
    nets_D = [getattr(model, 'net' + n) for n in DNAMES]
    model.set_requires_grad(nets_D, False)
    model.optimizer_G.zero_grad()
    model.backward_G()
    assert len(own) == 3 and max(own[:2]) < 2e-3 and own[2] < 2e-2, own     # mean |product TPS warp - reference fp64 warp|
    assert float((model.getlipline(model.target_B_lm_68).cpu()[..., ::2, ::2] != gd['liplinemask1_sub2']).float().sum()) == 0
    for k in G_TERMS:
        a, t = float(getattr(model, 'loss_' + k)), float(gd['loss_' + k])
        assert abs(a - t) <= 1e-3 * abs(t) + 1e-5, (k, a, t)
    bad = []

    def check(tag, mine, key, fl):
        g64 = gd[key]
        if key.endswith('.bias') and float(mine.abs().max()) == 0.0:
            assert float(g64.abs().max()) < 1e-6, key         # bias in front of InstanceNorm: exact zero here, noise there
            return
        e = _relerr(mine, g64)
        if e > 3.0 * float(gd[key + '_noise']) + fl:
            bad.append((tag, key, e, float(gd[key + '_noise'])))
    for k, p in model.netG_A.named_parameters():
        check('G', p.grad, 'gG/' + k, floor)
    model.set_requires_grad(nets_D, True)
    model.optimizer_D.zero_grad()
    model.backward_D_A(); model.backward_D_A_l(); model.backward_D_A_le(); model.backward_D_A_ll(); model.backward_D_A_coh()
    for n in DNAMES:
        a, t = float(getattr(model, 'loss_' + n)), float(gd['loss_' + n])
        assert abs(a - t) <= 1e-3 * abs(t) + 1e-6, (n, a, t)
        for k, p in getattr(model, 'net' + n).named_parameters():
            check(n, p.grad, 'gD/%s/%s' % (n, k), floor_d)
    assert not bad, bad


def test_optimize_parameters_sequence_vs_reference_class_golden(dev, golden):
    """Three optimize_parameters() calls at lr 1e-3 (G step, then the D step on the pre-update frames, two Adams), everything
    on the device and nothing replayed: every loss of every step against the reference class's sequence.  Adam's first steps
    are sign-like, so rounding differences grow (the reference's own fp32 run is 0.1-0.3 % from its fp64 run by step 2)."""
    from animateportrait_amd import standins
    from test_train_golden_cpu import G_TERMS, DNAMES
    gd = golden('train_step.npz')
    model, batch, _ = _golden_model(dev, gd, lr=float(gd['seq_lr']))
    model.aux['netF'] = standins.StandinFlowNet().to(dev)
    model.aux['modnet'] = standins.StandinMatteNet().to(dev)
    worst = 0.0
    for it in range(int(gd['seq_steps'])):
        model.set_input(batch)
        model.optimize_parameters()
        for k in G_TERMS + DNAMES:
            a, t = float(getattr(model, 'loss_' + k)), float(gd['seq%d_loss_%s' % (it, k)])
            f32 = float(gd['seq%d_loss_%s_f32' % (it, k)])
            worst = max(worst, abs(a - t) / abs(t))
            assert abs(a - t) <= 3.0 * abs(f32 - t) + 2e-2 * abs(t), (it, k, a, t, f32)
    print('largest relative loss difference over the sequence: %.3e' % worst)


def test_train_step_with_reference_aux_architectures(dev, tmp_path, monkeypatch):
    """The three frozen nets at their REAL architectures (aux_nets.py; pinned to the reference classes on the CPU) loaded from
    checkpoint files by BaseModel.setup -> attach_aux_networks, as geomgm_ifw_fore_model.py:362-376 does in __init__: the matte
    comes from MODNet, the geometry and identity terms run through MobileFaceNet / Sphere20a, gradients reach the generator."""
    from animateportrait_amd import aux_nets, networks as N
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    from animateportrait_amd.options.base_options import TrainOptions
    from animateportrait_amd.models import create_model
    monkeypatch.chdir(tmp_path)
    os.makedirs('checkpoints')
    torch.manual_seed(3)
    torch.save({'state_dict': aux_nets.MobileFaceNet().state_dict()}, 'checkpoints/' + aux_nets.MOBILEFACENET_CKPT)
    torch.save({'module.' + k: v for k, v in aux_nets.MODNet().state_dict().items()}, 'checkpoints/' + aux_nets.MODNET_CKPT)
    torch.save(aux_nets.Sphere20a().state_dict(), 'checkpoints/sphere20a_20171020.pth')
    argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
            '--output_nc', '1', '--ngf', '8', '--ndf', '8', '--netg_resb_div', '3', '--netg_resb_disp', '3', '--lambda_geom', '50',
            '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2', '--lambda_face', '3.0', '--lambda_warp_inter', '10',
            '--blendbg', '1', '--batch_size', '2', '--gpu_ids', '0']
    opt = TrainOptions().parse(argv)
    model = create_model(opt)
    model.setup(opt)
    assert isinstance(model.aux['modnet'], aux_nets.MODNet) and isinstance(model.aux['landmarks'].net, aux_nets.MobileFaceNet)
    assert isinstance(model.aux['faceloss'], N.FaceLoss) and isinstance(model.aux['faceloss'].net.net, aux_nets.Sphere20a)
    batch = make_train_batch(2, seed=9)
    del batch['mask']                                   # the matte must come from MODNet now
    model.set_input(batch)
    with torch.no_grad():
        want = (model.aux['modnet'](batch['A'].to(dev), True)[2] > 0.5).float()
    assert torch.equal(model.mask, want)
    model.optimize_parameters()
    losses = model.get_current_losses()
    assert all(np.isfinite(v) for v in losses.values()) and losses['geom_B'] > 0 and losses['iden_B'] > 0, losses
    assert float(model.optimizer_G.flat_grad.abs().max()) > 0 if hasattr(model.optimizer_G, 'flat_grad') else True


def test_graphed_frozen_nets_match_eager(dev, monkeypatch):
    """aux_nets.GraphedFrozen (forward + backward of a frozen aux net as hipGraphs) == the eager module: outputs and the
    gradient w.r.t. the input, on fresh inputs after the capture (the graph reads its static buffers, not stale data: those
    would be off by O(1))."""
    from animateportrait_amd import aux_nets
    torch.manual_seed(1)
    for net, shape, pick in ((aux_nets.MobileFaceNet(), (4, 3, 112, 112), lambda o: o[0]), (aux_nets.Sphere20a(), (2, 3, 112, 96), tuple)):
        net = aux_nets._frozen(net, dev)
        g = aux_nets.GraphedFrozen(net, pick=pick)
        for it in range(3):
            x = torch.rand(shape, device=dev, generator=None)
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            oa, ob = g(xa), pick(net(xb))
            ta, tb = ([oa], [ob]) if torch.is_tensor(oa) else (list(oa), list(ob))
            # a LINEAR functional of the outputs (fixed weights): with |t| the gradient carries sign(t), and an output within
            # rounding of zero flips it between two runs -- 2.6 % of the largest gradient element in one of four runs of round 4
            ws = [torch.randn(t.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(11 + i)) / t.numel() ** 0.5
                  for i, t in enumerate(tb)]
            la = sum((t * w).sum() for t, w in zip(ta, ws))
            lb = sum((t * w).sum() for t, w in zip(tb, ws))
            la.backward(); lb.backward()
            assert abs(float(la.detach()) - float(lb.detach())) <= 1e-4 * abs(float(lb.detach())) + 1e-6, it
            # MIOpen picks its algorithms by timing them, so eager and captured runs (and two processes) round differently; a
            # PReLU / ReLU input within that rounding of zero then switches slope in one run only and moves the input gradient
            # in its receptive field (seen: 0.9 % of the largest element, 2 of 6 runs).  Sparse by nature: bound the L2 error
            # tightly and the largest element loosely.
            d = (xa.grad - xb.grad).double()
            assert float(d.norm()) <= 2e-3 * float(xb.grad.double().norm()), it
            assert float(d.abs().max()) <= 5e-2 * float(xb.grad.abs().max()), it
        assert len(g._graphs) == 1
        with torch.no_grad():                       # no gradient wanted: eager path, same values
            o = g(x)
        assert torch.is_tensor(o) or len(o) == 5
        # two grad-requiring calls of one shape BEFORE either backward: the second must not reuse the graph's static buffers
        # (ADVICE r4): both gradients equal the eager ones
        x1, x2 = torch.rand(shape, device=dev), torch.rand(shape, device=dev)
        grads = []
        for fn in (g, lambda t: pick(net(t))):
            a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            o1, o2 = fn(a), fn(b)
            first = lambda o: o if torch.is_tensor(o) else o[0]
            (first(o1).sum() + 2 * first(o2).sum()).backward()
            grads.append((a.grad, b.grad))
        for u, v in zip(grads[0], grads[1]):
            assert float((u - v).double().norm()) <= 2e-3 * float(v.double().norm())
        assert not g._in_flight
