"""GPU (-m gpu): the plain-bf16 arithmetic mode (AP_PRECISION_BF16, BASELINE configs[2-3] "bf16"): every wide
convolution / data gradient / weight gradient multiplies bf16-rounded operands with ONE MFMA per product and
accumulates in fp32.  The kernels are checked against exactly that definition (fp64 sums of bf16-rounded
operands), the composed train step against the fp64 oracle at bf16-autocast tolerances."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import linf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def r16(t):
    """round to bf16 (nearest even), back in fp64: what the head part of a split tensor / packed weight holds"""
    return t.float().bfloat16().double()


CASES = [
    # name, segs, cout, k, stride, pad, mode, transposed, H, W
    ('res3x3 reflect', [64], 64, 3, 1, 1, 'reflect', False, 40, 36),
    ('block2 3 segments zero', [64, 16, 16], 80, 3, 1, 1, 'zero', False, 33, 64),
    ('down 3x3 s2', [64], 128, 3, 2, 1, 'zero', False, 64, 64),
    ('patchgan 4x4 s1', [64], 96, 4, 1, 1, 'zero', False, 32, 32),
    ('patchgan 4x4 s2 (space-to-depth)', [64], 128, 4, 2, 1, 'zero', False, 64, 48),
    ('res3x3 reflect, 128-wide rows', [64], 64, 3, 1, 1, 'reflect', False, 32, 128),
    ('res3x3 zero, 256-wide rows, 2 segments', [16, 16], 48, 3, 1, 1, 'zero', False, 16, 256),
    ('down 3x3 s2, 128-wide rows', [64], 64, 3, 2, 1, 'zero', False, 32, 128),
    ('patchgan 4x4 s2, 256-wide rows', [64], 64, 4, 2, 1, 'zero', False, 16, 256),
    ('stem 7x7 (row form)', [3], 64, 7, 1, 3, 'reflect', False, 40, 70),
    ('up 3x3 s2 transposed (fused phases)', [64], 64, 3, 2, 1, 'zero', True, 32, 32),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_bf16_is_fp32_sum_of_bf16_products(dev, case):
    from animateportrait_amd import ops
    from animateportrait_amd.networks import ConvLayer
    name, segs, cout, k, stride, pad, mode, tr, H, W = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    n = 2
    xs = [torch.randn(n, c, H, W, generator=g) * 1.5 + 0.3 for c in segs]
    layer = ConvLayer(segs, cout, k, stride, pad, ops.PAD_REFLECT if mode == 'reflect' else ops.PAD_ZERO, tr,
                      1 if tr else 0).to(dev)
    layer.spec.precision = ops.PRECISION_BF16
    w = torch.randn(layer.weight.shape, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    with torch.no_grad():
        layer.weight.copy_(w); layer.bias.copy_(b)
    x16, w16 = r16(torch.cat(xs, 1)), r16(w)
    if tr:
        ref = F.conv_transpose2d(x16, w16, b.double(), stride=2, padding=1, output_padding=1)
    elif mode == 'reflect':
        ref = F.conv2d(F.pad(x16, (pad,) * 4, mode='reflect'), w16, b.double(), stride=stride)
    else:
        ref = F.conv2d(x16, w16, b.double(), stride=stride, padding=pad)
    y = layer.run([ops.Feat(x.to(dev)) for x in xs], act=ops.ACT_NONE)
    buf = __import__('ctypes').create_string_buffer(96)
    d = layer.spec.desc(n, H, W) if not ops.stem_rows_eligible(layer.spec) else None
    assert y.data.shape == ref.shape
    scale = float(ref.abs().max())
    assert linf(y.data, ref) < 2e-5 * scale, (name, linf(y.data, ref) / scale)
    # ... and it is NOT the fp32-class result: the plain-bf16 mode really is what ran
    full = (F.conv_transpose2d(torch.cat(xs, 1).double(), w.double(), b.double(), stride=2, padding=1, output_padding=1) if tr else
            F.conv2d(F.pad(torch.cat(xs, 1).double(), (pad,) * 4, mode='reflect') if mode == 'reflect' else
                     F.pad(torch.cat(xs, 1).double(), (pad,) * 4), w.double(), b.double(), stride=stride))
    assert linf(y.data, full) > 1e-4 * scale
    # statistics epilogue feeds InstanceNorm as in the other modes
    yn = layer.run([ops.Feat(x.to(dev)) for x in xs], norm_act=ops.ACT_NONE)
    got = (yn.data - yn.mean.view(n, cout, 1, 1)) * yn.rstd.view(n, cout, 1, 1)
    assert linf(got, F.instance_norm(ref - b.double().view(1, -1, 1, 1))) < 1e-3


@pytest.mark.parametrize('case', [c for c in CASES if c[3] in (3, 4, 7)], ids=[c[0] for c in CASES if c[3] in (3, 4, 7)])
def test_conv_bf16_backward_is_fp32_sum_of_bf16_products(dev, case, monkeypatch):
    """Data and weight gradients of a plain (bias + no norm) layer in bf16 mode.  Operators on the bf16 matrix path
    (stride-1 3x3 / 4x4 weight gradients, the 7x7 stems' in their opt-in row form, every wide data gradient) must equal the fp64 sum of bf16-rounded operands
    -- dgrad: bf16(dy) x bf16(w), wgrad: bf16(dy) x bf16(x); the ones that stay on the exact-fp32 kernels (strided /
    transposed weight gradients, 16-channel segments) must equal the exact result.  Nothing in between."""
    from animateportrait_amd import ops, autograd
    from animateportrait_amd.networks import ConvLayer
    name, segs, cout, k, stride, pad, mode, tr, H, W = case
    if k == 7:
        monkeypatch.setenv('APAMD_ROWS_WGRAD', '1')      # the stems' row form is opt-in
    g = torch.Generator().manual_seed(7 + sum(map(ord, name)))
    n = 2
    xs = [torch.randn(n, c, H, W, generator=g) for c in segs]
    layer = ConvLayer(segs, cout, k, stride, pad, ops.PAD_REFLECT if mode == 'reflect' else ops.PAD_ZERO, tr,
                      1 if tr else 0).to(dev)
    layer.spec.precision = ops.PRECISION_BF16
    w = torch.randn(layer.weight.shape, generator=g) * 0.05
    with torch.no_grad():
        layer.weight.copy_(w); layer.bias.zero_()
    tape = autograd.Tape()
    feats = [tape.track(ops.Feat(x.to(dev))) for x in xs]
    out = autograd.conv_forward(tape, layer, feats, act=ops.ACT_NONE)
    gy = torch.randn(out.data.shape, generator=g)
    tape.add(out, gy.to(dev), 0)
    tape.backward()
    # references
    gy16, w16 = r16(gy), r16(w)
    xcat = torch.cat(xs, 1).double().requires_grad_(True)
    wr = w16.clone().requires_grad_(True)

    def fwd(x, ww):
        if tr:
            return F.conv_transpose2d(x, ww, None, stride=2, padding=1, output_padding=1)
        xp = F.pad(x, (pad,) * 4, mode='reflect') if mode == 'reflect' else F.pad(x, (pad,) * 4)
        return F.conv2d(xp, ww, None, stride=stride)
    (fwd(xcat, w16) * gy16).sum().backward()                 # dgrad: bf16(dy) x bf16(w)
    dx_ref = xcat.grad
    (fwd(r16(torch.cat(xs, 1)), wr) * gy16).sum().backward()  # wgrad: bf16(dy) x bf16(x)
    dw_ref = wr.grad
    xe = torch.cat(xs, 1).double().requires_grad_(True)       # exact-fp32 operators: unrounded operands
    we = w.double().requires_grad_(True)
    (fwd(xe, we) * gy.double()).sum().backward()
    c0 = 0
    for f, c in zip(feats, segs):
        contribs = tape.take(f)
        g1, p, g2 = ops._split_contribs(contribs)
        dx = g1 if (p == 0 and g2 is None) else ops.fold_add(g1, p, g2)
        sc = float(dx_ref.abs().max())
        e16, eex = linf(dx, dx_ref[:, c0:c0 + c]) / sc, linf(dx, xe.grad[:, c0:c0 + c]) / sc
        assert min(e16, eex) < 3e-5, (name, 'dgrad', c, e16, eex)
        if c >= 32 and cout >= 48:
            assert e16 < 3e-5, (name, 'dgrad of a wide segment must run on the bf16 path', c, e16, eex)
        c0 += c
    dw = tape.param_grads[layer.weight]
    sc = float(dw_ref.abs().max())
    e16, eex = linf(dw, dw_ref) / sc, linf(dw, we.grad) / sc
    assert min(e16, eex) < 3e-5, (name, 'wgrad', e16, eex)
    if stride == 1 and not tr and (sum(segs) >= 32 or k == 7) and cout >= 48:     # (k = 7: the stems' row form)
        assert e16 < 3e-5, (name, 'stride-1 wgrad must run on the bf16 path', e16, eex)


def test_resnet_block_with_bf16_stored_activations(dev, monkeypatch):
    """Plain-bf16 training keeps the trunk's raw convolution outputs (and the data gradients that flow back into their
    InstanceNorm backward) as bf16 in HBM.  Definition checked here on one ResnetBlock at the trunk's shape:
    operands of every product rounded to bf16 AND the stored raw outputs rounded to bf16 -- the InstanceNorm statistics come from
    the fp32 accumulators (the convolution's epilogue), the normalisation reads the stored bf16 values.  The forward pass is
    compared with exactly that (fp64 emulation; a handful of elements may round to the neighbouring bf16 where the fp32 and
    fp64 accumulations differ, hence a quantile bar beside the L-inf one); the backward pass with the same block keeping its
    tensors fp32 (`APAMD_NO_BF16_RAW` semantics): the two differ by the bf16 rounding of the stored tensors and nothing else."""
    from animateportrait_amd import ops, autograd
    from animateportrait_amd.networks import ResnetBlock
    monkeypatch.setattr(ops, 'DEFAULT_PRECISION', ops.PRECISION_BF16)
    g = torch.Generator().manual_seed(77)
    n, c, h = 2, 256, 64
    x = torch.randn(n, c, h, h, generator=g)
    up = torch.randn(n, c, h, h, generator=g)

    def run(raw16):
        monkeypatch.setattr(ops, 'BF16_RAW', raw16)
        torch.manual_seed(3)
        blk = ResnetBlock(c).to(dev)
        for l in (blk.conv_block['1'], blk.conv_block['5']):
            l.spec.precision = ops.PRECISION_BF16
            torch.nn.init.normal_(l.weight, 0.0, 0.02)
        tape = autograd.Tape()
        f = tape.track(ops.Feat(x.to(dev)))
        prof = ops.LaunchProfiler()
        ops.PROFILER = prof
        try:
            out = blk.run(f, tape)
            tape.add(out, up.to(dev), 0)
            tape.backward()
        finally:
            ops.PROFILER = None
        g1, p1, g2 = ops._split_contribs(tape.take(f))
        dx = g1 if (p1 == 0 and g2 is None) else ops.fold_add(g1, p1, g2)
        ws = [blk.conv_block[k].weight.detach().cpu() for k in ('1', '5')]
        dws = [tape.param_grads[blk.conv_block[k].weight].detach().cpu().double() for k in ('1', '5')]
        return out.data.detach().cpu().double(), dx.detach().cpu().double(), ws, dws, prof

    out16, dx16, ws, dw16, prof16 = run(True)
    out32, dx32, _, dw32, prof32 = run(False)
    names16 = {r[0].replace(' ', '') for r in prof16.records}
    names32 = {r[0].replace(' ', '') for r in prof32.records}
    assert 'Bf3Cfg<1,3,1,2,4,4>bf16' in names16 or 'Bf3Cfg<1,3,1,2,4,1>bf16' in names16, sorted(names16)
    # ---- forward: fp64 emulation with bf16 operands and bf16-stored raw outputs
    r16 = lambda t: t.float().bfloat16().double()       # noqa: E731

    def conv(v, w):
        return F.conv2d(F.pad(r16(v), (1,) * 4, mode='reflect'), r16(w))

    def stats(raw):
        m = raw.mean((2, 3), keepdim=True)
        return m, 1.0 / torch.sqrt(raw.var((2, 3), unbiased=False, keepdim=True) + 1e-5)
    raw1 = conv(x.double(), ws[0].double())
    m1, r1 = stats(raw1)
    v1 = F.relu((r16(raw1) - m1) * r1)
    raw5 = conv(v1, ws[1].double())
    m5, r5 = stats(raw5)
    ref = (r16(raw5) - m5) * r5 + x.double()
    err = (out16 - ref).abs()
    scale = float(ref.abs().max())
    # (the bulk agrees to fp32 summation noise -- the definition is exact; one element of raw1 that rounds to the neighbouring bf16
    # where the fp32 and fp64 accumulations differ perturbs the 9 x 256 outputs it feeds by ~1e-4)
    assert float(err.median()) < 2e-6 * scale, float(err.median()) / scale
    assert float(err.quantile(0.999)) < 1e-3 * scale, float(err.quantile(0.999)) / scale
    assert float(err.max()) < 3e-2 * scale, float(err.max()) / scale       # (an element of raw1 one bf16 ulp off moves 9 x 256 products)
    # ... and it is NOT what the fp32-stored block gives: the stored tensors really are rounded
    assert linf(out16, out32) > 1e-4 * scale
    assert float((out16 - out32).abs().mean()) < 2e-3 * scale
    # ---- backward: the bf16-stored block against the fp32-stored one
    for a, b_, name in ((dx16, dx32, 'dx'), (dw16[0], dw32[0], 'dW1'), (dw16[1], dw32[1], 'dW5')):
        sc = float(b_.abs().max())
        assert float((a - b_).abs().mean()) < 2e-3 * sc, (name, float((a - b_).abs().mean()) / sc)
        cos = float((a * b_).sum() / (a.norm() * b_.norm()))
        assert cos > 0.9995, (name, cos)


@pytest.mark.parametrize('width,nb', [(8, 2), (64, 1)])
def test_train_step_bf16_mode_vs_fp64_oracle(dev, monkeypatch, width, nb):
    """The drawing-config train step with every wide layer in plain-bf16 arithmetic against the fp64 oracle, at
    bf16-autocast tolerances -- at ngf=ndf=8 / B=2 and at the FULL width the mode is timed at (BASELINE configs[2]:
    ngf=ndf=64, B=1 here): outputs within 3e-2 L-inf (the reference's own autocast(bf16) generator is 7.8e-2 from fp32,
    BASELINE.md; 1.5e-1 / mean 1e-2 at full width, measured 9.4e-2 / 5.5e-3), every loss term of backward_G and of the five
    D steps within 5 % (measured <= 0.3 %), gradient direction (cosine) >= 0.98 on every large weight tensor of G AND of the
    five discriminators (full-width G: >= 0.90, see below); fp32 master weights: one optimiser step moves the
    fp32 parameters by ~lr.  Step order of the reference: geomgm_ifw_fore_model.py:782-819."""
    from animateportrait_amd import ops
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    from oracle import generator as og, discriminator as od, train_step as ts
    import test_train_gpu as T
    monkeypatch.setattr(ops, 'DEFAULT_PRECISION', ops.PRECISION_BF16)
    torch.manual_seed(0)
    model, opt = T._make_model(dev, width, width)
    assert model.netG_A.model_tri_merge.spec.precision == ops.PRECISION_BF16
    sdG = og.init_params(og.generator_param_shapes(3, 1, width, 9, 3, 3), seed=11)
    model.netG_A.load_state_dict(sdG, strict=True)
    sdD, dnames = {}, ['D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh']
    for i, name in enumerate(dnames):
        sdD[name] = og.init_params(od.patchgan_param_shapes(1 if name == 'D_A' else 2, width), seed=20 + i)
        getattr(model, 'net' + name).load_state_dict(sdD[name], strict=True)
    batch = make_train_batch(nb, seed=5)
    model.set_input(batch)
    model.forward()
    nets_D = [getattr(model, 'net' + n) for n in model.model_names[1:]]
    model.set_requires_grad(nets_D, False)
    model.optimizer_G.zero_grad()
    model.backward_G()
    gG = {k: p.grad.detach().clone().cpu().double() for k, p in model.netG_A.named_parameters()}
    model.set_requires_grad(nets_D, True)
    model.optimizer_D.zero_grad()
    model.backward_D_A(); model.backward_D_A_l(); model.backward_D_A_le(); model.backward_D_A_ll(); model.backward_D_A_coh()
    gD = {n: {k: p.grad.detach().clone().cpu().double() for k, p in getattr(model, 'net' + n).named_parameters()}
          for n in dnames}
    ov = {k: getattr(model, k).detach().cpu().double() for k in ('mask1', 'mask2', 'fakeB_static_warp', 'fake_B_warp')}
    # the D step is checked on the product's own generated frames (their parity is asserted below)
    fakes = {k: getattr(model, k).detach().cpu().double() for k in ('fake_B', 'fake_B2', 'fake_B_l', 'fake_B2_l', 'fake_B_le',
                                                                     'fake_B2_le', 'fake_B_ll', 'fake_B2_ll')}
    cast = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t     # noqa: E731
    sG = {k: v.double().clone().requires_grad_(True) for k, v in sdG.items()}
    sD = {n: {k: v.double().clone() for k, v in sd.items()} for n, sd in sdD.items()}
    b = {k: cast(v) for k, v in batch.items()}
    o = ts.forward(sG, b, overrides=ov)
    terms = ts.g_loss(sD, o, b, overrides=ov)
    terms['G'].backward()
    for sd in sD.values():
        for v in sd.values():
            v.requires_grad_(True)
    od_ = dict(o)
    od_.update(fakes)
    dl = ts.d_losses(sD, od_, b)
    sum(dl.values()).backward()
    rep = {'linf': linf(model.fake_B_fore, o['fake_B_fore']),
           'mean': float((model.fake_B_fore.detach().cpu().double() - o['fake_B_fore'].detach()).abs().mean())}
    for k in ('G_A', 'G_A_l', 'G_A_le', 'G_A_ll', 'G_A_coh', 'geom_B_lipline', 'warp_B', 'warp_inter1', 'G'):
        a, t = float(getattr(model, 'loss_' + k)), float(terms[k])
        rep['loss_' + k] = (a, t)
    for name in dnames:
        rep['loss_' + name] = (float(getattr(model, 'loss_' + name)), float(dl[name]))
    cosG, cosD = {}, {}
    for k, v in sG.items():
        if k.endswith('.weight') and v.numel() >= 512:
            cosG[k] = float((gG[k] * v.grad).sum() / (gG[k].norm() * v.grad.norm()).clamp_min(1e-30))
    for n in dnames:
        for k, v in sD[n].items():
            if k.endswith('.weight') and v.numel() >= 512:
                cosD[n + '.' + k] = float((gD[n][k] * v.grad).sum() / (gD[n][k].norm() * v.grad.norm()).clamp_min(1e-30))
    rep['cosG_min'] = min(cosG.items(), key=lambda kv: kv[1])
    rep['cosD_min'] = min(cosD.items(), key=lambda kv: kv[1])
    print('bf16 train step width %d:' % width, rep)
    import os
    if os.environ.get('APAMD_TEST_DUMP'):
        open(os.environ['APAMD_TEST_DUMP'], 'a').write('bf16 %d %r %r %r\n' % (width, rep, cosG, cosD))
    # the full-width generator is deeper in rounding: the reference's own autocast(bf16) generator is 7.8e-2 L-inf from
    # fp32 at ngf=64; at ngf=8 the products are 8x shorter
    assert rep['linf'] < (3e-2 if width == 8 else 1.5e-1), rep
    assert rep['mean'] < (3e-3 if width == 8 else 1e-2), rep
    for k, v in rep.items():
        if k.startswith('loss_'):
            a, t = v
            assert abs(a - t) <= 5e-2 * abs(t) + 1e-3, (k, a, t)
    # gradient directions.  The discriminators (5 layers) and the ngf=8 generator keep cosine > 0.98.  At full width the
    # generator's gradient reaches its first layers through ~40 bf16-rounded layers (each product ~2^-9) and a B=1 batch
    # averages nothing: measured 0.938 (stems) .. 0.965 (merge conv) .. 0.99 (decoder), D's >= 0.996
    # (profiles/r03a_bf16_full_width.txt); the bar there is 0.90
    g_bar = 0.98 if width == 8 else 0.90
    for k, c in cosG.items():
        assert c > g_bar, ('G', k, c)
    for k, c in cosD.items():
        assert c > 0.98, (k, c)
    # fp32 master weights, finite step
    w0 = model.netG_A.model_tri_merge.weight.detach().clone()
    assert w0.dtype == torch.float32
    model.set_input(batch)
    model.optimize_parameters()
    losses = model.get_current_losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    step = float((model.netG_A.model_tri_merge.weight - w0).abs().max())
    assert 0 < step < 3 * 5e-5


K7_CASES = [
    # name, final form, wide channels, narrow channels, N, H, W
    ('stem 3->64', 0, 64, 3, 2, 20, 48),
    ('stem 3->32, odd rows', 0, 32, 3, 3, 37, 64),
    ('stem 1->64', 0, 64, 1, 1, 16, 16),
    ('stem 3->64 at 256 columns', 0, 64, 3, 1, 12, 256),
    ('final 64->1', 1, 64, 1, 2, 24, 32),
    ('final 32->1, odd rows', 1, 32, 1, 3, 19, 80),
    ('final 64->1 at 256 columns', 1, 64, 1, 1, 9, 256),
    ('stem 3->64, 300 small images (one block per image)', 0, 64, 3, 300, 4, 16),
    ('final 32->1, 290 small images', 1, 32, 1, 290, 5, 16),
    ('stem 3->32, one tall image', 0, 32, 3, 1, 301, 32),
]


@pytest.mark.parametrize('case', K7_CASES, ids=[c[0] for c in K7_CASES])
def test_k7_edge_layer_wgrad_on_the_matrix_pipe(dev, case):
    """ap_wgrad_k7_bf16 (csrc/wgrad_k7.h): the 7x7 reflection-padded edge layers' weight gradients in plain-bf16 arithmetic are the
    fp32-accumulated sums of bf16-rounded operands -- stems (networks.py:1251-1260) and the last layer (networks.py:1277-1279,
    whose input arrives with its InstanceNorm + ReLU still to be applied)."""
    from animateportrait_amd import ops
    name, final_form, mw, cn, n, H, W = case
    gen = torch.Generator().manual_seed(11 + sum(map(ord, name)))
    if final_form:
        x = torch.randn(n, mw, H, W, generator=gen) * 1.3 + 0.2
        mean = torch.randn(n * mw, generator=gen) * 0.1
        rstd = torch.rand(n * mw, generator=gen) + 0.5
        gy = torch.randn(n, 1, H, W, generator=gen)
        src = ops.Feat(x.to(dev), mean.to(dev), rstd.to(dev), ops.ACT_RELU)
        xin = F.relu((x.double() - mean.double().view(n, mw, 1, 1)) * rstd.double().view(n, mw, 1, 1))
        xin32 = F.relu((x - mean.view(n, mw, 1, 1)) * rstd.view(n, mw, 1, 1))      # as the kernel forms it, in fp32
        out_shape, cin = (1, mw, 7, 7), mw
    else:
        x = torch.randn(n, cn, H, W, generator=gen)
        gy = torch.randn(n, mw, H, W, generator=gen)
        src = ops.Feat(x.to(dev))
        xin, xin32 = x.double(), x
        out_shape, cin = (mw, cn, 7, 7), cn
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    try:
        dw = ops.wgrad(7, 1, 3, ops.PAD_REFLECT, ops.Feat(gy.to(dev)), [src], out_shape, precision=ops.PRECISION_BF16)
    finally:
        ops.PROFILER = None
    assert prof.calls.get('wgrad_k7<%s>' % ('final' if final_form else 'stem')) == 1, prof.calls

    def wgrad_ref(xv, gv):
        w = torch.zeros(out_shape, dtype=torch.float64, requires_grad=True)
        (F.conv2d(F.pad(xv, (3,) * 4, mode='reflect'), w) * gv).sum().backward()
        return w.grad
    ref16 = wgrad_ref(r16(xin32), r16(gy))
    exact = wgrad_ref(xin, gy.double())
    sc = float(ref16.abs().max())
    e16, eex = linf(dw, ref16) / sc, linf(dw, exact) / sc
    assert e16 < 3e-5, (name, e16, eex)
    # the same layer on the exact kernels differs from the bf16 sums by the operands' rounding: the route is not a relabelled fp32 one
    assert eex > 10 * e16, (name, e16, eex)
    # and the call is deterministic (fixed summation order)
    dw2 = ops.wgrad(7, 1, 3, ops.PAD_REFLECT, ops.Feat(gy.to(dev)), [src], out_shape, precision=ops.PRECISION_BF16)
    assert torch.equal(dw, dw2)


def test_stem_gradient_stored_as_bf16(dev):
    """Plain-bf16 train step: the InstanceNorm backward of a stem stores dy as bf16 (ap_instnorm_bwd act bit 8) and the weight
    gradient reads it (ap_wgrad_k7_bf16, wide->act bit 8) -- the same bits as rounding the fp32 dy, so both operators give
    exactly what the fp32-stored route gives."""
    from animateportrait_amd import ops
    gen = torch.Generator().manual_seed(5)
    n, c, H, W = 2, 32, 144, 128                      # a plane of the big-plane kernel's class (16384 < H W <= 65536)
    y = torch.randn(n, c, H, W, generator=gen) * 1.2 + 0.1
    g = torch.randn(n, c, H, W, generator=gen)
    x = torch.randn(n, 3, H, W, generator=gen)
    yd = y.to(dev)
    mean = yd.mean((2, 3)).flatten().contiguous()
    rstd = (yd.var((2, 3), unbiased=False) + 1e-5).rsqrt().flatten().contiguous()
    f = ops.Feat(yd, mean, rstd, ops.ACT_RELU)
    dy32 = ops.instnorm_bwd([(g.to(dev), 0)], f)
    dy16 = ops.instnorm_bwd([(g.to(dev), 0)], f, out_bf16=True)
    assert dy32.dtype == torch.float32 and dy16.dtype == torch.bfloat16
    assert torch.equal(dy16, dy32.bfloat16())
    src = ops.Feat(x.to(dev))
    dw32 = ops.wgrad(7, 1, 3, ops.PAD_REFLECT, ops.Feat(dy32), [src], (c, 3, 7, 7), precision=ops.PRECISION_BF16)
    dw16 = ops.wgrad(7, 1, 3, ops.PAD_REFLECT, ops.Feat(dy16), [src], (c, 3, 7, 7), precision=ops.PRECISION_BF16)
    assert torch.equal(dw16, dw32)
    # a small plane is not the big-plane kernel's: the request is ignored, fp32 comes back
    ys = torch.randn(1, 8, 32, 32, generator=gen).to(dev)
    fs = ops.Feat(ys, ys.mean((2, 3)).flatten().contiguous(), (ys.var((2, 3), unbiased=False) + 1e-5).rsqrt().flatten().contiguous(), ops.ACT_RELU)
    assert ops.instnorm_bwd([(torch.randn(1, 8, 32, 32, generator=gen).to(dev), 0)], fs, out_bf16=True).dtype == torch.float32


@pytest.mark.parametrize('case', [(2, 64, 24, 32), (3, 32, 19, 80), (1, 64, 9, 256), (1, 64, 5, 16), (270, 32, 4, 16), (1, 64, 4, 16)], ids=lambda c: 'N%d C%d %dx%d' % c)
def test_final_layer_dgrad_on_the_matrix_pipe(dev, case):
    """ap_conv_final_dgrad_bf16 (csrc/dgrad_k7.h): the gradient w.r.t. the reflection-padded input of the last layer
    (networks.py:1277-1279) in plain-bf16 arithmetic = fp32-accumulated sums of bf16(w) x bf16(g), in padded coordinates; folded
    over the reflection it is the layer's data gradient."""
    from animateportrait_amd import ops
    n, c, H, W = case
    gen = torch.Generator().manual_seed(3 + sum(case))
    w = torch.randn(1, c, 7, 7, generator=gen) * 0.05
    gy = torch.randn(n, 1, H, W, generator=gen)
    gp = ops.final_dgrad_k7(ops.Feat(gy.to(dev)), w.to(dev))
    assert tuple(gp.shape) == (n, c, H + 6, W + 6)

    def ref(wv, gv):
        xp = torch.zeros(n, c, H + 6, W + 6, dtype=torch.float64, requires_grad=True)
        (F.conv2d(xp, wv) * gv).sum().backward()
        return xp.grad
    ref16, exact = ref(r16(w), r16(gy)), ref(w.double(), gy.double())
    sc = float(ref16.abs().max())
    e16, eex = linf(gp, ref16) / sc, linf(gp, exact) / sc
    assert e16 < 3e-5, (case, e16, eex)
    assert eex > 10 * e16, (case, e16, eex)
    # folded over the reflection = the gradient w.r.t. the layer's input
    x = torch.zeros(n, c, H, W, dtype=torch.float64, requires_grad=True)
    (F.conv2d(F.pad(x, (3,) * 4, mode='reflect'), r16(w)) * r16(gy)).sum().backward()
    assert linf(ops.fold_add(gp, 3, None), x.grad) / sc < 3e-5


@pytest.mark.parametrize('case', [(2, 1, 256, 256, 2), (1, 2, 256, 256, 2), (3, 2, 64, 48, 0), (2, 1, 18, 8, 1), (48, 1, 256, 256, 2)],
                         ids=lambda c: 'N%d Cin%d %dx%d act%d' % c)
def test_patchgan_first_layer_on_the_matrix_pipe(dev, case, monkeypatch):
    """ap_conv_d0_fwd_bf16 (csrc/conv_d0.h): Conv2d(1 | 2, 64, 4, stride 2, pad 1) + bias + activation (networks.py:2620-2623) in
    plain-bf16 arithmetic = fp32-accumulated sums of bf16(x) x bf16(w); reached through ConvLayer.run in that mode."""
    from animateportrait_amd import ops
    from animateportrait_amd.networks import ConvLayer
    n, cin, H, W, act = case
    monkeypatch.setattr(ops, 'DEFAULT_PRECISION', ops.PRECISION_BF16)
    gen = torch.Generator().manual_seed(17 + sum(case))
    layer = ConvLayer([cin], 64, 4, 2, 1).to(dev)
    layer.spec.precision = ops.PRECISION_BF16
    w = torch.randn(64, cin, 4, 4, generator=gen) * 0.1
    b = torch.randn(64, generator=gen) * 0.1
    x = torch.randn(n, cin, H, W, generator=gen)
    with torch.no_grad():
        layer.weight.copy_(w); layer.bias.copy_(b)
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    try:
        y = layer.run(ops.Feat(x.to(dev)), act=act)
    finally:
        ops.PROFILER = None
    assert prof.calls.get('conv_d0<%d>' % cin) == 1, prof.calls
    assert not y.virtual and tuple(y.data.shape) == (n, 64, H // 2, W // 2)

    def ref(xv, wv):
        r = F.conv2d(xv, wv, b.double(), stride=2, padding=1)
        return F.relu(r) if act == 1 else (F.leaky_relu(r, 0.2) if act == 2 else r)
    ref16, exact = ref(r16(x), r16(w)), ref(x.double(), w.double())
    sc = float(ref16.abs().max())
    e16, eex = linf(y.data, ref16) / sc, linf(y.data, exact) / sc
    assert e16 < 3e-5, (case, e16, eex)
    assert eex > 10 * e16, (case, e16, eex)


@pytest.mark.parametrize('case', [(2, 1, 64, 64), (3, 2, 36, 96), (1, 2, 256, 256), (2, 1, 256, 256), (1, 1, 2, 32), (290, 2, 4, 32), (1, 1, 10, 512), (1, 2, 6, 448)],
                         ids=lambda c: 'N%d Cin%d %dx%d' % c)
def test_patchgan_first_layer_wgrad_on_the_matrix_pipe(dev, case):
    """ap_wgrad_d0_bf16 (form 2 of csrc/wgrad_k7.h): the weight gradient of Conv2d(1 | 2, 64, 4, stride 2, pad 1)
    (networks.py:2620-2623) in plain-bf16 arithmetic = fp32-accumulated sums of bf16(g) x bf16(x)."""
    from animateportrait_amd import ops
    n, cin, H, W = case
    gen = torch.Generator().manual_seed(23 + sum(case))
    x = torch.randn(n, cin, H, W, generator=gen)
    gy = torch.randn(n, 64, H // 2, W // 2, generator=gen)
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    try:
        dw = ops.wgrad(4, 2, 1, ops.PAD_ZERO, ops.Feat(gy.to(dev)), [ops.Feat(x.to(dev))], (64, cin, 4, 4), precision=ops.PRECISION_BF16)
    finally:
        ops.PROFILER = None
    assert prof.calls.get('wgrad_k7<d0>') == 1, prof.calls
    from animateportrait_amd import _capi
    assert _capi.lib().ap_wgrad_d0_bf16_ok(1, 64, 2, 10, 512) == 0       # (two channels of 256-pixel output rows: not served, the fp32 kernel runs)

    def wgrad_ref(xv, gv):
        w = torch.zeros(64, cin, 4, 4, dtype=torch.float64, requires_grad=True)
        (F.conv2d(xv, w, stride=2, padding=1) * gv).sum().backward()
        return w.grad
    ref16, exact = wgrad_ref(r16(x), r16(gy)), wgrad_ref(x.double(), gy.double())
    sc = float(ref16.abs().max())
    e16, eex = linf(dw, ref16) / sc, linf(dw, exact) / sc
    assert e16 < 3e-5, (case, e16, eex)
    assert eex > 10 * e16, (case, e16, eex)
    dw2 = ops.wgrad(4, 2, 1, ops.PAD_ZERO, ops.Feat(gy.to(dev)), [ops.Feat(x.to(dev))], (64, cin, 4, 4), precision=ops.PRECISION_BF16)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize('case', [(2, 512, 31, 31), (3, 64, 17, 30), (1, 32, 2, 2), (40, 96, 34, 34)], ids=lambda c: 'N%d C%d %dx%d' % c)
def test_patchgan_output_layer_dgrad_on_the_matrix_pipe(dev, case):
    """ap_conv_head_dgrad_bf16 (csrc/dgrad_k7.h: dgrad_head_kernel): the data gradient of Conv2d(C, 1, 4, 1, 1) (networks.py:2643) in
    plain-bf16 arithmetic = fp32-accumulated sums of bf16(w) x bf16(g)."""
    from animateportrait_amd import ops
    n, c, H, W = case
    gen = torch.Generator().manual_seed(31 + sum(case))
    w = torch.randn(1, c, 4, 4, generator=gen) * 0.05
    gy = torch.randn(n, 1, H - 1, W - 1, generator=gen)
    gx = ops.head_dgrad(ops.Feat(gy.to(dev)), w.to(dev), H, W)
    assert tuple(gx.shape) == (n, c, H, W)

    def ref(wv, gv):
        x = torch.zeros(n, c, H, W, dtype=torch.float64, requires_grad=True)
        (F.conv2d(x, wv, padding=1) * gv).sum().backward()
        return x.grad
    ref16, exact = ref(r16(w), r16(gy)), ref(w.double(), gy.double())
    sc = float(ref16.abs().max())
    e16, eex = linf(gx, ref16) / sc, linf(gx, exact) / sc
    assert e16 < 3e-5, (case, e16, eex)
    assert eex > 10 * e16, (case, e16, eex)
