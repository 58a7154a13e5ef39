"""CPU (-m "not gpu"): the C-ABI library loads and exports what include/*.h declares; host-side
registry / state_dict / planning logic; the product path refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from animateportrait_amd import _capi
    lib = _capi.lib()
    hdr = open(os.path.join(ROOT, 'include', 'animateportrait_amd.h')).read()
    declared = set(re.findall(r'\b(ap_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'ap_stream_t'}
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), name
        assert name in _capi.SIGNATURES, 'ctypes signature missing for %s' % name
    assert set(_capi.SIGNATURES) == declared
    assert b'gfx950' in lib.ap_version()


def test_ctypes_struct_layout_matches_header():
    from animateportrait_amd import _capi
    assert ctypes.sizeof(_capi.ApSrc) == 32
    assert ctypes.sizeof(_capi.ApConvDesc) == 18 * 4 + 3 * 32
    # ap_wgrad_desc: 12 ints, g + 3 sources, then (ABI 10) src_xs[3], src_xs_s2d, xs_parts (+ tail padding to 8)
    assert ctypes.sizeof(_capi.ApWgradDesc) == 12 * 4 + 4 * 32 + 4 * 8 + 8
    assert _capi.ApWgradDesc.src_xs.offset == 12 * 4 + 4 * 32 and _capi.ApWgradDesc.xs_parts.offset == 12 * 4 + 4 * 32 + 4 * 8


def test_planning_queries_need_no_gpu():
    from animateportrait_amd import ops
    s = ops.ConvSpec([256], 256, 3, 1, 1, ops.PAD_REFLECT)
    assert s.out_size(64, 64) == (64, 64)
    assert ops.ConvSpec([1], 64, 4, 2, 1).out_size(256, 256) == (128, 128)
    assert ops.ConvSpec([256], 512, 4, 1, 1).out_size(32, 32) == (31, 31)
    assert ops.ConvSpec([512], 1, 4, 1, 1).out_size(31, 31) == (30, 30)
    assert ops.ConvSpec([256], 128, 3, 2, 1, transposed=True, output_padding=1).out_size(64, 64) == (128, 128)
    assert ops.ConvSpec([3], 64, 7, 1, 3, ops.PAD_REFLECT).out_size(256, 256) == (256, 256)


def test_plans_of_the_special_layers_need_no_gpu():
    """Host-side planning of the layers with their own kernels: packed-image sizes are independent of the map size
    (weights are packed once per layer), workspaces are sized by the plan, unsupported forms are errors."""
    from animateportrait_amd import ops, _capi
    lib = _capi.lib()

    def packed_floats(spec, h, w):
        d = spec.desc(1, h, w)
        return _capi.check(lib.ap_conv2d_packed_floats(ctypes.byref(d)), 'packed_floats')

    head = ops.ConvSpec([512], 1, 4, 1, 1)                         # PatchGAN output layer: OIHW copy rides behind the image
    assert packed_floats(head, 31, 31) == packed_floats(head, 64, 64) >= 512 * 16
    first_dgrad = ops.ConvSpec([64], 2, 4, 2, 1, ops.PAD_ZERO, True, 0, ops.W_IOHW, False)
    assert packed_floats(first_dgrad, 128, 128) == packed_floats(first_dgrad, 16, 16) >= 64 * 2 * 16
    s2d = ops.s2d_spec(ops.ConvSpec([64], 128, 4, 2, 1))           # space-to-depth form: 2x2 stride 1 over 4C channels
    assert s2d.cin_segments == (256,) and s2d.k == 2 and s2d.out_size(65, 65) == (64, 64)
    with pytest.raises(RuntimeError, match='space-to-depth'):
        ops.ConvSpec([64], 64, 2, 2, 0).out_size(16, 16)
    assert ops.s2d_eligible(ops.ConvSpec([64], 128, 4, 2, 1), 128, 128)
    assert not ops.s2d_eligible(ops.ConvSpec([2], 64, 4, 2, 1), 256, 256)
    w = ops.s2d_weight(ops.torch.arange(2 * 3 * 16, dtype=ops.torch.float32).view(2, 3, 4, 4))
    assert w.shape == (2, 12, 2, 2) and float(w[1, (1 * 2 + 0) * 3 + 2, 1, 0]) == float(48 + 2 * 16 + (2 * 1 + 1) * 4 + 0)
    # weight-gradient workspaces: the streaming kernels need partial sums only
    n = _capi.check(lib.ap_conv_final_wgrad_workspace_floats(32, 64, 256, 256), 'final_ws')
    assert n % (64 * 49) == 0 and 0 < n <= 64 * 64 * 49
    d = _capi.ApWgradDesc()
    d.N, d.M, d.GH, d.GW, d.H, d.W, d.K, d.stride, d.pad, d.pad_mode, d.nsrc = 64, 8, 256, 256, 256, 256, 3, 1, 1, 0, 1
    d.precision = ops.PRECISION_BF16X3
    d.src[0].C = 1
    narrow = _capi.check(lib.ap_conv2d_wgrad_workspace_floats(ctypes.byref(d)), 'wgrad_ws')
    assert narrow % (8 * 9) == 0 and narrow < 64 * 256 * 256            # no padded operand copies
    d.src[0].C = 16
    assert _capi.check(lib.ap_conv2d_wgrad_workspace_floats(ctypes.byref(d)), 'wgrad_ws') > 64 * 16 * 256 * 256


def test_edge_layer_routes_served_shapes_need_no_gpu():
    """The plain-bf16 train step's narrow-sided layers (include/animateportrait_amd.h: ap_wgrad_k7_bf16, ap_wgrad_d0_bf16,
    ap_conv_final_dgrad_bf16, ap_conv_d0_fwd_bf16): which shapes the matrix-pipe kernels serve is host logic -- everything else
    stays on the general kernels -- and so is the workspace they ask for."""
    from animateportrait_amd import _capi
    lib = _capi.lib()
    k7 = lib.ap_wgrad_k7_bf16_ok
    assert k7(32, 64, 3, 256, 256, 0) == 1 and k7(32, 32, 3, 256, 256, 0) == 1 and k7(2, 64, 1, 16, 16, 0) == 1    # stems
    assert k7(32, 64, 1, 256, 256, 1) == 1 and k7(1, 32, 1, 9, 80, 1) == 1                                          # last layer
    assert k7(32, 64, 2, 256, 256, 0) == 0          # two narrow channels: no instantiation
    assert k7(32, 64, 3, 256, 256, 1) == 0          # final form: one gradient channel
    assert k7(32, 48, 3, 256, 256, 0) == 0          # wide channels: 32 or 64
    assert k7(32, 64, 3, 256, 250, 0) == 0 and k7(32, 64, 3, 256, 272, 0) == 0 and k7(32, 64, 3, 3, 256, 0) == 0
    d0 = lib.ap_wgrad_d0_bf16_ok
    assert d0(16, 64, 2, 256, 256) == 1 and d0(48, 64, 1, 256, 256) == 1 and d0(1, 64, 1, 10, 512) == 1 and d0(1, 64, 2, 6, 448) == 1
    assert d0(1, 64, 2, 10, 512) == 0 and d0(16, 32, 2, 256, 256) == 0 and d0(16, 64, 3, 256, 256) == 0 and d0(16, 64, 1, 255, 256) == 0
    dg = lib.ap_conv_final_dgrad_bf16_ok
    assert dg(32, 64, 256, 256) == 1 and dg(1, 32, 4, 16) == 1
    assert dg(32, 48, 256, 256) == 0 and dg(32, 64, 256, 272) == 0 and dg(32, 64, 256, 24) == 0
    hd = lib.ap_conv_head_dgrad_bf16_ok
    assert hd(16, 512, 31, 31) == 1 and hd(1, 32, 2, 2) == 1 and hd(16, 500, 31, 31) == 0 and hd(16, 512, 35, 35) == 0
    f0 = lib.ap_conv_d0_fwd_bf16_ok
    assert f0(16, 2, 64, 256, 256) == 1 and f0(3, 1, 64, 18, 8) == 1
    assert f0(16, 3, 64, 256, 256) == 0 and f0(16, 2, 32, 256, 256) == 0 and f0(16, 2, 64, 255, 256) == 0 and f0(16, 2, 64, 256, 258) == 0
    # workspace = the prepared narrow rows (bf16, two copies, A = H + 6 rows of W + 16) + one accumulator-order tile set per workgroup
    # (256 workgroups where no device answers: 8 row blocks per image at N = 32)
    narrow = 32 * 3 * (256 + 6) * 2 * (256 + 16) // 2
    assert lib.ap_wgrad_k7_bf16_workspace_floats(32, 64, 3, 256, 256, 0) == narrow + 256 * 2 * 5 * 1024
    assert lib.ap_conv_final_dgrad_bf16_workspace_floats(32, 64, 256, 256) == 32 * (256 + 12) * 2 * (256 + 16) // 2
    assert lib.ap_wgrad_k7_bf16_workspace_floats(32, 64, 2, 256, 256, 0) < 0 and b'not served' in lib.ap_last_error()


def test_planning_errors_are_reported():
    from animateportrait_amd import ops, _capi
    with pytest.raises(RuntimeError, match='stride'):
        ops.ConvSpec([8], 8, 3, 3, 1).out_size(16, 16)
    with pytest.raises(RuntimeError, match='reflection'):
        ops.ConvSpec([8], 8, 7, 1, 3, ops.PAD_REFLECT).out_size(2, 2)


def test_state_dict_keys_match_reference_layout():
    from animateportrait_amd import networks as N
    from oracle import generator as og, discriminator as od
    for disp in (1, 3):
        G = N.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [], div=3, disp=disp)
        want = og.generator_param_shapes(3, 1, 8, 9, 3, disp)
        got = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
        assert got == want
        blocks2 = [i for i in range(9) if isinstance(G.model2[str(i)], N.ResnetBlock2)]
        assert blocks2 == ([2, 5, 8] if disp == 1 else [0, 3, 6])
    G = N.define_G(3, 1, 64, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [], div=3, disp=3)
    assert sum(p.numel() for p in G.parameters()) == 15925553
    from oracle import static_generator as osg
    for ngf in (8, 64):
        S = N.define_G(3, 1, ngf, 'resnet_style2_9blocks', 'instance', use_dropout=False, gpu_ids=[])
        assert [(k, tuple(v.shape)) for k, v in S.state_dict().items()] == osg.static_param_shapes(3, 1, ngf)
    for cin, n in ((1, 2762689), (2, 2763713)):
        D = N.define_D(cin, 64, 'basic', 3, 'instance', 'normal', 0.02, [])
        assert [(k, tuple(v.shape)) for k, v in D.state_dict().items()] == od.patchgan_param_shapes(cin, 64)
        assert sum(p.numel() for p in D.parameters()) == n


def test_init_weights_semantics():
    from animateportrait_amd import networks as N
    torch.manual_seed(0)
    G = N.define_G(3, 1, 16, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [])
    w = torch.cat([p.flatten() for k, p in G.named_parameters() if k.endswith('weight')])
    assert abs(float(w.std()) - 0.02) < 5e-4 and abs(float(w.mean())) < 1e-4
    assert all(float(p.abs().max()) == 0 for k, p in G.named_parameters() if k.endswith('bias'))


def test_registry_error_behaviour():
    from animateportrait_amd import networks as N
    with pytest.raises(NotImplementedError, match='not recognized'):
        N.define_G(3, 1, 8, 'no_such_net', 'instance')
    with pytest.raises(NotImplementedError, match='outside the MI355X hot path'):
        N.define_G(3, 1, 8, 'unet_256', 'instance')
    with pytest.raises(NotImplementedError, match='not recognized'):
        N.define_D(1, 8, 'bogus', 3, 'instance')
    with pytest.raises(NotImplementedError):
        N.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'bogusnorm')


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    from animateportrait_amd import networks as N
    G = N.define_G(3, 1, 8, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [])
    x = torch.zeros(1, 3, 256, 256)
    l = torch.zeros(1, 1, 256, 256)
    with torch.no_grad(), pytest.raises(RuntimeError, match='no CPU path|MI355X'):
        G(x, l, l, torch.zeros(1, 256, 256, 2), torch.zeros(1, 2, 256, 256), l)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'animateportrait_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


def test_ganloss_and_scheduler(golden):
    from animateportrait_amd import networks as N
    gd = golden('losses.npz')
    crit = N.GANLoss('lsgan')
    # the lsgan reduction is a HIP kernel (ap_reduce_mean): no CPU path; its golden check is in test_losses_gpu.py
    with pytest.raises(RuntimeError, match='no CPU path'):
        crit(gd['pred'], True)
    ad = golden('adam.npz')

    class Opt:
        lr_policy = 'linear'; epoch_count = 1; niter = 3; niter_decay = 4
    w = torch.zeros(1, requires_grad=True)
    sched = N.get_scheduler(torch.optim.Adam([w], lr=1.0), Opt)
    facs = []
    for _ in range(8):
        facs.append(sched.get_last_lr()[0])
        sched.optimizer.step(); sched.step()
    assert facs == pytest.approx(list(ad['lr_factors']))


def test_opencv_rule_fixtures_are_what_the_restatements_draw():
    """tests/golden/opencv_rules.json (literal disc row tables and thick-line rasters; cv2 itself is absent here -- a
    maintainer checks them against cv2 with tests/golden/check_opencv_rules.py): the oracle restatement
    (oracle/cv_raster.py, quoting OpenCV 4.2 drawing.cpp), the older disc restatement in oracle/aux_glue.py and the
    library's host-side table all agree with the committed literals."""
    import json
    import numpy as np
    from oracle import cv_raster as cr, aux_glue as oa
    from animateportrait_amd import losses
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'opencv_rules.json')))
    assert d['circle_half_widths']['3'] == [3, 2, 2, 0] and d['circle_half_widths']['4'] == [4, 3, 3, 2, 0]
    assert d['circle_half_widths']['5'] == [5, 4, 4, 4, 3, 0] and d['circle_half_widths']['1'] == [1, 0]
    for r, hw in d['circle_half_widths'].items():
        assert cr.circle_half_widths(int(r)) == hw == oa.cv2_filled_circle_rows(int(r)) == losses.circle_rows(int(r))
    for ln in d['lines']:
        img = np.zeros((d['canvas'], d['canvas']), np.uint8)
        cr.thick_line(img, ln['p0'], ln['p1'], ln['thickness'])
        want = np.zeros_like(img)
        for y, a, b in ln['runs']:
            want[y, a:b + 1] = 255
        assert np.array_equal(img, want), ln
    # a horizontal thickness-2 line is three rows tall with one-pixel caps (ThickLine: half width 1 px + r = 1 end circles)
    h = d['lines'][0]
    assert h['runs'] == [[7, 5, 30], [8, 4, 31], [9, 5, 30]]
