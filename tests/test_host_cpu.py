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


def test_planning_queries_need_no_gpu():
    from animateportrait_amd import ops
    s = ops.ConvSpec([256], 256, 3, 1, 1, ops.PAD_REFLECT)
    assert s.out_size(64, 64) == (64, 64)
    assert ops.ConvSpec([1], 64, 4, 2, 1).out_size(256, 256) == (128, 128)
    assert ops.ConvSpec([256], 512, 4, 1, 1).out_size(32, 32) == (31, 31)
    assert ops.ConvSpec([512], 1, 4, 1, 1).out_size(31, 31) == (30, 30)
    assert ops.ConvSpec([256], 128, 3, 2, 1, transposed=True, output_padding=1).out_size(64, 64) == (128, 128)
    assert ops.ConvSpec([3], 64, 7, 1, 3, ops.PAD_REFLECT).out_size(256, 256) == (256, 256)


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
    assert abs(float(crit(gd['pred'], True)) - float(gd['gan_real'])) < 1e-6
    assert abs(float(crit(gd['pred'], False)) - float(gd['gan_fake'])) < 1e-6
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
