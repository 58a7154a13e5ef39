#!/usr/bin/env python3
"""Census of the convolution / weight-gradient calls of one train step: which layer shapes land on which kernel.
Usage: python tools/train_conv_census.py [batch]"""
import collections
import contextlib
import ctypes
import io
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops, _capi as C
from animateportrait_amd.options.base_options import TrainOptions
from animateportrait_amd.models import create_model
from animateportrait_amd.data.synthetic_dataset import make_train_batch


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
            '--output_nc', '1', '--netg_resb_div', '3', '--netg_resb_disp', '3', '--lambda_geom', '50',
            '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2', '--lambda_face', '3.0', '--lambda_warp_inter', '10',
            '--blendbg', '1', '--niter', '70', '--niter_decay', '0', '--batch_size', str(bs), '--gpu_ids', '0']
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(TrainOptions().parse(argv))
    batch = {k: (v.cuda() if torch.is_tensor(v) and not k.startswith('win') else v) for k, v in make_train_batch(bs, seed=3).items()}
    model.set_input(batch); model.optimize_parameters()
    convs, wgrads = collections.Counter(), collections.Counter()
    conv2d, wgrad = ops.conv2d, ops.wgrad

    def conv2d_logged(spec, srcs, *a, **k):
        n, _, h, w = srcs[0].data.shape
        d = spec.desc(n, h, w)
        buf = ctypes.create_string_buffer(96)
        C.check(C.lib().ap_conv2d_kernel_name(ctypes.byref(d), buf, 96), 'name')
        convs[(buf.value.decode(), n, h, w, spec.cin_segments, spec.cout, spec.k, spec.stride, spec.transposed,
               spec.w_layout, spec.w_flip)] += 1
        return conv2d(spec, srcs, *a, **k)

    def wgrad_logged(k, stride, pad, pad_mode, g, srcs, out_shape, **kw):
        copies = 'xs' if all(f.xs is not None for f in srcs) else ('s2d' if len(srcs) == 1 and srcs[0].s2d is not None else '-')
        wgrads[(tuple(g.data.shape), tuple(f.data.shape[1] for f in srcs), k, stride, tuple(out_shape), copies,
                'prepared G' if kw.get('g_t') is not None else 'G from dy')] += 1
        return wgrad(k, stride, pad, pad_mode, g, srcs, out_shape, **kw)
    ops.conv2d, ops.wgrad = conv2d_logged, wgrad_logged
    model.set_input(batch); model.optimize_parameters()
    ops.conv2d, ops.wgrad = conv2d, wgrad
    print('--- convolutions not on the split-bf16 path')
    for key, cnt in sorted(convs.items(), key=lambda kv: kv[0][0]):
        if not key[0].startswith('Bf3'):
            print('%3d x %s' % (cnt, key))
    print('--- weight gradients')
    for key, cnt in sorted(wgrads.items()):
        print('%3d x g%s srcC%s k%d s%d -> %s  forward copies: %s, %s' % ((cnt,) + key))


if __name__ == '__main__':
    main()
