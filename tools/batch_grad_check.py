#!/usr/bin/env python3
"""Where does the B = 16 train-step gradient differ from the mean of its single-sample gradients?  (debugging aid of
tests/test_train_gpu.py::test_train_step_at_the_reported_size_equals_mean_of_single_sample_steps)
Usage: python tools/batch_grad_check.py [precision] [width]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from animateportrait_amd import ops, parallel
from animateportrait_amd.options.base_options import TrainOptions
from animateportrait_amd.models import create_model
from animateportrait_amd.data.synthetic_dataset import make_train_batch
from test_train_gpu import _backward_both

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
width = sys.argv[2] if len(sys.argv) > 2 else '64'
ops.DEFAULT_PRECISION = {'bf16': ops.PRECISION_BF16, 'bf16x3': ops.PRECISION_BF16X3, 'fp32': ops.PRECISION_FP32}[precision]
argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
        '--output_nc', '1', '--ngf', width, '--ndf', width, '--netg_resb_div', '3', '--netg_resb_disp', '3',
        '--lr', '0.00005', '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2',
        '--lambda_face', '3.0', '--lambda_warp_inter', '10', '--blendbg', '1', '--select_target12_thre', '0.0',
        '--niter', '70', '--niter_decay', '0', '--batch_size', '16', '--gpu_ids', '0', '--precision', precision]
torch.manual_seed(1234)
model = create_model(TrainOptions().parse(argv))
B = 16
batch = make_train_batch(B, seed=1234)


def mean_of(chunk):
    accG = accD = None
    for r in range(B // chunk):
        model.fake_B_pool = type(model.fake_B_pool)(model.opt.pool_size)
        a, b = _backward_both(model, parallel.shard_batch(batch, r, B // chunk))
        accG = a.double() * chunk / B if accG is None else accG + a.double() * chunk / B
        accD = b.double() * chunk / B if accD is None else accD + b.double() * chunk / B
    return accG, accD


names = {}
for net in ['G_A', 'D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh']:
    for k, p in getattr(model, 'net' + net).named_parameters():
        names[p.data_ptr()] = net + '.' + k
ref = mean_of(1)
for chunk in (2, 4, 8, 16):
    got = mean_of(chunk)
    for nm, x, y, opt_ in (('G', got[0], ref[0], model.optimizer_G), ('D', got[1], ref[1], model.optimizer_D)):
        rows, off = [], 0
        for p in opt_._params:
            k = p.numel()
            a, b = x[off:off + k], y[off:off + k]
            off += k
            if float(b.abs().max()) == 0:
                continue
            rows.append((float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max()), names.get(p.data_ptr(), '?'), tuple(p.shape)))
        rel = float((x - y).norm() / y.norm())
        rows.sort(reverse=True)
        print('chunk %2d %s: whole gradient rel L2 %.2e; worst: %s' % (chunk, nm, rel, ', '.join('%s %.1e' % (r[2], r[0]) for r in rows[:5])), flush=True)
