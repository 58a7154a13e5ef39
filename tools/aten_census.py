#!/usr/bin/env python3
"""Which lines of the host code launch ATen kernels during a train step (torch.profiler, grouped by Python frame)."""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animateportrait_amd.options.base_options import TrainOptions      # noqa: E402
from animateportrait_amd.models import create_model                    # noqa: E402
from animateportrait_amd.data.synthetic_dataset import make_train_batch  # noqa: E402

argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
        '--output_nc', '1', '--ngf', '64', '--ndf', '64', '--netg_resb_div', '3', '--netg_resb_disp', '3', '--lr', '0.00005',
        '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2', '--lambda_face', '3.0',
        '--lambda_warp_inter', '10', '--blendbg', '1', '--niter', '70', '--niter_decay', '0', '--batch_size', '16',
        '--gpu_ids', '0']
model = create_model(TrainOptions().parse(argv))
batch = {k: (v.cuda() if torch.is_tensor(v) and k not in ('winA', 'winB', 'winB2', 'winBr') else v)
         for k, v in make_train_batch(16, seed=3).items()}
for _ in range(2):
    model.set_input(batch); model.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    model.set_input(batch); model.optimize_parameters()
    torch.cuda.synchronize()
cnt = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.name in ('aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::view',
                                                        'aten::as_strided', 'aten::reshape', 'aten::slice', 'aten::select',
                                                        'aten::detach', 'aten::alias', 'aten::expand', 'aten::permute',
                                                        'aten::transpose', 'aten::_unsafe_view', 'aten::unsqueeze', 'aten::squeeze',
                                                        'aten::t', 'aten::narrow', 'aten::resolve_conj', 'aten::resolve_neg',
                                                        'aten::lift_fresh', 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense'):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
        continue                                   # count outermost ATen calls only
    frame = next((f for f in ev.stack if root in f and 'tools/' not in f), '?')
    cnt[(ev.name, frame.replace(root + '/', '')[:110])] += 1
for (name, frame), n in cnt.most_common(45):
    print('%4d  %-22s %s' % (n, name, frame))
