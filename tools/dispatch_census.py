#!/usr/bin/env python3
"""Which lines of the host code issue ATen operators during one plain-bf16 train step (TorchDispatchMode + the Python stack):
    APAMD_PRECISION=bf16 python tools/dispatch_census.py [top]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
top = int(sys.argv[1]) if len(sys.argv) > 1 else 60

from animateportrait_amd.options.base_options import TrainOptions      # noqa: E402
from animateportrait_amd.models import create_model                    # noqa: E402
from animateportrait_amd.data.synthetic_dataset import make_train_batch  # noqa: E402
from animateportrait_amd import standins, networks                     # noqa: E402

argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
        '--output_nc', '1', '--ngf', '64', '--ndf', '64', '--netg_resb_div', '3', '--netg_resb_disp', '3', '--lr', '0.00005',
        '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2', '--lambda_face', '3.0',
        '--lambda_warp_inter', '10', '--blendbg', '1', '--niter', '70', '--niter_decay', '0', '--batch_size', '16', '--gpu_ids', '0']
model = create_model(TrainOptions().parse(argv))
model.aux['landmarks'] = standins.StandinLandmarkNet().cuda()
model.aux['faceloss'] = networks.FaceLoss(standins.StandinFaceNet().cuda())
batch = {k: (v.cuda() if torch.is_tensor(v) and k not in ('winA', 'winB', 'winB2', 'winBr') else v)
         for k, v in make_train_batch(16, seed=3).items()}
for _ in range(2):
    model.set_input(batch); model.optimize_parameters()
torch.cuda.synchronize()
SKIP = ('view', 'as_strided', 'reshape', 'slice', 'select', 'detach', 'alias', 'expand', 'permute', 'transpose', 'unsqueeze', 'squeeze',
        't.default', 'narrow', 'empty', '_unsafe_view', 'unbind', 'split', 'lift_fresh', 'is_', '_local_scalar', 'item', 'sym_', 'stride', 'size')
cnt = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = '?'
            for f in reversed(traceback.extract_stack()):
                if f.filename.startswith(ROOT) and '/tools/' not in f.filename:
                    fr = '%s:%d %s' % (f.filename[len(ROOT) + 1:], f.lineno, f.name)
                    break
            dev = next((a.device.type for a in args if torch.is_tensor(a)), '-')
            cnt[(name, fr, dev)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    model.set_input(batch); model.optimize_parameters()
torch.cuda.synchronize()
print('ATen calls in one step (outside views):', sum(cnt.values()))
for (name, fr, dev), n in cnt.most_common(top):
    print('%4d  %-34s %-5s %s' % (n, name, dev, fr))
