#!/usr/bin/env python3
"""Which packed weight images are rebuilt by the unbatched packer in a steady-state train step (they should all be table members).
Usage: APAMD_PRECISION=bf16 python tools/pack_census.py [batch]"""
import collections
import contextlib
import io
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import ops
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
    for _ in range(3):
        model.set_input(batch); model.optimize_parameters()
    seen = collections.Counter()
    orig = ops.packed_slot

    def logged(spec, weight, view, slot):
        before = None if slot is None else (slot.key, slot.batched)
        out = orig(spec, weight, view, slot)
        why = 'new slot' if slot is None else ('table' if out.batched and before[0] is not None else 'unbatched')
        if slot is not None and before[0] == out.key and before[0] is not None:
            why = 'current'
        seen[(why, tuple(spec.cin_segments), spec.cout, spec.k, spec.stride, spec.transposed, spec.w_layout, spec.w_flip,
              getattr(weight, '_flat_owner', None) is not None, out.batched)] += 1
        return out
    ops.packed_slot = logged
    model.set_input(batch); model.optimize_parameters()
    ops.packed_slot = orig
    for key, cnt in sorted(seen.items(), key=lambda kv: kv[0][0]):
        if key[0] != 'current':
            print('%3d x %s' % (cnt, key))
    print('current (no repack):', sum(c for k, c in seen.items() if k[0] == 'current'))


if __name__ == '__main__':
    main()
