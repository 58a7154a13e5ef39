#!/usr/bin/env python3
"""Workloads of tools/power_record.py: run one thing back to back for N seconds.  Prints READY after the warm-up, RESULT lines at the end.
    python tools/power_loads.py dominant|forward|train <seconds>"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

mode, seconds = sys.argv[1], float(sys.argv[2])
dev = torch.device('cuda:0')


def loop(step, unit, per_call):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print('READY', flush=True)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        torch.cuda.synchronize()          # (bounds the launch queue; 20 calls in flight hide the host)
        n += 20
    dt = time.perf_counter() - t0
    print('RESULT %s: %d calls in %.2f s = %.1f us per call, %.1f %s' % (mode, n, dt, dt / n * 1e6, per_call * n / dt, unit), flush=True)


if mode == 'dominant':
    from animateportrait_amd import ops
    from animateportrait_amd.networks import ConvLayer
    layer = ConvLayer([256], 256, 3, 1, 1, ops.PAD_REFLECT, False, 0).to(dev)
    torch.nn.init.normal_(layer.weight, 0, 0.02)
    x = torch.randn(16, 256, 64, 64, device=dev)
    src = [ops.Feat(x, torch.zeros(16 * 256, device=dev), torch.ones(16 * 256, device=dev), ops.ACT_RELU)]
    xs = ops.presplit(src[0]) if hasattr(ops, 'presplit') else None       # the split copy is made once: the loop is the conv kernel only
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    layer.run(src, norm_act=ops.ACT_RELU)
    ops.PROFILER = None
    print('RESULT kernels of one call:', sorted({r[0] for r in prof.records}), flush=True)
    gf = 2 * 16 * 256 * 256 * 9 * 64 * 64 / 1e12
    loop(lambda: layer.run(src, norm_act=ops.ACT_RELU), 'TFLOP/s algorithmic (x3 executed)', gf)
elif mode == 'forward':
    import bench
    from animateportrait_amd.synthetic import make_generator_inputs, generator_args
    g = bench.build_generator(dev)
    args = [a.to(dev) for a in generator_args(make_generator_inputs(16, seed=1234))]
    with torch.no_grad():
        loop(lambda: g(*args), 'frames/s', 16)
else:
    import subprocess
    os.environ['APAMD_PRECISION'] = 'bf16'
    sys.argv = [sys.argv[0]]
    from animateportrait_amd.options.base_options import TrainOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    from animateportrait_amd import standins, networks
    argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
            '--output_nc', '1', '--ngf', '64', '--ndf', '64', '--netg_resb_div', '3', '--netg_resb_disp', '3',
            '--lr', '0.00005', '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2',
            '--lambda_face', '3.0', '--lambda_warp_inter', '10', '--blendbg', '1', '--niter', '70', '--niter_decay', '0',
            '--batch_size', '16', '--gpu_ids', '0', '--precision', 'bf16']
    model = create_model(TrainOptions().parse(argv))
    model.aux['landmarks'] = standins.StandinLandmarkNet().cuda()
    model.aux['faceloss'] = networks.FaceLoss(standins.StandinFaceNet().cuda())
    batch = {k: (v.cuda() if torch.is_tensor(v) and not k.startswith('win') else v) for k, v in make_train_batch(16, seed=3).items()}

    def step():
        model.set_input(batch)
        model.optimize_parameters()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    print('READY', flush=True)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        step()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('RESULT train: %d steps in %.2f s = %.1f ms per step' % (n, dt, dt / n * 1e3), flush=True)
