#!/usr/bin/env python3
"""Times the B = 16 generator forward (the bench's headline leg without its bookkeeping): frames/s over 30 steps.
Usage: [APAMD_FUSED_NORM=1] python tools/gen_time.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from animateportrait_amd.synthetic import make_generator_inputs, generator_args

dev = torch.device('cuda:0')
g = bench.build_generator(dev)
args = [a.to(dev) for a in generator_args(make_generator_inputs(16, seed=1234))]
with torch.no_grad():
    for _ in range(5):
        g(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g(*args)
    torch.cuda.synchronize()
print('%.1f frames/s' % (16 * 30 / (time.perf_counter() - t0)))
