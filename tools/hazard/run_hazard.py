#!/usr/bin/env python3
"""Co-residency hazard lab: does a kernel on one HIP stream compute wrong values while a matrix kernel of this library (or a
synthetic stand-in for it) runs on another stream?  Every (victim, aggressor) pair: the victim's output, launched REPS x 40
times beside the looping aggressor, is compared BITWISE with its output when run alone.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/hazard/libhazard.so tools/hazard/hazard_kernels.hip
    python tools/hazard/run_hazard.py [markdown-out]          exit code = number of pairs with wrong results

Victims:    warp   the product's forward warp kernel at level 2 (ap_warp_concat_fwd: the kernel round 3 saw fail)
            interp the failing branch's address pattern alone (four float2 gathers of a coarse map)
            reduce a ring-reduce-shaped kernel (RCCL's reduceCopy shape: 32 workgroups x 512 threads, 16-byte lanes, a + b)
Aggressors: conv3x3 / wgrad3x3 / ph4  the product's split-bf16 kernels through its Python operators
            synth:<v>                 hazard_kernels.hip aggr<v> (register-footprint / ingredient variants, see that file)
            gemm_bf16                 rocBLAS bf16 GEMM (control)
"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from animateportrait_amd import ops                                        # noqa: E402
from animateportrait_amd.networks import ConvLayer                         # noqa: E402
from animateportrait_amd.ops import Feat                                   # noqa: E402
from animateportrait_amd.synthetic import make_generator_inputs, generator_args   # noqa: E402

HZ = ctypes.CDLL(os.path.join(HERE, 'libhazard.so'))
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
REPS = int(os.environ.get('REPS', '5'))
cur = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)     # noqa: E731
P = lambda t: ctypes.c_void_p(t.data_ptr())                                # noqa: E731

# ---------------------------------------------------------------- victims
args = [t.to(dev)[:8].contiguous() for t in generator_args(make_generator_inputs(16, seed=1234))]
motion, flow, ifmask = args[3], args[4], args[5]
xw = torch.randn(8, 128, 64, 64, generator=g).to(dev)
fw = Feat(xw, (torch.randn(8 * 128, generator=g) * 0.1).to(dev), (torch.rand(8 * 128, generator=g) + 0.5).to(dev), act=ops.ACT_RELU)
ra = torch.randn(16 << 20, generator=g).to(dev)
rb = torch.randn(16 << 20, generator=g).to(dev)
cmap = torch.randn(8, 256, 256, 2, generator=g).to(dev)


def v_warp():
    return ops.warp_concat(fw, motion, flow, ifmask, 2, emit_xs=False).data


def v_reduce():
    out = torch.empty_like(ra)
    assert HZ.hz_reduce_launch(P(ra), P(rb), P(out), ctypes.c_longlong(ra.numel()), 32, cur()) == 0
    return out


def v_interp():
    out = torch.empty(8, 64, 64, 2, device=dev)
    assert HZ.hz_interp_launch(P(cmap), P(out), 8, 256, 64, 64, cur()) == 0
    return out


VICTIMS = {'warp': v_warp, 'interp': v_interp, 'reduce': v_reduce}

# ---------------------------------------------------------------- aggressors
L3 = ConvLayer([256], 256, 3, 1, 1, ops.PAD_REFLECT).to(dev)

with torch.no_grad():
    L3.weight.copy_(torch.randn(L3.weight.shape, generator=g) * 0.02)
xa = Feat(torch.randn(8, 256, 64, 64, generator=g).to(dev))
ga = torch.randn(8, 256, 64, 64, generator=g).to(dev)
ma = torch.randn(4096, 4096, generator=g).to(dev).to(torch.bfloat16)
big = torch.randn(64 << 20, generator=g).to(dev)
sink = torch.zeros(16, device=dev)


def a_conv():
    return L3.run(xa, norm_act=ops.ACT_RELU)


def a_wgrad():
    return ops.wgrad(3, 1, 1, ops.PAD_REFLECT, Feat(ga), [xa], (256, 256, 3, 3))


def a_synth(v):
    def f():
        rc = HZ.hz_aggr_launch(v, P(big), P(sink), 60, cur())
        assert rc == 0, rc
    return f


AGGR = {'conv3x3 (conv_bf16x3)': a_conv, 'wgrad3x3 (wgrad_bf16x3)': a_wgrad, 'gemm_bf16 (rocBLAS, control)': lambda: ma @ ma}
for v, note in ((0, '128 + 128 regs'), (8, '136 + 128'), (9, '144 + 128'), (10, '152 + 128'), (1, '160 + 128'), (2, '192 + 128'),
                (4, 'waves_per_eu(1,1)'), (5, 'LDS-DMA only'), (6, 'MFMA only'), (7, 's_nop 7 after each DMA piece'),
                (11, 'builtin DMA + MFMA + indexed private array'), (12, 'raw-asm DMA + MFMA + indexed private array'),
                (13, 'as 11, 160 arch VGPRs')):
    n = ctypes.c_int(0)
    HZ.hz_aggr_registers(v, ctypes.byref(n))
    AGGR['synth:%d (%s; numRegs %d)' % (v, note, n.value)] = a_synth(v)

# ---------------------------------------------------------------- victim-side experiment: the victim ALONE behind a poison kernel
if os.environ.get('POISON'):
    psink = torch.zeros(4, dtype=torch.int32, device=dev)
    nrep = int(os.environ.get('POISON', '200'))
    for vn, vf in VICTIMS.items():
        with torch.no_grad():
            ref = vf().clone()
        torch.cuda.synchronize()
        bad = 0
        for _ in range(nrep):
            assert HZ.hz_poison_launch(P(psink), cur()) == 0
            with torch.no_grad():
                o = vf()
            torch.cuda.synchronize()
            bad += int(not torch.equal(o, ref))
        print('%-7s ALONE behind the poison kernel (NaN in every VGPR / AGPR / LDS dword of every CU): wrong launches %4d of %d' % (vn, bad, nrep), flush=True)
    sys.exit(0)

only_v = os.environ.get('VICTIMS')
only_a = os.environ.get('AGGRESSORS')
if only_v:
    VICTIMS = {k: v for k, v in VICTIMS.items() if k in only_v.split(',')}
if only_a:
    AGGR = {k: v for k, v in AGGR.items() if any(k.startswith(a) for a in only_a.split(','))}
rows = []
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for vn, vf in VICTIMS.items():
    with torch.no_grad():
        ref = vf().clone()
    torch.cuda.synchronize()
    for an, af in AGGR.items():
        with torch.no_grad():
            af()
        torch.cuda.synchronize()
        bad = tot = 0
        for rep in range(REPS):
            with torch.no_grad():
                with torch.cuda.stream(sb):
                    keep = [af() for _ in range(40)]
                with torch.cuda.stream(sa):
                    outs = [vf() for _ in range(40)]
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o, ref)) for o in outs)
            tot += len(outs)
            del keep, outs
        rows.append((vn, an, bad, tot))
        print('%-7s beside %-52s wrong launches %4d of %d' % (vn, an, bad, tot), flush=True)

if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        f.write('| victim | aggressor on the other stream | wrong victim launches | of |\n|---|---|---|---|\n')
        for r in rows:
            f.write('| %s | %s | %d | %d |\n' % r)
sys.exit(sum(1 for r in rows if r[2]))
