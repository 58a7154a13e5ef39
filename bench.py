#!/usr/bin/env python3
"""bench.py -- generator frames/s @256x256, bs=16 per GPU (BASELINE.json metric), MI355X.

One "step" = one forward pass of the full-width generator (ngf=64, fp32,
resnet_9blocks_rcatland32_full_ifw, disp=div=3) over a batch of 16 synthetic 256x256 frames that is
already resident in HBM (BASELINE config 2).  N > 1: one process per GPU, frame batches shard by sample with
no data-path collective (InstanceNorm makes samples independent) -> weak scaling; only the timing uses a
collective.  The train-step leg (BASELINE configs 2-3: bs=16 per GPU, gradients all-reduced over RCCL) runs on
the same ranks.  Launch: either through ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
or plainly as ``python bench.py --gpus N`` -- without WORLD_SIZE in the environment bench.py re-executes itself
under torch.distributed.run with N ranks.  A world size that differs from --gpus is a hard error.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying also
  "roofline":        the convolution kernel with the largest total time per step (today the split-bf16 3x3 matrix kernel
                     conv_bf16x3), HIP-event time on the launch stream; `achieved` = ALGORITHMIC FLOP/s (2 x MACs of the
                     convolution / launch time), `peak` = the dense bf16 MFMA peak, `frac` = achieved / peak;
                     `executed_tflops` / `mfma_pipe_utilisation` = the bf16 MFMA FLOPs the kernel EXECUTES (3 per multiply);
  "parity_linf_vs_oracle": the gate -- output of the last timed step vs the oracle on the same batch and weights (< 1e-3 or
                     no line at all);
  "cpu_baseline":    the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample of the
                     same workload;
  "exact_fp32" / "plain_bf16_inference": the same step in the two other arithmetic modes (the strict-fp32 figure and the
                     reduced-precision one; neither is `value`);
  "train_step" / "train_step_bf16": the full geomgm_ifw_fore drawing-config step (all nine backward_G terms: the frozen
                     aux nets MODNet / MobileFaceNet / Sphere20a run at the reference architectures, aux_nets.py), B=16 per GPU;
  "stream":          BASELINE configs[4], the 10 s clip end to end (N=1 only);
  "env_switches":    every APAMD_* variable that was set (they select alternative kernels; none skips work).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

GFLOP_PER_FRAME = 140.125          # 70.063 GMAC conv + conv-transpose
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
# what the chip SUSTAINS on this kernel's operand data: a register-only stream of v_mfma_f32_32x32x16_bf16 on all 256 CUs
# with random bf16 heads + tails (tools/mfma_peak.hip; profiles/r06_power.md): the matrix pipe is power-limited on real data
SUSTAINED_BF16_MFMA_TFLOPS = 1800.0
BATCH = 16


def build_generator(dev, ngf=64):
    from animateportrait_amd import networks
    torch.manual_seed(1234)
    g = networks.define_G(3, 1, ngf, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02,
                          [dev.index], div=3, disp=3)
    return g.eval()


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota, and counted as
    physical cores (one thread per core: oneDNN convolutions do not gain from SMT siblings)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        import subprocess
        out = subprocess.run(['lscpu', '-p=CORE,SOCKET'], capture_output=True, text=True, timeout=5).stdout
        phys = len({l for l in out.splitlines() if l and not l.startswith('#')})
        if phys:
            n = min(n, phys)
    except Exception:
        pass
    forced = os.environ.get('APAMD_CPU_THREADS')
    return int(forced) if forced else n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(sd, warmup=3, iters=5):
    """Oracle generator forward on the host cores, BASELINE.md section 3 protocol: fp32, B=1 and B=16, 3 warm-up +
    5 timed iterations each, median; all physical cores.  ~25-40 s of CPU work in total."""
    import statistics
    from oracle import generator as og
    from animateportrait_amd.synthetic import make_generator_inputs, generator_args
    cores = host_cores()
    torch.set_num_threads(cores)
    res = {}
    t_all = time.time()
    ref = None
    with torch.no_grad():
        for b in (1, BATCH):
            args = generator_args(make_generator_inputs(b, seed=1234))
            for _ in range(warmup if b == 1 else 1):          # B=16: one warm-up (2-3 s each) keeps the leg bounded
                og.generator_forward(sd, *args, div=3, disp=3)
            ts = []
            for _ in range(iters):
                t0 = time.perf_counter()
                ref = og.generator_forward(sd, *args, div=3, disp=3)
                ts.append(time.perf_counter() - t0)
            res[b] = b / statistics.median(ts)
    # value = the better of the two batch sizes (oneDNN on many cores is often faster per frame at B=1)
    # `ref` = the oracle's output on the B=16 batch of seed 1234: the checker of the parity gate in main()
    return ref, {'value': round(max(res.values()), 3), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'bs1_frames_per_s': round(res[1], 3), 'bs16_frames_per_s': round(res[BATCH], 3), 'cpu_model': cpu_model(),
            'sample': 'oracle.generator_forward ngf=64 fp32, B=1 (3 warm-up) and B=16 (1 warm-up), median of %d timed '
                      'iterations each (%.1f s in total), torch CPU %d threads; value = max(B=1, B=16) rate'
                      % (iters, time.time() - t_all, cores)}


def oracle_reference_output(sd):
    """The oracle's forward of the reported batch (B=16, seed 1234), untimed: the parity checker when the CPU-baseline
    leg -- which computes the same tensor -- does not run (N > 1, --no-cpu-baseline)."""
    from oracle import generator as og
    from animateportrait_amd.synthetic import make_generator_inputs, generator_args
    torch.set_num_threads(host_cores())
    with torch.no_grad():
        return og.generator_forward(sd, *generator_args(make_generator_inputs(BATCH, seed=1234)), div=3, disp=3)


def train_parity_gate(dev):
    """The train-step composition against the golden made by the REFERENCE's own model class (tests/golden/train_step.npz:
    GeomGMIFWForeModel.forward / backward_G / backward_D_* at ngf = ndf = 8, b = 1, README flags, stand-in aux nets): one
    product step on the golden's batch and weights, all sixteen loss values within 1 % (+ the reference's own fp32-vs-fp64
    distance) -- the fp32 TPS solve is the only ill-conditioned piece.  The train legs are not reported if this fails."""
    import contextlib
    import io
    from animateportrait_amd import networks as _nets, standins
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    from animateportrait_amd.models import create_model
    from animateportrait_amd.options.base_options import TrainOptions
    from oracle import generator as og, discriminator as od          # (checker side: the seeded weights the golden was made with)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_step.npz'))
    w = int(z['width'])
    argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
            '--output_nc', '1', '--ngf', str(w), '--ndf', str(w), '--netg_resb_div', '3', '--netg_resb_disp', '3', '--lr', '0.00005',
            '--lambda_geom', '50', '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2', '--lambda_face', '3.0',
            '--lambda_warp_inter', '10', '--blendbg', '1', '--batch_size', '1', '--gpu_ids', str(dev.index), '--precision', 'bf16x3']
    dnames = ['D_A', 'D_A_l', 'D_A_le', 'D_A_ll', 'D_A_coh']
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(TrainOptions().parse(argv))
        model.netG_A.load_state_dict(og.init_params(og.generator_param_shapes(3, 1, w, 9, 3, 3), seed=int(z['g_seed'])), strict=True)
        for i, n in enumerate(dnames):
            getattr(model, 'net' + n).load_state_dict(
                og.init_params(od.patchgan_param_shapes(1 if n == 'D_A' else 2, w), seed=int(z['d_seed0']) + i), strict=True)
        model.aux['landmarks'] = standins.StandinLandmarkNet().to(dev)
        model.aux['faceloss'] = _nets.FaceLoss(standins.StandinFaceNet().to(dev))
        model.aux['netF'] = standins.StandinFlowNet().to(dev)
        model.aux['modnet'] = standins.StandinMatteNet().to(dev)
        batch = make_train_batch(1, seed=int(z['batch_seed']))
        for k in ('winA', 'winB', 'winB2', 'winBr'):
            batch[k] = torch.from_numpy(np.asarray(z[k])).view(1, 4)
        model.set_input(batch)
        model.optimize_parameters()
        got = model.get_current_losses()
    worst, worst_k = 0.0, None
    for k in ['G_A', 'G_A_l', 'G_A_le', 'G_A_ll', 'G_A_coh', 'geom_B', 'geom_B_lipline', 'warp_B', 'warp_inter1', 'iden_B', 'G'] + dnames:
        ref, f32 = float(z['loss_' + k]), float(z['loss_' + k + '_f32'])
        err = abs(got[k] - ref) / abs(ref)
        if err > worst:
            worst, worst_k = err, k
        if abs(got[k] - ref) > 3.0 * abs(f32 - ref) + 1e-2 * abs(ref):
            raise SystemExit('bench.py: train-step parity gate FAILED -- loss %s = %.6f, the reference class gives %.6f' % (k, got[k], ref))
    return {'max_rel_loss_diff_vs_reference_class_golden': round(worst, 6), 'worst_term': worst_k, 'terms': 16,
            'golden': 'tests/golden/train_step.npz (reference GeomGMIFWForeModel, ngf = ndf = 8, b = 1, fp64)'}


PARITY_BUDGET = 1e-3      # every reported number needs the generator output within 1e-3 L-inf (fp32, outputs in [-1, 1])


def train_step_ms(dev, rank, world, dist, steps, precision='bf16x3', aux_kind='reference-architecture'):
    """Time `steps` full train steps (after 2 warm-up steps) of the drawing config (readme.md:65 flags), B=16/GPU.
    precision: 'bf16x3' (fp32-class arithmetic, fp32 tensors) or 'bf16' (plain bf16 products, fp32 accumulation and
    fp32 master weights -- BASELINE configs[2-3]).
    aux_kind: 'reference-architecture' = the three frozen nets the reference's step calls at their REAL architectures
    (MODNet matte in set_input, MobileFaceNet in the geometry term, Sphere20a in the identity term: aux_nets.py, stock
    PyTorch-ROCm, random init -- their checkpoints are not in the reference tree); 'standin' = the toy nets of standins.py
    (what rounds 1-3 timed; kept as the comparison that isolates the aux nets' cost)."""
    from animateportrait_amd.options.base_options import TrainOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    argv = ['--model', 'geomgm_ifw_fore', '--netG', 'resnet_9blocks_rcatland32_full_ifw', '--dataset_mode', 'synthetic',
            '--output_nc', '1', '--netg_resb_div', '3', '--netg_resb_disp', '3', '--lr', '0.00005', '--lambda_geom', '50',
            '--lambda_geom_lipline', '50', '--more_weight_for_lip', '2', '--lambda_face', '3.0',
            '--lambda_warp_inter', '10', '--blendbg', '1', '--select_target12_thre', '0.0', '--niter', '70',
            '--niter_decay', '0', '--batch_size', str(BATCH), '--gpu_ids', str(dev.index), '--precision', precision]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):      # the model prints its notices; keep stdout = one JSON line
        torch.manual_seed(1234)
        model = create_model(TrainOptions().parse(argv))
        # the frozen third-party nets (MobileFaceNet, Sphere20a; checkpoints absent from the reference tree) as fixed-seed
        # stand-ins with the same call contracts: the geometry and identity terms of backward_G
        # (geomgm_ifw_fore_model.py:704-713, :741-752) -- window crop + bicubic / bilinear resize, both aux forwards and
        # data gradients, FaceLoss -- are then INSIDE the timed step
        from animateportrait_amd import parallel, standins, aux_nets, networks as _nets
        if aux_kind == 'standin':
            model.aux['landmarks'] = standins.StandinLandmarkNet().to(dev)
            model.aux['faceloss'] = _nets.FaceLoss(standins.StandinFaceNet().to(dev))
        else:
            torch.manual_seed(4321)
            frozen = lambda net: aux_nets._frozen(net, dev)                                       # noqa: E731
            # (as attach_aux_networks builds them: forward + backward of the two differentiated nets replay as hipGraphs)
            model.aux['landmarks'] = aux_nets.GraphedFrozen(frozen(aux_nets.MobileFaceNet((112, 112), 136)), pick=lambda o: o[0])   # geomgm_ifw_fore_model.py:362
            model.aux['faceloss'] = _nets.FaceLoss(aux_nets.GraphedFrozen(frozen(aux_nets.Sphere20a()), pick=tuple))            # :374-376
            model.aux['modnet'] = frozen(aux_nets.MODNet())                                       # :369-373, called in forward :519
            # netF: set_input runs flow_network_warp twice per step (geomgm_ifw_fore_model.py:57-84, 503-505).  FlowUnet_v2 on the
            # HIP kernels at the hyper-parameters the clip leg uses (class defaults nf 64 / max_nf 256 / 2 residual blocks, the 4
            # scales a 224-px input admits; the reference's own live in an absent train_opt.json), random init, as one hipGraph
            from animateportrait_amd import flow_unet, flow_unet_hip
            netF_cpu = flow_unet.FlowUnetV2(136, nf=64, max_nf=256, start_scale=2, num_scales=4, n_residual_blocks=2, norm='batch').eval()
            model.aux['netF'] = flow_unet_hip.FlowUnetV2Hip(netF_cpu).to(dev)
            model.aux['netF'].heads_only = True
            model.aux['netF'].use_graph = True
        parallel.broadcast_model(model)                  # all ranks start from rank 0's weights (no-op at N=1)
        drift0 = parallel.replica_drift(model)
        batch = {k: (v.to(dev) if torch.is_tensor(v) and not k.startswith('win') else v)
                 for k, v in make_train_batch(BATCH, seed=1234, rank=rank).items()}
        for _ in range(2):                               # two untimed steps: graph captures, MIOpen's algorithm search, packed weights
            model.set_input(batch)
            model.optimize_parameters()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.set_input(batch)
            model.optimize_parameters()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    from animateportrait_amd import parallel
    drift = parallel.replica_drift(model)                # identical updates on identical weights on every rank: 0.0
    losses = model.get_current_losses()
    netF_ms = None
    if model.aux.get('netF') is not None:
        # what the two flow_network_warp calls of set_input cost inside the step: set_input alone, with and without netF
        def _set_input_ms(n=5):
            model.set_input(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                model.set_input(batch)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3
        with contextlib.redirect_stdout(io.StringIO()):
            with_f = _set_input_ms()
            keep, model.aux['netF'] = model.aux['netF'], None
            without = _set_input_ms()
            model.aux['netF'] = keep
        netF_ms = round(with_f - without, 2)
    return {'ms_per_step': round(dt * 1e3, 2), 'samples_per_s': round(world * BATCH / dt, 2), 'steps': steps,
            'world_size': world, 'global_batch': world * BATCH,
            'replica_drift': {'after_broadcast': drift0, 'after_steps': drift},
            'gradient_exchange': 'none (1 rank)' if world == 1 else
                                 '2 RCCL all-reduces / step (G 63.7 MB in flight under the D backward passes, D 55.3 MB)',
            'batch_per_gpu': BATCH,
            'dtype': 'f32 (split-bf16 products, fp32 accumulate)' if precision == 'bf16x3' else
                     'bf16 (bf16 products, fp32 accumulate, fp32 master weights)',
            'loss_G': round(losses.get('G', float('nan')), 4),
            'losses': {k: round(v, 5) for k, v in losses.items()},
            'gflop_per_sample_algorithmic': 1234.0,
            'algorithmic_tflops': round(BATCH * 1.234 / dt, 1),
            'frac_algorithmic': round(BATCH * 1.234 / dt / PEAK_BF16_MFMA_TFLOPS, 4),
            'aux_nets': aux_kind,
            'netF_ms_per_step': netF_ms,
            'note': 'geomgm_ifw_fore drawing config, all nine backward_G terms in the timed region; ' + (
                'MODNet (matte), MobileFaceNet (geometry term) and Sphere20a (identity term) run at the reference architectures '
                '(aux_nets.py, stock PyTorch-ROCm, random init: their checkpoints are absent from the reference tree); netF '
                '(flow_network_warp twice per set_input) is FlowUnet_v2 on the HIP kernels at the class-default hyper-parameters '
                '(the reference\'s are in an absent train_opt.json), random init: netF_ms_per_step'
                if aux_kind != 'standin' else
                'the frozen landmark / identity nets are the toy stand-ins of standins.py, matte / intrinsic flow are batch inputs')}


def stream_leg(dev, frames=625, batch=16, cpu_frames=6):
    """BASELINE configs[4]: a 10 s clip (625 frames at the reference's 62.5 fps, main_end2end_module2.py:123-124)
    through the in-process pipeline (Module1 content LSTM -> landmarks -> motion grids -> landmark maps -> netF pre/post
    -> static drawing (once) -> generator -> blend), wall-clock, next to the reference-style CPU data path (per frame:
    txt round trip, scipy.griddata motion, static drawing at 512^2, generator at batch 1) timed on `cpu_frames` frames of
    the same clip and extrapolated.  Random-init weights; FlowUnet_v2 netF on the HIP kernels, stand-in matte (no checkpoints in the reference tree)."""
    import contextlib
    import io
    import tempfile
    from animateportrait_amd import standins, stream, module1, audio
    from animateportrait_amd.options.base_options import TestOptions
    from animateportrait_amd.models import create_model
    from animateportrait_amd.synthetic import make_landmarks
    opt = TestOptions().parse(['--model', 'geomcgt_ifw_test', '--netG', 'resnet_9blocks_rcatland32_full_ifw',
                               '--dataset_mode', 'synthetic', '--name', 'drawing_stream', '--output_nc', '1', '--ngf', '64',
                               '--netg_resb_div', '3', '--netg_resb_disp', '3', '--gpu_ids', str(dev.index)])
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(1234)
        model = create_model(opt)
    # netF: FlowUnet_v2 (intrinsic_flow_models/networks.py:647-744) on the HIP conv kernels, BatchNorm folded.  Its
    # hyper-parameters live in the absent checkpoints/FlowReg_id_flow_faces/train_opt.json: the class defaults (nf 64,
    # max_nf 256, start_scale 2, 2 residual blocks) with the 4 scales a 224-px input admits (112 -> 56 -> 28 -> 14 -> 7)
    from animateportrait_amd import flow_unet, flow_unet_hip
    torch.manual_seed(4321)
    netF_cpu = flow_unet.FlowUnetV2(136, nf=64, max_nf=256, start_scale=2, num_scales=4, n_residual_blocks=2, norm='batch').eval()
    model.aux['netF'] = flow_unet_hip.FlowUnetV2Hip(netF_cpu).to(dev)
    model.aux['netF'].heads_only = True        # as BaseModel.attach_flow_network sets it: flow_out / vis_out only
    model.aux['netF'].use_graph = True         # ... and one hipGraph launch per call
    content = module1.Audio2LandmarkContent(use_prior_net=True, drop_out=0.5).to(dev).eval()      # train_audio2landmark.py:71-73
    pose = module1.Audio2LandmarkPos(drop_out=0.5).to(dev).eval()                                  # :55-59
    spk = torch.randn(256, generator=torch.Generator().manual_seed(7))
    # the AutoVC converter between the mel spectrogram and the windows (main_end2end_module2.py:218-224): the reference
    # architecture (autovc.py), random init; synthetic speaker embeddings and an all-voiced synthetic f0 track (RAPT / resemblyzer
    # are third-party packages absent from this image)
    from animateportrait_amd import autovc
    torch.manual_seed(99)
    vc = autovc.Generator(16, 256, 512, 16).to(dev).eval()
    e_src = np.abs(np.random.RandomState(3).randn(256)).astype(np.float32) * 0.1
    e_trg = np.abs(np.random.RandomState(4).randn(256)).astype(np.float32) * 0.1
    convert = lambda mel: autovc.convert_mel(vc, mel, 0.5 + 0.4 * np.sin(np.arange(mel.shape[0]) / 9.0), e_src, e_trg, dev)   # noqa: E731
    g = torch.Generator().manual_seed(1234)
    photo = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 256), torch.linspace(-1, 1, 256), indexing='ij')
    matte = (((yy / 0.8) ** 2 + (xx / 0.6) ** 2) < 1).float().view(1, 1, 256, 256)
    lm0 = make_landmarks(1, g)[0]
    t = torch.arange(frames).view(frames, 1, 1).float()
    seq = lm0.unsqueeze(0) + 3.0 * torch.sin(0.1 * t + lm0.unsqueeze(0) / 40.0)
    wav = os.path.join(ROOT, 'tests', 'golden', 'female12.wav')      # the reference's example clip (examples/female12.wav)
    fid = torch.randn(204, generator=g) * 0.1
    streamer = stream.ClipStreamer(model, batch=batch)

    front = {}
    _convert_raw = convert

    def convert(mel):                      # (the AutoVC stage inside clip_audio_features, timed on its own when profiling)
        t1 = time.perf_counter()
        out = _convert_raw(mel)
        torch.cuda.synchronize()
        front['autovc_converter'] = time.perf_counter() - t1
        return out

    def clip(profile=False):
        # Module1 over the whole clip: both networks + the landmark post-processing of Audio2landmark_model.test and
        # main_end2end_module2.py:262-272 (timed; its output is not fed on -- random weights do not draw faces -- the
        # synthetic sequence of the same length is)
        # audio front end on the host, as in the reference: wav -> loudness -> mel (62.5 frames/s) -> 18-frame windows
        t0 = time.perf_counter()
        au = audio.clip_audio_features(wav, max_frames=frames, converter=convert)
        assert au.shape == (frames, 18, 80)
        if profile:
            torch.cuda.synchronize()
            front['audio_host_mel_windows'] = time.perf_counter() - t0 - front.get('autovc_converter', 0.0)
            t0 = time.perf_counter()
        fl = module1.predict_landmarks_speaker_aware(pose, content, au, spk, fid)
        module1.to_image_landmarks(fl, scale=0.01, shift=(-128.0, -128.0), rng=np.random.RandomState(0))
        if profile:
            torch.cuda.synchronize()
            front['module1_landmarks'] = time.perf_counter() - t0
        return streamer.run(photo, lm0, seq, matte=matte, profile=profile)
    clip()                                                   # warm-up (first-use compilation of torch LSTM kernels etc.)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = clip()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    clip(profile=True)
    stages = {k: round(v, 4) for k, v in list(front.items()) + list(streamer.timing.items())}      # front end + the streamer's stages
    # ---- reference-style CPU path on a bounded sample
    from oracle import generator as og, static_generator as osg, motion as om, aux_glue as oa
    cores = host_cores()
    torch.set_num_threads(cores)
    sd_g = og.init_params(og.generator_param_shapes(3, 1, 64, 9, 3, 3), seed=1234)
    sd_s = og.init_params(osg.static_param_shapes(3, 1, 64), seed=4321)
    netF = netF_cpu                                          # the same FlowUnet_v2 weights as stock PyTorch modules on the host
    a_lm = oa.draw2(256, 256, lm0.numpy(), 3).unsqueeze(0)

    def frame(ori, lm):
        with torch.no_grad():
            motion = torch.from_numpy(om.cal_motion256(ori, lm)).unsqueeze(0)
            tb = oa.draw2(256, 256, lm, 3).unsqueeze(0)
            flow, ifm = oa.flow_network_warp(netF, photo, torch.from_numpy(ori).unsqueeze(0), torch.from_numpy(lm).unsqueeze(0))
            static = osg.static_drawing(sd_s, photo)         # the reference recomputes it per frame (geomcgt_ifw_test_model.py:282-285)
            return osg.streaming_forward(lambda *a: og.generator_forward(sd_g, *a, div=3, disp=3), photo, matte, static,
                                         a_lm, tb, motion, flow, ifm)[0]
    with tempfile.TemporaryDirectory() as td:
        stream.reference_style_cpu_clip(frame, lm0.numpy(), seq[:1].numpy(), td)      # warm-up
        t0 = time.perf_counter()
        stream.reference_style_cpu_clip(frame, lm0.numpy(), seq[:cpu_frames].numpy(), td)
        cpu_s = (time.perf_counter() - t0) / cpu_frames
    return {'metric': 'end-to-end clip wall-clock (10 s clip = %d frames @62.5 fps)' % frames,
            'value': round(wall, 3), 'unit': 's', 'frames_per_s': round(frames / wall, 1), 'batch': batch,
            'realtime_factor': round(10.0 / wall, 2), 'stage_seconds_profiled': stages,
            'cpu_baseline': {'value': round(cpu_s * frames, 1), 'unit': 's (extrapolated from %d frames)' % cpu_frames,
                             's_per_frame': round(cpu_s, 3), 'cores': cores, 'kind': 'port',
                             'sample': 'oracle per-frame path: landmark txt round trip, scipy.griddata motion, cv2-rule landmark '
                                       'maps, netF pre/post, static drawing @512^2, generator, blend; batch 1'},
            'speedup_vs_cpu': round(cpu_s * frames / wall, 1),
            'note': 'random-init weights; netF = FlowUnet_v2 (nf 64, 4 scales) on the HIP conv kernels, stand-in matte; Module1 = both landmark networks (content + speaker-aware '
                    'pose branch) and their post-processing on the mel windows of the reference\'s example clip '
                    '(tests/golden/female12.wav; mel stage and the AutoVC converter -- reference architecture, random init -- inside '
                    'the timed region); synthetic speaker embeddings and f0 track (resemblyzer / RAPT are third-party packages '
                    'absent from this image); no checkpoint of any of these nets is in the reference tree'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-gate', action='store_true',
                    help='development only: skip the oracle comparison of the timed output; the line then says value_unverified')
    ap.add_argument('--no-exact-fp32', action='store_true', help='skip the exact-fp32 comparison leg')
    ap.add_argument('--train-steps', type=int, default=3, help='timed train steps (0 = skip the train-step leg)')
    ap.add_argument('--no-stream', action='store_true', help='skip the 10 s clip sub-record (BASELINE configs[4]) of the default run')
    ap.add_argument('--stream-batch', type=int, default=16, help='frames per batch of the clip leg (the reference runs batch 1)')
    ap.add_argument('--stream', action='store_true',
                    help='BASELINE configs[4] instead of the default legs: wall-clock of a 10 s (625-frame) clip through the '
                         'in-process pipeline, next to the reference-style CPU path (prints its own JSON line)')
    a = ap.parse_args()
    if a.stream:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
        torch.cuda.set_device(0)
        print(json.dumps(stream_leg(torch.device('cuda', 0), batch=a.stream_batch)))
        return

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # test plumbing for boxes with fewer GPUs than ranks: APAMD_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # APAMD_DIST_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device); the JSON line says so
    share_gpu = bool(os.environ.get('APAMD_BENCH_SHARE_GPU'))
    backend = os.environ.get('APAMD_DIST_BACKEND', 'nccl')
    if a.gpus < 1 or (a.gpus > torch.cuda.device_count() and not share_gpu):
        raise SystemExit('--gpus %d but this node exposes %d GPU(s)' % (a.gpus, torch.cuda.device_count()))
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks (one per GPU) under torch.distributed.run
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                  '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
                                  '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if share_gpu else int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: refusing to report a run whose rank count differs from the '
                         'one asked for' % (a.gpus, world))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist = None
    rccl_world = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                    # the world RCCL really sees: every rank contributes 1
        rccl_world = int(probe.item())
        if rccl_world != a.gpus:
            raise SystemExit('RCCL all-reduce saw %d ranks, --gpus %d' % (rccl_world, a.gpus))

    from animateportrait_amd import ops
    from animateportrait_amd.synthetic import make_generator_inputs, generator_args
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):      # keep stdout = the one JSON line
        G = build_generator(dev)
    # ---- CPU baseline first (rank 0, N=1 only): the GPU legs then run last, back to back.  The oracle runs on a host copy
    # of the product generator's own weights, and its B=16 output is the checker of the parity gate below
    sd_host = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    cpu = ref16 = None
    if not a.no_cpu_baseline and world == 1:
        ref16, cpu = cpu_baseline(sd_host)
    elif rank == 0 and not a.no_parity_gate:
        ref16 = oracle_reference_output(sd_host)
    args =[t.to(dev) for t in generator_args(make_generator_inputs(BATCH, seed=1234, rank=rank))]

    def step():
        with torch.no_grad():
            return G(*args)

    for _ in range(a.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(y).all())
    # ---- parity gate ("every reported number: generator output within 1e-3 L-inf"): the output of the LAST timed
    # step on rank 0's batch (seed 1234) against the oracle's forward of the same batch with the same weights.  No `value`
    # is printed for a run that fails it
    parity = None
    if ref16 is not None:
        parity = float((y.detach().cpu() - ref16).abs().max())
        if not parity < PARITY_BUDGET:
            raise SystemExit('bench.py: parity gate FAILED -- generator output differs from the oracle by %.3e L-inf '
                             '(budget %.0e) at ngf=64 B=%d; no number is reported' % (parity, PARITY_BUDGET, BATCH))

    # ---- per-kernel attribution: every conv launch bracketed by events on the launch stream
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    psteps = min(a.steps, 5)
    for _ in range(psteps):
        step()
    ops.PROFILER = None
    agg = prof.summary()
    if rank == 0 and os.environ.get('APAMD_BENCH_VERBOSE'):
        for name, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
            print('  %-34s launches/step %3d  ms/step %7.3f  TFLOP/s %6.1f' % (
                name, v['launches'] // psteps, v['ms'] / psteps, v['flops'] / (v['ms'] * 1e-3) / 1e12),
                'outliers %d' % v['outliers'] if v['outliers'] else '', file=sys.stderr)
    dom = max(agg.items(), key=lambda kv: kv[1]['ms'])
    dname, dk = dom
    alg = dk['flops'] / (dk['ms'] * 1e-3) / 1e12          # algorithmic TFLOP/s (2 * MACs of the convolution)
    if dname.startswith('Bf3Cfg'):
        # split-bf16 kernel: operands are split 2-way (bf16 head + tail) and every algorithmic FLOP is executed as 3 bf16
        # MFMA FLOPs (xh*wh + xh*wl + xl*wh); the pipe that bounds it is the bf16 matrix pipe
        kname, mult, peak, pipe = 'conv_bf16x3<%s>' % dname, 3.0, PEAK_BF16_MFMA_TFLOPS, \
            'bf16 MFMA (fp32 operands as bf16 head + tail, 3 products per multiply)'
    elif dname.startswith('DirectCfg'):
        kname, mult, peak, pipe = 'conv_direct_f32<%s>' % dname, 1.0, PEAK_FP32_MFMA_TFLOPS, 'fp32 VALU'
    else:
        kname, mult, peak, pipe = 'conv_igemm_f32<%s>' % dname, 1.0, PEAK_FP32_MFMA_TFLOPS, 'fp32 MFMA'
    # achieved = ALGORITHMIC FLOP/s (2 x MACs of the convolution) / launch time; frac = achieved / peak
    # of the pipe the kernel multiplies on.  The matrix-pipe utilisation (executed MFMA FLOP/s / peak: what the PMC
    # counters of profiles/ measure) is carried next to it under its own name
    roofline = {'bound': 'mfma', 'kernel': kname, 'pipe': pipe,
                'achieved': round(alg, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(alg / peak, 4), 'frac_algorithmic': round(alg / peak, 4),
                'frac_vs_sustained': round(alg * mult / SUSTAINED_BF16_MFMA_TFLOPS, 4) if mult == 3.0 else None,
                'sustained_note': 'frac_vs_sustained = executed FLOP/s over the %.0f TFLOP/s a register-only MFMA stream sustains on random '
                                  'bf16 operands on this chip (tools/mfma_peak.hip, profiles/r06_power.md); frac stays against the '
                                  'nominal peak' % SUSTAINED_BF16_MFMA_TFLOPS,
                'executed_tflops': round(alg * mult, 2), 'mfma_pipe_utilisation': round(alg * mult / peak, 4),
                'executed_flops_per_algorithmic_flop': mult,
                'launches_per_step': dk['launches'] // psteps,
                'avg_launch_us': round(dk['ms'] * 1e3 / dk['launches'], 2),
                'gflop_per_launch': round(dk['flops'] / dk['launches'] / 1e9, 3),
                'conv_ms_per_step': round(sum(v['ms'] for v in agg.values()) / psteps, 3),
                'bracket_outliers_replaced': sum(v['outliers'] for v in agg.values()),
                'traffic': None}
    # HBM bytes per launch of the dominant kernel: PMC passes cannot run inside this process, so the figure comes from the
    # committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes (tools/pmc_prof.sh -> profiles/hbm_traffic.json), stamped with the
    # profile round it was taken in.  A dominant kernel without an entry is an error, not a silent null
    tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    tab = {k.replace(' ', ''): v for k, v in json.load(open(tpath)).items()}
    ent = tab.get(dname.replace(' ', ''))
    if ent is None and os.environ.get('APAMD_BENCH_NO_TRAFFIC'):
        ent = {'hbm_bytes_per_launch': None, 'round': 'NONE (APAMD_BENCH_NO_TRAFFIC: development run before the PMC passes exist)'}
    if ent is None:
        raise SystemExit('bench.py: profiles/hbm_traffic.json has no HBM-traffic entry for the dominant kernel %s -- rerun '
                         'tools/round_profiles.sh and commit the PMC passes' % dname)
    roofline['traffic'] = ent['hbm_bytes_per_launch']
    roofline['traffic_unit'] = 'bytes/launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE)'
    roofline['traffic_source'] = 'profiles/hbm_traffic.json, PMC passes of profile round %s (not collected in this run)' % ent.get('round')
    train_hbm = {k: tab.get(k) for k in ('__train_step_bf16__', '__train_step_bf16x3__')}

    # ---- the same workload with every product on the exact-fp32 MFMA (APAMD_PRECISION=fp32): reported next to the
    # headline so that the split-bf16 arithmetic (3 bf16 MFMAs per fp32 product, fp32 accumulate) is an explicit,
    # measured choice -- and the largest output difference between the two paths on this batch
    def other_precision_leg(precision, note):
        old = ops.DEFAULT_PRECISION
        ops.DEFAULT_PRECISION = precision                        # the package mode stays set while the leg runs: it is
        try:                                                     # what `--precision` gives (split copies, tail planes)
            with contextlib.redirect_stdout(io.StringIO()):
                G2 = build_generator(dev)                        # same seed -> same weights
            with torch.no_grad():
                y2 = G2(*args)
                diff = float((y2 - y).abs().max())
                for _ in range(2):
                    G2(*args)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n2 = min(a.steps, 5) if precision == ops.PRECISION_FP32 else a.steps
                for _ in range(n2):
                    G2(*args)
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t1) / n2
        finally:
            ops.DEFAULT_PRECISION = old
        return {'value': round(BATCH / dt2, 2), 'unit': 'frames/s per GPU', 'ms_per_step': round(dt2 * 1e3, 3),
                'steps': n2, 'max_abs_diff_vs_headline_output': diff, 'note': note}

    exact = plain = None
    if not a.no_exact_fp32:
        exact = other_precision_leg(ops.PRECISION_FP32,
                                    'all convolutions on v_mfma_f32_32x32x2_f32 (APAMD_PRECISION=fp32); the headline '
                                    'path differs from it by max_abs_diff on outputs in [-1, 1] (budget 1e-3)')
        # REDUCED precision, reported for information only and never as `value`: one bf16 MFMA per product
        plain = other_precision_leg(ops.PRECISION_BF16,
                                    'REDUCED PRECISION, not the headline: operands rounded to bf16, one MFMA per product '
                                    '(--precision bf16), fp32 accumulate; max_abs_diff is against the headline output')

    # ---- second half of the BASELINE metric: "train step ms" -- the geomgm_ifw_fore drawing-config step
    # (G + 5 PatchGAN D's, warp / coherence losses, Adam), B=16 per GPU, fp32, gradients all-reduced over RCCL
    train = train_bf16 = None
    if a.train_steps > 0:
        del G, args, y
        torch.cuda.empty_cache()
        old = ops.DEFAULT_PRECISION
        train_gate = train_parity_gate(dev) if (rank == 0 and not a.no_parity_gate) else None
        try:
            train = train_step_ms(dev, rank, world, dist, a.train_steps, 'bf16x3')
            torch.cuda.empty_cache()
            train_bf16 = train_step_ms(dev, rank, world, dist, a.train_steps, 'bf16')
            torch.cuda.empty_cache()
            # the same bf16 step with the toy aux nets of rounds 1-3: the difference is what the three real nets cost
            toy = train_step_ms(dev, rank, world, dist, a.train_steps, 'bf16', aux_kind='standin')
            train_bf16['ms_per_step_with_standin_aux'] = toy['ms_per_step']
            train_bf16['aux_ms_per_step'] = round(train_bf16['ms_per_step'] - toy['ms_per_step'], 2)
        finally:
            ops.DEFAULT_PRECISION = old

    # ---- BASELINE configs[4]: the 10 s clip end to end, in the same JSON line (N = 1 only: one clip is one stream)
    stream_rec = None
    if world == 1 and not a.no_stream:
        import gc
        gc.collect()                                     # the train legs' models (reference cycles) hold ~18 GB of device blocks
        torch.cuda.empty_cache()
        stream_rec = stream_leg(dev, batch=a.stream_batch)

    if rank == 0:
        fps = world * BATCH * a.steps / dt
        out = {'metric': 'generator frames/sec @256x256 bs=16', 'value': round(fps, 2), 'unit': 'frames/s',
               'n_gpus': world, 'rccl_world_size': rccl_world, 'collective_backend': backend if world > 1 else None,
               'ranks_share_one_gpu': share_gpu, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f32 (fp32 tensors; wide convs multiply fp32-class on the bf16 pipe: operands as bf16 head + tail, '
                        '3 products of 16-bit significands per multiply, ~2^-17 relative per product, fp32 accumulate)',
               'data': 'synthetic',
               'config': {'workload': 'Module2 generator resnet_9blocks_rcatland32_full_ifw fwd-only, ngf=64, '
                                      'bs=16/GPU, 256x256, fp32 (BASELINE configs[1])',
                          'global_batch': BATCH * world, 'parallelism': 'dp%d (frame batches sharded, no collective)' % world,
                          'weights': 'random init N(0,0.02), seed 1234'},
               'parity_linf_vs_oracle': parity, 'parity_budget': PARITY_BUDGET,
               'parity_note': 'output of the last timed step (rank 0, B=%d, seed 1234) vs oracle.generator_forward on the same '
                              'batch and weights; a run that fails the gate prints no line' % BATCH if parity is not None
                              else 'SKIPPED (--no-parity-gate): value_unverified',
               'value_unverified': parity is None,
               'whole_generator_algorithmic_tflops': round(fps / world * GFLOP_PER_FRAME / 1e3, 2),
               'whole_net_frac_algorithmic': round(fps / world * GFLOP_PER_FRAME / 1e3 / PEAK_BF16_MFMA_TFLOPS, 4),
               'vs_fp32_mfma_conv_roofline': round(fps / world * GFLOP_PER_FRAME / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
               'roofline': roofline}
        if exact is not None:
            out['exact_fp32'] = exact
            out['plain_bf16_inference'] = plain
        if train is not None:
            train['parity_gate'] = train_gate if train_gate is not None else 'SKIPPED (--no-parity-gate)'
            out['train_step'] = train
        if train_bf16 is not None:
            if train is not None:
                # the nine backward_G terms (+ their sum) of the two arithmetic modes after the same steps on the same batch
                gk = [k for k in train['losses'] if k.startswith('G') or k in ('geom_B', 'geom_B_lipline', 'warp_B',
                                                                               'warp_inter1', 'iden_B')]
                train_bf16['max_rel_loss_diff_vs_bf16x3'] = round(max(
                    abs(train_bf16['losses'][k] - train['losses'][k]) / max(abs(train['losses'][k]), 1e-6) for k in gk), 5)
            out['train_step_bf16'] = train_bf16
        # the second half of the BASELINE metric ("train step ms") where the driver's record keeps it: sub-records of `roofline`
        for key, rec in (('train_step_bf16', train_bf16), ('train_step', train)):
            if rec is None:
                continue
            hb = train_hbm.get('__%s__' % (key if key.endswith('bf16') else key + '_bf16x3')) or {}
            roofline[key] = {
                'ms_per_step': rec['ms_per_step'], 'ms_per_step_with_standin_aux': rec.get('ms_per_step_with_standin_aux'),
                'netF_ms_per_step': rec.get('netF_ms_per_step'), 'aux_ms_per_step': rec.get('aux_ms_per_step'),
                'samples_per_s': rec['samples_per_s'], 'frac_algorithmic': rec['frac_algorithmic'],
                'frac_algorithmic_with_standin_aux': (round(BATCH * 1.234 / (rec['ms_per_step_with_standin_aux'] * 1e-3) / PEAK_BF16_MFMA_TFLOPS, 4)
                                                      if rec.get('ms_per_step_with_standin_aux') else None),
                'hbm_gb_per_step': hb.get('hbm_gb_per_step'),
                'hbm_source': ('profiles/hbm_traffic.json: FETCH_SIZE*2 + WRITE_SIZE over all kernels of the step with stand-in aux nets, '
                               'PMC passes of profile round %s (tools/train_hbm.sh; not collected in this run)' % hb.get('round')) if hb else None,
                'dtype': rec['dtype'], 'batch_per_gpu': BATCH, 'world_size': world,
                'max_rel_loss_diff_vs_bf16x3': rec.get('max_rel_loss_diff_vs_bf16x3')}
        if stream_rec is not None:
            out['stream'] = stream_rec
        if cpu is not None:
            out['cpu_baseline'] = cpu
            out['speedup_vs_cpu'] = round(fps / world / cpu['value'], 1)
        out['env_switches'] = {k: v for k, v in sorted(os.environ.items()) if k.startswith('APAMD_')}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
