"""Data parallelism: one process per GPU, gradients averaged with an all-reduce over RCCL (xGMI).

The reference uses single-process ``nn.DataParallel`` (Module2/models/networks.py:115-118); here every rank
holds G and the D's on its own GPU, runs the step on its shard of the frame batch, and the only exchange is the
mean of the gradients: once for G between ``backward_G`` and ``optimizer_G.step`` and once for the D's before
``optimizer_D.step`` (geomgm_ifw_fore_model.py:797-798, 810-819).  With ``FlatAdam`` the gradients of an
optimiser are one contiguous buffer, so each exchange is ONE collective (63.7 MB for G, 55.3 MB for the D's at
ngf=ndf=64); otherwise tensors are coalesced into buckets.  Works on any backend (``nccl`` = RCCL on ROCm;
``gloo`` in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torch.distributed.run environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


# tests: True makes every helper below behave as at world size 1 (a single-process reference run of the same model
# inside an initialised process group)
DISABLED = False


def world_size():
    if DISABLED:
        return 1
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_batch(batch, rank, world):
    """Split every tensor of a batch dict along dim 0 into ``world`` equal shards and return shard ``rank``."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0:
            if v.shape[0] % world:
                raise ValueError('batch entry %s of size %d does not divide over %d ranks' % (k, v.shape[0], world))
            n = v.shape[0] // world
            out[k] = v[rank * n:(rank + 1) * n]
        else:
            out[k] = v
    return out


# RCCL runs its kernels on its own HIP stream.  Measured on MI355X (round 3: tools/conc_warp.py; round 4: tools/hazard/run_hazard.py,
# profiles/r04_cohazard.md): while one of this library's LDS-DMA + MFMA kernels shares compute units with a kernel of ANOTHER
# stream, that kernel can compute wrong values -- the library's own warp kernel did in 50-80 % of its launches.  The trigger on the
# aggressor's side is narrowed down (VALU select traffic between the LDS-DMA pieces and the MFMAs; not the register count, not M0),
# the mechanism is not understood; a ring-reduce-shaped victim (RCCL's reduceCopy shape) stayed bit-exact over 1000 launches
# beside conv_bf16x3 and wgrad_bf16x3, but RCCL's real kernels were never tested (no multi-GPU node).  So by DEFAULT a collective is
# never in flight next to the backward pass: the compute stream is ordered behind every collective at once (``work.wait()`` is
# a stream wait on RCCL, no host block), ~1-2 ms of exposed exchange per train step on one node.  APAMD_OVERLAP_COLLECTIVES=1
# restores the overlapped form and says what it risks.
OVERLAP_COLLECTIVES = os.environ.get('APAMD_OVERLAP_COLLECTIVES', '0') == '1'
if OVERLAP_COLLECTIVES:
    import warnings
    warnings.warn('APAMD_OVERLAP_COLLECTIVES=1: gradient all-reduces will run NEXT TO the backward kernels.  Kernels sharing compute '
                  'units with this library\'s matrix kernels were measured to compute wrong values on MI355X (profiles/r04_cohazard.md); '
                  'an RCCL-shaped kernel was clean in that test, RCCL itself is untested.  Compare replica fingerprints '
                  '(parallel.assert_replicas_in_sync) and a serial run before trusting results.', RuntimeWarning, stacklevel=2)


def allreduce_flat_(flat, async_op=False):
    """In-place mean of one flat gradient buffer over all ranks.  async_op=True returns the work handle; on the GPU the
    compute stream has already been ordered behind it unless OVERLAP_COLLECTIVES (see above)."""
    w = world_size()
    if w == 1:
        return None
    flat.div_(w)
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
    if async_op and flat.is_cuda and not OVERLAP_COLLECTIVES:
        work.wait()
    return work


def allreduce_gradients(params, bucket_bytes=32 << 20):
    """Mean of ``p.grad`` over all ranks for every parameter, coalesced into buckets of ~bucket_bytes."""
    w = world_size()
    if w == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0
    for g in grads + [None]:
        if g is not None:
            bucket.append(g)
            size += g.numel() * g.element_size()
        if bucket and (g is None or size >= bucket_bytes):
            flat = torch.cat([b.reshape(-1) for b in bucket])
            flat.div_(w)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            off = 0
            for b in bucket:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            bucket, size = [], 0


def allreduce_optimizer_grads(optimizer, params=None, async_op=False):
    """One collective when the optimiser keeps a flat gradient buffer, bucketed otherwise.

    async_op=True (flat buffers only): the collective is enqueued and its work handle returned; the caller keeps
    launching independent work (the D backward passes while G's gradients travel) and calls ``wait_work`` before it
    reads the gradients."""
    flat = getattr(optimizer, 'flat_grad', None)
    if flat is not None:
        if hasattr(optimizer, '_rebind'):
            optimizer._rebind()
        return allreduce_flat_(flat, async_op=async_op)
    allreduce_gradients(params if params is not None else
                        [p for g in optimizer.param_groups for p in g['params']])
    return None


def net_grad_slice(optimizer, net):
    """The contiguous range of a FlatAdam's flat gradient buffer that holds the gradients of ``net``'s parameters
    (parameters of one network are adjacent: the optimiser is built over the concatenation of the networks' parameter
    lists, geomgm_ifw_fore_model.py:346-360), or None when the optimiser keeps no flat buffer / the range is not dense."""
    flat = getattr(optimizer, 'flat_grad', None)
    ps = [p for p in net.parameters() if getattr(p, '_flat_owner', None) is optimizer]
    if flat is None or not ps:
        return None
    lo = min(p._flat_off for p in ps)
    hi = max(p._flat_off + p.numel() for p in ps)
    if hi - lo != sum(p.numel() for p in ps):
        return None
    return flat[lo:hi]


def allreduce_net_grads(optimizer, net, async_op=True):
    """Mean over the ranks of ONE network's gradients inside a shared flat buffer, issued as soon as that network's
    backward pass has been enqueued: the five discriminators' exchanges of a step (geomgm_ifw_fore_model.py:810-819)
    travel under the backward passes of the discriminators that follow instead of as one exposed collective at the end.
    Returns the work handle (None at world size 1), or False when the optimiser has no dense slice for the network
    (the caller then falls back to allreduce_optimizer_grads)."""
    if world_size() == 1:
        return None
    sl = net_grad_slice(optimizer, net)
    if sl is None:
        return False
    return allreduce_flat_(sl, async_op=async_op)


def reduce_losses(losses):
    """Mean over the ranks of a dict of python floats with ONE small all-reduce (loss scalars for
    logging every print_freq).  Every rank must call it with the same keys; returns the averaged dict on every rank."""
    w = world_size()
    if w == 1:
        return dict(losses)
    keys = list(losses.keys())
    t = torch.tensor([float(losses[k]) for k in keys], dtype=torch.float64)
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = (t / w).cpu()
    return type(losses)((k, float(v)) for k, v in zip(keys, t))


def broadcast_model(model, src=0):
    """Make every rank start from rank ``src``'s weights (the reference's nn.DataParallel replicates ONE module,
    networks.py:115-118; here every rank built its own).  A ``FlatAdam`` optimiser owns its parameters as one flat
    buffer: one broadcast per optimiser; networks outside any optimiser (inference models) are broadcast tensor by
    tensor.  Call after ``create_model`` / ``setup`` -- ``BaseModel.setup`` does."""
    if world_size() == 1:
        return
    from . import ops
    covered = set()
    for opt in getattr(model, 'optimizers', []):
        flat = getattr(opt, 'flat', None)
        if flat is not None:
            dist.broadcast(flat, src=src)
            for extra in ('exp_avg', 'exp_avg_sq'):
                dist.broadcast(getattr(opt, extra), src=src)
            covered.update(id(p) for p in opt._params)
        else:
            for g in opt.param_groups:
                for p in g['params']:
                    dist.broadcast(p.data, src=src)
                    covered.add(id(p))
    for name in getattr(model, 'model_names', []):
        net = getattr(model, 'net' + name)
        for p in net.parameters():
            if id(p) not in covered:
                dist.broadcast(p.data, src=src)
        for b in net.buffers():
            dist.broadcast(b.data, src=src)
    ops.WEIGHTS_EPOCH += 1          # parameters changed through raw storage: packed-weight caches are stale


def state_fingerprint(model):
    """(sum, sum of squares) over all parameters of the model's networks as a 2-element fp64 tensor: equal on every
    rank iff the replicas hold the same weights (used by the tests and by ``assert_replicas_in_sync``)."""
    acc = torch.zeros(2, dtype=torch.float64)
    for name in getattr(model, 'model_names', []):
        for p in getattr(model, 'net' + name).parameters():
            d = p.detach().double()
            acc += torch.stack([d.sum(), (d * d).sum()]).cpu()
    return acc


def replica_drift(model):
    """Largest relative difference between this rank's weight fingerprint and any other rank's (0.0 at world size 1):
    identical all-reduced gradients applied to identical weights keep it at exactly 0."""
    w = world_size()
    if w == 1:
        return 0.0
    fp = state_fingerprint(model)
    dev = next(getattr(model, 'net' + model.model_names[0]).parameters()).device
    mine = fp.to(dev) if dist.get_backend() == 'nccl' else fp
    both = [torch.zeros_like(mine) for _ in range(w)]
    dist.all_gather(both, mine)
    scale = max(1.0, float(fp.abs().max()))
    return max(float((other.cpu() - fp).abs().max()) / scale for other in both)


def assert_replicas_in_sync(model, tol=0.0):
    """Raise when the ranks' weights differ (a data-parallel run whose replicas drifted is not SGD on one model)."""
    drift = replica_drift(model)
    if drift > tol:
        raise RuntimeError('data-parallel replicas diverged: relative fingerprint difference %.3e (tolerance %.1e)'
                           % (drift, tol))


def wait_work(work):
    """Order the current stream (RCCL) / block the host (gloo) behind an asynchronous collective; None is a no-op."""
    if work is not None:
        work.wait()
