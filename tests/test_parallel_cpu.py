"""CPU: the N>1 gradient exchange with world sizes 2, 4 and 8 over gloo (the GPU path uses the same code over RCCL; the first
real 8-GPU run must not also be the first 8-rank run)."""
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from animateportrait_amd import parallel
    r, w, _ = parallel.init_distributed('gloo')
    assert (r, w) == (rank, world) and parallel.world_size() == world
    g = torch.Generator().manual_seed(100 + rank)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((7, 3), (1000,), (2, 2, 5))]
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    parallel.allreduce_gradients(params, bucket_bytes=4096)        # several buckets
    flat = torch.randn(5000, generator=g)
    parallel.allreduce_flat_(flat)

    # the train step's form: flat-gradient optimiser, collective in flight while other work runs, then wait
    class _Opt:
        flat_grad = torch.randn(3000, generator=g)
    local = _Opt.flat_grad.clone()
    work = parallel.allreduce_optimizer_grads(_Opt, async_op=True)
    busy = torch.randn(64, 64, generator=g) @ torch.randn(64, 64, generator=g)      # independent work meanwhile
    parallel.wait_work(work)
    parallel.wait_work(None)
    both = [torch.zeros(3000) for _ in range(world)]
    dist.all_gather(both, local)
    assert torch.allclose(_Opt.flat_grad, sum(both) / world, atol=1e-6) and busy.shape == (64, 64)
    batch = {'x': torch.arange(4 * world).view(4 * world, 1), 'name': 'n'}
    shard = parallel.shard_batch(batch, rank, world)
    # numpy arrays travel by value; torch tensors would travel as file descriptors the parent has to fetch from this
    # process while it is still alive (a race with the exit below)
    q.put((rank, [p.grad.numpy().copy() for p in params], flat.numpy().copy(), shard['x'].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


WORLDS = [2, 4, 8]


@pytest.mark.parametrize('world', WORLDS)
def test_allreduce_mean(world):
    port = 29600 + random.randint(0, 300)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    res = [(r, [torch.from_numpy(g) for g in gs], torch.from_numpy(fl), torch.from_numpy(sh)) for r, gs, fl, sh in res]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # expected: mean over ranks of the per-rank tensors
    exp_grads, exp_flat = None, None
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        gs = [torch.randn(s, generator=g) for s in ((7, 3), (1000,), (2, 2, 5))]
        fl = torch.randn(5000, generator=g)
        exp_grads = gs if exp_grads is None else [a + b for a, b in zip(exp_grads, gs)]
        exp_flat = fl if exp_flat is None else exp_flat + fl
    for rank, grads, flat, shard in res:
        for a, b in zip(grads, exp_grads):
            assert torch.allclose(a, b / world, atol=1e-6)
        assert torch.allclose(flat, exp_flat / world, atol=1e-6)
        assert shard.flatten().tolist() == list(range(rank * 4, rank * 4 + 4))


def test_image_pool_matches_reference_sequence(golden):
    from animateportrait_amd.util.image_pool import ImagePool
    random.seed(0)
    pool = ImagePool(50)
    seq = [float(pool.query(torch.full((1, 1, 1, 1), float(i))).item()) for i in range(60)]
    assert seq == list(golden('imagepool.npz')['returned'])


def test_shard_batch_rejects_ragged():
    from animateportrait_amd import parallel
    with pytest.raises(ValueError):
        parallel.shard_batch({'x': torch.zeros(5, 1)}, 0, 2)


# ------------------------------------------------------------------ replicas start equal; DP grads == big-batch grads
class _FakeModel:
    """The attributes parallel.broadcast_model / state_fingerprint read from a BaseModel."""

    def __init__(self, seed):
        from animateportrait_amd.optim import FlatAdam
        torch.manual_seed(seed)                                  # every rank builds DIFFERENT weights, as train.py did
        self.netG_A = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.Conv2d(4, 2, 3))
        self.netS = torch.nn.Conv2d(2, 2, 1)                     # a frozen net outside every optimiser
        self.model_names = ['G_A', 'S']
        self.optimizers = [FlatAdam(self.netG_A.parameters(), lr=1e-3)]


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from animateportrait_amd import parallel, ops
    from oracle import discriminator as od, generator as og, losses as ol
    parallel.init_distributed('gloo')
    # ---- (1) initial weights: different per rank before, identical after broadcast_model
    m = _FakeModel(seed=1000 + rank)
    before = parallel.state_fingerprint(m)
    raised = False
    try:
        parallel.assert_replicas_in_sync(m)
    except RuntimeError:
        raised = True
    epoch0 = ops.WEIGHTS_EPOCH
    parallel.broadcast_model(m)
    after = parallel.state_fingerprint(m)
    parallel.assert_replicas_in_sync(m)
    views_ok = all(p.data_ptr() >= m.optimizers[0].flat.data_ptr() for p in m.netG_A.parameters())
    # ---- (2) SURVEY.md section 4: all-reduced grads of N shards == single-rank grads on the concatenated batch
    # (InstanceNorm keeps samples independent, mean-reduced losses average).  Net = the oracle's PatchGAN.
    sd = og.init_params(od.patchgan_param_shapes(2, 4), seed=3)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(world * (2 if world == 2 else 1), 2, 64, 64, generator=g)
    shard = parallel.shard_batch({'x': x}, rank, world)['x']
    params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    sdp = dict(zip(sd.keys(), params))
    ol.gan_loss_lsgan(od.patchgan_forward(sdp, shard), True).backward()
    parallel.allreduce_gradients(params, bucket_bytes=1 << 12)
    # numpy payloads: tensors would travel as file descriptors that die with this process
    q.put((rank, before.numpy(), after.numpy(), raised, views_ok, ops.WEIGHTS_EPOCH - epoch0,
           [p.grad.numpy().copy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_broadcast_initial_weights_and_dp_gradient_equivalence(world):
    from oracle import discriminator as od, generator as og, losses as ol
    port = 29950 + random.randint(0, 300)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    _, b0, a0, r0, v0, e0, g0 = res[0]
    for _, b1, a1, r1, v1, e1, g1 in res[1:]:
        assert not np.array_equal(b0, b1) and r0 and r1       # different seeds -> replicas differed and that was detected
        assert np.array_equal(a0, a1) and np.array_equal(a0, b0)   # after the broadcast every rank holds rank 0's weights
        assert v0 and v1 and e0 == 1 and e1 == 1              # still views of the flat buffer; packed caches invalidated
        for a, b in zip(g0, g1):
            assert np.array_equal(a, b)
    sd = og.init_params(od.patchgan_param_shapes(2, 4), seed=3)
    x = torch.randn(world * (2 if world == 2 else 1), 2, 64, 64, generator=torch.Generator().manual_seed(9))
    params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    ol.gan_loss_lsgan(od.patchgan_forward(dict(zip(sd.keys(), params)), x), True).backward()
    for a, ref in zip(g0, params):
        r = ref.grad.numpy()            # (biases in front of InstanceNorm have pure rounding-noise gradients: atol)
        assert np.abs(a - r).max() <= 1e-4 * np.abs(r).max() + 1e-6, float(np.abs(a - r).max())


# ------------------------------------------------------------------ per-network in-flight collectives, loss means
def _slices_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import itertools
    from collections import OrderedDict
    from animateportrait_amd import parallel
    from animateportrait_amd.optim import FlatAdam
    parallel.init_distributed('gloo')
    torch.manual_seed(7)                                        # same structure on every rank
    nets = [torch.nn.Conv2d(1, 3, 3), torch.nn.Conv2d(2, 5, 1), torch.nn.Linear(4, 1)]
    opt = FlatAdam(list(itertools.chain(*[n.parameters() for n in nets])), lr=1e-3)     # as optimizer_D: one buffer, five nets
    g = torch.Generator().manual_seed(40 + rank)
    opt.flat_grad.copy_(torch.randn(opt.flat_grad.shape, generator=g))
    local = opt.flat_grad.clone()
    works, lens = [], []
    for n in nets:                                              # one collective per network, all in flight together
        sl = parallel.net_grad_slice(opt, n)
        lens.append(sl.numel())
        works.append(parallel.allreduce_net_grads(opt, n))
    for w in works:
        parallel.wait_work(w)
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    ok_mean = bool(torch.allclose(opt.flat_grad, sum(both) / world, atol=1e-6))
    # a network that is not (entirely) inside the optimiser has no slice
    no_slice = parallel.allreduce_net_grads(opt, torch.nn.Conv2d(1, 1, 1))
    parallel.DISABLED = True
    off = parallel.allreduce_net_grads(opt, nets[0]) is None and parallel.world_size() == 1
    parallel.DISABLED = False
    losses = parallel.reduce_losses(OrderedDict([('G_A', 1.0 + rank), ('D_A', 10.0 * (rank + 1))]))
    q.put((rank, ok_mean, lens, no_slice is False, off, list(losses.items())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_per_network_collectives_and_loss_means(world):
    """the discriminators' gradients travel as one in-flight collective per network over its slice
    of the shared flat buffer; loss scalars for logging are averaged with one small all-reduce."""
    port = 30300 + random.randint(0, 300)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_slices_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_mean, lens, no_slice, off, losses in res:
        assert ok_mean and no_slice and off
        assert lens == [3 * 9 + 3, 5 * 2 + 5, 4 + 1]
        assert losses == [('G_A', 1.0 + (world - 1) / 2.0), ('D_A', 10.0 * (world + 1) / 2.0)]
