"""CPU: the N>1 gradient exchange with world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
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
    batch ={'x': torch.arange(8).view(8, 1), 'name': 'n'}
    shard = parallel.shard_batch(batch, rank, world)
    q.put((rank, [p.grad.clone() for p in params], flat.clone(), shard['x'].clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_world2():
    world = 2
    port = 29600 + random.randint(0, 300)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
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
