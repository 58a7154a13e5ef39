"""Worker of tests/test_dp_gpu.py: one rank of a two-rank data-parallel run of the PRODUCT model (optimize_parameters
with real collectives).  Both ranks share cuda:0 and the collectives travel over gloo -- RCCL refuses two ranks on one
device, and this box has one GPU; the model code path (async G all-reduce under the D backward passes, one in-flight
collective per discriminator, broadcast of the initial weights, loss all-reduce) is the one an 8-GPU RCCL run takes.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; argv[1] = file the rank writes its verdict to.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from animateportrait_amd import parallel
    from animateportrait_amd.data.synthetic_dataset import make_train_batch
    import test_train_gpu as T
    res = {}

    def to_dev(b):
        return {k: (v.to(dev) if torch.is_tensor(v) and not k.startswith('win') else v) for k, v in b.items()}

    # every rank draws DIFFERENT initial weights; setup() must leave all of them on rank 0's
    torch.manual_seed(100 + rank)
    model, opt = T._make_model(dev)
    res['drift_before_broadcast'] = parallel.replica_drift(model)
    model.setup(opt)
    res['drift_after_broadcast'] = parallel.replica_drift(model)
    sd0 = {n: {k: v.detach().clone() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in model.model_names}

    batch = make_train_batch(2 * world, seed=77)                         # the global batch
    model.set_input(to_dev(parallel.shard_batch(batch, rank, world)))
    model.optimize_parameters()                                          # step 1: collectives in flight
    # after one Adam step from zero moments: exp_avg = (1 - beta1) * (all-reduced mean gradient)
    b1 = model.optimizer_G.param_groups[0]['betas'][0]
    g_dp = model.optimizer_G.exp_avg.clone() / (1 - b1)
    d_dp = model.optimizer_D.exp_avg.clone() / (1 - b1)
    res['drift_after_step1'] = parallel.replica_drift(model)
    losses = parallel.reduce_losses(model.get_current_losses())
    res['losses_finite'] = all(v == v and abs(v) < 1e9 for v in losses.values())
    res['n_losses'] = len(losses)
    model.set_input(to_dev(parallel.shard_batch(make_train_batch(2 * world, seed=78), rank, world)))
    model.optimize_parameters()                                          # step 2
    res['drift_after_step2'] = parallel.replica_drift(model)

    # ---- single-process reference on the SAME initial weights and the WHOLE batch (collectives disabled)
    parallel.DISABLED = True
    torch.manual_seed(5)
    ref, _ = T._make_model(dev)
    for n in ref.model_names:
        getattr(ref, 'net' + n).load_state_dict(sd0[n], strict=True)
    ref.set_input(to_dev(batch))
    ref.optimize_parameters()
    g_big = ref.optimizer_G.exp_avg.clone() / (1 - b1)
    d_big = ref.optimizer_D.exp_avg.clone() / (1 - b1)
    parallel.DISABLED = False

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

    def worst_tensor(opt_, a, b):
        worst, off = 0.0, 0
        for p in opt_._params:
            k = p.numel()
            x, y = a[off:off + k].double(), b[off:off + k].double()
            off += k
            scale = float(y.abs().max())
            if scale > 0:
                worst = max(worst, float((x - y).abs().max()) / scale)
            else:
                worst = max(worst, float(x.abs().max()))
        return worst
    res['G_rel'], res['D_rel'] = rel(g_dp, g_big), rel(d_dp, d_big)
    res['G_worst_tensor'] = worst_tensor(model.optimizer_G, g_dp, g_big)
    res['D_worst_tensor'] = worst_tensor(model.optimizer_D, d_dp, d_big)
    res['G_norm'] = float(g_big.double().norm())
    json.dump(res, open(sys.argv[1], 'w'))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
