"""Training entry point -- counterpart of Module2/train.py:7-64 (epoch / iteration loop, set_input ->
optimize_parameters, loss printing, checkpointing).  Launch one process per GPU:

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m animateportrait_amd.train \
        --model geomgm_ifw_fore --netG resnet_9blocks_rcatland32_full_ifw --dataset_mode synthetic --output_nc 1 ...
"""
import time

import torch

from . import parallel
from .data import create_dataset
from .models import create_model
from .options.base_options import TrainOptions


def main(argv=None):
    rank, world, local = parallel.init_distributed()
    opt = TrainOptions().parse(argv)
    if world > 1:
        opt.gpu_ids = [opt.gpu_ids[local] if local < len(opt.gpu_ids) else local]
    opt.rank = rank
    torch.cuda.set_device(opt.gpu_ids[0])
    dataset = create_dataset(opt)
    print('The number of training images = %d' % len(dataset))
    model = create_model(opt)
    model.setup(opt)            # also broadcasts rank 0's initial weights / optimiser state to every rank
    total_iters = 0
    for epoch in range(opt.epoch_count, opt.niter + opt.niter_decay + 1):
        epoch_start = time.time()
        model.update_process(epoch)
        for data in dataset:
            t0 = time.time()
            total_iters += opt.batch_size
            model.set_input(data)
            model.optimize_parameters()
            if total_iters % opt.print_freq < opt.batch_size:
                # every rank reads its shard's loss scalars; one small all-reduce makes the printed line the mean over
                # the global batch (Module2/train.py:44-49 prints the losses of the whole nn.DataParallel batch)
                torch.cuda.synchronize()
                losses = parallel.reduce_losses(model.get_current_losses())
                if rank == 0:
                    print('(epoch: %d, iters: %d, time: %.3f) ' % (epoch, total_iters, (time.time() - t0) / opt.batch_size)
                          + ' '.join('%s: %.3f' % kv for kv in losses.items()))
            if total_iters % opt.save_latest_freq < opt.batch_size and rank == 0:
                model.save_networks('iter_%d' % total_iters if opt.save_by_iter else 'latest')
        if epoch % opt.save_epoch_freq == 0 and rank == 0:
            model.save_networks('latest')
            model.save_networks(epoch)
        if rank == 0:
            print('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay, time.time() - epoch_start))
        model.update_learning_rate()
        parallel.assert_replicas_in_sync(model, tol=1e-6)      # identical updates on identical weights: any drift is a bug


if __name__ == '__main__':
    main()
