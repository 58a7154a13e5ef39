"""Inference entry point -- counterpart of Module2/test.py:38-66: build the model, load ``G_A``, run the
generator over the dataset and write the frames (``.npy`` per frame; the PNG/HTML writers of the reference's
util/visualizer.py are outside the hot path)."""
import os

import numpy as np
import torch

from .data import create_dataset
from .models import create_model
from .options.base_options import TestOptions


def main(argv=None):
    opt = TestOptions().parse(argv)
    opt.num_threads, opt.serial_batches, opt.no_flip = 0, True, True     # test.py:41-45
    torch.cuda.set_device(opt.gpu_ids[0])
    dataset = create_dataset(opt)
    model = create_model(opt)
    model.setup(opt)       # loads '<epoch>_net_G_A.pth'; a missing file is an error (test.py:48) unless --allow_random_init
    if opt.eval:
        model.eval()
    out_dir = os.path.join(opt.results_dir, opt.name, '%s_%s' % (opt.phase, opt.epoch), opt.imagefolder)
    os.makedirs(out_dir, exist_ok=True)
    n = 0
    for data in dataset:
        if n >= opt.num_test:
            break
        model.set_input(data)
        model.test()
        fake = model.fake_B.detach().cpu().numpy()
        for i, path in enumerate(model.get_image_paths()):
            np.save(os.path.join(out_dir, os.path.basename(str(path)) + '_fake_B.npy'), fake[i])
            n += 1
    print('wrote %d frames to %s' % (n, out_dir))


if __name__ == '__main__':
    main()
