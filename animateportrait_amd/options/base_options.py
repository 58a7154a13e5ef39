"""Option surface of Module2 (options/base_options.py:20-58, train_options.py:10-40, test_options.py:10-25):
same flag names, types and defaults, same three-stage gathering (base -> model -> dataset,
base_options.py:61-87).  ``--dataroot`` is only required for the file-based datasets of the reference;
the built-in ``--dataset_mode synthetic`` needs none."""
import argparse

from .. import models


class BaseOptions:
    def __init__(self):
        self.initialized = False
        self.isTrain = False

    def initialize(self, p):
        p.add_argument('--dataroot', default='', help='path to images (unused by --dataset_mode synthetic)')
        p.add_argument('--name', type=str, default='experiment_name')
        p.add_argument('--gpu_ids', type=str, default='0', help='gpu ids; one process per GPU: rank r uses the r-th entry')
        p.add_argument('--gpu_ids_p', type=str, default='0', help='accepted for compatibility; D and aux nets share the rank\'s GPU')
        p.add_argument('--checkpoints_dir', type=str, default='./checkpoints')
        p.add_argument('--model', type=str, default='geomgm_ifw_fore')
        p.add_argument('--input_nc', type=int, default=3)
        p.add_argument('--output_nc', type=int, default=3)
        p.add_argument('--ngf', type=int, default=64)
        p.add_argument('--ndf', type=int, default=64)
        p.add_argument('--netD', type=str, default='basic')
        p.add_argument('--netG', type=str, default='resnet_9blocks')
        p.add_argument('--n_layers_D', type=int, default=3)
        p.add_argument('--norm', type=str, default='instance')
        p.add_argument('--init_type', type=str, default='normal')
        p.add_argument('--init_gain', type=float, default=0.02)
        p.add_argument('--no_dropout', action='store_true')
        p.add_argument('--dataset_mode', type=str, default='unaligned')
        p.add_argument('--direction', type=str, default='AtoB')
        p.add_argument('--serial_batches', action='store_true')
        p.add_argument('--num_threads', default=4, type=int)
        p.add_argument('--batch_size', type=int, default=1)
        p.add_argument('--load_size', type=int, default=286)
        p.add_argument('--crop_size', type=int, default=256)
        p.add_argument('--max_dataset_size', type=int, default=float('inf'))
        p.add_argument('--preprocess', type=str, default='resize_and_crop')
        p.add_argument('--no_flip', action='store_true')
        p.add_argument('--display_winsize', type=int, default=256)
        p.add_argument('--epoch', type=str, default='latest')
        p.add_argument('--load_iter', type=int, default=0)
        p.add_argument('--verbose', action='store_true')
        p.add_argument('--suffix', default='', type=str)
        p.add_argument('--precision', type=str, default=None, choices=['fp32', 'bf16x3', 'bf16'],
                       help='(not in the reference) arithmetic of the wide convolutions: fp32 = exact fp32 MFMA; '
                            'bf16x3 = fp32-class split-bf16 (default, or $APAMD_PRECISION); bf16 = plain bf16 products with '
                            'fp32 accumulation and fp32 master weights (training configurations)')
        self.initialized = True
        return p

    def gather_options(self, argv=None):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        parser = self.initialize(parser)
        opt, _ = parser.parse_known_args(argv)
        setter = models.get_option_setter(opt.model)          # base_options.py:76-78
        parser = setter(parser, self.isTrain)
        opt, _ = parser.parse_known_args(argv)
        from .. import data
        parser = data.get_option_setter(opt.dataset_mode)(parser, self.isTrain)   # :81-83
        self.parser = parser
        return parser.parse_args(argv)

    def parse(self, argv=None):
        opt = self.gather_options(argv)
        opt.isTrain = self.isTrain
        if opt.suffix:
            opt.name = opt.name + ('_' + opt.suffix.format(**vars(opt)))
        if opt.precision is not None:            # picked up by every layer created afterwards
            from .. import ops
            ops.DEFAULT_PRECISION = ops.PRECISION_BY_NAME[opt.precision]
        opt.gpu_ids = [int(i) for i in str(opt.gpu_ids).split(',') if int(i) >= 0]      # :127-137
        opt.gpu_ids_p = [int(i) for i in str(opt.gpu_ids_p).split(',') if int(i) >= 0]  # :138-143
        self.opt = opt
        return opt


class TrainOptions(BaseOptions):
    def initialize(self, p):
        p = BaseOptions.initialize(self, p)
        p.add_argument('--display_freq', type=int, default=400)
        p.add_argument('--display_ncols', type=int, default=4)
        p.add_argument('--display_id', type=int, default=1)
        p.add_argument('--display_server', type=str, default='http://localhost')
        p.add_argument('--display_env', type=str, default='main')
        p.add_argument('--display_port', type=int, default=8097)
        p.add_argument('--update_html_freq', type=int, default=1000)
        p.add_argument('--print_freq', type=int, default=100)
        p.add_argument('--no_html', action='store_true')
        p.add_argument('--save_latest_freq', type=int, default=5000)
        p.add_argument('--save_epoch_freq', type=int, default=10)
        p.add_argument('--save_by_iter', action='store_true')
        p.add_argument('--continue_train', action='store_true')
        p.add_argument('--epoch_count', type=int, default=1)
        p.add_argument('--phase', type=str, default='train')
        p.add_argument('--niter', type=int, default=100)
        p.add_argument('--niter_decay', type=int, default=100)
        p.add_argument('--beta1', type=float, default=0.5)
        p.add_argument('--lr', type=float, default=0.0002)
        p.add_argument('--gan_mode', type=str, default='lsgan')
        p.add_argument('--pool_size', type=int, default=50)
        p.add_argument('--lr_policy', type=str, default='linear')
        p.add_argument('--lr_decay_iters', type=int, default=50)
        self.isTrain = True
        return p


class TestOptions(BaseOptions):
    def initialize(self, p):
        p = BaseOptions.initialize(self, p)
        p.add_argument('--ntest', type=int, default=float('inf'))
        p.add_argument('--results_dir', type=str, default='./results/')
        p.add_argument('--aspect_ratio', type=float, default=1.0)
        p.add_argument('--phase', type=str, default='test')
        p.add_argument('--eval', action='store_true')
        p.add_argument('--num_test', type=int, default=50)
        p.add_argument('--imagefolder', type=str, default='images')
        p.add_argument('--allow_random_init', action='store_true',
                       help='smoke mode (not in the reference): run with freshly initialised weights when a checkpoint '
                            'is missing instead of failing')
        p.set_defaults(model='geomgm_ifw_fore')
        p.set_defaults(load_size=p.get_default('crop_size'))
        self.isTrain = False
        return p
