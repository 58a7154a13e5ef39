"""Dataset registry (Module2/data/__init__.py:47-91).  The reference's file-based datasets
(umlvd_ifw, umlvdfw_test) read a Data/ tree that is not in the tree and are outside the hot path; the
``synthetic`` mode produces batches with the same dict keys / shapes / value ranges."""
from .synthetic_dataset import SyntheticDataset


def find_dataset_using_name(name):
    if name == 'synthetic':
        return SyntheticDataset
    raise NotImplementedError('dataset_mode [%s] is outside the MI355X hot path; use --dataset_mode synthetic or feed '
                              'model.set_input() with batches from the reference data layer' % name)


def get_option_setter(name):
    if name == 'synthetic':
        return SyntheticDataset.modify_commandline_options
    return lambda parser, is_train: parser   # the model sets dataset_mode=umlvd_ifw by default; tolerate it at parse time


def create_dataset(opt):
    return find_dataset_using_name(opt.dataset_mode)(opt)
