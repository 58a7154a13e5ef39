"""Synthetic batches with the input contract of GeomGMIFWForeModel.set_input
(Module2/models/geomgm_ifw_fore_model.py:443-505; produced in the reference by
UMLVDIFWDataset.__getitem__, Module2/data/umlvd_ifw_dataset.py:149-428).  See SURVEY.md section 8(d)."""
import torch

from ..synthetic import make_generator_inputs, jitter_landmarks, _discs, make_ifmask


def _boxes(batch, gen, size, lo, hi):
    """random axis-aligned box masks in {0,1}, (B,1,size,size)"""
    c = torch.randint(size // 4, 3 * size // 4, (batch, 2), generator=gen)
    r = torch.randint(lo, hi, (batch, 2), generator=gen)
    yy = torch.arange(size).view(1, size, 1)
    xx = torch.arange(size).view(1, 1, size)
    m = ((xx - c[:, 0].view(-1, 1, 1)).abs() <= r[:, 0].view(-1, 1, 1)) & ((yy - c[:, 1].view(-1, 1, 1)).abs() <= r[:, 1].view(-1, 1, 1))
    return m.float().unsqueeze(1)


def make_train_batch(batch, seed=1234, rank=0, size=256):
    """One batch dict of CPU tensors (all keys set_input reads, incl. the aux-net outputs mask / iw_flow / if_mask)."""
    d = make_generator_inputs(batch, seed=seed, rank=rank, size=size)
    gen = torch.Generator().manual_seed(seed * 7919 + rank + 1)
    lmA = d['lm1']
    tB = d['lm2']
    tB2 = jitter_landmarks(tB, gen, 1.5, size)
    ident = d['motion'] - torch.randn(d['motion'].shape, generator=torch.Generator().manual_seed(1)) * 0  # same tensor
    out = {
        'A': d['input'], 'B': torch.rand(batch, 1, size, size, generator=gen) * 2 - 1,
        'A_lm': d['land1'], 'tB_lm': d['land2'], 'tB2_lm': _discs(tB2, size),
        'A_lm_68': lmA, 'tB_lm_68': tB, 'tB2_lm_68': tB2,
        'warp_motion': ident, 'warp_motion2': (ident + torch.randn(ident.shape, generator=gen) * 0.01).clamp(-1.1, 1.1),
        'iw_flow': d['flow'], 'if_mask': d['ifmask'],
        'fakeB_static': torch.rand(batch, 1, size, size, generator=gen) * 2 - 1,
        'mask': make_ifmask(batch, gen, size),
        'image_paths': ['synthetic_r%d_s%d_%d' % (rank, seed, i) for i in range(batch)],
    }
    m2 = make_ifmask(batch, gen, size)
    out['if_mask2'] = m2
    out['iw_flow2'] = torch.randn(batch, 2, size, size, generator=gen) * 3.0 * m2
    for k in ('B1', 'B2', 'B3', 'B4'):
        out[k] = torch.rand(batch, 1, size, size, generator=gen) * 2 - 1
    for suf, lo, hi in (('', 20, 40), ('e', 10, 24), ('l', 10, 24)):
        for pre in ('Br', 'B', 'B2'):
            out['%s_mask%s' % (pre, suf)] = _boxes(batch, gen, size, lo, hi)
    win = torch.tensor([[size // 8, size - size // 8, size // 8, size - size // 8]]).repeat(batch, 1)
    for k in ('winA', 'winB', 'winB2', 'winBr'):
        out[k] = win.clone()
    return out


class SyntheticDataset:
    """Iterable of batches (already batched: ``opt.batch_size`` samples per item)."""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--synthetic_batches', type=int, default=8, help='batches per epoch')
        parser.add_argument('--synthetic_seed', type=int, default=1234)
        return parser

    def __init__(self, opt):
        self.opt = opt
        self.rank = getattr(opt, 'rank', 0)

    def __len__(self):
        return self.opt.synthetic_batches * self.opt.batch_size

    def __iter__(self):
        for i in range(self.opt.synthetic_batches):
            yield make_train_batch(self.opt.batch_size, seed=self.opt.synthetic_seed + i, rank=self.rank,
                                   size=self.opt.crop_size)
