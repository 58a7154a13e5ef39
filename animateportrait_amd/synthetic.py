"""Seeded synthetic inputs with the shapes/ranges of the hot path's input contract.

The reference ships no data; benchmarks and parity tests
therefore use synthetic tensors shaped like what ``GeomGMIFWForeModel.set_input``
(Module2/models/geomgm_ifw_fore_model.py:443-505) receives from
``UMLVDIFWDataset.__getitem__`` (Module2/data/umlvd_ifw_dataset.py:149-428):

* ``input``   B x 3 x S x S in [-1, 1]
* ``land1/2`` B x 1 x S x S in {-1, +1}: 68 filled discs r=3
  (Module2/data/umlvdfw_test_dataset.py:36-41)
* ``motion``  B x S x S x 2 (x, y), identity grid normalised by /127.5-1
  (umlvd_ifw_dataset.py:60-74) plus a small perturbation
* ``flow``    B x 2 x S x S pixels (dx, dy), ``ifmask`` B x 1 x S x S in [0, 1]
  (netF outputs, geomgm_ifw_fore_model.py:69-84)

Everything is generated on a CPU ``torch.Generator`` so the CPU baseline and the
GPU see bit-identical values.
"""
import torch
import torch.nn.functional as F


def _discs(points, size, radius=3):
    """points (B, P, 2) as (x, y) -> (B,1,size,size) in {-1,+1}."""
    b = points.shape[0]
    yy = torch.arange(size, dtype=torch.float32).view(1, 1, size, 1)
    xx = torch.arange(size, dtype=torch.float32).view(1, 1, 1, size)
    px = points[..., 0].view(b, -1, 1, 1)
    py = points[..., 1].view(b, -1, 1, 1)
    hit = ((xx - px) ** 2 + (yy - py) ** 2 <= radius * radius).any(dim=1, keepdim=True)
    return hit.float() * 2 - 1


def make_landmarks(batch, gen, size=256, n_points=68):
    # jittered 9 x 8 lattice (first n_points cells): distinct, well separated control points -- coincident
    # landmarks would make the TPS system of the warp losses singular (sparse_image_warp.py:124-128 drops to pdb)
    cols, rows = 9, 8
    assert n_points <= cols * rows
    step_x, step_y = (size * 0.8) / cols, (size * 0.8) / rows
    idx = torch.arange(n_points)
    cx = (idx % cols).float() * step_x + size * 0.1 + step_x / 2
    cy = (idx // cols).float() * step_y + size * 0.1 + step_y / 2
    grid = torch.stack([cx, cy], -1).unsqueeze(0).expand(batch, -1, -1)
    jit = (torch.rand(batch, n_points, 2, generator=gen) - 0.5) * torch.tensor([step_x, step_y]) * 0.5
    return (grid + jit).round()


def jitter_landmarks(base, gen, sigma=2.0, size=256):
    return (base + torch.randn(base.shape, generator=gen) * sigma).round().clamp(0, size - 1)


def make_ifmask(batch, gen, size=256):
    yy = torch.arange(size, dtype=torch.float32).view(1, size, 1)
    xx = torch.arange(size, dtype=torch.float32).view(1, 1, size)
    c = torch.rand(batch, 2, generator=gen) * (size * 0.25) + size * 0.375
    r = torch.rand(batch, 2, generator=gen) * (size * 0.15) + size * 0.2
    m = ((((xx - c[:, 0].view(-1, 1, 1)) / r[:, 0].view(-1, 1, 1)) ** 2
          + ((yy - c[:, 1].view(-1, 1, 1)) / r[:, 1].view(-1, 1, 1)) ** 2) <= 1.0).float().unsqueeze(1)
    m = F.interpolate(m, size=(size // 2, size // 2), mode='bilinear', align_corners=True)
    return F.interpolate(m, size=(size, size), mode='bilinear', align_corners=True).clamp(0, 1)


def make_generator_inputs(batch, seed=1234, rank=0, size=256):
    """Returns a dict of CPU fp32 tensors: input, land1, land2, motion, flow, ifmask,
    plus the landmark sets lm1/lm2 (B,68,2) (x,y) the discs were drawn from."""
    gen = torch.Generator().manual_seed(seed + rank)
    inp = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    lm1 = make_landmarks(batch, gen, size)
    lm2 = jitter_landmarks(lm1, gen, 2.0, size)
    land1 = _discs(lm1, size)
    land2 = _discs(lm2, size)
    lin = torch.linspace(-1, 1, size)
    ident = torch.stack([lin.view(1, size).expand(size, size), lin.view(size, 1).expand(size, size)], -1)
    motion = (ident.unsqueeze(0) + torch.randn(batch, size, size, 2, generator=gen) * 0.02).clamp(-1.1, 1.1)
    ifmask = make_ifmask(batch, gen, size)
    flow = torch.randn(batch, 2, size, size, generator=gen) * 3.0 * ifmask
    return dict(input=inp, land1=land1, land2=land2, motion=motion.contiguous(),
                flow=flow.contiguous(), ifmask=ifmask.contiguous(), lm1=lm1, lm2=lm2)


GEN_ARG_ORDER = ('input', 'land1', 'land2', 'motion', 'flow', 'ifmask')


def generator_args(d):
    """Positional args of ``netG(input, land1, land2, motion, flow, ifmask)`` (networks.py:1315)."""
    return tuple(d[k] for k in GEN_ARG_ORDER)
