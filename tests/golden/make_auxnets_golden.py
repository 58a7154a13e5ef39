#!/usr/bin/env python3
"""Golden for the mirrors of the three frozen aux networks (tests/golden/auxnets.npz), from the REFERENCE's own classes
(Module2/models/mobilefacenet.py:104-159, facenet.py:200-282, modnet.py:204-236 + backbones/) run in the build container:
state_dict key lists (names, shapes, dtypes, registration order) and outputs for seeded weights and seeded inputs.

    python tests/golden/make_auxnets_golden.py

No checkpoint of these nets is in the reference tree, so both sides load ``seeded_state_scaled`` (fan-in scaled normal weights,
BatchNorm statistics / affine terms away from their defaults, so a swapped or missing layer shows).  Nothing of the
reference's source is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

SEEDS = dict(mobilefacenet=301, sphere20a=302, modnet=303)


def seeded_state_scaled(keys_shapes, seed):
    """Weights both sides load, one draw per tensor in key order: conv / linear weights N(0, 2 / fan_in); PReLU slopes
    0.25 + 0.05 N; BatchNorm weight 1 + 0.1 N, bias 0.1 N, running_mean 0.1 N, running_var 1 + |0.2 N|, counters 0;
    other biases 0.05 N."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape, dtype in keys_shapes:
        if 'num_batches_tracked' in k:
            sd[k] = torch.zeros(shape, dtype=torch.int64)
            continue
        t = torch.randn(shape, generator=g)
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'running_var':
            sd[k] = 1.0 + 0.2 * t.abs()
        elif leaf == 'running_mean':
            sd[k] = 0.1 * t
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            sd[k] = t * (2.0 / fan_in) ** 0.5
        elif 'relu' in k or 'prelu' in k:
            sd[k] = 0.25 + 0.05 * t
        elif leaf == 'weight':                  # 1-d weight that is not a PReLU slope: a BatchNorm scale
            sd[k] = 1.0 + 0.1 * t
        else:
            sd[k] = (0.1 if ('bn' in k or 'norm' in k) else 0.05) * t
    return sd


def keys_of(net):
    return [(k, tuple(v.shape), str(v.dtype)) for k, v in net.state_dict().items()]


def inputs():
    g = torch.Generator().manual_seed(300)
    return dict(mobilefacenet=torch.rand(2, 3, 112, 112, generator=g),
                sphere20a=torch.rand(2, 3, 112, 96, generator=g) * 2 - 1,
                modnet=torch.rand(2, 3, 96, 96, generator=g) * 2 - 1)


def main():
    import warnings
    warnings.filterwarnings('ignore')
    from make_golden import import_reference, save
    import_reference()
    from models.mobilefacenet import MobileFaceNet
    from models.facenet import Sphere20a
    from models.modnet import MODNet
    torch.set_num_threads(8)
    x = inputs()
    out = {}

    def record(tag, net):
        ks = keys_of(net)
        net.load_state_dict(seeded_state_scaled(ks, SEEDS[tag]), strict=True)
        net.eval()
        out[tag + '_keys'] = np.array([k for k, _, _ in ks])
        out[tag + '_shapes'] = np.array([str(s) for _, s, _ in ks])
        out[tag + '_dtypes'] = np.array([d for _, _, d in ks])
        return net

    with torch.no_grad():
        net = record('mobilefacenet', MobileFaceNet([112, 112], 136))
        emb, feat = net(x['mobilefacenet'])
        out.update(mobilefacenet_out=emb, mobilefacenet_feat_sub=feat[:, ::16], mobilefacenet_feat_abs=feat.double().abs().sum())
        net = record('sphere20a', Sphere20a())
        feats = net(x['sphere20a'])
        for i, f in enumerate(feats):
            out['sphere20a_f%d' % i] = f if f.dim() == 2 else f[:, ::16, ::2, ::2]
            out['sphere20a_f%d_abs' % i] = f.double().abs().sum()
        net = record('modnet', MODNet(backbone_pretrained=False))
        sem, det, matte = net(x['modnet'], False)
        _, _, matte_inf = net(x['modnet'], True)
        assert torch.equal(matte, matte_inf)
        out.update(modnet_matte=matte, modnet_semantic=sem, modnet_detail_sub=det[:, :, ::2, ::2])
    save('auxnets.npz', **out)
    print({k: (float(v.abs().max()) if torch.is_tensor(v) else None) for k, v in out.items() if k.endswith(('_out', '_matte', '_f4'))})


if __name__ == '__main__':
    main()
