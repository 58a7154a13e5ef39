#!/usr/bin/env python3
"""Golden vectors of the landmark -> motion grid map from the REFERENCE's own
``cal_motion256`` (Module2/data/umlvdfw_test_dataset.py:67-81), imported read-only with import-time stubs for the
packages this image lacks (cv2, torchvision) and for the module-level ``np.load('faceLmarkLookup.npy')``.

    python tests/golden/make_motion_golden.py          (build container only)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = '/root/reference/Module2'


def landmarks(seed, n=68, size=256):
    """seeded face-like landmark pairs: a jittered lattice inside the image and a small smooth displacement of it"""
    rng = np.random.RandomState(seed)
    gy, gx = np.meshgrid(np.linspace(60, 200, 9), np.linspace(70, 190, 8), indexing='ij')
    base = np.stack([gx.ravel(), gy.ravel()], 1)[:n] + rng.uniform(-3, 3, (n, 2))
    moved = base + rng.uniform(-6, 6, (n, 2)) + np.array([rng.uniform(-4, 4), rng.uniform(-4, 4)])
    return base.astype(np.float64), np.clip(moved, 2, size - 3).astype(np.float64)


def main():
    for name in ['cv2', 'torchvision', 'torchvision.transforms']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    real_load = np.load
    np.load = lambda path, *a, **k: np.zeros((1, 2), dtype=np.int64) if str(path).endswith('faceLmarkLookup.npy') else real_load(path, *a, **k)
    sys.path.insert(0, REF)
    try:
        from data.umlvdfw_test_dataset import cal_motion256
    finally:
        np.load = real_load
    from oracle import motion as om
    out = {}
    for i, seed in enumerate((11, 12)):
        lm0, lm = landmarks(seed)
        ref = cal_motion256(lm0.copy(), lm.copy())
        assert np.abs(om.cal_motion256(lm0, lm) - ref).max() == 0.0
        out['lm0_%d' % i], out['lm_%d' % i], out['motion_%d' % i] = lm0, lm, ref.astype(np.float32)
    path = os.path.join(HERE, 'motion.npz')
    np.savez_compressed(path, **out)
    print('motion.npz %.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
