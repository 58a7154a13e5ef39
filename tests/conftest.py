import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: takes more than a few seconds on CPU')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == 'f' and z[k].ndim > 0 else z[k]) for k in z.files}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def sd_sha(sd):
    import hashlib
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def linf(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
