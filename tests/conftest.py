import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ood-gan-inversion_amd')
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    import torch

    def load(name):
        z = np.load(os.path.join(GOLDEN, name))
        return {k: torch.from_numpy(z[k]) for k in z.files}

    return load


@pytest.fixture
def tunable():
    """Set a dispatch tunable of liboodgan_hip.so (include/oodgan.h, oodgan_set_tunable) for one test; restored afterwards."""
    from oodgan import _lib
    saved = {}

    def set_(name, value):
        old = _lib.set_tunable(name, value)
        saved.setdefault(name, old)

    yield set_
    for name, old in saved.items():
        _lib.set_tunable(name, old)
