import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "python-fluid-simulation_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
