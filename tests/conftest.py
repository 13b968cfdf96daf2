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


# MFS_* variables set OUTSIDE the test run select non-default engine forms (MFS_FUSE_D=0, MFS_RESIDENT=0, MFS_VISC_MARCH=0 ...).
# Tests that assert "the default form is what runs" cannot hold under them: they skip, with the reason, instead of failing
# -- a suite run under a knob is then green or red for real reasons (round 2: 49 spurious failures under MFS_FUSE_D=0).
_HARMLESS_KNOBS = {"MFS_COLLECTIVE_TIMEOUT_S", "MFS_P2P_TIMEOUT_MS", "MFS_BENCH_SHARED_GPU"}
EXTERNAL_KNOBS = sorted(k for k in os.environ if k.startswith("MFS_") and k not in _HARMLESS_KNOBS)


def require_default_engine(what="this test"):
    """skip when an engine knob is set in the environment of the test run (tests set their own knobs with monkeypatch
    AFTER this point; those do not count)"""
    if EXTERNAL_KNOBS:
        pytest.skip(f"{what} asserts the DEFAULT engine form; overridden by {', '.join(EXTERNAL_KNOBS)}")


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
