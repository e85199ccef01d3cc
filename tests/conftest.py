import os
import sys

# before the HIP runtime initialises (same switch fastvim_amd/__init__.py and bench.py set)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _reset_flat_training_switches():
    """FlatTrainingState switches the kernels' wrappers to deferred reductions / grouped weight gradients process-wide
    until its close(); a test that forgets to close one must not change what the next test measures."""
    yield
    import sys
    mo, mf = sys.modules.get("fastvim_amd.mixer_ops"), sys.modules.get("fastvim_amd.mamba_simple_faster")
    if mo is not None:
        mo._Deferred.jobs = []
        mo._Deferred.enabled = False
    if mf is not None:
        mf._GroupedWgrad.jobs = []
        mf._GroupedWgrad.sums = []
        mf._GroupedWgrad.enabled = False
        mf._SideStream.enabled = False
