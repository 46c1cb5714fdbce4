import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads():
    """The CPU oracle runs in many tests; on the 256-thread hosts of the GPU boxes torch's default (all threads) is 3-5x SLOWER than 16
    (profiles/r03_cpu_thread_sweep.json: 5.7 s at 16 threads, 16 s at 64, 30 s at 128 for one 160 x 160 crop)."""
    import torch
    from bfsr_amd import hostenv
    torch.set_num_threads(max(1, min(16, hostenv.effective_cpus())))       # (and never more than the container's CPU quota: hostenv.py)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
