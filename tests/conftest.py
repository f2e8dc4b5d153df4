import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference_available():
    import ref_shim
    return ref_shim.available()


def pytest_sessionstart(session):
    # the CPU oracle is many small torch ops: a few threads beat the 256-thread default of the GPU box
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
