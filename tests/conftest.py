import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable_reason():
    """None when the `gpu` tests can run here: a HIP device is visible and libmmdfn_hip.so loads."""
    import torch
    if not torch.cuda.is_available():
        return "needs an MI355X (no HIP device visible)"
    try:
        from mm_dfn_amd import _hip
        _hip.lib()
    except Exception as exc:   # a GPU box without the library is a hard error for the product, a skip reason here
        return "libmmdfn_hip.so unavailable: %s" % exc
    return None


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a GPU-less host skips the `gpu` tests instead of failing them; an explicit
    `-m gpu` run on a box WITHOUT a working device/library still fails loudly (the driver must not see a green
    run that executed nothing)."""
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in config.getoption("-m"):
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def reference_available():
    import ref_shim
    return ref_shim.available()


def pytest_sessionstart(session):
    # the CPU oracle is many small torch ops: a few threads beat the 256-thread default of the GPU box
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture
def kernel_variants(monkeypatch):
    """For tests that force a specific kernel variant through the MMDFN_* switches: those exist only in the
    -DMMDFN_TUNING build (lib/libmmdfn_hip_tuning.so), which is selected for the duration of the test; yields
    ``monkeypatch`` for the setenv / delenv calls."""
    from mm_dfn_amd import _hip
    prev = _hip.set_tuning(True)
    yield monkeypatch
    _hip.set_tuning(prev)
