import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # tests/_margins.py

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long tail of the GPU suite (tests/_slow.py): deselected unless -m names `slow` or SPCIES_RUN_SLOW=1")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """A GPU session initialises torch's HIP runtime BEFORE the library's first call: the PyTorch wheel bundles its own ROCm user space,
    and initialised second (after libspcies_hip.so has brought up the installed one) it reports "No HIP GPUs are available" - the tests
    that hand torch device buffers to the C-ABI would then depend on which test ran before them."""
    import _slow
    for item in items:
        if _slow.is_slow(item.nodeid):
            item.add_marker(pytest.mark.slow)
    if "slow" not in (config.getoption("-m") or "") and os.environ.get("SPCIES_RUN_SLOW", "0") != "1":
        keep, drop = [], []
        for item in items:
            (drop if item.get_closest_marker("slow") else keep).append(item)
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep
    if any(item.get_closest_marker("gpu") for item in items):
        try:
            import torch
            if torch.cuda.device_count() > 0 and torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # noqa: BLE001 - a CPU box: the gpu tests are deselected or fail on their own terms
            pass


def pytest_sessionfinish(session, exitstatus):
    """Worst differences seen by the parity helpers -> gpurun_out/parity_margins.json (tests/_margins.py)."""
    import _margins
    _margins.dump(ROOT)
