import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-GiB inputs; GPU box only")


def _gpu_count():
    try:
        from kfunca_amd import hip_abi
        return hip_abi.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than pass vacuously; plain runs skip.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if _gpu_count() == 0:
        skip = pytest.mark.skip(reason="no GPU visible")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)
