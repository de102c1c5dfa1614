import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the reference's own deprecations, raised on purpose by the legacy API under test
    for msg in ("Setting samples_per_run different to 1", ".*'NoiseModel.runs' is deprecated"):
        config.addinivalue_line("filterwarnings", f"ignore:{msg}:DeprecationWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip:
    # the gpu tests themselves raise when the device or librydemu is missing.
    return
