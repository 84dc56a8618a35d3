import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU wave-emulator runs (seconds each)")


@pytest.fixture(scope="session")
def hiplib():
    """The product library (HIP).  Fails loudly when it is not built -- there is no CPU fallback."""
    import mpcqp
    return mpcqp.load_library()
