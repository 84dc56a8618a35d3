import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU wave-emulator runs (seconds each)")
    config.addinivalue_line("markers", "expects_generic_fallback: the test provokes the fallback to the runtime-dimension kernel")


def pytest_collection_modifyitems(config, items):
    # On the GPU a handle that ends up on the runtime-dimension kernel although a specialisation was eligible (object
    # rejected by mpcqp_prepare's comparison, compiler failure, distrusted cache) FAILS its test: the library reports it
    # with a RuntimeWarning (api.Handle.prepare), which the GPU tests treat as an error.
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("expects_generic_fallback"):
            it.add_marker(pytest.mark.filterwarnings("error:mpcqp. specialised kernel not available:RuntimeWarning"))


@pytest.fixture(scope="session")
def hiplib():
    """The product library (HIP).  Fails loudly when it is not built -- there is no CPU fallback."""
    import mpcqp
    return mpcqp.load_library()
