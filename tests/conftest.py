import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The bridge's client seeds its key generation and encryption streams from the OS; the tests pin them so that two benchmark
# objects (host client / device client, two runs) draw the same keys and noise.
os.environ.setdefault("HE355_SEED", "0x5EA1C0DE")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as ho
    ho.build()
    return ho
