import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the tests choose per decoder; an inherited opt-in would change what the default-path tests cover
os.environ.pop("JPEGGPU_DEVICE_SCAN", None)
os.environ.pop("JPEGGPU_SUBSEQ_BYTES", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def photo_bytes():
    """The reference's only test image (images/IMG_6510.JPG), committed as a fixture."""
    with open(os.path.join(GOLDEN, "IMG_6510.JPG"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def gpu_lib():
    """Build (if stale) and load the C-ABI library; GPU tests fail loudly if it cannot be loaded."""
    import jpeggpu_amd
    from jpeggpu_amd import build as jbuild

    jbuild.build()
    return jpeggpu_amd.lib()
