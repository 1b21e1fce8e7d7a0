import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import binding
    binding.lib()
    binding.set_threads(min(8, os.cpu_count() or 1))
    return binding


@pytest.fixture(scope="session")
def dev():
    """A usable MI355X through the product's own HIP runtime (no torch on the data path)."""
    from emfusion_amd import devmem
    if devmem.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    devmem.set_device(0)
    return devmem
