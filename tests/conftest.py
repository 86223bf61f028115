import torch  # noqa: F401  (FIRST: torch bundles its own HIP/HSA runtime under un-versioned NEEDED names; loaded after
#                       libcammiq_hip.so pulled in /opt/rocm's copies, a second runtime would come up and see no GPU)
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the product library and the oracle exist (no-op when already built)."""
    import subprocess
    from cammiq_amd import binding
    if not os.path.exists(binding.lib_path()):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cammiq_amd", "csrc"), "-j4"])
    import oracle_lib
    oracle_lib.lib()
    yield
