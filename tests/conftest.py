import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), bound through the product's ctypes binder."""
    from tests.oracle_lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def product():
    """The HIP library; GPU tests fail loudly if it is missing or no GPU is visible."""
    import torch
    from warpx_amd import load_product
    if os.environ.get("WXA_HIP_ON_CPU") == "1":   # logic check without a GPU, see tests/hipcpu
        from tests.oracle_lib import load_hip_on_cpu
        return load_hip_on_cpu()
    lib = load_product()
    assert torch.cuda.is_available(), "gpu-marked test without a visible GPU"
    return lib
