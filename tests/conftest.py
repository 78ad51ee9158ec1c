import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "first_gpu_run: no MI355X has executed this test yet (runs last)")


def pytest_collection_modifyitems(config, items):
    first = [it for it in items if it.get_closest_marker("first_gpu_run")]
    if not first:
        return
    if os.environ.get("WXA_SKIP_FIRST_GPU_RUN") == "1":
        for it in first:
            it.add_marker(pytest.mark.skip(reason="WXA_SKIP_FIRST_GPU_RUN=1"))
    rest = [it for it in items if not it.get_closest_marker("first_gpu_run")]
    items[:] = rest + first


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
