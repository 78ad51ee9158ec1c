import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "first_gpu_run: no MI355X has executed this test yet (runs last)")


def pytest_sessionstart(session):
    """Thread teams sized to what the cgroup grants (tests/oracle_lib.py::available_cpus): the GPU boxes show 256 logical
    CPUs under a quota of 16, and a team of 256 then costs 50-100x."""
    from tests.oracle_lib import available_cpus
    n = available_cpus()
    for var in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(n))
    if "OMP_NUM_THREADS" not in os.environ:
        try:
            import torch
            torch.set_num_threads(n)
        except Exception:   # noqa: BLE001 -- torch is only plumbing here
            pass


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


def _guarded_device_tensors():
    """WXA_HIP_ON_CPU=1 HIPCPU_GUARD_PAGES=1: the "device" tensors the tests allocate (torch.zeros(..., device="cpu:0"))
    come from the execution model's hipMalloc, i.e. they end on an inaccessible page: a kernel that reads or writes
    past a buffer it was handed faults instead of touching whatever the allocator placed next to it."""
    import ctypes as C
    import torch
    from tests.oracle_lib import load_hip_on_cpu
    dll = load_hip_on_cpu()._dll
    hip_malloc = getattr(dll, "_Z9hipMallocPPvm")
    hip_malloc.restype, hip_malloc.argtypes = C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]
    plain_zeros = torch.zeros

    def zeros(*size, **kw):
        if str(kw.get("device", "cpu")) != "cpu:0":
            return plain_zeros(*size, **kw)
        dtype = kw.get("dtype", torch.float32)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        n = 1
        for v in shape:
            n *= int(v)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        if nbytes == 0:
            return plain_zeros(*size, **kw)
        # end-aligned only up to 256 B: pad the request so that the tensor's last element is the buffer's last
        nalloc = (nbytes + 255) // 256 * 256
        p = C.c_void_p()
        assert hip_malloc(C.byref(p), nalloc) == 0
        buf = (C.c_uint8 * nalloc).from_address(p.value)
        t = torch.frombuffer(buf, dtype=torch.uint8)[nalloc - nbytes:].view(dtype).reshape(shape)
        t.zero_()
        return t

    torch.zeros = zeros


if os.environ.get("WXA_HIP_ON_CPU") == "1" and os.environ.get("HIPCPU_GUARD_PAGES"):
    _guarded_device_tensors()
