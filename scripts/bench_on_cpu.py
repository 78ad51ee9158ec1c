"""Dry run of bench.py's control flow where there is no GPU (tests only; never a measurement).

bench.py is executed unmodified; this wrapper (i) hands it the library built from the same .hip sources on the
CPU execution model of tests/hipcpu instead of libwarpx_amd.so, (ii) redirects the torch "cuda" device strings and
the torch.cuda calls bench.py makes to host memory / no-ops.  What it checks: argument handling, particle set-up,
the timed loop, the phase pass, the roofline / kernels / config assembly and the JSON line.  The numbers it prints
are meaningless.

    python scripts/bench_on_cpu.py --ncell 16 --steps 2 --warmup 1 --preroll 2 --no-cpu-baseline
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WXA_HIP_ON_CPU"] = "1"

import torch  # noqa: E402


def _host(dev):
    return "cpu:0" if dev is not None and str(dev).startswith("cuda") else dev


def _redirect(fn):
    def wrapped(*a, **kw):
        if "device" in kw:
            kw["device"] = _host(kw["device"])
        return fn(*a, **kw)
    return wrapped


for name in ("arange", "empty", "zeros", "randn", "tensor", "ones"):
    setattr(torch, name, _redirect(getattr(torch, name)))
_Generator = torch.Generator
torch.Generator = lambda device=None: _Generator(device="cpu")


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
torch.cuda.empty_cache = lambda: None
torch.cuda.Event = _Event

import warpx_amd  # noqa: E402
from tests.oracle_lib import load_hip_on_cpu  # noqa: E402

warpx_amd.load_product = load_hip_on_cpu

# N > 1 (python -m torch.distributed.run --nproc-per-node N scripts/bench_on_cpu.py --gpus N ...): gloo instead of RCCL,
# host buffers in the torch transport; the in-library RCCL transport is not in the CPU build, so bench.py takes its
# documented fall-back to the torch.distributed callbacks.
import torch.distributed as _dist  # noqa: E402
import warpx_amd.distributed as _wd  # noqa: E402

_init = _dist.init_process_group
_dist.init_process_group = lambda backend=None, **kw: _init("gloo", **{k: v for k, v in kw.items() if k != "device_id"})
_Torch = _wd.TorchBrickTransport
_wd.TorchBrickTransport = lambda on_device=True: _Torch(on_device=False)

if __name__ == "__main__":
    # --script <path>: another bench script of the same kind (scripts/bench_lwfa_boosted.py) instead of bench.py
    if len(sys.argv) > 2 and sys.argv[1] == "--script":
        import importlib.util
        path = sys.argv[2]
        del sys.argv[1:3]
        spec = importlib.util.spec_from_file_location("bench_script", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.main()
    else:
        import bench  # noqa: E402
        bench.main()
