"""warpx_amd: MI355X-native PIC inner loop (WarpX's FDTD/gather-push/deposition hot path).

The product is the HIP library `libwarpx_amd.so` (C-ABI: include/warpx_amd.h) plus the
C++17 host layer in csrc/host/.  This Python package is the thin driver used by
bench.py and the test-suite: ctypes binding, array containers, initial conditions and
the torch.distributed halo plumbing.
"""
from . import _capi  # noqa: F401
from ._capi import (DEPOSIT_DIRECT, DEPOSIT_ESIRKEPOV, PUSHER_BORIS, PUSHER_VAY,  # noqa: F401
                    WxaError, load_product)

__version__ = "0.1.0"
