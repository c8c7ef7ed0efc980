"""cl_ops_amd — MI355X-native sort/scan primitives behind the cl_ops C API.

The product is cl_ops_amd/lib/libcl_ops_hip.so (C-ABI, see include/): host
drivers in C (cl_ops_amd/csrc) + hand-written HIP kernels for gfx950
(cl_ops_amd/csrc/hip). This package is only the ctypes view of that library
used by the tests and bench.py; importing it fails loudly when the library has
not been built — there is no CPU or PyTorch fallback path.
"""
from . import _hip  # noqa: F401  (raises ImportError if the .so is missing)
from .api import (CloError, Context, Queue, Buffer, Sorter, Scanner, Profiler, HipEventTimer,  # noqa: F401
                  ShardTransport, ShardSort, CLO_TYPES, clo_type, wait_for_events)

__all__ = ["CloError", "Context", "Queue", "Buffer", "Sorter", "Scanner", "Profiler", "HipEventTimer",
           "ShardTransport", "ShardSort",
           "CLO_TYPES", "clo_type", "wait_for_events"]
