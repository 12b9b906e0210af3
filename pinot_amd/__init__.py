"""pinot_amd -- MI355X-native segment scan-filter-aggregate engine for Apache Pinot's per-segment hot path.

The product is the C-ABI shared library `pinot_amd/csrc/libpinot_gpu.so` (include/pinot_gpu.h) and the C++
host mirror `libpinot_host.so`; this package is the ctypes plumbing used by tests and bench.py.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi"]
