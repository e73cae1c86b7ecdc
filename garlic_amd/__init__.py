"""garlic_amd -- MI355X-native Phase-I window LOD engine for GARLIC (szpiech/garlic).

The product is garlic_amd/libgarlic_hip.so (HIP kernels + C ABI, include/garlic_hip.h) and the
C++ host adapter in garlic_amd/host.  The Python modules here are thin ctypes plumbing for the
tests and bench.py.
"""
from . import abi  # noqa: F401
