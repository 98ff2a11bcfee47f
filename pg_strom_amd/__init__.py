"""
pg_strom_amd -- MI355X-native execution path for PG-Strom's GpuScan /
GpuHashJoin / GpuPreAgg.  The product is pg_strom_amd/libstrom_hip.so
(C ABI in include/); this package is its Python host side.
"""
from . import _lib            # noqa: F401  (fails loudly when the .so is absent)
from . import kds, runtime     # noqa: F401
from .gpuscan import GpuScan   # noqa: F401

__all__ = ["kds", "runtime", "GpuScan"]
