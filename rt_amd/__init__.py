"""rt_amd -- MI355X-native trace loop behind the call surface of Chang-Liu-TAMU/RT's PbrtEngine.

The product is the C-ABI library ``librt_amd.so`` (``include/rt_abi.h``); this package is the thin
Python host side over it (ctypes), mirroring ``PbrtEngine`` (reference include/RT/pbrt_engine.h:51-203).
There is no CPU fallback: importing :mod:`rt_amd.engine` fails loudly if the library is not built.
"""
from .engine import (  # noqa: F401
    PbrtEngine,
    RtError,
    RtStats,
    TRI_DTYPE,
    NODE_DTYPE,
    MAT_DTYPE,
    default_materials,
    load_library,
    write_ppm,
)
