"""sympgpr_amd -- MI355X (gfx950) implementation of SympGPR's GP training core.

Only what the hot path needs: the ctypes binding of libsympgpr_hip.so (``_lib``), the
device-resident fit (``fit.SympFit``) and the mirror of the reference's Python call surface
(``func``, ``kernels``, ``fortran.sympgpr`` -- python/functions/func.py of the reference).
There is no CPU fallback: every compute call raises if the HIP library or a GPU is missing.
"""
from ._lib import (FAMILIES, SympGPRError, NoDeviceError, lib_path, load_library,  # noqa: F401
                   device_count)

__all__ = ["FAMILIES", "SympGPRError", "NoDeviceError", "lib_path", "load_library", "device_count"]
