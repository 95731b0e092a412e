"""sdr-j-fm_amd -- MI355X-native FM demodulation hot path (drop-in for sdr-j-fm's fmProcessor chain).

The package holds only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/fmx.h),
``build.py`` (hipcc, gfx950) and ``fmx.py`` (ctypes binding + host-side mirror of the reference's
fmProcessor interface).  Import with ``importlib.import_module("sdr-j-fm_amd")``.
"""
import os as _os


from .fmx import (Fmx, FmProcessor, FmxError, load_library, EXPORTS, LIB_PATH)  # noqa: F401
from . import fmx, shard  # noqa: F401
from .filesource import WavFileSource  # noqa: F401
