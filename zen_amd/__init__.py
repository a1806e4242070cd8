"""zen_amd -- MI355X (gfx950) native HPSS engine behind sevagh/Zen's GPU-backend interface.

The product is the C-ABI shared library zen_amd/libzen_hip.so (sources in zen_amd/csrc, interface in
include/zen_hip.h) plus the C++ host mirror of the reference's libzen in zen_amd/libzen.  This package
is a thin ctypes binding used by tests/ and bench.py; it never falls back to a CPU path: if the HIP
library is missing or no GPU is present the calls raise.
"""
from .lib import (  # noqa: F401
    FREQUENCY, OUTPUT_HARMONIC, OUTPUT_PERCUSSIVE, OUTPUT_RESIDUAL, TIME_ANTICAUSAL, TIME_CAUSAL,
    BoxFilterGPU, DeviceBuffer, Event, FFTC2CWrapperGPU, HPR, HPRIOffline, HPRRealtime, IOGPU, MedianFilterGPU, PinnedHost,
    ZenHipError, ZgException, debug_poke, device_name, init, load, memcheck, run_plan, set_option, synchronize,
)
