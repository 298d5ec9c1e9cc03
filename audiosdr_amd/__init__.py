"""audiosdr_amd -- batched AudioSDR update() demodulation chain on AMD Instinct MI355X.

The product is libasdr_hip.so (hand-written HIP kernels for gfx950 behind the C ABI of
include/asdr.h).  This package is only a thin ctypes mirror of that ABI whose class,
`AudioSDRBatch`, keeps the reference's method names (AudioSDR.h:88-156) so that test code
reads like reference usage.  There is no CPU fallback: if the library or a HIP device is
missing, construction raises.
"""
from .binding import (STREAM_BATCH, AGCfast, AGCmedium, AGCoff, AGCslow, ALL, AMmode, BLOCK, CW_LSBmode, CW_USBmode, LSBmode, SAMmode,
                      TAPS, USBmode, WSPRmode, AudioSDRBatch, AsdrError, audio2100, audio2300, audio2500, audio2700,
                      audio2900, audio3100, audio3300, audioAM, audioBypass, audioCW, audioWSPR, library_path,
                      library_sha256, load_library, host_alloc, host_free)
from .front import (NO_DEVICE, AudioGrabberComplex256Batch, AudioIQgeneratorBatch, AudioSDRpreProcessorBatch,  # noqa: E402
                    FRONT_EXPORTS)

__all__ = [n for n in dir() if not n.startswith("_")]
