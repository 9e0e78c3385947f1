"""usdm_amd — MI355X (gfx950) native hot path of USDM inference.

Python here is host glue only (device memory, streams, drop-in API surface of the reference's
src/inference.py + src/decoder); all arithmetic on the path runs in hand-written HIP kernels
behind the C-ABI of include/usdm_hip.h (libusdm_hip.so).  There is no CPU fallback: importing
usdm_amd._lib fails loudly when the library is missing.
"""
__version__ = "0.1.0"
