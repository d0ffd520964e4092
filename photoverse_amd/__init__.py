"""photoverse_amd: MI355X-native (gfx950) implementation of PhotoVerse's denoising hot path.

Python host layer over ``libphotoverse_hip.so`` (hand-written HIP kernels, C-ABI in ``include/photoverse_hip.h``).
There is no CPU or eager fallback: ops raise if the library or a HIP device is missing.
"""
__version__ = "0.1.0"
