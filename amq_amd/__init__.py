"""amq_amd -- MI355X-native mixed-precision (2/3/4-bit) dequantize-matmul path
for AMQ, behind the reference's QuantLinear / prepare_for_inference surface.

The compute lives in ``libamq_hip.so`` (hand-written HIP for gfx950, C ABI in
``include/amq_hip.h``); this package is the host-side mirror of the reference
interface.  Importing the package does not load the library; the first op does
and raises if it has not been built (no CPU fallback).
"""
__version__ = "0.1.0"
