"""jpeggpu_amd -- MI355X-native baseline JPEG decoder behind the jpeggpu C API.

The product is the C-ABI shared library (include/jpeggpu/jpeggpu.h, built from jpeggpu_amd/csrc by
hipcc for gfx950). This package is the thin host-side mirror used by the tests and the benchmark:
ctypes bindings with the reference's function names, argument meaning and status codes.
"""
from .api import (  # noqa: F401
    Batch,
    Decoder,
    ImgInfo,
    JpegGpuError,
    Status,
    decode_to_planes,
    fused_tail_timeouts,
    lib,
    parse_headers,
    self_test,
    status_string,
)
