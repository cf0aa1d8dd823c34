"""jpeg-quantsmooth coefficient recovery on AMD MI355X (gfx950).

Host-side Python mirror of the flat C ABI in include/jpegqs_hip.h.  The product
is the shared library `libjpegqs_hip.so` (hand-written HIP kernels + C++ host
driver, built in-tree by csrc/Makefile); this package only binds it (ctypes)
and provides seeded synthetic inputs.  There is no CPU compute path here: if
the library is missing, loading fails loudly.
"""
from . import synth  # noqa: F401
from .hipqs import (  # noqa: F401
    FLAGS, HipQS, QsHipError, flags_for_quality, lib_path, load_library,
)

__all__ = ["synth", "FLAGS", "HipQS", "QsHipError", "flags_for_quality", "lib_path", "load_library"]
