"""iefvad_amd -- MI355X (gfx950) implementation of IEF-VAD's image-event fusion inference path.

The package is a thin Python host over `libiefvad.so` (hand-written HIP kernels behind the C ABI
of include/iefvad.h).  Importing it does not load the library; the first forward does, and fails
loudly if the library is missing -- there is no CPU fallback.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
