"""iefvad_amd -- MI355X (gfx950) implementation of IEF-VAD's image-event fusion path (inference, and the training step around it).

A thin Python host over `libiefvad.so` (hand-written HIP kernels behind the C ABI of
include/iefvad.h).  Importing the package does not load the library; the first forward does, and
fails loudly if the library is missing -- there is no CPU fallback.
"""
from . import harness, layers, lib, losses, module, synth, trainer  # noqa: F401
from .model import MMFMIL, OUTPUT_KEYS  # noqa: F401

__all__ = ["MMFMIL", "OUTPUT_KEYS", "harness", "layers", "lib", "losses", "module", "synth", "trainer"]
