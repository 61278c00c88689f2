"""monica_amd -- MI355X-native engine for monica's aligner hot path.

Host side: Python mirror of ``monica.genomes.aligner`` (``monica_amd.aligner``) over a thin
ctypes C-ABI (``include/monica_amd.h``) to hand-written gfx950 HIP kernels
(``monica_amd/csrc``).  Importing the package does not load the shared library; the first
call that needs it does, and fails loudly if it has not been built.
"""
__all__ = ["_capi", "synth"]
__version__ = "0.1"
