"""MI355X-native implementation of the baryon_painter CVAE hot path.

The arithmetic lives in ``csrc/`` (hand-written HIP for gfx950 behind the C ABI
declared in ``include/bp_hip.h``); this package is the host-side mirror of the
reference's Python surface (``painter.CVAEPainter``, ``models.cvae.CVAE``).
"""
__version__ = "0.1.0"
