"""shape_based_matching_amd — MI355X-native LINE-2D / linemod matching engine.

The product is ``libsbm_hip.so`` (hand-written HIP kernels for gfx950 behind the
C ABI of ``include/sbm.h``) plus the C++ ``line2Dup::Detector`` facade of
``include/line2Dup.h``.  This Python package is the thin host mirror used by the
tests and ``bench.py``: ctypes bindings (``capi``), template containers and the
OpenCV-FileStorage YAML subset (``templates``), the Detector mirror
(``detector``), template sharding + match-list gathering (``sharding``) and
seeded synthetic inputs (``synth``).
"""
from .templates import FEATURE_DTYPE, LEVEL_DTYPE, MATCH_DTYPE, TemplateSet  # noqa: F401

__all__ = ["TemplateSet", "FEATURE_DTYPE", "LEVEL_DTYPE", "MATCH_DTYPE"]
