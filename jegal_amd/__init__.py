"""jegal_amd -- MI355X-native JEGAL embedding-extraction engine (HIP/CDNA4 behind a C ABI).

Public surface mirrors the reference's hot-path interface: ``GestSync`` (models/gestsync.py),
``JEGAL`` (models/jegal.py), metric functions (evaluation/evaluate_*.py).  Nothing here falls back
to PyTorch math: without libjegal_hip.so and a HIP device the engine raises.
"""
__all__ = ["GestSync", "JEGAL", "XLMRoberta", "Engine"]


def __getattr__(name):
    if name == "GestSync":
        from .gestsync import GestSync
        return GestSync
    if name == "JEGAL":
        from .jegal import JEGAL
        return JEGAL
    if name == "XLMRoberta":
        from .xlmr import XLMRoberta
        return XLMRoberta
    if name == "Engine":
        from ._lib import Engine
        return Engine
    raise AttributeError(name)
