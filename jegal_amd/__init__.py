"""jegal_amd -- MI355X-native JEGAL embedding-extraction engine (HIP/CDNA4 behind a C ABI).

Public surface mirrors the reference's hot-path interface: ``GestSync`` (models/gestsync.py),
``JEGAL`` (models/jegal.py), metric functions (evaluation/evaluate_*.py).  Nothing here falls back
to PyTorch math: without libjegal_hip.so and a HIP device the engine raises.
"""
#: hardware queues a streaming caller should ask the HIP runtime for (see want_hw_queues)
RECOMMENDED_HW_QUEUES = 8


def want_hw_queues(n=RECOMMENDED_HW_QUEUES):
    """The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
    serialise.  A streaming caller has more than four: H2D, D2H and compute streams plus the engine's two internal lanes -- measured
    on MI355X (tools/stream_timeline.py): with an unlucky assignment the upload of batch k+1 queues behind the compute of batch k and
    GestureStreamer drops from 2 150 to 1 230 clips/s.  The runtime reads the variable when it initialises, i.e. at the first HIP
    call, so an APPLICATION calls this before it touches the GPU (bench.py and the CLI drivers do); importing the package does not
    change the process environment.  Never overrides the caller's own setting.  Returns the value in force."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(n)))
    return os.environ["GPU_MAX_HW_QUEUES"]


__all__ = ["GestSync", "JEGAL", "XLMRoberta", "Engine", "want_hw_queues"]


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
