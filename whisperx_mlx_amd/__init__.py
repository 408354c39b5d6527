"""MI355X-native WhisperX hot path (HIP kernels behind a C ABI).  The public names are the ones the reference exposes
lazily from `whisperx/__init__.py:9-41` for the pieces this package provides; importing the package does not load
torch or the HIP library (it does ask the HIP runtime for 8 hardware queues, see _request_hw_queues)."""
import importlib
import os
import sys


def _request_hw_queues(n: int = 8) -> int:
    """The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says
    otherwise) and reads the variable when it initialises the GPU.  The backend keeps one stream per pass in flight: with
    4 queues a fourth pass shares a queue with another stream, its kernels line up behind that stream's, and the job gets
    SLOWER (1 880x against 2 190x with three passes); with 8 queues four passes reach 2 330x (DESIGN.md 5a,
    tools/ab_rows_lanes.py).  So the package asks for 8 queues -- unless the variable is already set (the user's
    choice) or the GPU is already initialised (too late).  Either way the backend asks its streams before it settles
    (WhisperHipBackend._default_lanes) and swaps colliding ones, so this is a help, not a requirement.
    Returns the number of hardware queues the backend may count on."""
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if v is not None:
        try:
            return max(1, int(v))
        except ValueError:
            return 4
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_initialized():
        return 4
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return n


HW_QUEUES = _request_hw_queues()


def _lazy(module, name):
    def call(*args, **kwargs):
        return getattr(importlib.import_module(f"{__name__}.{module}"), name)(*args, **kwargs)
    call.__name__ = name
    call.__doc__ = f"{__name__}.{module}.{name} (imported on first use)"
    return call


load_model = _lazy("backend", "load_model")                 # whisperx/__init__.py:19-21 -> asr.load_model
load_audio = _lazy("backend", "load_audio")                 # :24-26
load_align_model = _lazy("alignment", "load_align_model")   # :9-11
align = _lazy("alignment", "align")                         # :14-16

__all__ = ["load_model", "load_audio", "load_align_model", "align"]
