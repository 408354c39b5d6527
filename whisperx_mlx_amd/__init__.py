"""MI355X-native WhisperX hot path (HIP kernels behind a C ABI).  The public names are the ones the reference exposes
lazily from `whisperx/__init__.py:9-41` for the pieces this package provides; importing the package does not load
torch or the HIP library, and does not touch the environment."""
import importlib
import os


def _hw_queues() -> int:
    """Hardware queues the HIP runtime gives this process's streams: GPU_MAX_HW_QUEUES when the user has set it, else the
    runtime's default of 4.  Read, never written: importing the package leaves the environment alone.  (Round 2 set the
    variable to 8 here: four 16-row passes in flight needed four streams with queues of their own.  The default scheduler
    now keeps three wide passes in flight, small jobs four, and the backend asks its streams whether they run side by side
    before it settles -- WhisperHipBackend._default_lanes swaps colliding ones -- so the default 4 queues do: 2 685x with
    GPU_MAX_HW_QUEUES=4 against 2 670-2 697x with 8.)"""
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


HW_QUEUES = _hw_queues()


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
