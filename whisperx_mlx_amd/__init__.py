"""MI355X-native WhisperX hot path (HIP kernels behind a C ABI).  The public names are the ones the reference exposes
lazily from `whisperx/__init__.py:9-41` for the pieces this package provides; importing the package does not load
torch or the HIP library."""
import importlib


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
