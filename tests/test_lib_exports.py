"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/wxhip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

from tests.conftest import ROOT


def _declared(headers=("wxhip.h", "wxhip_test.h")):
    """entry points declared by the boundary header and by the test / measurement header"""
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names.update(re.findall(r"\b(wx_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_builds_and_exports_header_symbols():
    import torch  # noqa: F401  (its HIP runtime must be loaded before libwxhip.so)
    from whisperx_mlx_amd.build import build_library
    lib = build_library()
    so = ctypes.CDLL(lib)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(so, n), f"{n} declared in wxhip.h but not exported"


def test_python_binding_table_matches_header():
    from whisperx_mlx_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()
    _lib.lib()          # resolves every symbol with its prototype
    # the drop-in boundary itself stays small: the building blocks live in wxhip_test.h
    boundary = _declared(("wxhip.h",))
    assert len(boundary) <= 26 and not any(n in boundary for n in ("wx_gemm_f16", "wx_skinny_f16", "wx_probe", "wx_sample_step"))


def test_no_gpu_means_loud_failure():
    import torch
    import pytest
    from whisperx_mlx_amd import _lib, weights
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from whisperx_mlx_amd.engine import WhisperHipEngine
    with pytest.raises(_lib.WxError):
        WhisperHipEngine(weights.MODEL_DIMS["tiny"], {}, max_batch=1)
    # and the C side refuses too (no CPU fallback anywhere)
    d = _lib.ModelDims(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4)
    h = ctypes.c_void_p()
    assert _lib.lib().wx_create(0, ctypes.byref(d), 1, ctypes.byref(h)) != 0
