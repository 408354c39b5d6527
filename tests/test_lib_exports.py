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


def test_device_code_has_no_packed_fp32_ops(tmp_path):
    """v_pk_fma_f32 & co. return wrong bits in lanes 48-63 while another wave of the SIMD issues MFMAs
    (tools/pk_fp32_mfma_probe.hip, profiles/r02_pk_fp32_mfma_probe.txt): with several engine contexts in
    flight that changed log-mel values from call to call.  The shipped gfx950 code must not contain them."""
    import shutil
    import subprocess
    from whisperx_mlx_amd.build import build_library
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        import pytest
        pytest.skip("llvm-objdump not in this image")
    so = shutil.copy(build_library(), tmp_path / "libwxhip.so")
    subprocess.run([objdump, "--offloading", os.path.basename(so)], cwd=tmp_path, check=True, capture_output=True)
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert objs, "no gfx950 code object found in libwxhip.so"
    n_mfma = 0
    for f in objs:
        asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        bad = sorted(set(re.findall(r"\bv_pk_[a-z0-9]+_f32\b", asm)))
        assert not bad, f"{f}: packed-fp32 VALU ops in device code: {bad}"
        n_mfma += len(re.findall(r"\bv_mfma_", asm))
    assert n_mfma > 1000          # the disassembly really covered the kernels


def test_importing_the_package_leaves_the_environment_alone(monkeypatch):
    """whisperx_mlx_amd/__init__.py reads GPU_MAX_HW_QUEUES (the user's, else the runtime's default 4) and never writes it;
    whatever it says, the backend asks its streams before it settles on the passes in flight (tests/test_gpu_backend.py)."""
    import importlib
    import whisperx_mlx_amd as pkg
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    importlib.reload(pkg)
    assert "GPU_MAX_HW_QUEUES" not in os.environ and pkg.HW_QUEUES == 4
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    assert pkg._hw_queues() == 6 and os.environ["GPU_MAX_HW_QUEUES"] == "6"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "many")
    assert pkg._hw_queues() == 4


def test_ctypes_structs_have_the_header_layout(tmp_path):
    """the Python mirrors of the boundary's structs (whisperx_mlx_amd/_lib.py) against the header itself: a C program
    compiled with gcc prints sizeof and every field's offset of wx_decode_opts / wx_model_dims / wx_w2v_dims"""
    import subprocess
    from whisperx_mlx_amd import _lib
    pairs = (("wx_decode_opts", _lib.DecodeOpts), ("wx_model_dims", _lib.ModelDims), ("wx_w2v_dims", _lib.W2vDims),
             ("wx_tuning", _lib.Tuning))
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "wxhip_test.h"', 'int main(void) {',
           'wx_tuning d = WX_TUNING_DEFAULTS;',
           'printf("defaults %d %d %d %d %d %d %d\\n", d.use_graph, d.check_every, d.cross_split, d.step_variant, d.fc2_tile_n, d.profile_launches, d.max_steps_ahead);']
    for cname, cls in pairs:
        src.append(f'printf("{cname} %zu", sizeof({cname}));')
        for f, _t in cls._fields_:
            src.append(f'printf(" %zu", offsetof({cname}, {f}));')
        src.append('printf("\\n");')
    src += ['return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    dflt = _lib.Tuning.defaults()
    assert [int(v) for v in out[0].split()[1:]] == [getattr(dflt, f) for f, _t in _lib.Tuning._fields_]      # WX_TUNING_DEFAULTS
    # the boundary header by itself knows the tuning struct by name only
    hdr = open(os.path.join(ROOT, "include", "wxhip.h")).read()
    assert "struct wx_tuning;" in hdr and "step_variant" not in hdr and "fc2_tile_n" not in hdr and "max_steps_ahead" not in hdr
    for line, (cname, cls) in zip(out[1:], pairs):
        vals = line.split()
        assert vals[0] == cname
        assert int(vals[1]) == ctypes.sizeof(cls), cname
        assert [int(v) for v in vals[2:]] == [getattr(cls, f).offset for f, _t in cls._fields_], cname


def test_the_shipped_library_reads_no_environment_variable():
    """the lab knobs of the A/B scripts (WX_DL_POLL, WX_NO_WIDE_GEMV, WX_GEMM_*, ...) are compiled in by -DWX_LAB_ENV only
    (csrc/common.h, tools/build_lab.py): the product library does not import getenv at all"""
    import subprocess
    from whisperx_mlx_amd.build import build_library
    out = subprocess.run(["nm", "-D", "--undefined-only", build_library()], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out
    src = os.path.join(ROOT, "whisperx_mlx_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")) and f != "common.h":
            assert "getenv(" not in open(os.path.join(src, f)).read(), f
