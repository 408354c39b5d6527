"""Builds libwxhip.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m whisperx_mlx_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so stays next to this file so
that it travels to the GPU box with the source snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwxhip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# No packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in the library.  Measured on MI355X
# (tools/pk_fp32_mfma_probe.hip): a wave running v_pk_fma_f32 while ANOTHER wave on its SIMD issues
# v_mfma_f32_16x16x32_f16 gets wrong results in lanes 48-63.  One engine context alone never paired the two in a
# way that showed; with several contexts in flight the log-mel DFT of one shared CUs with the GEMM tiles of another
# and a few hundred log-mel values per call came out different.  Without the feature the compiler emits two
# v_fma_f32 instead, which is also what issues faster next to MFMAs (attention.hip already avoided them for that).
FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# per-file additions: attention.hip keeps scalar f32 VALU ops unpacked (see the note in attn_full_kernel)
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("wxhip.h", "wxhip_test.h")]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime()):
        return obj
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    return obj


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    newest = max([os.path.getmtime(s) for s in srcs] + [_deps_mtime()])
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {LIB} from {len(srcs)} sources", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
