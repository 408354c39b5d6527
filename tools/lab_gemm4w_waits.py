"""LAB: how much of the four-wave GEMM kernel's k-loop is waiting (tools/build_lab.py env WX_LAB_ENV; WX_GEMM_4W=1 WX_GEMM_STAMPS=1)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WX_GEMM_STAMPS"] = "1"; os.environ["WX_GEMM_4W"] = "1"
from whisperx_mlx_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_bin", "libwxhip_env.so")
import numpy as np, torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.synth import speechlike_audio
B = 16
dims = weights.MODEL_DIMS["large-v3"]
ck = weights.random_checkpoint(dims, seed=0, device="cuda")
eng = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=B, alignment_heads=weights.default_alignment_heads("large-v3", dims))
pcm = torch.from_numpy(speechlike_audio(30.0 * B, seed=1234).reshape(B, 480000)).cuda()
eng.encode(eng.logmel(pcm, torch.full((B,), 480000, dtype=torch.int32, device="cuda")))
torch.cuda.synchronize()
L = _lib.lib()
L.wx_lab_read_gemm_stamps.argtypes = [ctypes.c_void_p]; L.wx_lab_read_gemm_stamps.restype = ctypes.c_int
for label, kind, arg in (("FC1 without GELU (K 1280)", 1, 1), ("FC2 (K 5120)", 6, 0), ("FC1 + GELU", 1, 0)):
    for rep in range(3):
        ms = eng.probe(kind, B, 1, arg)
    st = np.zeros(2 * 3 * 16 * 8, dtype=np.uint64)
    assert L.wx_lab_read_gemm_stamps(st.ctypes.data) == 0
    st = st[:16].reshape(2, 2, 4).astype(np.int64)
    print(f"== {label}: launch {ms * 1e3:.1f} us")
    for bi, blk in enumerate((0, 700)):
        for wi, wv in enumerate((0, 3)):
            tot, wait, bar, nk = st[bi, wi]
            if tot:
                print(f"   block {blk:3d} wave {wv}: k-loop {tot} cycles for {nk} k-tiles of 32 = {tot / nk:.0f} per k-tile (MFMAs alone: 512); waiting at the top {wait} ({wait / tot:.0%}), barrier {bar} ({bar / tot:.0%})")
