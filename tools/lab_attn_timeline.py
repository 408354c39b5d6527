"""LAB (round 5): where a key tile of the encoder attention (attn_full_kernel) spends its cycles.  Needs the lab library
(python tools/build_lab.py env WX_LAB_ENV) and WX_ATTN_STAMPS=1: the first wave of units 0 / 1500 / 3000 stamps s_memtime
(core cycles) per key tile:  0 start | 1 scores in registers | 2 exponentials done | 3 PV MFMAs issued | 4 next tile parked
in LDS (vmcnt wait + ds_writes) | 5 barrier passed.    python tools/lab_attn_timeline.py [rows]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WX_ATTN_STAMPS"] = "1"
from whisperx_mlx_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_bin", "libwxhip_env.so")
import numpy as np
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.synth import speechlike_audio

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dims = weights.MODEL_DIMS["large-v3"]
ck = weights.random_checkpoint(dims, seed=0, device="cuda")
eng = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=B, alignment_heads=weights.default_alignment_heads("large-v3", dims))
pcm = torch.from_numpy(speechlike_audio(30.0 * B, seed=1234).reshape(B, 480000)).cuda()
eng.encode(eng.logmel(pcm, torch.full((B,), 480000, dtype=torch.int32, device="cuda")))
torch.cuda.synchronize()
L = _lib.lib()
L.wx_lab_read_attn_stamps.argtypes = [ctypes.c_void_p]
L.wx_lab_read_attn_stamps.restype = ctypes.c_int
names = ["scores (10 MFMA)", "exponentials", "P cvt + 8 MFMA", "park next tile", "barrier"]
for rep in range(3):
    ms = eng.probe(2, B, 1)
st = np.zeros(3 * 24 * 8, dtype=np.uint64)
assert L.wx_lab_read_attn_stamps(st.ctypes.data) == 0
st = st.reshape(3, 24, 8).astype(np.int64)
units = ((1500 + 127) // 128) * dims.n_audio_head * B
print(f"encoder attention, {B} rows: launch {ms * 1e3:.1f} us, {units} blocks of 4 waves, 24 key tiles each; core cycles per key tile (first wave of the block)")
for bi, unit in enumerate((0, 1500, 3000)):
    s = st[bi]
    if s[0, 0] == 0:
        continue
    d = np.diff(s[:, :6], axis=1).astype(float)          # [tile][phase]
    period = np.diff(s[:, 0]).astype(float)
    print(f"  unit {unit:5d}: median cycles  " + "  ".join(f"{n} {np.median(d[1:23, k]):6.0f}" for k, n in enumerate(names)) +
          f"   tile period {np.median(period[1:22]):6.0f}   block: {(s[23, 5] - s[0, 0])} cycles for 24 tiles")
    for t in (0, 1, 2, 12, 23):
        print(f"      tile {t:2d}: " + "  ".join(f"{x:6.0f}" for x in d[t]))
