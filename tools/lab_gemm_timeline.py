"""LAB (round 5): where a tile of the tile-pipelined 256 x 256 GEMM spends its time.  Needs the lab library
(python tools/build_lab.py env WX_LAB_ENV) and WX_GEMM_STAMPS=1: gemm_pipe_kernel then stamps s_memrealtime (10 ns ticks) along
the tiles of blocks 0 and 128, waves 0 / 4 / 7:
  0 tile start | 1 k-loop done | 2 wave groups met, LDS free | 3 epilogue done (stores issued) | 4 vmcnt(0) lgkmcnt(0) | 5 seam barrier
   python tools/lab_gemm_timeline.py [rows]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WX_GEMM_STAMPS"] = "1"
from whisperx_mlx_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_bin", "libwxhip_env.so")
import numpy as np
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.synth import speechlike_audio

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CAPS = [int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]      # wx_set_encoder_cap: blocks (= CUs) the GEMM may use; 0 = all
dims = weights.MODEL_DIMS["large-v3"]
ck = weights.random_checkpoint(dims, seed=0, device="cuda")
eng = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=B, alignment_heads=weights.default_alignment_heads("large-v3", dims))
pcm = torch.from_numpy(speechlike_audio(30.0 * B, seed=1234).reshape(B, 480000)).cuda()
mel = eng.logmel(pcm, torch.full((B,), 480000, dtype=torch.int32, device="cuda"))
eng.encode(mel)
torch.cuda.synchronize()
L = _lib.lib()
L.wx_lab_read_gemm_stamps.argtypes = [ctypes.c_void_p]
L.wx_lab_read_gemm_stamps.restype = ctypes.c_int
names = ["k-loop", "groups meet", "epilogue", "drain vmcnt(0)", "seam barrier"]
for cap, (label, kind, arg) in [(c, x) for c in CAPS for x in ((("FC1 without GELU (first)", 1, 1), ("FC1 + GELU (K 1280, N 5120)", 1, 0), ("FC1 without GELU", 1, 1), ("FC2 (K 5120, N 1280)", 6, 0), ("FC1 + GELU again", 1, 0)) if len(CAPS) == 1 else (("FC1 without GELU", 1, 1), ("FC2 (K 5120, N 1280)", 6, 0)))]:
    eng.set_encoder_cap(cap)
    label = f"{label}, at most {cap or 'all'} CUs"
    for rep in range(3):
        ms = eng.probe(kind, B, 1, arg)
    st = np.zeros(2 * 3 * 16 * 8, dtype=np.uint64)
    rc = L.wx_lab_read_gemm_stamps(st.ctypes.data)
    assert rc == 0, rc
    st = st.reshape(2, 3, 16, 8).astype(np.int64)
    print(f"== {label}, {B} rows: launch {ms * 1e3:.1f} us")
    for bi, blk in enumerate((0, 128)):
        for wi, wv in enumerate((0, 4, 7)):
            rows = []
            for t in range(16):
                s = st[bi, wi, t]
                if s[0] == 0 or s[3] == 0:
                    break
                d = [(s[k + 1] - s[k]) / 100.0 if s[k + 1] and s[k] else float("nan") for k in range(5)]
                rows.append(d)
            if not rows:
                continue
            a = np.array(rows)
            full = a[~np.isnan(a).any(axis=1)]
            med = np.nanmedian(a, axis=0)
            cyc = np.median([st[bi, wi, t, 6] for t in range(len(rows))])
            print(f"  block {blk:3d} wave {wv}: {len(rows)} tiles; median us  " + "  ".join(f"{n} {m:6.2f}" for n, m in zip(names, med)) +
                  (f"   tile period {np.median(full.sum(axis=1)):6.2f}" if len(full) else "") + f"   k-loop {cyc:.0f} cycles = {cyc / med[0] / 1e3:.2f} GHz")
            if bi == 0 and wi == 0:
                for t, d in enumerate(rows[:8]):
                    print("      tile", t, "  ".join(f"{x:6.2f}" for x in d))
