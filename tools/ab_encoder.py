"""Encoder alone on the GPU box: ms per 16 chunks at 16 and 64 rows per pass, and the FC1 / FC2 / attention probes.
    python tools/ab_encoder.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    print("library:", sys.argv[1])
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine

dims = weights.MODEL_DIMS["large-v3"]
ck = weights.random_checkpoint(dims, seed=0, device="cuda")
eng = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=64)
g = torch.Generator().manual_seed(1)
for B in (16, 64):
    mel = (torch.randn(B, 3000, dims.n_mels, generator=g) * 0.5).half().cuda()
    enc = eng.encode(mel)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        with torch.cuda.stream(eng.stream):
            e0.record(eng.stream)
            enc2 = eng.encode(mel)
            e1.record(eng.stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    d, T = dims.n_audio_state, dims.n_audio_ctx
    flops = B * (2 * T * 2 * d * 3 * dims.n_mels + 2 * T * d * 3 * d + dims.n_audio_layer * (4 * T * d * d * 2 + 2 * dims.n_audio_head * T * T * 64 * 2 + 2 * T * d * 4 * d * 2))
    print(f"encoder {B} rows: {best:.2f} ms ({best * 16 / B:.2f} ms per 16 chunks) = {flops / best / 1e9:.0f} TFLOP/s = {flops / best / 1e9 / 2500:.3f} of the MFMA peak; "
          f"bit-identical repeat {bool(torch.equal(enc, enc2))}; checksum {float(enc.float().abs().sum()):.1f}", flush=True)
    for name, kind in (("FC1+GELU", 1), ("FC2", 6), ("attention", 2)):
        print(f"   {name}: {eng.probe(kind, B, 8) * 1e3:.1f} us", flush=True)
