"""Round 5: the tile-pipelined 256 x 256 GEMM (gemm_pipe_kernel) against the one-tile-per-block kernel it replaces
(wx_set_encoder_cap(-1)), same process, same operands: whole encoder at 16 / 112 rows and the FC1 / FC2 probes.
    python tools/ab_gemm_pipe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd import _lib
if os.environ.get("AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])      # a lab build (tools/build_lab.py)
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine

dims = weights.MODEL_DIMS["large-v3"]
ck = weights.random_checkpoint(dims, seed=0, device="cuda")
eng = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=112)
g = torch.Generator().manual_seed(1)
d, T = dims.n_audio_state, dims.n_audio_ctx
for B in (16, 112):
    mel = (torch.randn(B, 3000, dims.n_mels, generator=g) * 0.5).half().cuda()
    outs = {}
    runs = (("one tile per block", -1), ("tile-pipelined", 0), ("one tile per block", -1), ("tile-pipelined", 0))
    if os.environ.get("AB_PIPE_ONLY"):
        runs = (("one tile per block", -1), ("tile-pipelined", 0), ("tile-pipelined", 0))
    for name, cap in runs:
        eng.set_encoder_cap(cap)
        enc = eng.encode(mel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            with torch.cuda.stream(eng.stream):
                e0.record(eng.stream)
                enc = eng.encode(mel)
                e1.record(eng.stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        flops = B * (2 * T * 2 * d * 3 * dims.n_mels + 2 * T * d * 3 * d + dims.n_audio_layer * (4 * T * d * d * 2 + 2 * dims.n_audio_head * T * T * 64 * 2 + 2 * T * d * 4 * d * 2))
        outs.setdefault(name, enc.clone())
        same = bool(torch.equal(enc, outs["one tile per block"]))
        probes = "  ".join(f"{n} {eng.probe(k, min(B, 112), 8) * 1e3:.1f} us" for n, k in (("FC1+GELU", 1), ("FC2", 6), ("attention", 2)))
        lab = " ".join(f"{k}={os.environ[k]}" for k in ("WX_GEMM_STAGGER_US", "WX_GEMM_NT", "WX_GEMM_4W") if k in os.environ)
        print(f"{lab} {B:4d} rows  {name:20s} encoder {best:8.2f} ms ({best * 16 / B:.2f} per 16 chunks) = {flops / best / 1e9:5.0f} TFLOP/s = "
              f"{flops / best / 1e9 / 2500:.3f} of peak   {probes}   bits equal to one-tile-per-block: {same}", flush=True)
eng.set_encoder_cap(0)
eng.check_status()
