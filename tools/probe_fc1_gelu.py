"""What the GELU of the encoder's FC1 GEMM costs (wx_probe 1 with and without it), at 16 and 112 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.backend import WhisperHipBackend
be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
eng = be.engine
mel = (torch.randn(112, 3000, 128, generator=torch.Generator().manual_seed(1)) * 0.5).half().cuda()
eng.encode(mel)          # real activations in the workspace (the clock the chip holds depends on the data)
for B in (16, 112):
    d = be.dims.n_audio_state
    fl = 2.0 * B * 1500 * d * 4 * d
    for name, arg in (("FC1 + GELU", 0), ("FC1 without GELU", 1)):
        ms = min(eng.probe(1, B, 16, arg) for _ in range(3))
        print(f"B {B:3d} {name:18s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
    ms = min(eng.probe(6, B, 16) for _ in range(3))
    print(f"B {B:3d} {'FC2 (K = 4d)':18s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
