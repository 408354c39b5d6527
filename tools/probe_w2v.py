#!/usr/bin/env python3
"""wav2vec2-base CTC forward + CTC DP throughput on the GPU box (random weights)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import wav2vec2_ref as OWV     # only for seeded random weights of the base architecture
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.w2v import W2VConfig, W2VHipModel

cfg = W2VConfig()
w = OWV.random_weights(OWV.W2VDims(), seed=0)
m = W2VHipModel.from_state_dict(w, cfg)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
waves = [speechlike_audio(30.0, seed=i) for i in range(S)]
pcm = torch.from_numpy(np.stack(waves)).cuda()
lens = [480000] * S
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    with torch.cuda.stream(m.stream):
        ev[0].record(m.stream)
        logp, T = m.emissions_device(pcm, lens)
        ev[1].record(m.stream)
    torch.cuda.synchronize()
dt = ev[0].elapsed_time(ev[1]) * 1e-3
print(f"w2v-base emissions (resident PCM): {S} x 30 s in {dt*1e3:.2f} ms ({dt*1e3*16/S:.2f} ms per 16 x 30 s) -> {S*30/dt:.0f}x realtime, "
      f"{1.4e10*S*30/dt/1e12:.0f} TFLOP/s = {1.4e10*S*30/dt/2.5e15:.3f} of the MFMA peak, T={T[0]}, finite={bool(torch.isfinite(logp).all())}")
for _ in range(2):
    m.emissions(waves)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    logp, T = m.emissions(waves)
torch.cuda.synchronize()
print(f"w2v-base emissions (host arrays in, pinned staging): {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms for {S} x 30 s")
N = 3
tok = torch.randint(1, 32, (S, 400), dtype=torch.int32)
Nn = torch.full((S,), 400, dtype=torch.int32)
for _ in range(2):
    out = m.ctc_align(logp, torch.tensor(T, dtype=torch.int32), tok, Nn, 0, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    out = m.ctc_align(logp, torch.tensor(T, dtype=torch.int32), tok, Nn, 0, 2)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / N
print(f"ctc trellis+backtrack: {S} segments (T={T[0]}, N=400) in {dt2*1e3:.1f} ms, ok={out[2].cpu().tolist()}")
