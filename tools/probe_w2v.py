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
S = 16
waves = [speechlike_audio(30.0, seed=i) for i in range(S)]
for _ in range(2):
    logp, T = m.emissions(waves)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 3
for _ in range(N):
    logp, T = m.emissions(waves)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"w2v-base emissions: {S} x 30 s in {dt*1e3:.1f} ms -> {S*30/dt:.0f}x realtime, T={T[0]}, finite={bool(torch.isfinite(logp).all())}")
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
