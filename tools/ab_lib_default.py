"""A/B of library builds on the DEFAULT scheduler (tools/ab_lib.py pins rows and passes): K requests of 16 chunks, large-v3,
random weights, 145 forced tokens, DTW words; a fresh process per library, best of three.
    python tools/ab_lib_default.py K lib [lib ...]     ("-" = the product library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, sys.argv[1])
from whisperx_mlx_amd import _lib
if sys.argv[2] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend
K = int(sys.argv[3])
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145)
be.transcribe_batch(segs, **kw)
torch.cuda.synchronize()
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    best = max(best, K * 480 / (time.perf_counter() - t0))
print(os.path.basename(sys.argv[2]), f"{best:8.1f}x  plan {be.last_plan['rows']} x {be.last_plan['passes_in_flight']}", flush=True)
'''
K = sys.argv[1]
for lib in sys.argv[2:]:
    subprocess.run([sys.executable, "-c", CHILD, ROOT, lib, K])
