"""A/B of a backend option in the product configuration (default passes in flight): K requests of 16 chunks, large-v3,
random weights, 145 forced tokens, DTW words.   python tools/ab_backend_opt.py K name=value [name=value ...]"""
import sys, time, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

K = int(sys.argv[1])
opts = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    opts[k] = int(v)
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0, **opts)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, return_chunks=True)
be.transcribe_batch(segs[:64], **kw)
torch.cuda.synchronize()
best = 0.0
for rep in range(2):
    t0 = time.perf_counter()
    be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    best = max(best, K * 480 / (time.perf_counter() - t0))
print(f"{opts or 'defaults'}: {best:.1f}x  (passes in flight {be.passes_in_flight})", flush=True)
