"""A/B of library builds on the driver's `value` job (320 chunks, default scheduler) and the 16 x 4 mode, fresh process per
library, median of three.   python tools/ab_lib_value.py lib [lib ...]   ("-" = the product library)"""
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
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(320)]
out = []
lanes16 = be._default_lanes(16)          # streams that really run side by side (the hardware queues are asked)
for name, kw in (("default", {}), (f"16x{lanes16}", dict(rows_per_pass=16, passes_in_flight=lanes16))):
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, **kw)
    be.transcribe_batch(segs, **kw)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        be.transcribe_batch(segs, **kw)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out.append(f"{name}: {320 * 30 / sorted(ts)[1]:7.1f}x ({', '.join(f'{t * 1e3:.0f}' for t in ts)} ms)")
print(os.path.basename(sys.argv[2]), " | ".join(out), flush=True)
'''
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", CHILD, ROOT, lib])
