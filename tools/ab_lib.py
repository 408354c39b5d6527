"""A/B of library builds (lab variants under tools/_bin/) on the GPU box: for each lib given, a fresh process measures
16 rows x 1, 16 rows x 4 and 64 rows x 3 (large-v3, random weights, 145 forced tokens, K requests) and prints the
fused launch's self-computed-query count.   python tools/ab_lib.py K lib [lib ...]   ("-" = the product library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from whisperx_mlx_amd import _lib
if sys.argv[2] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend
K = int(sys.argv[3])
be = WhisperHipBackend("large-v3", max_batch=16, coalesce=4, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
out = []
for rows, inflight, k in ((16, 1, max(2, K // 4)), (16, 4, K), (64, 3, K)):
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, rows_per_pass=rows, passes_in_flight=inflight)
    be.transcribe_batch(segs[: inflight * rows], **kw)
    torch.cuda.synchronize()
    s0 = be.selfq_blocks
    t0 = time.perf_counter()
    be.transcribe_batch(segs[: k * 16], **kw)
    torch.cuda.synchronize()
    out.append(f"{rows}x{inflight}: {k * 480 / (time.perf_counter() - t0):7.1f}x selfq {be.selfq_blocks - s0}")
print(os.path.basename(sys.argv[2]), " | ".join(out), flush=True)
'''
K = sys.argv[1]
for lib in sys.argv[2:]:
    subprocess.run([sys.executable, "-c", CHILD, ROOT, lib, K])
