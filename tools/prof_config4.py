"""Where config 4's alignment stage spends its time on the host (cProfile of transcribe_batch(align_words=True) on the 81
VAD-shaped chunks, after a warm-up call).   python tools/prof_config4.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                                  # noqa: E402
from whisperx_mlx_amd.backend import WhisperHipBackend             # noqa: E402
from whisperx_mlx_amd.synth import speechlike_audio                # noqa: E402

dev = torch.device("cuda", 0)
be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
B._bench_align_model(be, dev)
audio = speechlike_audio(1800.0, seed=1234)
segs, lens, secs = B._vad_segments(torch.from_numpy(audio).to(dev))
kw = dict(batch_size=16, language="en", forced_len=max(lens), forced_lens=lens)
plain = be.transcribe_batch(segs, **kw)
be.transcribe_batch(segs, align_words=True, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
plain = be.transcribe_batch(segs, **kw)
torch.cuda.synchronize()
t1 = time.perf_counter()
import copy
res = copy.deepcopy(plain)
pr = cProfile.Profile()
pr.enable()
out = be._align_batch_words(res, segs)
torch.cuda.synchronize()
pr.disable()
t2 = time.perf_counter()
print(f"ASR {1e3 * (t1 - t0):.1f} ms, align stage alone (profiled) {1e3 * (t2 - t1):.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
