"""A/B of pass plans for SMALL jobs (a 30-minute file = 60 chunks, the reference run's 81 VAD windows, 100 chunks) on the
GPU box: large-v3, random weights, 145 forced tokens, DTW words -- the shipped plan against equal passes, fewer and wider
passes, and 16-row passes on four contexts.   python tools/ab_small_jobs.py [N ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd.backend import WhisperHipBackend, plan_passes        # noqa: E402
from whisperx_mlx_amd.synth import speechlike_audio                         # noqa: E402

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
print("streams that run side by side for four lanes:", be._default_lanes(16, need=4), flush=True)     # (an explicit passes_in_flight=4 does not ask the hardware queues)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()


def equal(n, k):
    return [n // k + (1 if i < n % k else 0) for i in range(k)]


for N in [int(a) for a in sys.argv[1:]] or [60, 81, 100, 30]:
    segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(N)]
    shipped, lanes = plan_passes(N, 128)
    plans = [("shipped", shipped, lanes), ("3 equal", equal(N, 3), 3), ("2 equal", equal(N, 2), 2), ("one pass", [N], 1),
             ("4 equal", equal(N, 4), 4), ("6 equal x3", equal(N, 6), 3), ("4 equal x2", equal(N, 4), 2), ("8 equal x4", equal(N, 8), 4),
             ("16s x4", [16] * (N // 16) + ([N % 16] if N % 16 else []), 4)]
    seen = set()
    for name, rows, fl in plans:
        if (tuple(rows), fl) in seen or max(rows) > 128:
            continue
        seen.add((tuple(rows), fl))
        kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, pass_rows=rows, passes_in_flight=fl)
        be.transcribe_batch(segs, **kw)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            be.transcribe_batch(segs, **kw)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[1]
        print(f"N {N:4d} {name:12s} {str(rows):34s} x{fl}: {t * 1e3:7.1f} ms  {N * 30 / t:8.1f}x", flush=True)
