"""Sweep on the GPU box: rows per pass (requests of 16 chunks coalesced into one pass) x passes in flight, large-v3,
seeded random weights, 145 forced tokens, DTW words, K requests of 16 chunks through transcribe_batch.
    python tools/ab_rows_inflight.py [K] [rows,inflight ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or \
    [(16, 4), (16, 3), (32, 4), (32, 3), (32, 2), (48, 4), (48, 3), (48, 2), (64, 4), (64, 3), (64, 2), (64, 1)]
be = WhisperHipBackend("large-v3", max_batch=16, coalesce=4, random_init=True, seed=0)
audio = speechlike_audio(1800.0, seed=1234).reshape(60, 480000)
dev = torch.from_numpy(audio).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
ref = None
for rows, inflight in combos:
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, rows_per_pass=rows, passes_in_flight=inflight, return_chunks=True)
    be.transcribe_batch(segs[: inflight * rows], **kw)
    torch.cuda.synchronize()
    be.stage_ms = {}
    t0 = time.perf_counter()
    r = be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = {k: round(v / K, 1) for k, v in be.stage_ms.items()}
    be.stage_ms = None
    toks = [c["tokens"] for c in r["chunks"]]
    ref = ref or toks
    print(f"rows {rows:2d} x {inflight} in flight: {K * 480 / dt:8.1f}x  ({dt / K * 1e3:6.1f} ms / 16 chunks)  stages/16 chunks {st}  tokens identical {toks == ref}", flush=True)
