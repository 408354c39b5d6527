"""A/B on the GPU box: decode step variants (0 = fused launches, 1 = one kernel per stage) on large-v3, seeded random
weights, forced 145 tokens: single stream and 3 passes in flight; tokens must be identical."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
audio = speechlike_audio(1800.0, seed=1234).reshape(60, 480000)
dev = torch.from_numpy(audio).cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 9
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 0]
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
out = {}
for inflight in (1, 3):
    for v in variants:
        be.step_variant = v
        kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, passes_in_flight=inflight, return_chunks=True)
        be.transcribe_batch(segs[: 3 * 16], **kw)
        torch.cuda.synchronize()
        be.stage_ms = {}
        t0 = time.perf_counter()
        r = be.transcribe_batch(segs if inflight > 1 else segs[: 3 * 16], **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = K if inflight > 1 else 3
        st = {k: round(x / n, 2) for k, x in be.stage_ms.items()}
        be.stage_ms = None
        print(f"variant {v} in flight {inflight}: {n * 480 / dt:8.1f}x  ({dt / n * 1e3:.1f} ms/step)  stages {st}", flush=True)
        out[(inflight, v)] = [c["tokens"] for c in r["chunks"]]
for inflight in (1, 3):
    ks = [k for k in out if k[0] == inflight]
    print("in flight", inflight, "tokens identical across variants:", all(out[k] == out[ks[0]] for k in ks))
