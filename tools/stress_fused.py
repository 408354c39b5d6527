"""Stress of the product configuration (the backend's default scheduler, fused decode launch): N requests of 16 chunks; reports
the throughput, how many attention blocks of the fused launches computed their query themselves (forward progress without
the producer blocks, declayer.hip) and whether every repeat of a chunk decoded to the same tokens."""
import sys, time, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, return_chunks=True)   # the backend's own passes in flight
be.transcribe_batch(segs[:48], **kw)
torch.cuda.synchronize()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    t0 = time.perf_counter()
    r = be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
same = all(r["chunks"][i]["tokens"] == r["chunks"][i % 60 if i >= 60 else i]["tokens"] for i in range(len(r["chunks"])))
print(f"{K} requests: {K * 480 / dt:.1f}x, plan {be.last_plan['rows'][:6]}... x {be.last_plan['passes_in_flight']} in flight, self-computed queries: {be.selfq_blocks}, "
      f"key-split give-ups: {be.split_giveups}, warnings: {len(w)}, every repeat of a chunk decoded to the same tokens: {same}")
