"""Ragged jobs in the product configuration (default passes in flight, one launch shape per context, padding rows): every
chunk must decode to the same tokens whatever job it is part of."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend, pass_sizes

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=60, return_chunks=True)
ref = be.transcribe_batch([{"start": 0.0, "end": 30.0, "audio": dev[i]} for i in range(60)], passes_in_flight=1, **kw)["chunks"]
ref_tok = [c["tokens"] for c in ref]
ref_words = [c.get("words") for c in ref]
bad = 0
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for n in (1, 7, 17, 33, 47, 81, 100, 129, 200):
        idx = [(3 * i + n) % 60 for i in range(n)]
        t0 = time.perf_counter()
        out = be.transcribe_batch([{"start": 0.0, "end": 30.0, "audio": dev[j]} for j in idx], **kw)["chunks"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = sum(o["tokens"] == ref_tok[j] and o.get("words") == ref_words[j] for o, j in zip(out, idx))
        bad += n - same
        print(f"{n:4d} chunks, cut {pass_sizes(n, 16, be._default_lanes(16))[-6:]}: {n * 30 / dt:7.1f}x, chunks identical to the reference run {same}/{n}", flush=True)
print("give-ups:", len([x for x in w if "gave up" in str(x.message)]), " mismatching chunks:", bad, " passes in flight:", be.passes_in_flight)
