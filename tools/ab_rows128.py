"""128-row passes against 64-row passes on contexts that take 128 rows: K requests of 16 chunks, large-v3, random weights,
145 forced tokens, DTW words, three passes in flight; best of two; tokens and log-probabilities compared.
    python tools/ab_rows128.py K [K ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0, max_rows=128)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
for K in [int(a) for a in sys.argv[1:]] or [20]:
    segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
    ref = None
    for rows in (64, 128, None):
        kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, return_chunks=True)
        if rows:
            kw.update(rows_per_pass=rows, passes_in_flight=3)
        be.transcribe_batch(segs[: 3 * (rows or 128)], **kw)
        torch.cuda.synchronize()
        best, out = 0.0, None
        for rep in range(2):
            t0 = time.perf_counter()
            out = be.transcribe_batch(segs, **kw)
            torch.cuda.synchronize()
            best = max(best, K * 480 / (time.perf_counter() - t0))
        sig = [(tuple(c["tokens"]), float(c["avg_logprob"])) for c in out["chunks"]]
        ref = ref or sig
        print(f"K {K}: rows per pass {rows or 'default plan'}: {best:8.1f}x   plan {be.last_plan['rows'] if be.last_plan else None} x {be.last_plan['passes_in_flight'] if be.last_plan else None}"
              f"   same tokens and log-probabilities as 64-row passes: {sig == ref}", flush=True)
print("memory allocated by torch:", round(torch.cuda.memory_allocated() / 2**30, 1), "GiB; device free/total:", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info()])
