"""Round 5: how should a job be cut when the decoder is shallow?  whisper-tiny (config 2: 60 chunks, batch 8), large-v3-turbo
and distil-large-v3 (4 / 2 decoder layers) under explicit cuts: one wide pass, two, three, four 15-row passes.
    python tools/ab_small_model_plans.py [model ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from whisperx_mlx_amd.backend import WhisperHipBackend
from whisperx_mlx_amd.synth import speechlike_audio

models = sys.argv[1:] or ["tiny", "large-v3-turbo", "distil-large-v3"]
audio = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
for model in models:
    be = WhisperHipBackend(model, device="cuda", max_batch=8 if model == "tiny" else 16, random_init=True, seed=0)
    for n in (60, 120, 320):
        segs = [{"start": 30.0 * j, "end": 30.0 * (j + 1), "audio": audio[j % 60]} for j in range(n)]
        cuts = {"default": None}
        for k in (1, 2, 3, 4, 6):
            per = -(-n // k)
            if per <= 128:
                cuts[f"{k} x {per}"] = [n // k + (1 if i < n % k else 0) for i in range(k)]
        ref = None
        for name, cut in cuts.items():
            for lanes in ((None,) if cut is None else sorted({min(len(cut), 4), min(len(cut), 3), min(len(cut), 2), 1})):
                kw = dict(batch_size=be.max_batch, language="en", word_timestamps="dtw", forced_len=145, return_chunks=True)
                if cut is not None:
                    kw.update(pass_rows=cut, passes_in_flight=lanes)
                try:
                    be.transcribe_batch(segs, **kw)
                    ts = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        res = be.transcribe_batch(segs, **kw)
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t0)
                except Exception as e:      # noqa: BLE001
                    print(f"{model} {n} chunks {name}: {type(e).__name__}: {str(e)[:100]}", flush=True)
                    continue
                toks = [c["tokens"] for c in res["chunks"]]
                ref = ref or toks
                dt = sorted(ts)[1]
                print(f"{model:16s} {n:4d} chunks  {name:10s} lanes {str(lanes):5s} plan {be.last_plan['rows']} x {be.last_plan['passes_in_flight']}: "
                      f"{dt * 1e3:8.1f} ms = {n * 30.0 / dt:8.0f}x   tokens equal {toks == ref}", flush=True)
