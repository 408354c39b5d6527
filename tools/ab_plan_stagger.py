"""Round 5: the encoders of a job's first passes now run one after the other (shared encoder workspace), so the first pass
starts decoding ~0.27 s before the second and ~0.54 s before the third.  Does an UNEQUAL deal of the rows (more to the pass
that starts first) end the passes together and beat 112 + 112 + 96?   python tools/ab_plan_stagger.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
N = 320
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(N)]
plans = [[112, 112, 96], [128, 112, 80], [128, 96, 96], [120, 104, 96], [128, 104, 88], [96, 112, 112], [112, 112, 96]]
ref = None
for rows in plans:
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, pass_rows=rows, passes_in_flight=3, return_chunks=True)
    be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        res = be.transcribe_batch(segs, **kw)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    toks = [c["tokens"] for c in res["chunks"]]
    ref = ref or toks
    print(f"{str(rows):18s} median {N * 30 / sorted(ts)[1]:8.1f}x  best {N * 30 / min(ts):8.1f}x  runs ms {[round(t * 1e3) for t in ts]}  tokens equal {toks == ref}", flush=True)
