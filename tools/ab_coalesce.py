"""A/B on the GPU box: 48-row passes with and without a key split of the decode cross-attention, and whether tokens
depend on the pass size at the same split (large-v3, seeded random weights, forced 145 tokens)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, coalesce=3, random_init=True, seed=0)
audio = speechlike_audio(1800.0, seed=1234).reshape(60, 480000)
dev = torch.from_numpy(audio).cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]


def run(rows, inflight, split):
    be.cross_split = split
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, rows_per_pass=rows, passes_in_flight=inflight, return_chunks=True)
    be.transcribe_batch(segs[: max(3 * 16, inflight * rows)], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rows {rows} in flight {inflight} split {split}: {K * 480 / dt:8.1f}x  ({dt / K * 1e3:.1f} ms/step)", flush=True)
    return [c["tokens"] for c in r["chunks"]]


a = run(16, 3, 2)
b = run(48, 2, 2)
c = run(48, 2, 1)
d = run(16, 3, 1)
e = run(16, 1, 2)
print("16x3 s2 == 48x2 s2:", a == b, " 48x2 s1 == 16x3 s1:", c == d, " s2 == s1:", a == d, " 16x3 == 16x1:", a == e)
n = sum(x != y for x, y in zip(a, d))
print("rows differing between split 2 and split 1:", n, "of", len(a))
