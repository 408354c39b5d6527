"""A/B of pass plans on the GPU box (large-v3, random weights, 145 forced tokens, DTW words) for several job sizes:
balanced equal passes (a multiple of three), full 64-row passes with the remainder first / last.   python tools/ab_plan.py [N ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()


def balanced(n, lanes=3, cap=64):
    p = lanes * -(-n // (lanes * cap))
    return [n // p + (1 if i < n % p else 0) for i in range(p)]


def full(n, first, cap=64):
    k, r = divmod(n, cap)
    rest = [r] if r else []
    return rest + [cap] * k if first else [cap] * k + rest


def full_split(n, cap=64):
    """full passes; a remainder is cut in multiples of 16 so that at least three passes exist"""
    k, r = divmod(n, cap)
    out = [cap] * k
    if r:
        out = [r] + out
    while len(out) < 3 and max(out) > 16:
        m = max(out)
        out.remove(m)
        a = 16 * (-(-m // 2) // 16) or m // 2
        out += [m - a, a] if a and m - a else [m]
        if a == 0 or m - a == 0:
            break
    return sorted(out)


for N in [int(a) for a in sys.argv[1:]] or [320, 384, 200, 160, 100, 81]:
    segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(N)]
    plans = {"balanced": balanced(N), "full, rest first": full(N, True), "full, rest last": full(N, False), "full split": full_split(N)}
    seen = set()
    for name, rows in plans.items():
        if tuple(rows) in seen:
            continue
        seen.add(tuple(rows))
        kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, pass_rows=rows, passes_in_flight=3)
        be.transcribe_batch(segs, **kw)
        torch.cuda.synchronize()
        best = 0
        for rep in range(2):
            t0 = time.perf_counter()
            be.transcribe_batch(segs, **kw)
            torch.cuda.synchronize()
            best = max(best, N * 30 / (time.perf_counter() - t0))
        print(f"N {N:4d} {name:18s} {str(rows):44s} {best:8.1f}x", flush=True)
