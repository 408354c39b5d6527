"""A/B of pass plans on the GPU box (large-v3, random weights, 145 forced tokens, DTW words) for several job sizes:
the shipped plan_passes against candidates.   python tools/ab_plan.py [N ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend, plan_passes

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()


def even_units(n, lanes=3, cap_units=4):
    """groups of 16 rows dealt evenly to the lanes, each lane's share cut into passes of <= cap_units groups"""
    units = -(-n // 16)
    per = [units // lanes + (1 if i < units % lanes else 0) for i in range(lanes)]
    lane_passes = []
    for u in per:
        k = -(-u // cap_units) if u else 0
        lane_passes.append(sorted([u // k + (1 if i < u % k else 0) for i in range(k)]) if k else [])
    rows, left = [], n
    depth = max(len(p) for p in lane_passes)
    order = []
    for d in range(depth):
        for l in range(lanes):
            if d < len(lane_passes[l]):
                order.append(lane_passes[l][d] * 16)
    # the ragged group (n % 16 rows) comes off the first pass
    if n % 16:
        order[0] -= 16 - n % 16
    return [r for r in order if r > 0]


for N in [int(a) for a in sys.argv[1:]] or [320, 200, 160, 100, 81, 448]:
    segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(N)]
    plans = {"shipped": plan_passes(N, 128)[0], "even units": even_units(N, 3, 8)}
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
        print(f"N {N:4d} {name:12s} {str(rows):50s} {best:8.1f}x", flush=True)
