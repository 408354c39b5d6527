"""Does the launch timer of the fused decode launch cost anything, and does it agree with rocprof?  16 rows x 1 and 64 rows
x 3, timer off / on, large-v3 random weights, 145 forced tokens.   python tools/ab_launch_timer.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
K = 24
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
for rows, lanes, k in ((16, 1, 6), (64, 3, K)):
    for prof in (False, True, False, True):
        be.profile_launches = prof
        kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, rows_per_pass=rows, passes_in_flight=lanes)
        be.transcribe_batch(segs[: lanes * rows], **kw)
        torch.cuda.synchronize()
        for e in be.engines:
            e.launch_profile()
        t0 = time.perf_counter()
        be.transcribe_batch(segs[: k * 16], **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        recs = [e.launch_profile() for e in be.engines[:lanes]]
        n = sum(r[1] for r in recs)
        avg = sum(r[0] * r[1] for r in recs) / max(n, 1)
        print(f"rows {rows} x {lanes}, timer {'on ' if prof else 'off'}: {k * 480 / dt:8.1f}x   timed launches {n:7d}  average {avg:7.2f} us", flush=True)
