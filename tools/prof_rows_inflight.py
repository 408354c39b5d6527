"""One configuration of tools/ab_rows_inflight.py for the profiler: rows per pass, passes in flight, K requests.
    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_rows_inflight.py 64 3 24"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

rows, inflight, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
be = WhisperHipBackend("large-v3", max_batch=16, coalesce=max(1, rows // 16), random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, rows_per_pass=rows, passes_in_flight=inflight)
be.transcribe_batch(segs[: inflight * rows], **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
be.transcribe_batch(segs, **kw)
torch.cuda.synchronize()
print(f"rows {rows} x {inflight}: {K * 480 / (time.perf_counter() - t0):.1f}x", flush=True)
