"""A/B on the GPU box: passes in flight and the FC2 tile under the default (fused) decode step.  large-v3, random weights,
145 forced tokens, DTW, filters on; K requests of 16 chunks."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
for inflight, fc2 in ((3, 16), (3, 0), (2, 16), (4, 16), (3, 16)):
    be.fc2_tile_n = fc2
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, passes_in_flight=inflight)
    be.transcribe_batch(segs[: inflight * 16], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {inflight} fc2_tile_n {fc2:2d}: {K * 480 / dt:8.1f}x  ({dt / K * 1e3:.1f} ms/step)", flush=True)
