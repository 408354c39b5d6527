"""Per-kernel time of the decode step at a given pass width, single stream: rocprofv3 --kernel-trace --stats over one
pass of R rows (large-v3, random weights).   python tools/probe_rows.py R [positions]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend
R = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
be = WhisperHipBackend("large-v3", max_batch=16, coalesce=max(1, R // 16), random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(R)]
kw = dict(batch_size=16, language="en", word_timestamps=False, forced_len=T, rows_per_pass=R, passes_in_flight=1, return_chunks=True)
r0 = be.transcribe_batch(segs, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
r1 = be.transcribe_batch(segs, **kw)
torch.cuda.synchronize()
print(f"rows {R}: {(time.perf_counter() - t0) * 1e3:.1f} ms for {T} tokens; checksum {sum(sum(c['tokens']) for c in r1['chunks'])}", flush=True)
