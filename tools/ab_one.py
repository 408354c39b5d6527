"""one measurement of the product configuration (passes in flight as given, fused step) + single stream; prints tokens checksum"""
import sys, time, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(K * 16)]
for inflight, n in ((3, K), (1, 3), (3, K)):
    kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, passes_in_flight=inflight, return_chunks=True)
    be.transcribe_batch(segs[: 3 * 16], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = be.transcribe_batch(segs[: n * 16], **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    crc = zlib.crc32(str([c["tokens"] for c in r["chunks"][:48]]).encode())
    print(f"{sys.argv[2] if len(sys.argv) > 2 else ''} in flight {inflight}: {n * 480 / dt:8.1f}x  ({dt / n * 1e3:.1f} ms/step)  tokens crc {crc:08x}", flush=True)
