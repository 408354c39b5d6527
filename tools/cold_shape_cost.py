"""What does a pass size that a context has not seen yet cost?  (hipGraph capture + instantiation of the prompt and step
graphs for that row count, enqueued from the calling thread before the launcher threads start.)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend, pass_sizes

be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
dev = torch.from_numpy(speechlike_audio(1800.0, seed=1234).reshape(60, 480000)).cuda()
kw = dict(batch_size=16, language="en", word_timestamps="dtw", forced_len=145, return_chunks=True)


def job(n):
    segs = [{"start": 0.0, "end": 30.0, "audio": dev[i % 60]} for i in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    be.transcribe_batch(segs, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


job(128)
job(128)
for n in (128 + 36, 128 + 44, 128 + 52):
    cold = job(n)
    warm = min(job(n), job(n))
    print(f"{n} chunks, cut {pass_sizes(n, 16, 4)[-5:]}: first time {cold:.0f} ms, again {warm:.0f} ms -> new pass sizes cost {cold - warm:.0f} ms", flush=True)
