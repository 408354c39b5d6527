"""Where the alignment stage's time goes, forward by forward: host time inside _submit (enqueue), wait inside _collect (GPU not
done yet) and host assembly between them, for ten consecutive stages.   python tools/prof_align_stage.py"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                                  # noqa: E402
from whisperx_mlx_amd import alignment as AL                        # noqa: E402
from whisperx_mlx_amd.backend import WhisperHipBackend             # noqa: E402
from whisperx_mlx_amd.synth import speechlike_audio                # noqa: E402

dev = torch.device("cuda", 0)
be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
B._bench_align_model(be, dev)
audio = speechlike_audio(1800.0, seed=1234)
segs, lens, secs = B._vad_segments(torch.from_numpy(audio).to(dev))
kw = dict(batch_size=16, language="en", forced_len=max(lens), forced_lens=lens)
plain = be.transcribe_batch(segs, **kw)
log = []
_sub, _col = AL._HipAligner._submit, AL._HipAligner._collect


def sub(self, *a, **k):
    t0 = time.perf_counter()
    r = _sub(self, *a, **k)
    log.append(("submit", time.perf_counter() - t0))
    return r


def col(handle):
    t0 = time.perf_counter()
    r = _col(handle)
    log.append(("collect", time.perf_counter() - t0))
    return r


AL._HipAligner._submit = sub
AL._HipAligner._collect = staticmethod(col)
for it in range(10):
    res = copy.deepcopy(plain)
    log.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    be._align_batch_words(res, segs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"stage {dt * 1e3:6.1f} ms   " + "  ".join(f"{n} {t * 1e3:.1f}" for n, t in log), flush=True)
