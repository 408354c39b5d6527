"""Round 5: config 4's alignment stage with the forwards cut for host / GPU overlap (27 + 27 + 27 segments) against round
4's cut (64 + 17), same process: wall of backend._align_batch_words on the 81 VAD-shaped chunks, median of 5, and a
cProfile of one run.    python tools/ab_align_cuts.py"""
import copy, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                                  # noqa: E402
from whisperx_mlx_amd.backend import WhisperHipBackend             # noqa: E402
from whisperx_mlx_amd.synth import speechlike_audio                # noqa: E402

dev = torch.device("cuda", 0)
be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
B._bench_align_model(be, dev)
audio = speechlike_audio(1800.0, seed=1234)
segs, lens, secs = B._vad_segments(torch.from_numpy(audio).to(dev))
kw = dict(batch_size=16, language="en", forced_len=max(lens), forced_lens=lens)
plain = be.transcribe_batch(segs, **kw)
ref = None
from whisperx_mlx_amd import alignment as AL                        # noqa: E402
_r05 = AL._HipAligner._cuts
_r04 = lambda self, order: [order[a: a + self.max_batch] for a in range(0, len(order), self.max_batch)]      # noqa: E731  (round 4: max_batch segments per forward)
for cuts in ("r05", "r04", "r05", "r04"):
    AL._HipAligner._cuts = _r05 if cuts == "r05" else _r04
    ts = []
    for _ in range(6):
        res = copy.deepcopy(plain)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = be._align_batch_words(res, segs)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ref = ref or out["segments"]
    print(f"cuts {cuts}: align stage {sorted(ts)[len(ts) // 2] * 1e3:.1f} ms (runs {[round(t * 1e3, 1) for t in ts]}); same dict as the first: {out['segments'] == ref}", flush=True)
AL._HipAligner._cuts = _r05
res = copy.deepcopy(plain)
pr = cProfile.Profile()
pr.enable()
be._align_batch_words(res, segs)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
