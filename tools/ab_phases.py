"""How fast do N decodes run in flight when no encoder runs beside them?  (large-v3, random weights, 145 tokens.)
Pre-encodes, then times K decodes spread over N engine contexts / launcher threads; and the encoders alone."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 9
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
engines = be._get_engines(N)
tok = be.tokenizer
prompt = tok.sot_sequence("en", "transcribe")
audio = torch.from_numpy(speechlike_audio(480.0, seed=1234).reshape(16, 480000)).cuda()
nv = torch.full((16,), 480000, dtype=torch.int32, device="cuda")
encs = []
for e in engines:
    with torch.cuda.stream(e.stream):
        encs.append(e.encode(e.logmel(audio, nv)))
torch.cuda.synchronize()


def dec(e, enc, split_fc2):
    with torch.cuda.stream(e.stream):
        return e.decode(enc, tok, prompt, rules=127, suppress_ids=be.suppress, capture_qk=True, forced_len=145,
                        cross_split=2, fc2_tile_n=split_fc2, step_variant=1)


for n in (1, N):
    fc2 = 16 if n > 1 else 0
    for e, enc in zip(engines[:n], encs):
        dec(e, enc, fc2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()

    def work(k):
        torch.cuda.set_device(0)
        for _ in range(k, K, n):
            dec(engines[k], encs[k], fc2)
    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"decode only, {n} in flight: {dt / K * 1e3:.1f} ms per 16-row decode  ({841.0 / (dt / K) / 1e3:.2f} TB/s algorithmic)", flush=True)

# encoders alone, back to back on one stream
e = engines[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(e.stream):
    for _ in range(6):
        e.encode(e.logmel(audio, nv))
torch.cuda.synchronize()
print(f"log-mel + encoder alone: {(time.perf_counter() - t0) / 6 * 1e3:.1f} ms per 16 chunks")
