"""Is the decode of several passes in flight bound by the GPU or by launching?  Decodes (large-v3, random weights, 145
tokens, no encoder beside them) with B rows per pass on N engine contexts: if the aggregate step rate stays put when
the rows -- and with them the bytes per step -- shrink, the bound is not HBM."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

K = int(sys.argv[1]) if len(sys.argv) > 1 else 9
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
NMAX = int(os.environ.get('LANES_MAX', '4'))
engines = be._get_engines(NMAX)
tok = be.tokenizer
if os.environ.get("SELECT_STREAMS"):
    # give every context a stream that runs side by side with the ones before it (what WhisperHipBackend._default_lanes does up to 4)
    import ctypes as C
    from whisperx_mlx_amd import _lib
    L = _lib.lib()

    def side_by_side(streams):
        arr = (C.c_void_p * len(streams))(*[C.c_void_p(st.cuda_stream) for st in streams])
        f = C.c_float(0.0)
        L.wx_streams_overlap(0, arr, len(streams), 300, C.byref(f))
        return f.value

    chosen = [engines[0].stream]
    for e in engines[1:]:
        for st in [e.stream] + [torch.cuda.Stream() for _ in range(24)]:
            if all(st.cuda_stream != c.cuda_stream for c in chosen) and side_by_side(chosen + [st]) < 1.35:
                e.stream = st
                chosen.append(st)
                break
        else:
            print("no further stream runs side by side after", len(chosen))
            break
    print("selected", len(chosen), "streams, factor all together", round(side_by_side(chosen), 2), flush=True)
prompt = tok.sot_sequence("en", "transcribe")
audio = torch.from_numpy(speechlike_audio(480.0, seed=1234).reshape(16, 480000)).cuda()
nv = torch.full((16,), 480000, dtype=torch.int32, device="cuda")
encs = []
for e in engines:
    with torch.cuda.stream(e.stream):
        encs.append(e.encode(e.logmel(audio, nv)))
torch.cuda.synchronize()


def dec(e, enc, fc2):
    with torch.cuda.stream(e.stream):
        return e.decode(enc, tok, prompt, rules=127, suppress_ids=be.suppress, capture_qk=True, forced_len=145,
                        cross_split=2, fc2_tile_n=fc2, step_variant=int(os.environ.get('STEP_VARIANT', '0')))


for rows in [int(x) for x in os.environ.get('ROWS', '16,4,1').split(',')]:
    for n in [int(x) for x in os.environ.get('LANES', ','.join(str(i) for i in range(1, NMAX + 1))).split(',')]:
        fc2 = 16 if n > 1 else 0
        for e, enc in zip(engines[:n], encs):
            dec(e, enc[:rows], fc2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()

        def work(k):
            torch.cuda.set_device(0)
            for _ in range(k, K, n):
                dec(engines[k], encs[k][:rows], fc2)
        th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"rows {rows:2d}, {n} in flight: {dt / K * 1e3:7.1f} ms per decode, {K * 147 / dt:7.0f} steps/s aggregate", flush=True)
