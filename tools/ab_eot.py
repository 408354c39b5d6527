"""Decode time of batches whose rows END AT DIFFERENT LENGTHS, drawn from the length distribution of the reference's own
run (tests/golden/gold30m_windows.json: 81 windows, 4..199 tokens, mean 108): large-v3, random weights, per-row forced
lengths (wx_decode_opts.forced_lens).  A finished row takes no part in the attention kernels any more, so the step gets
cheaper as rows end; before, every step streamed the cross K/V of all 16 rows until the longest row was done."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0)
eng, tok = be.engine, be.tokenizer
dev = torch.from_numpy(speechlike_audio(480.0, seed=77).reshape(16, 480000)).cuda()
enc = eng.encode(eng.logmel(dev, torch.full((16,), 480000, dtype=torch.int32, device="cuda")))
prompt = tok.sot_sequence("en", "transcribe")
lens_all = [len(w["tokens"]) for w in json.load(open(os.path.join(ROOT, "tests", "golden", "gold30m_windows.json")))["windows"]]
uniform = None
for g in range(0, 80, 16):
    lens = lens_all[g: g + 16]
    fl = torch.tensor(lens, dtype=torch.int32).cuda()
    for mode in ("uniform", "per-row"):
        kw = dict(rules=127, suppress_ids=be.suppress, forced_len=max(lens), capture_qk=True)
        if mode == "per-row":
            kw["forced_lens"] = fl
        eng.decode(enc, tok, prompt, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.decode(enc, tok, prompt, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if mode == "uniform":
            uniform = dt
        else:
            t = out.tokens.cpu().numpy()[:, len(prompt): len(prompt) + out.n_sampled]
            got = [int(np.argmax(r == tok.eot)) if (r == tok.eot).any() else out.n_sampled for r in t]
            assert got == lens, (got, lens)
            print(f"windows {g:2d}-{g + 15}: lengths {min(lens)}..{max(lens)} (mean {np.mean(lens):.0f}); every row to the longest "
                  f"{uniform * 1e3:6.1f} ms, rows sitting out after their EOT {dt * 1e3:6.1f} ms  ({100 * (1 - dt / uniform):.0f} % less)", flush=True)
eng.check_status()

# ---- the same with 3 decodes in flight (the product's configuration): 15 decodes over 3 contexts
import threading
engines = be._get_engines(3)
encs = [enc] + [e.encode(e.logmel(dev, torch.full((16,), 480000, dtype=torch.int32, device="cuda"))) for e in engines[1:]]
groups = [lens_all[g: g + 16] for g in range(0, 80, 16)]
fls = [torch.tensor(g, dtype=torch.int32).cuda() for g in groups]
for mode in ("uniform", "per-row", "uniform", "per-row"):
    def work(k):
        torch.cuda.set_device(0)
        e = engines[k]
        with torch.cuda.stream(e.stream):
            for i in range(k, 15, 3):
                g = i % 5
                kw = dict(rules=127, suppress_ids=be.suppress, forced_len=max(groups[g]), capture_qk=True, fc2_tile_n=16)
                if mode == "per-row":
                    kw["forced_lens"] = fls[g]
                e.decode(encs[k], tok, prompt, **kw)
    for k in range(3):          # graphs of every shape, serially
        for g in range(5):
            kw = dict(rules=127, suppress_ids=be.suppress, forced_len=max(groups[g]), capture_qk=True, fc2_tile_n=16)
            if mode == "per-row":
                kw["forced_lens"] = fls[g]
            with torch.cuda.stream(engines[k].stream):
                engines[k].decode(encs[k], tok, prompt, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"3 decodes in flight, 15 batches of the reference's window lengths, {mode:8s}: {dt * 1e3 / 15:.1f} ms per batch", flush=True)
for e in engines:
    e.check_status()
