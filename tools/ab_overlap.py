"""Does the MFMA-bound encoder overlap with the HBM-bound decode AT ALL on this chip?  

Three decode passes of 112 rows in flight (pre-encoded, 145 forced tokens, exactly the launches of bench.py's `value`)
against an encoder loop on a fourth context, alone and together:

    efficiency = (decode alone / decode beside the encoder) + (encoder iterations done beside the decode x encoder alone / that time)

1.0 = work-conserving (running them side by side buys nothing, whatever the schedule); 2.0 = perfectly orthogonal resources.
Variants: encoder stream at low priority / decode streams at high priority; encoder batch 16 / 64 rows.

    python tools/ab_overlap.py [rows_enc ...]
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd import _lib                                              # noqa: E402
if os.environ.get("WX_LAB_LIB"):          # a lab build of the library (tools/build_lab.py)
    _lib.LIB_PATH = os.path.abspath(os.environ["WX_LAB_LIB"])
    print("library:", _lib.LIB_PATH, flush=True)
from whisperx_mlx_amd.backend import WhisperHipBackend, _new_context          # noqa: E402
from whisperx_mlx_amd.engine import RULES_LIGHTNING                            # noqa: E402

ROWS = 112
be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
tok = be.tokenizer
prompt = tok.sot_sequence("en", "transcribe")
sup = tok.suppress_tokens()
lanes = be._default_lanes(ROWS, need=3)
dec = be._get_engines(3, rows=ROWS)
print(f"decode contexts: {len(dec)} (side by side: {lanes})", flush=True)
g = torch.Generator().manual_seed(1)
mel = (torch.randn(ROWS, 3000, 128, generator=g) * 0.5).half().cuda()
encs = [e.encode(mel) for e in dec]
torch.cuda.synchronize()


def decode_all(kw=None):
    def one(e, x):
        torch.cuda.set_device(e.device)
        e.decode(x, tok, prompt, rules=RULES_LIGHTNING, suppress_ids=sup, forced_len=145, capture_qk=True, rows=ROWS,
                 fc2_tile_n=16, max_steps_ahead=32)
        e.stream.synchronize()
    th = [threading.Thread(target=one, args=(e, x)) for e, x in zip(dec, encs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


decode_all()            # graphs
t_dec = min(decode_all() for _ in range(2))
print(f"decode alone: 3 x {ROWS} rows x 145 tokens in {t_dec * 1e3:.1f} ms", flush=True)

for rows_enc in [int(a) for a in sys.argv[1:]] or [16, 64]:
    enc_ctx = _new_context(be.dims, be.engine.packed, rows_enc, 0, be.engine.alignment_heads)
    mel_e = mel[:rows_enc].contiguous()
    for prio_name, enc_prio, dec_prio in (("default priorities", 0, 0), ("encoder low / decode high", 0, -1)):
        enc_ctx.stream = torch.cuda.Stream(priority=enc_prio)
        old = [e.stream for e in dec]
        if dec_prio:
            continue_ = True
            for e in dec:
                e.stream = torch.cuda.Stream(priority=dec_prio)
            # a new stream: graphs are launched on whatever stream the context has; check they still run side by side
            decode_all()
            t_dec_p = min(decode_all() for _ in range(2))
            print(f"  [{prio_name}] decode alone on high-priority streams: {t_dec_p * 1e3:.1f} ms", flush=True)
        else:
            t_dec_p = t_dec
        enc_ctx.encode(mel_e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            enc_ctx.encode(mel_e)
        enc_ctx.stream.synchronize()
        t_enc = (time.perf_counter() - t0) / 8
        stop = threading.Event()
        count = [0]

        def enc_loop():
            torch.cuda.set_device(enc_ctx.device)
            while not stop.is_set():
                enc_ctx.encode(mel_e)
                enc_ctx.stream.synchronize()
                count[0] += 1

        th = threading.Thread(target=enc_loop)
        th.start()
        time.sleep(0.05)
        c0 = count[0]
        t_both = decode_all()
        c1 = count[0]
        stop.set()
        th.join()
        n = c1 - c0
        eff = t_dec_p / t_both + n * t_enc / t_both
        print(f"encoder {rows_enc} rows, {prio_name}: encoder alone {t_enc * 1e3:.1f} ms/iter; together: decode {t_both * 1e3:.1f} ms "
              f"(x{t_both / t_dec_p:.2f}), {n} encoder iterations = {n * t_enc * 1e3:.0f} ms of encoder work -> efficiency {eff:.2f}", flush=True)
        if dec_prio:
            for e, s in zip(dec, old):
                e.stream = s
    enc_ctx.close()

# host facts for the CPU-side tests / cpu_baseline
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
a = torch.randn(4096, 4096)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    a @ a
    t0 = time.perf_counter()
    for _ in range(3):
        a @ a
    dt = (time.perf_counter() - t0) / 3
    print(f"fp32 matmul 4096^3, {nt} threads: {2 * 4096 ** 3 / dt / 1e12:.2f} TFLOP/s", flush=True)
