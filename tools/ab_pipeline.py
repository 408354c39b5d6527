"""Lab: an ENCODER LANE on a partition of the chip against the product's schedule (every lane encodes its own pass, then
decodes it), for the driver's 320-chunk job and longer ones.  (after tools/ab_overlap.py and ab_cu_hog.py.)

  product-like   pass i on lane i % L: encode (all CUs) -> decode; L lanes run side by side
  pipelined      a fourth context encodes pass after pass on its own stream -- the first pass with all CUs, the later ones
                 capped to C CUs (wx_set_encoder_cap: persistent GEMM / attention blocks) while earlier passes decode on the
                 rest; lane i % L waits for pass i's encoder output (an event), then decodes

Encoder outputs travel as tensors; log-mel, DTW and the host halves are left out of both schedules (they are the same in
both).  Random log-mel input, bench.py's weights, 145 forced tokens, default filters, alignment-head capture on.

    GPU_MAX_HW_QUEUES=8 python tools/ab_pipeline.py [N_chunks ...]
"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd.backend import WhisperHipBackend, _new_context          # noqa: E402
from whisperx_mlx_amd.engine import RULES_LIGHTNING                            # noqa: E402
from whisperx_mlx_amd.scheduler import plan_passes                             # noqa: E402

be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
tok = be.tokenizer
prompt = tok.sot_sequence("en", "transcribe")
sup = tok.suppress_tokens()
L = be._default_lanes(112, need=3)
dec = be._get_engines(3, rows=112)
enc_lane = _new_context(be.dims, be.engine.packed, 128, 0, be.engine.alignment_heads)
print(f"decode lanes {len(dec)} (side by side {L}); GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}", flush=True)
g = torch.Generator().manual_seed(1)
mel_all = (torch.randn(128, 3000, 128, generator=g) * 0.5).half().cuda()
DKW = dict(rules=RULES_LIGHTNING, suppress_ids=sup, forced_len=145, capture_qk=True, fc2_tile_n=16, max_steps_ahead=32)


def launch_rows(n):
    return 16 if n <= 16 else 16 * -(-n // 16)


def product_like(sizes):
    def lane(k):
        e = dec[k]
        torch.cuda.set_device(e.device)
        with torch.cuda.stream(e.stream):
            for i in range(k, len(sizes), len(dec)):
                enc = e.encode(mel_all[: sizes[i]])
                e.decode(enc, tok, prompt, rows=launch_rows(sizes[i]), **DKW)
            e.stream.synchronize()
    th = [threading.Thread(target=lane, args=(k,)) for k in range(len(dec))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


def pipelined(sizes, cap, first_uncapped=True):
    n = len(sizes)
    encs, evs = [None] * n, [torch.cuda.Event() for _ in range(n)]
    ready = [threading.Event() for _ in range(n)]

    def encoder():
        torch.cuda.set_device(enc_lane.device)
        with torch.cuda.stream(enc_lane.stream):
            for i in range(n):
                enc_lane.set_encoder_cap(0 if (i == 0 and first_uncapped) else cap)
                # (the lane's workspace is reused pass after pass; its output tensor is the pass's own)
                encs[i] = enc_lane.encode(mel_all[: sizes[i]])
                evs[i].record(enc_lane.stream)
                ready[i].set()
        enc_lane.set_encoder_cap(0)

    def lane(k):
        e = dec[k]
        torch.cuda.set_device(e.device)
        with torch.cuda.stream(e.stream):
            for i in range(k, n, len(dec)):
                ready[i].wait()
                e.stream.wait_event(evs[i])
                e.decode(encs[i], tok, prompt, rows=launch_rows(sizes[i]), **DKW)
            e.stream.synchronize()
    th = [threading.Thread(target=encoder)] + [threading.Thread(target=lane, args=(k,)) for k in range(len(dec))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    enc_lane.stream.synchronize()
    return time.perf_counter() - t0


def best(fn, *a):
    fn(*a)
    return min(fn(*a) for _ in range(2))


def equal_units(n, k):
    """n chunks as k passes of whole 16-row groups, as equal as the groups allow; the ragged group comes off the first"""
    units = -(-n // 16)
    sizes = [16 * (units // k + (1 if i < units % k else 0)) for i in range(k)]
    sizes[0] -= 16 * units - n
    return [s for s in sizes if s > 0]


with torch.cuda.stream(enc_lane.stream):
    a = enc_lane.encode(mel_all[:32])
    enc_lane.set_encoder_cap(64)
    b = enc_lane.encode(mel_all[:32])
    enc_lane.set_encoder_cap(0)
torch.cuda.synchronize()
print("capped encoder output bit-identical to the uncapped one:", bool(torch.equal(a, b)), flush=True)
del a, b
for cap in (0, 64, 96, 128):         # the encoder alone at each cap (112 rows)
    enc_lane.set_encoder_cap(cap)
    with torch.cuda.stream(enc_lane.stream):
        enc_lane.encode(mel_all[:112])
        enc_lane.stream.synchronize()
        t0 = time.perf_counter()
        enc_lane.encode(mel_all[:112])
        enc_lane.stream.synchronize()
    print(f"encoder of 112 rows alone, cap {cap}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
enc_lane.set_encoder_cap(0)
for N in [int(a) for a in sys.argv[1:]] or [320, 768]:
    shipped = plan_passes(N, 128)[0]
    t_ref = best(product_like, shipped)
    print(f"N {N}: product-like {shipped}: {t_ref * 1e3:.0f} ms = {N * 30 / t_ref:.0f}x (encode + decode only)", flush=True)
    cands = [shipped]
    for k in ((4, 5, 6) if N <= 400 else (6, 9)):
        c = equal_units(N, k)
        if max(c) <= 128 and c not in cands:
            cands.append(c)
    for sizes in cands:
        if sizes is not shipped:
            t = best(product_like, sizes)
            print(f"   product-like {sizes}: {t * 1e3:.0f} ms = {N * 30 / t:.0f}x", flush=True)
        for cap in (0, 64, 96, 128):
            t = best(pipelined, sizes, cap)
            print(f"   pipelined    {sizes} cap {cap:3d}: {t * 1e3:.0f} ms = {N * 30 / t:.0f}x  ({t_ref / t:.3f} of the product-like shipped plan)", flush=True)
enc_lane.close()
