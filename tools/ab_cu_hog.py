"""How many compute units does the HBM-bound decode need?  (can the encoder have some?)

N CUs are taken away by a hog kernel (tools/cu_hog.hip: one 1024-thread block holding all 160 KB of a CU's LDS, asleep)
while the product decodes three 112-row passes in flight (bench.py's launches), and while the fused cross-attention
launch runs alone back to back.  If the decode keeps its speed on ~176 CUs, an encoder confined to the other ~80 could run
under it; if it slows in proportion, no schedule can hide the encoder.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/cu_hog.hip -o tools/_bin/libcuhog.so
    python tools/ab_cu_hog.py [N ...]
"""
import ctypes as C
import os
import sys
import threading
import time
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd.backend import WhisperHipBackend          # noqa: E402
from whisperx_mlx_amd.engine import RULES_LIGHTNING             # noqa: E402

HOG = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libcuhog.so"))
HOG.hog_launch.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
ROWS = 112
be = WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)
tok = be.tokenizer
prompt = tok.sot_sequence("en", "transcribe")
sup = tok.suppress_tokens()
be._default_lanes(ROWS, need=3)
dec = be._get_engines(3, rows=ROWS)
g = torch.Generator().manual_seed(1)
mel = (torch.randn(ROWS, 3000, 128, generator=g) * 0.5).half().cuda()
encs = [e.encode(mel) for e in dec]
torch.cuda.synchronize()


def decode_all(tokens=145):
    def one(e, x):
        torch.cuda.set_device(e.device)
        e.decode(x, tok, prompt, rules=RULES_LIGHTNING, suppress_ids=sup, forced_len=tokens, capture_qk=True, rows=ROWS,
                 fc2_tile_n=16, max_steps_ahead=32)
        e.stream.synchronize()
    th = [threading.Thread(target=one, args=(e, x)) for e, x in zip(dec, encs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


decode_all()
hog_stream = torch.cuda.Stream()
flag = torch.zeros(16, dtype=torch.int32).pin_memory()        # host memory the hog polls directly (no copy on a third stream)
where = torch.zeros(2 * 256, dtype=torch.int32, device="cuda")
base = {}
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for n in [int(a) for a in sys.argv[1:]] or [0, 1, 32, 64, 80, 96, 128]:
    res = {}
    for what in ("decode", "attn", "encode"):
        flag.zero_()
        torch.cuda.synchronize()
        if n:
            rc = HOG.hog_launch(C.c_void_p(hog_stream.cuda_stream), n, 8.0, C.c_void_p(flag.data_ptr()), C.c_void_p(where.data_ptr()))
            assert rc == 0, rc
            time.sleep(0.05)
        if what == "decode":
            res[what] = decode_all() * 1e3
        elif what == "attn":
            res[what] = dec[0].probe(13, ROWS, 128) * 1e3          # the fused [LN + cross-Q GEMV] -> [cross attention] launch, us
        else:
            e = dec[0]
            e.encode(mel[:16].contiguous())
            e.stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                e.encode(mel[:16].contiguous())
            e.stream.synchronize()
            res[what] = (time.perf_counter() - t0) / 4 * 1e3
        flag[0] = 1
        t_h = time.perf_counter()
        hog_stream.synchronize()
        waited = time.perf_counter() - t_h
        if waited > 0.5:
            print(f"   (the hog ran {waited:.1f} s beyond the measurement: it did not see the flag)", flush=True)
    if n:
        w = where[: 2 * n].cpu().view(-1, 2)
        xcc = Counter(int(x) for x in w[:, 1].tolist())
        distinct = len({(int(a), int(b)) for a, b in w.tolist()})
    else:
        xcc, distinct = {}, 0
        base = dict(res)
    print(f"hog {n:3d} CUs ({distinct} distinct hw ids, per XCC {sorted(xcc.items())}): decode 3x{ROWS} rows {res['decode']:.0f} ms "
          f"(x{res['decode'] / base['decode']:.2f})  fused attention launch alone {res['attn']:.1f} us (x{res['attn'] / base['attn']:.2f})  "
          f"encoder 16 rows {res['encode']:.1f} ms (x{res['encode'] / base['encode']:.2f})   [CUs left {256 - n}: x{256 / (256 - n):.2f} if proportional]",
          flush=True)
