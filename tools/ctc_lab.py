"""Time the CTC alignment kernel over (T, N, beam) to see which phase dominates.  python tools/ctc_lab.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as G  # noqa: E402

eng = G.tiny_engine()[0]
g = torch.Generator().manual_seed(0)
V = 32
for (T, N, beam, S) in [(1499, 400, 2, 16), (1499, 400, 1, 16), (1499, 40, 2, 16), (750, 400, 2, 16), (1499, 400, 2, 1), (375, 100, 2, 16), (1499, 1000, 2, 16)]:
    logp = torch.log_softmax(torch.randn(S, T, V, generator=g), -1).cuda()
    tok = torch.randint(1, V, (S, N), generator=g, dtype=torch.int32).cuda()
    Tt = torch.full((S,), T, dtype=torch.int32)
    Nt = torch.full((S,), N, dtype=torch.int32).cuda()
    for _ in range(2):
        out = eng.ctc_align(logp, Tt, tok, Nt, 0, beam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = eng.ctc_align(logp, Tt, tok, Nt, 0, beam)
    torch.cuda.synchronize()
    print(f"T={T} N={N} beam={beam} S={S}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call, ok={out[2].cpu().tolist()[:4]}")
