#!/usr/bin/env python3
"""Determinism stress of the decode loop (tagged-granule merge of the cross-attention key splits, hipGraph replay):
N decodes of the same input must give identical tokens and never raise the device error flag."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.tokenizer import get_tokenizer

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dims = weights.MODEL_DIMS[name]
ck = weights.random_checkpoint(dims, seed=0, std=0.05, device="cuda")
engs = [WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=B) for _ in range(2)]
tok = get_tokenizer(dims.n_vocab)
g = torch.Generator().manual_seed(5)
mel = (torch.randn(B, 3000, dims.n_mels, generator=g) * 0.5).half().cuda()
ref = None
for it in range(n_iter):
    e = engs[it % 2]
    enc = e.encode(mel)
    out = e.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=40, cross_split=(4, 2)[(it // 2) % 2])
    e.check_status()
    t = out.tokens.cpu().numpy().copy()
    key = (it // 2) % 2
    if ref is None:
        ref = {}
    if key not in ref:
        ref[key] = t
    elif not np.array_equal(ref[key], t):
        print("MISMATCH at iteration", it, "rows", np.nonzero((ref[key] != t).any(1))[0])
        sys.exit(1)
    if it % 20 == 0:
        print("iteration", it, "ok", flush=True)
print("STRESS OK", n_iter, "decodes,", name, "B =", B)

# ---- the same with three passes in flight (one launcher thread per engine, as bench.py runs them)
import threading
engs.append(WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=B))
results = [[] for _ in engs]
for e in engs:      # capture each context's graphs for these options before the threads start
    e.decode(e.encode(mel), tok, tok.sot_sequence(), rules=0, forced_len=40, cross_split=2, fc2_tile_n=16)
    e.check_status()


def worker(k):
    torch.cuda.set_device(0)
    e = engs[k]
    with torch.cuda.stream(e.stream):
        for _ in range(max(2, n_iter // 25)):
            enc = e.encode(mel)
            out = e.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=40, cross_split=2, fc2_tile_n=16)
            e.check_status()
            results[k].append(out.tokens.cpu().numpy().copy())


th = [threading.Thread(target=worker, args=(k,)) for k in range(len(engs))]
[t.start() for t in th]
[t.join() for t in th]
for k, rs in enumerate(results):
    for t in rs:
        if not np.array_equal(t, ref[1]):
            print("MISMATCH with passes in flight, engine", k)
            sys.exit(1)
print("STRESS OK with", len(engs), "passes in flight:", sum(len(r) for r in results), "decodes")
